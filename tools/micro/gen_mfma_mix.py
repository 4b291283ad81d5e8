"""Generates mfma_mix.cpp: v_mfma_f32_32x32x2_f32 rate with other instructions interleaved (hand-written register numbers).
Usage: python gen_mfma_mix.py > mfma_mix.cpp"""
def body(extra):
    out = []
    for u in range(16):
        acc = f"a[{16 * (u % 2)}:{16 * (u % 2) + 15}]"
        out.append(f"v_mfma_f32_32x32x2_f32 {acc}, a{40+u}, v{30+u}, {acc}")
        out += extra(u)
    return out
variants = {
    "MFMA only": lambda u: [],
    "+2 v_max on idle VGPRs": lambda u: [f"v_max_f32 v{60+u}, 0, v{60+u}", f"v_max_f32 v{80+u}, 0, v{80+u}"],
    "+8 v_max on idle VGPRs": lambda u: [f"v_max_f32 v{60+(u*8+j)%40}, 0, v{60+(u*8+j)%40}" for j in range(8)],
    "+14 v_max on idle VGPRs": lambda u: [f"v_max_f32 v{60+(u*14+j)%40}, 0, v{60+(u*14+j)%40}" for j in range(14)],
    "+1 accvgpr_read (idle AGPR)": lambda u: [f"v_accvgpr_read_b32 v{60+u}, a{100+u}"],
    "+4 accvgpr_read (idle AGPR)": lambda u: [f"v_accvgpr_read_b32 v{60+(4*u+j)%40}, a{100+(4*u+j)%20}" for j in range(4)],
    "+1 accvgpr_write (idle AGPR)": lambda u: [f"v_accvgpr_write_b32 a{100+u}, v{60+u}"],
    "+v_max writing NEXT B operand": lambda u: [f"v_max_f32 v{30+(u+1)%16}, 0, v{60+u}"],
    "+v_max writing B operand 2 ahead": lambda u: [f"v_max_f32 v{30+(u+2)%16}, 0, v{60+u}"],
    "+v_max writing CURRENT B operand (WAR)": lambda u: [f"v_max_f32 v{30+u}, 0, v{60+u}"],
    "+v_max writing PREVIOUS B operand": lambda u: [f"v_max_f32 v{30+(u-1)%16}, 0, v{60+u}"],
    "+1 ds_read_b64": lambda u: [f"ds_read_b64 v[{60+2*u}:{61+2*u}], v8"],
    "+2 ds_write_b32": lambda u: [f"ds_write_b32 v8, v{60+u}", f"ds_write_b32 v8, v{61+u} offset:256"],
    "+2 v_max +1 acc_read +1 ds_read": lambda u: [f"v_max_f32 v{100+u}, 0, v{100+u}", f"v_accvgpr_read_b32 v{80+u}, a{100+u}", f"ds_read_b64 v[{60+2*u}:{61+2*u}], v8", f"v_max_f32 v{116+u%8}, 0, v{116+u%8}"],
    "+1 v_exp_f32": lambda u: [f"v_exp_f32 v{60+u}, v{60+u}"],
    "+4 v_fma_f32": lambda u: [f"v_fma_f32 v{60+(4*u+j)%40}, v{60+(4*u+j)%40}, v9, v9" for j in range(4)],
    "+2 v_cmp/v_cndmask pairs": lambda u: [f"v_cmp_lt_f32 vcc, 0, v{60+u}", f"v_cndmask_b32 v{80+u}, 0, v{80+u}, vcc", f"v_cmp_lt_f32 vcc, 0, v{61+u}", f"v_cndmask_b32 v{100+u}, 0, v{100+u}, vcc"],
    "+s_nop 0 x4": lambda u: ["s_nop 0"] * 4,
}
print("#include <hip/hip_runtime.h>\n#include <cstdio>")
names = list(variants)
for vi, name in enumerate(names):
    init = ["v_mov_b32 v8, 0"] + [f"v_mov_b32 v{r}, %1" for r in range(9, 128)] + [f"v_accvgpr_write_b32 a{r}, %1" for r in range(0, 128)]
    asm = init + ["s_mov_b32 s20, %2", "s_memtime s[22:23]", "s_waitcnt lgkmcnt(0)", "1:"] + body(variants[name]) + \
          ["s_waitcnt lgkmcnt(0)", "s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 7", "s_nop 7", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)",
           "s_sub_u32 s22, s24, s22", "s_subb_u32 s23, s25, s23", "v_mov_b32 %0, s22"]
    clob = ", ".join([f'"v{r}"' for r in range(8, 128)] + [f'"a{r}"' for r in range(128)] + ['"s20"', '"s22"', '"s23"', '"s24"', '"s25"', '"scc"', '"vcc"', '"memory"'])
    text = "\\n\\t".join(asm)
    print(f'__global__ void __launch_bounds__(256, 1) k{vi}(unsigned* out, float x, int iters) {{\n  __shared__ float lds[1024]; lds[threadIdx.x] = x;\n  unsigned t;\n  asm volatile("{text}" : "=v"(t) : "v"(x + threadIdx.x), "s"(iters) : {clob});\n  if (threadIdx.x == 0 && blockIdx.x == 7) *out = t + (unsigned)lds[5] * 0;\n}}')
print("int main() {\n  unsigned* out; hipMalloc(&out, 4); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);\n  const int iters = 5000;")
for vi, name in enumerate(names):
    print(f'  {{ k{vi}<<<256, 256>>>(out, 1.f, iters); hipDeviceSynchronize(); hipEventRecord(e0); k{vi}<<<256, 256>>>(out, 1.f, iters); hipEventRecord(e1); hipEventSynchronize(e1);\n'
          f'    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned t; hipMemcpy(&t, out, 4, hipMemcpyDeviceToHost);\n'
          f'    printf("%-42s %7.2f ticks/MFMA\\n", "{name}", t / (iters * 16.0)); }}')
print("  return 0;\n}")
