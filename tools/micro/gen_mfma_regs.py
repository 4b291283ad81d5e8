"""Generates tools/micro/mfma_regs.cpp: v_mfma_f32_32x32x2_f32 issue rate as a function of WHICH registers hold the operands
(explicit register numbers through inline asm).  Usage: python gen_mfma_regs.py > mfma_regs.cpp"""
variants = {
    # name: (n_acc, acc_kind, list of (A, B) operand register names for the 16 MFMAs of the body)
    "same a/b regs, 2 acc (AGPR)": (2, "a", [("v10", "v11")] * 16),
    "distinct consecutive, 2 acc": (2, "a", [(f"v{10+u}", f"v{30+u}") for u in range(16)]),
    "distinct, A,B adjacent pair": (2, "a", [(f"v{10+2*u}", f"v{11+2*u}") for u in range(16)]),
    "A == B register": (2, "a", [(f"v{10+u}", f"v{10+u}") for u in range(16)]),
    "A,B same bank (mod 4)": (2, "a", [(f"v{8+4*(u%8)}", f"v{48+4*(u%8)}") for u in range(16)]),
    "A bank0, B bank1": (2, "a", [(f"v{8+4*(u%8)}", f"v{49+4*(u%8)}") for u in range(16)]),
    "A bank0, B bank2": (2, "a", [(f"v{8+4*(u%8)}", f"v{50+4*(u%8)}") for u in range(16)]),
    "A in AGPR, B VGPR distinct": (2, "a", [(f"a{40+u}", f"v{30+u}") for u in range(16)]),
    "A,B both AGPR distinct": (2, "a", [(f"a{40+u}", f"a{60+u}") for u in range(16)]),
    "distinct, 2 acc in VGPR": (2, "v", [(f"v{10+u}", f"v{30+u}") for u in range(16)]),
    "distinct, 4 acc (AGPR)": (4, "a", [(f"v{10+u}", f"v{30+u}") for u in range(16)]),
    "distinct, 1 acc (AGPR)": (1, "a", [(f"v{10+u}", f"v{30+u}") for u in range(16)]),
    "B = other acc reg (chained)": (2, "a", [(f"v{10+u}", f"a{96+u}") for u in range(16)]),
}
print("#include <hip/hip_runtime.h>\n#include <cstdio>")
names = list(variants)
for vi, name in enumerate(names):
    nacc, kind, ops = variants[name]
    base = 0 if kind == "a" else 64
    body = []
    for u, (A, B) in enumerate(ops):
        acc = f"{kind}[{base + 16 * (u % nacc)}:{base + 16 * (u % nacc) + 15}]"
        body.append(f"v_mfma_f32_32x32x2_f32 {acc}, {A}, {B}, {acc}")
    init = [f"v_mov_b32 v{r}, %1" for r in range(8, 128)] + [f"v_accvgpr_write_b32 a{r}, %1" for r in range(0, 128)]
    asm = init + ["s_mov_b32 s20, %2", "s_memtime s[22:23]", "s_waitcnt lgkmcnt(0)", "1:"] + body + \
          ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 7", "s_nop 7", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)",
           "s_sub_u32 s22, s24, s22", "s_subb_u32 s23, s25, s23", "v_mov_b32 %0, s22"]
    clob = ", ".join([f'"v{r}"' for r in range(8, 128)] + [f'"a{r}"' for r in range(128)] + ['"s20"', '"s22"', '"s23"', '"s24"', '"s25"', '"scc"'])
    text = "\\n\\t".join(asm)
    print(f'__global__ void __launch_bounds__(256, 1) k{vi}(unsigned* out, float x, int iters) {{\n  unsigned t;\n  asm volatile("{text}" : "=v"(t) : "v"(x + threadIdx.x), "s"(iters) : {clob});\n  if (threadIdx.x == 0 && blockIdx.x == 7) *out = t;\n}}')
print("int main() {\n  unsigned* out; hipMalloc(&out, 4); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);\n  const int iters = 20000;")
for vi, name in enumerate(names):
    print(f'  {{ k{vi}<<<256, 256>>>(out, 1.f, iters); hipDeviceSynchronize(); hipEventRecord(e0); k{vi}<<<256, 256>>>(out, 1.f, iters); hipEventRecord(e1); hipEventSynchronize(e1);\n'
          f'    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned t; hipMemcpy(&t, out, 4, hipMemcpyDeviceToHost);\n'
          f'    printf("%-34s %7.2f ticks/MFMA  %6.1f TFLOP/s\\n", "{name}", t / (iters * 16.0), 1024.0 * iters * 16 * 4096 / (ms * 1e-3) / 1e12); }}')
print("  return 0;\n}")
