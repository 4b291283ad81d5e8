"""Generates mfma_bf16_mix.cpp: issue cost of v_mfma_f32_32x32x16_bf16 with VALU instructions interleaved, one and two
waves per SIMD (is the VALU work of a 3-way bf16 split hidden under the matrix pipe?).
Usage: python gen_mfma_bf16_mix.py > mfma_bf16_mix.cpp"""
def body(n_valu, mfma="bf16"):
    out = []
    for u in range(16):
        acc = f"a[{16 * (u % 2)}:{16 * (u % 2) + 15}]"
        if mfma == "bf16":
            out.append(f"v_mfma_f32_32x32x16_bf16 {acc}, v[{32 + 4 * (u % 4)}:{35 + 4 * (u % 4)}], v[{48 + 4 * (u % 4)}:{51 + 4 * (u % 4)}], {acc}")
        elif mfma == "f32":
            out.append(f"v_mfma_f32_32x32x2_f32 {acc}, v{32 + u}, v{48 + u}, {acc}")
        for j in range(n_valu):
            r = 70 + (u * n_valu + j) % 50
            out.append([f"v_and_b32 v{r}, 0xffff0000, v{r}", f"v_sub_f32 v{r}, v{r}, v9", f"v_perm_b32 v{r}, v{r}, v9, v10"][j % 3])
    return out
variants = [("f32 MFMA only", 0, "f32"), ("bf16 MFMA only", 0, "bf16"), ("bf16 MFMA + 1 VALU", 1, "bf16"), ("bf16 MFMA + 2 VALU", 2, "bf16"),
            ("bf16 MFMA + 4 VALU", 4, "bf16"), ("bf16 MFMA + 8 VALU", 8, "bf16"), ("bf16 MFMA + 12 VALU", 12, "bf16"), ("8 VALU only", 8, "none")]
import sys
RANDOM = len(sys.argv) > 1
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#define RANDOM_ARG " + ("__uint_as_float(threadIdx.x * 2246822519u + blockIdx.x * 3266489917u + 374761393u)" if RANDOM else "(x + threadIdx.x)"))
clob = ", ".join([f'"v{r}"' for r in range(8, 128)] + [f'"a{r}"' for r in range(64)] + ['"s20"', '"s26"', '"s22"', '"s23"', '"s24"', '"s25"', '"scc"', '"vcc"', '"memory"'])
for vi, (name, n, kind) in enumerate(variants):
    init = ["v_mov_b32 v8, 0"] + [f"v_mov_b32 v{r}, %1" for r in range(9, 128)] + [f"v_accvgpr_write_b32 a{r}, %1" for r in range(0, 64)]
    if RANDOM:
        init += ["v_mov_b32 v9, %1"] + sum([[f"s_mov_b32 s26, {(2654435761 + 2 * r * 40503) & 0xffffffff}", f"v_mul_lo_u32 v{r}, v9, s26", f"v_xor_b32 v{r}, v{r}, v{r-1}" if r > 32 else f"v_mov_b32 v{r}, v{r}",
                                              f"v_and_b32 v{r}, 0x807f807f, v{r}", f"v_or_b32 v{r}, 0x3f003f00, v{r}"] for r in range(32, 64)], [])
    asm = init + ["s_mov_b32 s20, %2", "s_memtime s[22:23]", "s_waitcnt lgkmcnt(0)", "1:"] + body(n, kind) + \
          ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 7", "s_nop 7", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)",
           "s_sub_u32 s22, s24, s22", "s_subb_u32 s23, s25, s23", "v_mov_b32 %0, s22"]
    text = "\\n\\t".join(asm)
    for waves in (4, 8):
        print(f'__global__ void __launch_bounds__({64 * waves}, 1) k{vi}_{waves}(unsigned* out, float x, int iters) {{\n  unsigned t;\n'
              f'  asm volatile("{text}" : "=v"(t) : "v"(RANDOM_ARG), "s"(iters) : {clob});\n  if (threadIdx.x == 0 && blockIdx.x == 7) *out = t;\n}}')
# two waves per SIMD, waves 0-3 MFMA only, waves 4-7 VALU only
init = ["v_mov_b32 v8, 0"] + [f"v_mov_b32 v{r}, %1" for r in range(9, 128)] + [f"v_accvgpr_write_b32 a{r}, %1" for r in range(0, 64)]
def wrap(b):
    return "\\n\\t".join(init + ["s_mov_b32 s20, %2", "s_memtime s[22:23]", "s_waitcnt lgkmcnt(0)", "1:"] + b +
          ["s_sub_u32 s20, s20, 1", "s_cmp_lg_u32 s20, 0", "s_cbranch_scc1 1b", "s_nop 7", "s_nop 7", "s_memtime s[24:25]", "s_waitcnt lgkmcnt(0)",
           "s_sub_u32 s22, s24, s22", "s_subb_u32 s23, s25, s23", "v_mov_b32 %0, s22"])
valu_only = [l for l in body(8, "none")]
print(f'__global__ void __launch_bounds__(512, 1) ksplit(unsigned* out, float x, int iters, int mode) {{\n  unsigned t = 0;\n  if ((threadIdx.x >> 6) < 4) {{\n    if (mode == 2) return;\n'
      f'    asm volatile("{wrap(body(0, "bf16"))}" : "=v"(t) : "v"(RANDOM_ARG), "s"(iters) : {clob});\n    if (threadIdx.x == 0 && blockIdx.x == 7) out[0] = t;\n  }} else {{\n    if (mode == 1) return;\n'
      f'    asm volatile("{wrap(valu_only)}" : "=v"(t) : "v"(RANDOM_ARG), "s"(iters) : {clob});\n    if (threadIdx.x == 256 && blockIdx.x == 7) out[1] = t;\n  }}\n}}')
print("#undef RANDOM_ARG")
print("int main() {\n  unsigned* out; hipMalloc(&out, 8); hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);\n  const int iters = 5000;")
for vi, (name, n, kind) in enumerate(variants):
    for waves in (4, 8):
        print(f'  {{ k{vi}_{waves}<<<256, {64 * waves}>>>(out, 1.f, iters); hipDeviceSynchronize(); hipEventRecord(e0); k{vi}_{waves}<<<256, {64 * waves}>>>(out, 1.f, iters); hipEventRecord(e1); hipEventSynchronize(e1);\n'
              f'    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned t; hipMemcpy(&t, out, 4, hipMemcpyDeviceToHost);\n'
              f'    printf("%-24s %d waves/SIMD  %8.2f ticks per MFMA slot (per wave)  %8.3f ms  -> %7.2f ns per slot per SIMD\\n", "{name}", {waves // 4}, t / (iters * 16.0), ms, ms * 1e6 / (iters * 16.0 * {waves // 4})); }}')
print('  const char* names[] = {"MFMA waves + VALU waves", "MFMA waves alone", "VALU waves alone"};\n  for (int mode = 0; mode < 3; ++mode) {\n    hipMemset(out, 0, 8); ksplit<<<256, 512>>>(out, 1.f, iters, mode); hipDeviceSynchronize();\n'
      '    hipEventRecord(e0); ksplit<<<256, 512>>>(out, 1.f, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);\n    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned t[2]; hipMemcpy(t, out, 8, hipMemcpyDeviceToHost);\n'
      '    printf("%-26s %8.3f ms   %7.2f ticks/MFMA   %6.2f ticks per 8 VALU\\n", names[mode], ms, t[0] / (iters * 16.0), t[1] / (iters * 16.0)); }')
print("  return 0;\n}")
