// microbenchmark: two waves per SIMD.  Waves 0-3 of a 512-thread workgroup issue only MFMAs, waves 4-7 only VALU (or LDS)
// work: do the two streams overlap?  Prints ticks per MFMA of the MFMA waves and ticks per instruction of the others.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void __launch_bounds__(512, 1) k(unsigned* out, float x, int iters, int mode) {
  __shared__ float lds[2048];
  lds[threadIdx.x] = x;
  __syncthreads();
  unsigned t = 0;
  const int wave = threadIdx.x >> 6;
  if (wave < 4 || mode == 3) {
    if (mode == 2) return;                               // VALU waves alone
    asm volatile(
        "s_mov_b32 s20, %2\n\ts_memtime s[22:23]\n\ts_waitcnt lgkmcnt(0)\n1:\n\t"
        REP16("v_mfma_f32_32x32x2_f32 a[0:15], %1, %1, a[0:15]\n\tv_mfma_f32_32x32x2_f32 a[16:31], %1, %1, a[16:31]\n\t")
        "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_nop 7\n\ts_nop 7\n\ts_memtime s[24:25]\n\ts_waitcnt lgkmcnt(0)\n\t"
        "s_sub_u32 s22, s24, s22\n\tv_mov_b32 %0, s22"
        : "=v"(t) : "v"(x), "s"(iters)
        : "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","s20","s22","s23","s24","s25","scc");
    if (threadIdx.x == 0 && blockIdx.x == 7) out[0] = t;
    if (threadIdx.x == 256 && blockIdx.x == 7 && mode == 3) out[1] = t;
  } else {
    if (mode == 1) return;                               // MFMA waves alone
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
    asm volatile(
        "s_mov_b32 s20, %5\n\ts_memtime s[22:23]\n\ts_waitcnt lgkmcnt(0)\n1:\n\t"
        REP16("v_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\tv_max_f32 %4, 0, %4\n\t")
        "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\ts_memtime s[24:25]\n\ts_waitcnt lgkmcnt(0)\n\t"
        "s_sub_u32 s22, s24, s22\n\tv_mov_b32 %0, s22"
        : "=v"(t), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(iters * 8) : "s20","s22","s23","s24","s25","scc");
    if (threadIdx.x == 256 && blockIdx.x == 7) out[1] = t + (unsigned)(v0 + v1 + v2 + v3) * 0;
  }
}
int main() {
  unsigned* out; hipMalloc(&out, 8);
  const int iters = 4000;
  const char* names[] = {"MFMA waves + VALU waves", "MFMA waves alone", "VALU waves alone", "MFMA in all 8 waves"};
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(out, 0, 8);
    k<<<256, 512>>>(out, 1.f, iters, mode); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<<<256, 512>>>(out, 1.f, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("[%.3f ms; MFMA-only ideal %.3f ms] ", ms, iters * 32.0 * 64 / 2.4e6);
    unsigned t[2]; hipMemcpy(t, out, 8, hipMemcpyDeviceToHost);
    printf("%-26s  %7.2f ticks/MFMA (per wave)   %6.2f ticks per VALU instruction\n", names[mode], t[0] / (iters * 32.0), t[1] / (iters * 8 * 64.0));
  }
  return 0;
}
