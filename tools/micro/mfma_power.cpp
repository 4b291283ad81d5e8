// microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate with constant vs random operand data (switching activity drives
// power, power drives the clock).  Prints wall-clock TFLOP/s and s_memtime ticks per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void __launch_bounds__(256, 1) k(const float* __restrict__ in, float* out, unsigned long long* ticks, int iters) {
  f32x16 acc[2];
  acc[0] = 0; acc[1] = 0;
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 4095]; b[i] = in[(blockIdx.x * 64 + threadIdx.x * 16 + i + 1) & 4095]; }
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[u], a[u], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { asm volatile("" : "+v"(a[u]), "+v"(b[u])); }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) *ticks = t1 - t0;
}
int main() {
  float *in, *out; unsigned long long* ticks;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 8);
  std::vector<float> h(4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < 4096; ++i) h[i] = mode == 0 ? 1.0f : (mode == 1 ? (float)rand() / RAND_MAX - 0.5f : ((rand() & 1) ? 1.f : -1.f) * (float)rand());
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int iters : {500, 20000, 200000}) {
      k<<<256, 256>>>(in, out, ticks, iters); hipDeviceSynchronize();
      hipEventRecord(e0); k<<<256, 256>>>(in, out, ticks, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      const double n = iters * 32.0;
      printf("data=%-8s iters=%7d  %9.3f ms  %6.1f TFLOP/s  %6.2f ticks/MFMA  tick rate %.3f GHz\n", mode == 0 ? "const" : (mode == 1 ? "uniform" : "wide"), iters, ms,
             1024.0 * n * 4096 / (ms * 1e-3) / 1e12, t / n, t / (ms * 1e6));
    }
  }
  return 0;
}
