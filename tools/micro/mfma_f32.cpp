// microbenchmark: v_mfma_f32_32x32x2_f32 issue rate, one wave per SIMD, N independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// chained: B operand of each MFMA is an accumulator register of the OTHER tile pair (as in the decoder)
__global__ void __launch_bounds__(256, 1) kchain(float* out, int iters, float a0) {
  f32x16 x[2], y[2];
  x[0] = 1; x[1] = 2; y[0] = 0; y[1] = 0;
  float a = a0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float b = x[s >> 4][s & 15];
      y[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, y[0], 0, 0, 0);
      y[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + 1, b, y[1], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float b = y[s >> 4][s & 15];
      x[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, x[0], 0, 0, 0);
      x[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + 1, b, x[1], 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += x[0][r] + x[1][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> void timeit(const char* name, F launch, double mfma_per_wave) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s %8.3f ms  %7.1f ns per MFMA per SIMD  (= %5.1f cycles @2.4GHz)  %6.1f TFLOP/s\n", name, ms, ms * 1e6 / mfma_per_wave,
         ms * 1e6 / mfma_per_wave * 2.4, 1024.0 * mfma_per_wave * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 2000;
  timeit("1 accumulator", [&] { k<1><<<256, 256>>>(out, iters, 1.f, 2.f); }, iters * 16.0 * 1);
  timeit("2 accumulators", [&] { k<2><<<256, 256>>>(out, iters, 1.f, 2.f); }, iters * 16.0 * 2);
  timeit("4 accumulators", [&] { k<4><<<256, 256>>>(out, iters, 1.f, 2.f); }, iters * 16.0 * 4);
  timeit("chained 2+2 (decoder)", [&] { kchain<<<256, 256>>>(out, iters / 4, 1.f); }, iters / 4 * 128.0);
  return 0;
}
