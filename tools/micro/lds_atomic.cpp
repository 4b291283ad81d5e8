// microbenchmark: LDS atomic add throughput (float vs uint), conflict-free vs random vs same-address
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE, bool FLT>
__global__ void k(float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    int a;
    if (MODE == 0) a = (threadIdx.x + it * 64) & 4095;            // conflict-free, lane-linear
    else if (MODE == 1) { r = r * 1664525u + 1013904223u; a = (r >> 10) & 4095; }   // random
    else if (MODE == 2) a = (it & 15);                               // all lanes same address
    else a = ((threadIdx.x & 3) + it * 4) & 4095;                    // 16 lanes per address (4 distinct / wave)
    if (FLT) atomicAdd(&lds[a], 1.0f); else atomicAdd(reinterpret_cast<unsigned*>(&lds[a]), 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[1];
}
template <int MODE>
__global__ void k64(float* out, int iters) {
  __shared__ unsigned long long lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    int a;
    if (MODE == 0) a = (threadIdx.x + it * 64) & 4095;
    else if (MODE == 1) { r = r * 1664525u + 1013904223u; a = (r >> 10) & 4095; }
    else a = ((threadIdx.x & 3) + it * 4) & 4095;
    atomicAdd(&lds[a], (unsigned long long)(r | 1));
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(lds[0] + lds[1]);
}
template <int MODE>
__global__ void kf64(float* out, int iters) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
  for (int it = 0; it < iters; ++it) {
    int a;
    if (MODE == 0) a = (threadIdx.x + it * 64) & 4095;
    else if (MODE == 1) { r = r * 1664525u + 1013904223u; a = (r >> 10) & 4095; }
    else a = ((threadIdx.x & 3) + it * 4) & 4095;
    atomicAdd(&lds[a], 1.0);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(lds[0] + lds[1]);
}
template <int MODE> void runf64(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  const int iters = 4096, blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kf64<MODE><<<blocks, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  kf64<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.3f ms  -> %6.1f cycles per wave-instr per CU\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters / 256));
  hipFree(out);
}
template <int MODE> void run64(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  const int iters = 4096, blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k64<MODE><<<blocks, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  k64<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.3f ms  -> %6.1f cycles per wave-instr per CU\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters / 256));
  hipFree(out);
}
template <int MODE, bool FLT> void run(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  const int iters = 4096, blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, FLT><<<blocks, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, FLT><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)blocks * 4 * iters;            // per chip
  double cyc_per_cu = ms * 1e-3 * 2.4e9;                      // cycles elapsed
  printf("%-28s %8.3f ms  -> %6.1f cycles per wave-instr per CU\n", name, ms, cyc_per_cu / (wave_instr / 256));
  hipFree(out);
}
int main() {
  run<0, true>("f32 conflict-free"); run<0, false>("u32 conflict-free");
  run<1, true>("f32 random"); run<1, false>("u32 random");
  run<3, true>("f32 16 lanes/address"); run<3, false>("u32 16 lanes/address");
  run<2, true>("f32 same address"); run<2, false>("u32 same address");
  run64<0>("u64 conflict-free"); run64<1>("u64 random"); run64<3>("u64 16 lanes/address");
  runf64<0>("f64 conflict-free"); runf64<1>("f64 random"); runf64<3>("f64 16 lanes/address");
  return 0;
}
