// micro-benchmark (round 3): global float atomic adds into a table larger than one XCD's L2, by memory scope.
//   agent scope (what atomicAdd emits) vs workgroup scope (executed in the issuing XCD's L2) — the latter is only CORRECT if every
//   address is touched from one XCD during the kernel; here each workgroup reads its XCC id and skips rows of other partitions.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_scope.cpp -o /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int MODE>   // 0: agent scope, all rows; 1: workgroup scope, rows of my XCD only (8x redundant index work); 2: agent scope, partitioned too
__global__ void k(float* table, long long T, long long n, int* xcc_hist) {
  const unsigned me = xcc_id();
  if (threadIdx.x == 0) atomicAdd(xcc_hist + (me & 7), 1);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned h = (unsigned)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const long long row = (long long)(h % (unsigned long long)T);
    float* p = table + row * 2;
    if (MODE == 0) { atomicAdd(p, 1.0f); atomicAdd(p + 1, 1.0f); }
    else {
      if (((row >> 4) & 7) != (me & 7)) continue;
      if (MODE == 1) {
        __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(p + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else { atomicAdd(p, 1.0f); atomicAdd(p + 1, 1.0f); }
    }
  }
}
template <int MODE> void run(const char* name, float* table, long long T, long long n, int* hist, int passes) {
  hipMemset(table, 0, T * 2 * sizeof(float));
  hipMemset(hist, 0, 8 * sizeof(int));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int ps = 0; ps < passes; ++ps) k<MODE><<<grid, 256>>>(table, T, n, hist);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // check: total of the table must equal 2 n per pass (x 2 reps) when MODE covers all rows once
  std::vector<float> h(T * 2);
  hipMemcpy(h.data(), table, T * 2 * sizeof(float), hipMemcpyDeviceToHost);
  double sum = 0; for (float v : h) sum += v;
  int hh[8]; hipMemcpy(hh, hist, sizeof(hh), hipMemcpyDeviceToHost);
  printf("%-44s %8.3f ms per pass   sum/expected %.6f   workgroups per XCC id:", name, ms / passes, sum / (2.0 * n * passes * 2));
  for (int i = 0; i < 8; ++i) printf(" %d", hh[i]);
  printf("\n");
}
int main() {
  const long long T = 1ll << 24, n = 1ll << 24;     // 128 MiB table (F = 2), 16.7 M row updates
  float* table; int* hist;
  hipMalloc(&table, T * 2 * sizeof(float)); hipMalloc(&hist, 8 * sizeof(int));
  run<0>("agent scope, every workgroup every row", table, T, n, hist, 1);
  run<2>("agent scope, partitioned by XCC (8 passes)", table, T, n, hist, 1);
  run<1>("workgroup scope, partitioned by XCC", table, T, n, hist, 1);
  return 0;
}
