// microbenchmark: how fast can 1536 workgroups add their 2462-element sub-grid images (64-bit fixed point) into one dense
// 11 MB array with global 64-bit integer atomics (no return value)?  This is what the tiled encoder backward would do if it
// accumulated dG directly instead of writing per-item partial images that gather_partials sums (25 us per step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k(unsigned long long* dG, int64_t n, int per_item, int spread) {
  // item b adds per_item consecutive-ish elements starting at a pseudo-random, overlapping offset
  const int64_t base = ((int64_t)blockIdx.x * 2654435761u) % (n - (int64_t)per_item * spread);
  for (int e = threadIdx.x; e < per_item; e += 256)
    __hip_atomic_fetch_add(dG + base + (int64_t)e * spread, (unsigned long long)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
  const int64_t n = 720000 * 2;          // sum_l (N_l + 2)^2 * F at cfg2
  unsigned long long* d; hipMalloc(&d, n * 8); hipMemset(d, 0, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int items : {1536, 3072}) for (int per : {2462, 4924}) for (int spread : {1, 7}) {
    for (int w = 0; w < 3; ++w) k<<<items, 256>>>(d, n, per, spread);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) k<<<items, 256>>>(d, n, per, spread);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("items %5d x %5d elements, stride %d: %7.2f us per launch  (%.1f G atomics/s)\n", items, per, spread, ms / 20 * 1e3,
           (double)items * per / (ms / 20 * 1e-3) / 1e9);
  }
  return 0;
}
