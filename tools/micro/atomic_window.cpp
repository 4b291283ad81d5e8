// micro-benchmark (round 4): global float atomic adds when the rows a workgroup touches lie in a WINDOW of the table that fits one
// XCD's L2 (4 MiB), with and without XCD affinity (all workgroups that touch a window run on one XCD: blockIdx % 8 is the XCD).
// Question: is a partition-by-window pass + window-local atomics a way around the 20.6 G/s of random atomics (atomic_scope.cpp)?
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_window.cpp -o /tmp/atomic_window
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: every workgroup random rows of the whole table (baseline)
// MODE 1: window w of NW; the workgroups of a window all have blockIdx % 8 == w % 8 (one XCD), agent scope
// MODE 2: as 1, workgroup scope
// MODE 3: windows, but a window's workgroups are consecutive block indices (spread over all 8 XCDs), agent scope
template <int MODE, int F>
__global__ void __launch_bounds__(256) k(float* table, long long T, long long per_block, int log2_window_rows, int blocks_per_window) {
  const long long NW = T >> log2_window_rows;
  long long w;
  if (MODE == 0) w = 0;
  else if (MODE == 3) w = blockIdx.x / blocks_per_window;
  else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;                 // j-th workgroup of this XCD
    w = (long long)(j / blocks_per_window) * 8 + xcd;                    // windows xcd, xcd + 8, ... in launch order
  }
  if (MODE != 0 && w >= NW) return;
  const unsigned long long mask = MODE == 0 ? (unsigned long long)T - 1 : (1ull << log2_window_rows) - 1;
  const long long base = MODE == 0 ? 0 : (w << log2_window_rows);
  const long long i0 = (long long)blockIdx.x * per_block;
  for (long long i = i0 + threadIdx.x / F; i < i0 + per_block; i += 256 / F) {
    unsigned h = (unsigned)i * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const long long row = base + (long long)(h & mask);
    float* p = table + row * F + (threadIdx.x % F);
    if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(p, 1.0f);
  }
}

template <int MODE, int F> void run(const char* name, float* table, long long T, long long n, int log2_window_rows) {
  hipMemset(table, 0, T * F * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long long NW = T >> log2_window_rows;
  const long long per_block = 8192;                                     // row updates per workgroup
  long long blocks = n / per_block;
  int bpw = (int)(blocks / NW); if (bpw < 1) bpw = 1;
  if (MODE != 0) blocks = (long long)bpw * NW;
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE, F><<<(unsigned)blocks, 256>>>(table, T, per_block, log2_window_rows, bpw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<float> h(T * F);
  hipMemcpy(h.data(), table, T * F * sizeof(float), hipMemcpyDeviceToHost);
  double sum = 0; for (float v : h) sum += v;
  const double updates = (double)blocks * per_block;
  printf("%-64s F=%d window %6.2f MiB  %8.3f ms  %7.2f G row-updates/s  sum/expected %.6f\n", name, F,
         (double)(1ll << log2_window_rows) * F * 4 / 1048576.0, ms, updates / ms * 1e-6, sum / (updates * F * 3));
}

template <int F> void suite(long long T, long long n) {
  float* table;
  hipMalloc(&table, T * F * sizeof(float));
  printf("table %lld rows x %d floats = %.0f MiB, %lld row updates\n", T, F, (double)T * F * 4 / 1048576.0, n);
  run<0, F>("whole table, agent scope", table, T, n, 0);
  for (int lw = 16; lw <= 20; ++lw) {
    if ((1ll << lw) > T) break;
    run<1, F>("window, one XCD per window, agent scope", table, T, n, lw);
    run<2, F>("window, one XCD per window, workgroup scope", table, T, n, lw);
    run<3, F>("window, all XCDs on a window, agent scope", table, T, n, lw);
  }
  hipFree(table);
}

// the partition pass that would feed the windows: every lane writes one 16-byte item to bucket b (random), at the position the block
// reserved in that bucket — runs of (items per block / NB) items per (block, bucket)
__global__ void __launch_bounds__(256) scatter_items(float4* out, long long cap, int NB, int per_block) {
  const int run = per_block / NB < 1 ? 1 : per_block / NB;               // items of this block per bucket (expected)
  for (int i = threadIdx.x; i < per_block; i += 256) {
    unsigned h = (unsigned)(blockIdx.x * per_block + i) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const int b = (int)(h % (unsigned)NB);
    const long long pos = (long long)b * cap + (long long)blockIdx.x * run + (i / NB) % run;
    out[pos] = make_float4(1.f, 2.f, 3.f, (float)i);
  }
}
static void scatter_suite(long long n) {
  const int per_block = 8192;
  const long long blocks = n / per_block;
  for (int NB : {8, 64, 256, 1024, 4096}) {
    const int run = per_block / NB < 1 ? 1 : per_block / NB;
    const long long cap = blocks * run;
    float4* out; hipMalloc(&out, sizeof(float4) * cap * NB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      scatter_items<<<(unsigned)blocks, 256>>>(out, cap, NB, per_block);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("scatter %lld 16-byte items into %5d buckets (runs of %4d per block): %8.3f ms  %7.2f G items/s  %7.1f GB/s\n", n, NB, run, ms,
           n / ms * 1e-6, n * 16.0 / ms * 1e-6);
    hipFree(out);
  }
}

int main() {
  scatter_suite(1ll << 23);
  suite<2>(1ll << 22, 1ll << 22);      // cfg4 level: 32 MiB table, 4.2 M corner contributions
  suite<4>(1ll << 24, 1ll << 22);      // cfg5 level: 256 MiB of fp32 gradient, 4.2 M corner contributions
  return 0;
}
