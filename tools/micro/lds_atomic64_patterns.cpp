// micro-benchmark (round 3): cost of ds_add_u64 and ds_read_b64 per wave-instruction under the address patterns the tiled
// encoder kernels can produce, to choose the LDS layout of the privatised sub-grid images.
//   lanes = 4 pixels x 16 levels per wave (lane = 16 * pixel + level), 16 waves per workgroup, one workgroup per CU
//   pattern 0  lane-linear (slot = lane): the conflict-free reference
//   pattern 1  random slot in a 20 KB image (today's back-to-back sub-grids)
//   pattern 2  level-interleaved, 16 columns: slot = row * 16 + level, random row  (every 16-lane group hits 16 distinct columns)
//   pattern 3  level-interleaved, 32 columns: slot = row * 32 + 16 * (pixel & 1) + level
//   pattern 4  as 2, but the 4 pixels of a wave share the row for levels < 6 (coarse levels: same cell => same address)
//   pattern 5  as 3, same sharing
// -DF32 measures ds_add_f32 on 4-byte slots of the same patterns instead (would fp32 LDS atomics be cheaper than the fixed-point ones?)
// measured (MI355X): ds_add_u64 7.6 (lane-linear) / 10.1 (interleaved) / 15.1 (interleaved, shared coarse cells) cycles per
// wave instruction; ds_add_f32 193 cycles in EVERY pattern (3 cycles per lane: the LDS float atomic is serialised) -- 25 x slower.
// build: hipcc -O3 [-DF32 -munsafe-fp-atomics] --offload-arch=gfx950 tools/micro/lds_atomic64_patterns.cpp -o tools/micro/lds_atomic64_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifdef F32
typedef float slot_t;
#define ADDNAME "ds_add_f32"
#else
typedef unsigned long long slot_t;
#define ADDNAME "ds_add_u64"
#endif

constexpr int kSlots = 16384;      // 128 KB of 8-byte slots

template <int PAT>
__device__ __forceinline__ int slot_of(unsigned& r, int lane, int it) {
  const int level = lane & 15, pixel = lane >> 4;
  r = r * 1664525u + 1013904223u;
  const unsigned rnd = r >> 8;
  if (PAT == 0) return (lane + 64 * it) & (kSlots - 1);
  if (PAT == 1) return rnd % 2560;
  int row = rnd % 648;
  if (PAT == 4 || PAT == 5) {
    if (level < 6) row = (it * 7 + level) % 18;      // the four pixels of the wave in the same cell
  }
  if (PAT == 2 || PAT == 4) return row * 16 + level;
  return (row * 32 + 16 * (pixel & 1) + level) & (kSlots - 1);
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_add(float* out, int iters) {
  extern __shared__ slot_t lds[];
  for (int i = threadIdx.x; i < kSlots; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    const int a = slot_of<PAT>(r, lane, it);
    #ifdef F32
    __hip_atomic_fetch_add(&lds[a], __uint_as_float((r & 0x007fffffu) | 0x3f800000u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    atomicAdd(&lds[a], (unsigned long long)(r | 1));
#endif
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(lds[0] + lds[1]);
}

template <int PAT>
__global__ void __launch_bounds__(1024) k_read(float* out, int iters) {
  extern __shared__ slot_t lds[];
  for (int i = threadIdx.x; i < kSlots; i += blockDim.x) lds[i] = i;
  __syncthreads();
  unsigned r = threadIdx.x * 2654435761u + blockIdx.x;
  const int lane = threadIdx.x & 63;
  slot_t acc = 0;
  for (int it = 0; it < iters; it += 4) {
    int a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = slot_of<PAT>(r, lane, it + u);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += lds[a[u]];
  }
  if (acc == (slot_t)0x1234567) out[blockIdx.x] = 1.f;
}

template <int PAT> void run(const char* name) {
  float* out;
  hipMalloc(&out, 4096 * 4);
  const int iters = 8192, blocks = 256;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_add<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * sizeof(slot_t));
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_read<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, kSlots * sizeof(slot_t));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms[2];
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) k_add<PAT><<<blocks, 1024, kSlots * sizeof(slot_t)>>>(out, iters);
      else k_read<PAT><<<blocks, 1024, kSlots * sizeof(slot_t)>>>(out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[which], e0, e1);
    }
  }
  // per CU: 16 waves x iters wave-instructions
  const double winstr = 16.0 * iters;
  printf("%-44s " ADDNAME " %7.2f cycles/wave-instr   ds_read_b64 %7.2f (incl. ~6 VALU of address generation per instr, 16 waves/CU)\n", name,
         ms[0] * 1e-3 * 2.4e9 / winstr, ms[1] * 1e-3 * 2.4e9 / winstr);
  hipFree(out);
}

int main() {
  run<0>("0 lane-linear");
  run<1>("1 random in 20 KB (today)");
  run<2>("2 level-interleaved, 16 columns");
  run<3>("3 level-interleaved, 32 columns");
  run<4>("4 as 2, coarse levels share addresses");
  run<5>("5 as 3, coarse levels share addresses");
  return 0;
}
