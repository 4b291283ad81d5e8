// Collision statistics: the per-level number of DISTINCT table slots among the indices a batch used,
// which GeneralNeuralGaugeFields.calc_hash_collisions (reference models.py:568-619) turns into
// (#vertices of the level) - (#distinct slots).  The reference calls torch.unique per level (and per top-K rank): a sort of
// 4 P int64 values and a host synchronisation each.  Here every (rank, level) owns a T-bit map: one pass sets bits (a
// plain read first — after the first few thousand entries almost every bit is already set, so few atomics are issued),
// one pass counts them.
#include "gngf_common.h"

namespace gngf {

__global__ void __launch_bounds__(256)
slot_mark_kernel(const int64_t* __restrict__ idx, int64_t n, int L, int V, int K, int64_t T, uint32_t* __restrict__ bitmap,
                 int64_t words) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int k = (int)(e % K);
    const int l = (int)((e / ((int64_t)K * V)) % L);
    const int64_t slot = idx[e];
    if (slot < 0 || slot >= T) continue;
    uint32_t* w = bitmap + ((int64_t)k * L + l) * words + (slot >> 5);
    const uint32_t bit = 1u << (slot & 31);
    if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & bit)) atomicOr(w, bit);   // a stale 0 only costs a redundant atomic
  }
}

__global__ void __launch_bounds__(256)
slot_count_kernel(const uint32_t* __restrict__ bitmap, int64_t words, int32_t* __restrict__ counts) {
  __shared__ int red[4];
  const uint32_t* row = bitmap + (int64_t)blockIdx.y * words;
  int c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) c += __popc(row[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(counts + blockIdx.y, (red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace gngf

using namespace gngf;

extern "C" int64_t gngf_slot_bitmap_words(int L, int K, int64_t T) { return (int64_t)L * (K > 0 ? K : 1) * ((T + 31) / 32); }

// counts (K,L) int32 = number of distinct values in [0,T) of indices[:, l, :, k] for indices (P,L,V,K) int64, contiguous
// (K = 1 for the hash source's (P,L,V)).  bitmap: gngf_slot_bitmap_words(L, K, T) 32-bit words of workspace.
extern "C" int gngf_distinct_slot_counts(const int64_t* indices, int64_t P, int L, int V, int K, int64_t T, uint32_t* bitmap,
                                         int32_t* counts, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && V > 0 && K > 0 && T > 0 && bitmap && counts && (P == 0 || indices));
  hipStream_t s = as_stream(stream);
  const int64_t words = (T + 31) / 32;
  hipError_t e = hipMemsetAsync(bitmap, 0, sizeof(uint32_t) * (size_t)(words * L * K), s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)(L * K), s);
  if (e != hipSuccess) return (int)e;
  const int64_t n = P * L * V * K;
  if (n > 0) {
    const int64_t want = ceil_div(n, 256 * 8);
    slot_mark_kernel<<<dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, s>>>(indices, n, L, V, K, T, bitmap, words);
  }
  const int64_t cw = ceil_div(words, 256 * 4);
  slot_count_kernel<<<dim3((unsigned)(cw > 64 ? 64 : (cw < 1 ? 1 : cw)), (unsigned)(L * K)), dim3(256), 0, s>>>(bitmap, words, counts);
  GNGF_RETURN_LAUNCH();
}
