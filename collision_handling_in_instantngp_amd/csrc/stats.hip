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

// ---- the same statistic WITHOUT the (P,L,4[,K]) index tensor (2 GiB per step at the headline shape).  Both index sources
// depend on (level, vertex) only, so the distinct slots of a batch are the slots of its TOUCHED vertices:
//   pass 1  touched[l][vid] |= 1 for the four corners of every (pixel, level)        (vid = gy * vstride + gx)
//   pass 2  every touched (level, vertex) sets the bits of its slot(s): _fast_hash(gx, gy), or the K slots of the
//           per-vertex table — into slot maps that ACCUMULATE over the batches of an epoch (a trainable HPD changes the
//           table from batch to batch: each batch marks with its own), counted once at the end (slot_count_kernel).
__global__ void __launch_bounds__(256)
touched_mark_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, int64_t total /* P*L */, int L, int vstride,
                    int64_t NV, uint32_t* __restrict__ touched, int64_t vwords) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t p = gid / L;
  const int l = (int)(gid - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
  uint32_t* row = touched + (int64_t)l * vwords;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    // the two corners of a grid row are neighbouring bits: one word (one read, at most one atomic) unless they straddle a word
    const int64_t v0 = (int64_t)(cell.gy + dy) * vstride + cell.gx;
    if (cell.gx < 0 || cell.gx + 1 >= vstride || v0 < 0 || v0 + 1 >= NV) continue;        // outside the table: not a vertex of it
    const int64_t w0 = v0 >> 5, w1 = (v0 + 1) >> 5;
    const uint32_t b0 = 1u << (v0 & 31), b1 = 1u << ((v0 + 1) & 31);
    if (w0 == w1) {
      const uint32_t m = b0 | b1;
      if ((__atomic_load_n(row + w0, __ATOMIC_RELAXED) & m) != m) atomicOr(row + w0, m);
    } else {
      if (!(__atomic_load_n(row + w0, __ATOMIC_RELAXED) & b0)) atomicOr(row + w0, b0);
      if (!(__atomic_load_n(row + w1, __ATOMIC_RELAXED) & b1)) atomicOr(row + w1, b1);
    }
  }
}

__global__ void __launch_bounds__(256)
touched_slots_kernel(const uint32_t* __restrict__ touched, int64_t vwords, const int32_t* __restrict__ vert_idx, int L, int K,
                     int64_t T, bool pow2, int vstride, int64_t NV, uint32_t* __restrict__ bitmap, int64_t words) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;              // over L * vwords words of the touched map
  if (e >= (int64_t)L * vwords) return;
  uint32_t m = touched[e];
  if (!m) return;
  const int l = (int)(e / vwords);
  const int64_t base = (e - (int64_t)l * vwords) << 5;
  while (m) {
    const int b = __ffs(m) - 1;
    m &= m - 1;
    const int64_t vid = base + b;
    if (vid >= NV) break;
    if (vert_idx) {
      for (int k = 0; k < K; ++k) {
        const int64_t slot = vert_idx[vid * K + k];
        if (slot < 0 || slot >= T) continue;
        uint32_t* w = bitmap + ((int64_t)k * L + l) * words + (slot >> 5);
        const uint32_t bit = 1u << (slot & 31);
        if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & bit)) atomicOr(w, bit);
      }
    } else {
      const int gy = (int)(vid / vstride), gx = (int)(vid - (int64_t)gy * vstride);
      const int64_t slot = spatial_hash(gx, gy, T, pow2);
      uint32_t* w = bitmap + (int64_t)l * words + (slot >> 5);
      const uint32_t bit = 1u << (slot & 31);
      if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & bit)) atomicOr(w, bit);
    }
  }
}

}  // namespace gngf

using namespace gngf;

// Marks, into `bitmap` (gngf_slot_bitmap_words(L, K, T) words; ACCUMULATED — the caller clears it once per epoch), the table
// slots the batch `xy` uses: vert_idx (NV,K) int32 per-vertex table with vid = gy * vstride + gx, or NULL for the spatial hash
// (K = 1; vstride >= N_max + 2, NV = vstride^2).  touched: workspace of L * ceil(NV / 32) words (cleared here).
// = what gngf_distinct_slot_counts sees in the (P,L,4[,K]) index tensor of the same batch (models.py:568-619), without it.
extern "C" int gngf_mark_batch_slots(const float* xy, const int32_t* n_ls, int64_t P, int L, const int32_t* vert_idx, int K,
                                     int64_t T, int vstride, int64_t NV, uint32_t* touched, uint32_t* bitmap, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && K > 0 && K <= GNGF_MAX_TOPK && T > 0 && vstride > 0 && NV > 0 && touched && bitmap);
  GNGF_CHECK_ARG(vert_idx || K == 1);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(xy && n_ls);
  hipStream_t s = as_stream(stream);
  const int64_t vwords = (NV + 31) / 32, words = (T + 31) / 32;
  hipError_t e = zero_async(touched, sizeof(uint32_t) * (size_t)(vwords * L), s);
  if (e != hipSuccess) return (int)e;
  const int64_t total = P * L;
  touched_mark_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s>>>(reinterpret_cast<const float2*>(xy), n_ls, total, L,
                                                                                vstride, NV, touched, vwords);
  touched_slots_kernel<<<dim3((unsigned)ceil_div((int64_t)L * vwords, 256)), dim3(256), 0, s>>>(touched, vwords, vert_idx, L, K, T,
                                                                                               (T & (T - 1)) == 0, vstride, NV, bitmap, words);
  GNGF_RETURN_LAUNCH();
}

// counts (K,L) int32 = set bits of each (rank, level) map of `bitmap` (as the second half of gngf_distinct_slot_counts)
extern "C" int gngf_count_slot_bits(const uint32_t* bitmap, int L, int K, int64_t T, int32_t* counts, void* stream) {
  GNGF_CHECK_ARG(L > 0 && K > 0 && T > 0 && bitmap && counts);
  hipStream_t s = as_stream(stream);
  const int64_t words = (T + 31) / 32;
  hipError_t e = zero_async(counts, sizeof(int32_t) * (size_t)(L * K), s);
  if (e != hipSuccess) return (int)e;
  const int64_t cw = ceil_div(words, 256 * 4);
  slot_count_kernel<<<dim3((unsigned)(cw > 64 ? 64 : (cw < 1 ? 1 : cw)), (unsigned)(L * K)), dim3(256), 0, s>>>(bitmap, words, counts);
  GNGF_RETURN_LAUNCH();
}

extern "C" int64_t gngf_slot_bitmap_words(int L, int K, int64_t T) { return (int64_t)L * (K > 0 ? K : 1) * ((T + 31) / 32); }

// counts (K,L) int32 = number of distinct values in [0,T) of indices[:, l, :, k] for indices (P,L,V,K) int64, contiguous
// (K = 1 for the hash source's (P,L,V)).  bitmap: gngf_slot_bitmap_words(L, K, T) 32-bit words of workspace.
extern "C" int gngf_distinct_slot_counts(const int64_t* indices, int64_t P, int L, int V, int K, int64_t T, uint32_t* bitmap,
                                         int32_t* counts, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && V > 0 && K > 0 && T > 0 && bitmap && counts && (P == 0 || indices));
  hipStream_t s = as_stream(stream);
  const int64_t words = (T + 31) / 32;
  hipError_t e = zero_async(bitmap, sizeof(uint32_t) * (size_t)(words * L * K), s);
  if (e != hipSuccess) return (int)e;
  e = zero_async(counts, sizeof(int32_t) * (size_t)(L * K), s);
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
