// Backward of the DIRECT levels (the fine levels whose sub-grids do not fit the LDS: two at the 4096^2 shape, four at the 8192^2
// one), spatial-hash source — the scatter-add of /root/reference/models.py:382-392's autograd backward (index_put_ with
// accumulate=True of the gathered rows' gradients) WITHOUT one memory-side atomic per contribution.
//
// Why: this chip retires 20.6 G atomic ROW updates per second whatever the rows are (a 32 MiB table, a 0.5 MiB window of it, one
// XCD per window or all of them, agent or workgroup scope: tools/micro/atomic_window.cpp) — it is a per-request cost of the
// atomic path, not a cache effect — while plain 16-byte stores into thousands of open runs go at 80–300 G items/s and the LDS
// adds 64-bit integers at one wave instruction per 8–15 cycles.  So the contributions are PARTITIONED by table slice
// (a counting sort by bucket = slot >> bucket_shift: count, prefix, scatter), then ONE workgroup per (level, bucket) sums its
// bucket's contributions in a 64-bit fixed-point image of the slice in the LDS and adds the slice to the table gradient with plain
// coalesced stores.
//
//   bucket_count     per (pixel block, level): histogram of the buckets of its 4 corners per pixel            -> matrix
//   bucket_prefix    per (level, bucket): exclusive prefix over the pixel blocks; the totals scanned per group
//                    of 64 columns, the group sums left to the readers                                         -> matrix, local, group sums
//   bucket_scatter   per (pixel block, group of levels): items {slot in bucket, g * c (fp32, rounded as the reference rounds it)}
//   bucket_sum       per (level, bucket): max |term| and the fullest row's number of terms n -> scale 2^S,
//                    S = min(50, 61 - ceil log2 n) - exponent - 1 (no overflow; quantum 2^-50 of the bucket's largest term up to
//                    2048 terms per row); integer adds (order-free: the result is bitwise reproducible, which the float atomics
//                    were not); one rounding to fp32 at the end.
//                    ACCURACY, stated as what it is: the bound is ABSOLUTE per bucket — every term is rounded to a
//                    multiple of q = 2^-50 (2^-(61 - ceil log2 n) above 2048 terms per row) of the bucket's LARGEST |term|, so a
//                    row's error is <= (its terms) * q / 2 + half an fp32 ulp of its sum.  A row whose terms are all below q / 2
//                    of a term elsewhere in the same 4096-row (F = 2) bucket comes out as 0, where fp32 atomics would have kept it
//                    at fp32 relative precision: 15 decimal orders below the bucket's largest term (tests/test_gpu_bucket.py::
//                    test_a_tiny_row_next_to_a_large_one_in_the_same_bucket).  Against the double-precision sum of the same
//                    terms the result is within half an fp32 ulp + that quantum.  A bucket that holds a non-finite term is summed in fp32
//                    (NaN / inf propagate to the rows they belong to, as with atomics).
// + gngf_clear_hashed_rows: the sparse clear of a table gradient that lives from step to step (ops.PERSISTENT_TABLE_GRAD).
#include "gngf_common.h"

namespace gngf {

constexpr int kBkThreads = 1024;         // count / scatter workgroups: one per (pixel block, level)
constexpr int kBkPixels = 4;             // pixels per thread -> 4096 pixels per workgroup
constexpr int kBkChunk = kBkThreads * kBkPixels;
constexpr int kBkMaxBuckets = 8192;      // per level: 32 KB of LDS counters
constexpr int kBkCols = 64;              // (level, bucket) columns per prefix workgroup
constexpr int kBkSegs = 16;              // row segments per prefix workgroup (64 x 16 threads)
constexpr int kBkMaxGroups = 1024;       // column groups (nl * B / 64): their sums are scanned in every scatter workgroup's prologue
constexpr int kSumThreads = 512;
// items per thread kept in registers between the two passes of the summing kernel (F = 4: five registers per item)
template <int F> constexpr int sum_keep() { return F == 4 ? 4 : 12; }

// `order` (optional): the batch's pixels in TILE order — the binned records {x, y, bits(original index), 0} of the tiled form's
// workspace (encode_tiled.hip) — instead of xy in the caller's order.  The spatial hash keeps the low bits of gx: the vertices of one
// grid row (fixed gy) fall into ONE aligned block of table rows as wide as the row, so a workgroup whose 4096 pixels come from a
// handful of neighbouring tiles sends its contributions to a few hundred buckets instead of to all of them, and its items leave as
// runs of tens of records — whole lines that the L2 combines — instead of two isolated 16-byte stores per bucket (WRITE_SIZE at the
// 8192^2 shape: 664 MiB for 320 MiB of items before).  Both passes must walk the pixels in the same order.
__device__ __forceinline__ float2 bucket_pixel(const float2* __restrict__ xy, const float4* __restrict__ order, int64_t p, int64_t& row) {
  if (order) { const float4 s = order[p]; row = (int64_t)__float_as_int(s.z); return make_float2(s.x, s.y); }
  row = p;
  return xy[p];
}

__global__ void __launch_bounds__(kBkThreads)
bucket_count_kernel(const float2* __restrict__ xy, const float4* __restrict__ order, const int32_t* __restrict__ n_ls,
                    int32_t* __restrict__ matrix, int64_t P, int l0, int64_t T, bool pow2, int bshift, int B, int nblk) {
  extern __shared__ int bk_hist[];
  const int lv = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * kBkChunk + threadIdx.x;
  for (int b = threadIdx.x; b < B; b += kBkThreads) bk_hist[b] = 0;
  __syncthreads();
  const int n = n_ls[l0 + lv];
#pragma unroll
  for (int j = 0; j < kBkPixels; ++j) {
    const int64_t p = p0 + (int64_t)j * kBkThreads;
    if (p >= P) continue;
    int64_t row_;
    const float2 c = bucket_pixel(xy, order, p, row_);
    const Cell cell = make_cell(c.x, c.y, n);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int64_t h = spatial_hash(cell.gx + (v & 1), cell.gy + (v >> 1), T, pow2);
      atomicAdd(&bk_hist[(int)(h >> bshift)], 1);
    }
  }
  __syncthreads();
  int32_t* row = matrix + ((int64_t)lv * nblk + blockIdx.x) * B;
  for (int b = threadIdx.x; b < B; b += kBkThreads) row[b] = bk_hist[b];
}

// Exclusive prefix down the pixel blocks for 64 (level, bucket) columns per workgroup, 16 row segments side by side; the columns'
// totals are scanned inside the group (local[]) and the group's sum goes to groupsum[] (scanned by its readers: <= 1024 groups).
// The matrix is (nl, nblk, B): column col = lv * B + b lives at matrix[(lv * nblk + r) * B + b].
__global__ void __launch_bounds__(kBkCols * kBkSegs)
bucket_prefix_kernel(int32_t* __restrict__ matrix, int32_t* __restrict__ local, int32_t* __restrict__ groupsum, int ncols, int B, int nblk) {
  __shared__ int seg_sum[kBkSegs][kBkCols];
  __shared__ int col_tot[kBkCols];
  const int c = threadIdx.x % kBkCols, sg = threadIdx.x / kBkCols;
  const int col = blockIdx.x * kBkCols + c;
  const bool live = col < ncols;
  const int lv = live ? col / B : 0, b = live ? col - lv * B : 0;
  int32_t* m = matrix + (int64_t)lv * nblk * B + b;
  const int per = (nblk + kBkSegs - 1) / kBkSegs;
  const int r0 = sg * per, r1 = r0 + per < nblk ? r0 + per : nblk;
  int s = 0;
  if (live)
    for (int r = r0; r < r1; ++r) s += m[(int64_t)r * B];
  seg_sum[sg][c] = s;
  __syncthreads();
  int run = 0;
  for (int q = 0; q < sg; ++q) run += seg_sum[q][c];
  if (live)
    for (int r = r0; r < r1; ++r) {
      const int v = m[(int64_t)r * B];
      m[(int64_t)r * B] = run;
      run += v;
    }
  if (sg == kBkSegs - 1) col_tot[c] = live ? run : 0;                     // the last segment ends on the column's total
  __syncthreads();
  if (threadIdx.x < kBkCols) {                                            // one wave: exclusive scan of the 64 totals
    const int t = col_tot[threadIdx.x];
    int inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if ((int)threadIdx.x >= o) inc += u; }
    if (live) local[col] = inc - t;
    if (threadIdx.x == 63) groupsum[blockIdx.x] = inc;
  }
}

// every reader of the layout: exclusive scan of the <= 1024 group sums into LDS (blockDim.x >= 64; ngroups <= kBkMaxGroups);
// gbase[ngroups] = the number of items
__device__ __forceinline__ void bucket_group_scan(const int32_t* __restrict__ groupsum, int ngroups, int* gbase) {
  __shared__ int wave_tot[16];
  const int nthreads = blockDim.x;
  int run_base = 0;
  for (int g0 = 0; g0 < ngroups; g0 += nthreads) {                        // (one trip when the workgroup has 1024 threads)
    const int g = g0 + threadIdx.x;
    const int t = g < ngroups ? groupsum[g] : 0;
    int inc = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if ((int)(threadIdx.x & 63) >= o) inc += u; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
    __syncthreads();
    int before = run_base;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wave_tot[w];
    if (g < ngroups) gbase[g] = before + inc - t;
    int all = 0;
    for (int w = 0; w < (nthreads >> 6); ++w) all += wave_tot[w];
    run_base += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) gbase[ngroups] = run_base;
  __syncthreads();
}

// items: F = 1: float2 {slot bits, v}; F = 2: float4 {slot bits, v0, v1, 0}; F = 4: float4 vals[] followed by uint32 slots[]
template <int F>
__device__ __forceinline__ void put_item(void* items, int64_t total, int64_t pos, unsigned slot, const float* v) {
  if constexpr (F == 1) static_cast<float2*>(items)[pos] = make_float2(__uint_as_float(slot), v[0]);
  else if constexpr (F == 2) static_cast<float4*>(items)[pos] = make_float4(__uint_as_float(slot), v[0], v[1], 0.f);
  else {
    static_cast<float4*>(items)[pos] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<unsigned*>(static_cast<float4*>(items) + total)[pos] = slot;
  }
}
template <int F>
__device__ __forceinline__ unsigned get_item(const void* items, int64_t total, int64_t pos, float* v) {
  if constexpr (F == 1) { const float2 t = static_cast<const float2*>(items)[pos]; v[0] = t.y; return __float_as_uint(t.x); }
  else if constexpr (F == 2) { const float4 t = static_cast<const float4*>(items)[pos]; v[0] = t.y; v[1] = t.z; return __float_as_uint(t.x); }
  else {
    const float4 t = static_cast<const float4*>(items)[pos];
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    return reinterpret_cast<const unsigned*>(static_cast<const float4*>(items) + total)[pos];
  }
}

// One workgroup per (pixel block, group of LG levels): a pixel's gradient row is read ONCE for the group (the levels' F-vectors are
// adjacent in the (P, L*F) row: one 16-byte load at F = 2, LG = 2) — read level by level, every lane of a wave asks for its own line
// twice (measured at the 4096^2 shape: 109 us, 52 us without the gradient loads, 44 us without the item stores, 11 us without both).
template <int F, int LG>
__global__ void __launch_bounds__(kBkThreads)
bucket_scatter_kernel(const float2* __restrict__ xy, const float4* __restrict__ order, const int32_t* __restrict__ n_ls,
                      const float* __restrict__ genc,
                      const int32_t* __restrict__ matrix, const int32_t* __restrict__ local, const int32_t* __restrict__ groupsum,
                      int32_t* __restrict__ base, void* __restrict__ items, int64_t total, int64_t P, int L, int l0, int nl, int64_t T,
                      bool pow2, int bshift, int B, int nblk) {
  extern __shared__ int bk_offs[];                                      // LG x B offsets, then the scanned group sums
  const int lv0 = blockIdx.y * LG;
  const int ngroups = (nl * B + kBkCols - 1) / kBkCols;
  int* gbase = bk_offs + LG * B;
  bucket_group_scan(groupsum, ngroups, gbase);
#pragma unroll
  for (int i = 0; i < LG; ++i) {
    const int lv = lv0 + i;
    if (lv >= nl) break;
    const int32_t* row = matrix + ((int64_t)lv * nblk + blockIdx.x) * B;
    for (int b = threadIdx.x; b < B; b += kBkThreads) {
      const int col = lv * B + b;
      const int start = gbase[col / kBkCols] + local[col];              // where the bucket's items begin
      if (blockIdx.x == 0) base[col] = start;                           // (the summing kernel reads base[])
      bk_offs[i * B + b] = start + row[b];
    }
    if (blockIdx.x == 0 && lv == nl - 1 && threadIdx.x == 0) base[nl * B] = gbase[ngroups];
  }
  __syncthreads();
  const int64_t p0 = (int64_t)blockIdx.x * kBkChunk + threadIdx.x;
  const unsigned mask = (1u << bshift) - 1u;
  int n[LG];
#pragma unroll
  for (int i = 0; i < LG; ++i) n[i] = lv0 + i < nl ? n_ls[l0 + lv0 + i] : 1;
  const bool vec4 = LG * F == 4 && lv0 + LG <= nl && (((int64_t)L * F) & 3) == 0 && (((l0 + lv0) * F) & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(genc) & 15) == 0;
#pragma unroll
  for (int j = 0; j < kBkPixels; ++j) {
    const int64_t p = p0 + (int64_t)j * kBkThreads;
    if (p >= P) continue;
    int64_t row;
    const float2 c = bucket_pixel(xy, order, p, row);
    float g[LG][F];
    const float* gp = genc + (row * L + l0 + lv0) * F;
    if (vec4) {
      const float4 t = *reinterpret_cast<const float4*>(gp);
      const float tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int i = 0; i < LG; ++i)
#pragma unroll
        for (int f = 0; f < F; ++f) g[i][f] = tt[i * F + f];
    } else {
#pragma unroll
      for (int i = 0; i < LG; ++i)
#pragma unroll
        for (int f = 0; f < F; ++f) g[i][f] = lv0 + i < nl ? gp[i * F + f] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < LG; ++i) {
      if (lv0 + i >= nl) break;
      const Cell cell = make_cell(c.x, c.y, n[i]);
      int64_t h[4];
      int pos[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        h[v] = spatial_hash(cell.gx + (v & 1), cell.gy + (v >> 1), T, pow2);
        pos[v] = atomicAdd(&bk_offs[i * B + (int)(h[v] >> bshift)], 1);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float t[F];
#pragma unroll
        for (int f = 0; f < F; ++f) t[f] = g[i][f] * cell.c[v];
        put_item<F>(items, total, pos[v], (unsigned)h[v] & mask, t);
      }
    }
  }
}

__device__ __forceinline__ unsigned long long bucket_fixed(float v, double scale) {
  const double r = __builtin_fma((double)v, scale, 6755399441055744.0);       // 1.5 * 2^52: RNE(v * scale) in the low bits
  return (unsigned long long)__double_as_longlong(r) - 0x4338000000000000ull;
}

// the first sum_keep<F>() items of every thread, kept in registers from the pass that finds the scale to the pass that adds
template <int F>
__device__ __forceinline__ void bucket_load(const void* __restrict__ items, int64_t total, int beg, int n,
                                            float (&keep_v)[sum_keep<F>()][F], unsigned (&keep_s)[sum_keep<F>()]) {
#pragma unroll
  for (int j = 0; j < sum_keep<F>(); ++j) {
    const int k = threadIdx.x + j * kSumThreads;
    keep_s[j] = 0u;
#pragma unroll
    for (int f = 0; f < F; ++f) keep_v[j][f] = 0.f;
    if (k < n) keep_s[j] = get_item<F>(items, total, (int64_t)beg + k, keep_v[j]);
  }
}

// one (level, bucket): items in registers (bucket_load) -> 64-bit image in the LDS -> the slice of the table gradient
template <int F>
__device__ __forceinline__ void bucket_sum_one(const void* __restrict__ items, int64_t total, float* __restrict__ dtables, int l0, int64_t T,
                                               int bshift, int B, int accumulate, int i, int beg, int n,
                                               float (&keep_v)[sum_keep<F>()][F], unsigned (&keep_s)[sum_keep<F>()],
                                               unsigned long long* bk_img, unsigned* bk_red) {
  constexpr int kSumKeep = sum_keep<F>();
  const int lv = i / B, b = i - lv * B;
  const int slots = 1 << bshift;
  const int64_t slot0 = (int64_t)b << bshift;
  float* out = dtables + ((int64_t)(l0 + lv) * T + slot0) * F;
  const int live = (int)((T - slot0) < slots ? (T - slot0) : slots);            // the last bucket of a table that is no multiple of it
  // the largest |term| of the bucket (bit patterns of absolute values order like the values; NaN and inf sort last) and the
  // largest number of terms any ROW receives (what bounds a cell's sum: ~10 where the bucket holds thousands).
  // Round 5: a bucket of at most 2048 items needs NO count — no row can hold more than the bucket does, and up to 2^11 terms per
  // row the scale is the same (room = 50 below) — so the usual bucket of the 8192^2 shape (512 items; the ~4000 of the 4096^2 one
  // keep the count) goes: image cleared (the item loads were issued a whole bucket ago), one barrier, adds, one barrier, write-out
  // (was: five barriers with the loads' round trip exposed between the first two).
  const bool counted = n > 2048;                                                // (workgroup-uniform)
  unsigned* cnt = reinterpret_cast<unsigned*>(bk_img);
  if (counted) {
    for (int k = threadIdx.x; k < slots; k += kSumThreads) cnt[k] = 0u;
    __syncthreads();
  }
  unsigned mx = 0;
  if (!counted) {                                                               // the image is cleared under the loads
    ulonglong2* z = reinterpret_cast<ulonglong2*>(bk_img);
    const ulonglong2 zero = {0ull, 0ull};
    for (int k = threadIdx.x; k < slots * F / 2; k += kSumThreads) z[k] = zero;
    if ((slots * F) & 1) { if (threadIdx.x == 0) bk_img[slots * F - 1] = 0ull; }
  }
#pragma unroll
  for (int j = 0; j < kSumKeep; ++j) {
    if (threadIdx.x + j * kSumThreads < n) {
      if (counted) atomicAdd(&cnt[keep_s[j]], 1u);
#pragma unroll
      for (int f = 0; f < F; ++f) { const unsigned a = __float_as_uint(keep_v[j][f]) & 0x7fffffffu; mx = a > mx ? a : mx; }
    }
  }
  unsigned mc = counted ? 0u : 2048u;
  if (counted) {
    for (int k = threadIdx.x + kSumKeep * kSumThreads; k < n; k += kSumThreads) {
      float v[F];
      const unsigned s = get_item<F>(items, total, (int64_t)beg + k, v);
      atomicAdd(&cnt[s], 1u);
#pragma unroll
      for (int f = 0; f < F; ++f) { const unsigned a = __float_as_uint(v[f]) & 0x7fffffffu; mx = a > mx ? a : mx; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < slots; k += kSumThreads) mc = cnt[k] > mc ? cnt[k] : mc;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(mx, o, 64), u = __shfl_xor(mc, o, 64);
    mx = t > mx ? t : mx;
    mc = u > mc ? u : mc;
  }
  if ((threadIdx.x & 63) == 0) { bk_red[threadIdx.x >> 6] = mx; bk_red[kSumThreads / 64 + (threadIdx.x >> 6)] = mc; }
  __syncthreads();                            // (counted: every read of cnt is done before the image is cleared; else: the image IS clear)
  mx = 0; mc = 0;
#pragma unroll
  for (int w = 0; w < kSumThreads / 64; ++w) {
    mx = bk_red[w] > mx ? bk_red[w] : mx;
    mc = bk_red[kSumThreads / 64 + w] > mc ? bk_red[kSumThreads / 64 + w] : mc;
  }
  if (mx == 0) {                                                                // no contribution, or only zeros
    if (!accumulate)
      for (int k = threadIdx.x; k < live * F; k += kSumThreads) out[k] = 0.f;
    return;
  }
  if (mx >= 0x7f800000u) {                                                      // a non-finite term: fp32 sums, as the atomics formed them
    float* img = reinterpret_cast<float*>(bk_img);
    for (int k = threadIdx.x; k < slots * F; k += kSumThreads) img[k] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += kSumThreads) {
      float v[F];
      const unsigned s = get_item<F>(items, total, (int64_t)beg + k, v);
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(&img[s * F + f], v[f]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < live * F; k += kSumThreads) out[k] = accumulate ? out[k] + img[k] : img[k];
    return;
  }
  if (counted)
    for (int k = threadIdx.x; k < slots * F; k += kSumThreads) bk_img[k] = 0ull;
  const int e = (int)(mx >> 23) - 127;                                          // |term| < 2^(e + 1)  (denormals: e = -127, still true)
  const int lg = mc <= 1 ? 0 : 32 - __clz((int)mc - 1);                         // ceil(log2 of the terms of the fullest row)
  const int room = 61 - lg < 50 ? 61 - lg : 50;
  const int S = room - (e + 1);
  const double scale = ldexp(1.0, S), inv = ldexp(1.0, -S);
  if (counted) __syncthreads();
#pragma unroll
  for (int j = 0; j < kSumKeep; ++j) {
    if (threadIdx.x + j * kSumThreads < n) {
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(&bk_img[keep_s[j] * F + f], bucket_fixed(keep_v[j][f], scale));
    }
  }
  for (int k = threadIdx.x + kSumKeep * kSumThreads; k < n; k += kSumThreads) {
    float v[F];
    const unsigned s = get_item<F>(items, total, (int64_t)beg + k, v);
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(&bk_img[s * F + f], bucket_fixed(v[f], scale));
  }
  __syncthreads();
  for (int k = threadIdx.x; k < live; k += kSumThreads) {
    float r[F];
#pragma unroll
    for (int f = 0; f < F; ++f) r[f] = (float)((double)(long long)bk_img[k * F + f] * inv);
    float* o = out + (int64_t)k * F;
    // (write mode: the slice is written once and read next by the optimizer, a whole step later — streamed past the caches)
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    if constexpr (F == 2) {
      if (accumulate) {
        float2 t = *reinterpret_cast<float2*>(o);
        t.x += r[0]; t.y += r[1];
        *reinterpret_cast<float2*>(o) = t;
      } else {
        __builtin_nontemporal_store((f2v){r[0], r[1]}, reinterpret_cast<f2v*>(o));
      }
    } else if constexpr (F == 4) {
      if (accumulate) {
        float4 t = *reinterpret_cast<float4*>(o);
        t.x += r[0]; t.y += r[1]; t.z += r[2]; t.w += r[3];
        *reinterpret_cast<float4*>(o) = t;
      } else {
        __builtin_nontemporal_store((f4v){r[0], r[1], r[2], r[3]}, reinterpret_cast<f4v*>(o));
      }
    } else {
      o[0] = (accumulate ? o[0] : 0.f) + r[0];
    }
  }
}


// Workgroups are PERSISTENT (two per CU: the image is 64 KB) and walk the (level, bucket) list with their stride; the items of
// the NEXT bucket are requested before the current one is processed, so that their round trip — a third of a bucket's 6 us when
// every bucket started with it — runs under the current bucket's LDS work and write-out.
// (F = 4 only: with the twelve kept items of the F <= 2 item formats the second register set costs a workgroup per CU.)
template <int F>
__global__ void __launch_bounds__(kSumThreads)
bucket_sum_kernel(const void* __restrict__ items, int64_t total, const int32_t* __restrict__ base, float* __restrict__ dtables, int l0,
                  int64_t T, int bshift, int B, int accumulate, int nbuckets) {
  extern __shared__ unsigned long long bk_img[];
  __shared__ unsigned bk_red[2 * (kSumThreads / 64)];
  constexpr int kSumKeep = sum_keep<F>();
  int i = blockIdx.x;
  if (i >= nbuckets) return;
  float cur_v[kSumKeep][F], nxt_v[kSumKeep][F];
  unsigned cur_s[kSumKeep], nxt_s[kSumKeep];
  int beg = base[i], n = base[i + 1] - beg;
  bucket_load<F>(items, total, beg, n, cur_v, cur_s);
  for (;;) {
    const int inext = i + (int)gridDim.x;
    const bool more = inext < nbuckets;
    int nbeg = 0, nn = 0;
    if (more) {
      nbeg = base[inext]; nn = base[inext + 1] - nbeg;
      bucket_load<F>(items, total, nbeg, nn, nxt_v, nxt_s);
    }
    bucket_sum_one<F>(items, total, dtables, l0, T, bshift, B, accumulate, i, beg, n, cur_v, cur_s, bk_img, bk_red);
    if (!more) break;
    __syncthreads();                                   // every read of this bucket's image is done before the next one's clear
#pragma unroll
    for (int j = 0; j < kSumKeep; ++j) {
      cur_s[j] = nxt_s[j];
#pragma unroll
      for (int f = 0; f < F; ++f) cur_v[j][f] = nxt_v[j][f];
    }
    i = inext; beg = nbeg; n = nn;
  }
}

// ONE workgroup per (level, bucket) — the kernel of rounds 4 / 5 before the persistent form, kept for the shapes whose buckets are
// big (the 4096^2 one: ~4000 items, unequal: the hardware's dispatch balances them better than a stride) and for F <= 2 (restated
// through bucket_sum_one the F = 2 instance took 255 VGPRs against this one's 74: one workgroup per CU, 43 -> 90 us).
template <int F>
__global__ void __launch_bounds__(kSumThreads)
bucket_sum_single_kernel(const void* __restrict__ items, int64_t total, const int32_t* __restrict__ base, float* __restrict__ dtables, int l0,
                  int64_t T, int bshift, int B, int accumulate) {
  extern __shared__ unsigned long long bk_img[];
  __shared__ unsigned bk_red[2 * (kSumThreads / 64)];
  constexpr int kSumKeep = sum_keep<F>();
  const int i = blockIdx.x;
  const int lv = i / B, b = i - lv * B;
  const int beg = base[i], n = base[i + 1] - beg;
  const int slots = 1 << bshift;
  const int64_t slot0 = (int64_t)b << bshift;
  float* out = dtables + ((int64_t)(l0 + lv) * T + slot0) * F;
  const int live = (int)((T - slot0) < slots ? (T - slot0) : slots);            // the last bucket of a table that is no multiple of it
  // the largest |term| of the bucket (bit patterns of absolute values order like the values; NaN and inf sort last) and the
  // largest number of terms any ROW receives (what bounds a cell's sum: ~10 where the bucket holds thousands).
  // Round 5: a bucket of at most 2048 items needs NO count — no row can hold more than the bucket does, and up to 2^11 terms per
  // row the scale is the same (room = 50 below) — so the usual bucket of the 8192^2 shape (512 items; the ~4000 of the 4096^2 one
  // keep the count) goes: item loads issued, image cleared WHILE they are in flight, one barrier, adds, one barrier, write-out
  // (was: five barriers with the loads' round trip exposed between the first two).
  const bool counted = n > 2048;                                                // (workgroup-uniform)
  unsigned* cnt = reinterpret_cast<unsigned*>(bk_img);
  if (counted) {
    for (int k = threadIdx.x; k < slots; k += kSumThreads) cnt[k] = 0u;
    __syncthreads();
  }
  unsigned mx = 0;
  // the first kSumKeep items of every thread stay in registers for the second pass (a bucket of the usual size is read once)
  float keep_v[kSumKeep][F];
  unsigned keep_s[kSumKeep];
#pragma unroll
  for (int j = 0; j < kSumKeep; ++j) {
    const int k = threadIdx.x + j * kSumThreads;
    keep_s[j] = 0u;
#pragma unroll
    for (int f = 0; f < F; ++f) keep_v[j][f] = 0.f;
    if (k < n) keep_s[j] = get_item<F>(items, total, (int64_t)beg + k, keep_v[j]);
  }
  if (!counted) {                                                               // the image is cleared under the loads
    ulonglong2* z = reinterpret_cast<ulonglong2*>(bk_img);
    const ulonglong2 zero = {0ull, 0ull};
    for (int k = threadIdx.x; k < slots * F / 2; k += kSumThreads) z[k] = zero;
    if ((slots * F) & 1) { if (threadIdx.x == 0) bk_img[slots * F - 1] = 0ull; }
  }
#pragma unroll
  for (int j = 0; j < kSumKeep; ++j) {
    if (threadIdx.x + j * kSumThreads < n) {
      if (counted) atomicAdd(&cnt[keep_s[j]], 1u);
#pragma unroll
      for (int f = 0; f < F; ++f) { const unsigned a = __float_as_uint(keep_v[j][f]) & 0x7fffffffu; mx = a > mx ? a : mx; }
    }
  }
  unsigned mc = counted ? 0u : 2048u;
  if (counted) {
    for (int k = threadIdx.x + kSumKeep * kSumThreads; k < n; k += kSumThreads) {
      float v[F];
      const unsigned s = get_item<F>(items, total, (int64_t)beg + k, v);
      atomicAdd(&cnt[s], 1u);
#pragma unroll
      for (int f = 0; f < F; ++f) { const unsigned a = __float_as_uint(v[f]) & 0x7fffffffu; mx = a > mx ? a : mx; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < slots; k += kSumThreads) mc = cnt[k] > mc ? cnt[k] : mc;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(mx, o, 64), u = __shfl_xor(mc, o, 64);
    mx = t > mx ? t : mx;
    mc = u > mc ? u : mc;
  }
  if ((threadIdx.x & 63) == 0) { bk_red[threadIdx.x >> 6] = mx; bk_red[kSumThreads / 64 + (threadIdx.x >> 6)] = mc; }
  __syncthreads();                            // (counted: every read of cnt is done before the image is cleared; else: the image IS clear)
  mx = 0; mc = 0;
#pragma unroll
  for (int w = 0; w < kSumThreads / 64; ++w) {
    mx = bk_red[w] > mx ? bk_red[w] : mx;
    mc = bk_red[kSumThreads / 64 + w] > mc ? bk_red[kSumThreads / 64 + w] : mc;
  }
  if (mx == 0) {                                                                // no contribution, or only zeros
    if (!accumulate)
      for (int k = threadIdx.x; k < live * F; k += kSumThreads) out[k] = 0.f;
    return;
  }
  if (mx >= 0x7f800000u) {                                                      // a non-finite term: fp32 sums, as the atomics formed them
    float* img = reinterpret_cast<float*>(bk_img);
    for (int k = threadIdx.x; k < slots * F; k += kSumThreads) img[k] = 0.f;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += kSumThreads) {
      float v[F];
      const unsigned s = get_item<F>(items, total, (int64_t)beg + k, v);
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(&img[s * F + f], v[f]);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < live * F; k += kSumThreads) out[k] = accumulate ? out[k] + img[k] : img[k];
    return;
  }
  if (counted)
    for (int k = threadIdx.x; k < slots * F; k += kSumThreads) bk_img[k] = 0ull;
  const int e = (int)(mx >> 23) - 127;                                          // |term| < 2^(e + 1)  (denormals: e = -127, still true)
  const int lg = mc <= 1 ? 0 : 32 - __clz((int)mc - 1);                         // ceil(log2 of the terms of the fullest row)
  const int room = 61 - lg < 50 ? 61 - lg : 50;
  const int S = room - (e + 1);
  const double scale = ldexp(1.0, S), inv = ldexp(1.0, -S);
  if (counted) __syncthreads();
#pragma unroll
  for (int j = 0; j < kSumKeep; ++j) {
    if (threadIdx.x + j * kSumThreads < n) {
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(&bk_img[keep_s[j] * F + f], bucket_fixed(keep_v[j][f], scale));
    }
  }
  for (int k = threadIdx.x + kSumKeep * kSumThreads; k < n; k += kSumThreads) {
    float v[F];
    const unsigned s = get_item<F>(items, total, (int64_t)beg + k, v);
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(&bk_img[s * F + f], bucket_fixed(v[f], scale));
  }
  __syncthreads();
  for (int k = threadIdx.x; k < live; k += kSumThreads) {
    float r[F];
#pragma unroll
    for (int f = 0; f < F; ++f) r[f] = (float)((double)(long long)bk_img[k * F + f] * inv);
    float* o = out + (int64_t)k * F;
    // (write mode: the slice is written once and read next by the optimizer, a whole step later — streamed past the caches)
    typedef float f2v __attribute__((ext_vector_type(2)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    if constexpr (F == 2) {
      if (accumulate) {
        float2 t = *reinterpret_cast<float2*>(o);
        t.x += r[0]; t.y += r[1];
        *reinterpret_cast<float2*>(o) = t;
      } else {
        __builtin_nontemporal_store((f2v){r[0], r[1]}, reinterpret_cast<f2v*>(o));
      }
    } else if constexpr (F == 4) {
      if (accumulate) {
        float4 t = *reinterpret_cast<float4*>(o);
        t.x += r[0]; t.y += r[1]; t.z += r[2]; t.w += r[3];
        *reinterpret_cast<float4*>(o) = t;
      } else {
        __builtin_nontemporal_store((f4v){r[0], r[1], r[2], r[3]}, reinterpret_cast<f4v*>(o));
      }
    } else {
      o[0] = (accumulate ? o[0] : 0.f) + r[0];
    }
  }
}

// Rows of the table gradient that the STAGED levels of a spatial-hash encoder can ever touch: hash(gx, gy) of every vertex of
// every staged level — a set that depends on the level resolutions only, not on the batch.  A gradient buffer that lives from
// step to step (ops.PERSISTENT_TABLE_GRAD) needs only these rows cleared before the next backward adds into it: 4.6 M rows of
// 16 bytes at the 8192^2 shape instead of a dense 3 GiB.  One lane per (level, vertex), F floats per row in one store.
template <int F>
__global__ void __launch_bounds__(256)
clear_hashed_rows_kernel(float* __restrict__ dtables, const int32_t* __restrict__ n_ls, int Ls, int64_t T, bool pow2) {
  int l = 0, gw = n_ls[0] + 2;
  int64_t goff = 0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  while (l + 1 < Ls && e >= goff + (int64_t)gw * gw) { goff += (int64_t)gw * gw; ++l; gw = n_ls[l] + 2; }
  if (e - goff >= (int64_t)gw * gw) return;
  const int i = (int)(e - goff);
  const int gy = i / gw, gx = i - gy * gw;
  float* r = dtables + ((int64_t)l * T + spatial_hash(gx, gy, T, pow2)) * F;
  if constexpr (F == 2) *reinterpret_cast<float2*>(r) = make_float2(0.f, 0.f);
  else if constexpr (F == 4) *reinterpret_cast<float4*>(r) = make_float4(0.f, 0.f, 0.f, 0.f);
  else {
#pragma unroll
    for (int f = 0; f < F; ++f) r[f] = 0.f;
  }
}

// slots per bucket: the 64-bit image of a bucket is image_bytes of LDS
static int bucket_shift_for(int F, int image_bytes) {
  int s = 0;
  while ((8ll * F << (s + 1)) <= image_bytes) ++s;
  return s;
}

static int bucket_compute_units() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else { (void)hipGetLastError(); cus = 256; }
  }
  return cus;
}

// levels per scatter workgroup: as many as keep the offsets of the group within 64 KB of LDS
static int bucket_levels_per_group(int nl, int64_t B) { return (nl >= 4 && 4 * B <= 16384) ? 4 : ((nl >= 2 && 2 * B <= 16384) ? 2 : 1); }

template <int F>
static bool bucket_lds_grantable_f(int LG, size_t offs, size_t img) {
  auto ok = [](hipError_t e) {
    if (e == hipSuccess) return true;
    (void)hipGetLastError();
    return e == hipErrorNoDevice || e == hipErrorInsufficientDriver || e == hipErrorNotInitialized || e == hipErrorInvalidDevice;
  };
  const void* sc = LG == 4 ? reinterpret_cast<const void*>(bucket_scatter_kernel<F, 4>)
                           : (LG == 2 ? reinterpret_cast<const void*>(bucket_scatter_kernel<F, 2>) : reinterpret_cast<const void*>(bucket_scatter_kernel<F, 1>));
  if (offs > 48 * 1024 && !ok(hipFuncSetAttribute(sc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)offs))) return false;
  if (img > 48 * 1024 && !(ok(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sum_single_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img)) &&
                           ok(hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sum_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img))))
    return false;
  return true;
}
static bool bucket_lds_grantable(int F, int LG, size_t offs, size_t img) {
  return F == 1 ? bucket_lds_grantable_f<1>(LG, offs, img) : (F == 2 ? bucket_lds_grantable_f<2>(LG, offs, img) : bucket_lds_grantable_f<4>(LG, offs, img));
}

}  // namespace gngf

using namespace gngf;

// The library's own sizing of the bucketed backward for levels [l0, l1) of a hash-indexed encoder: returns 1 and fills
//   plan[0] bucket_shift   plan[1] buckets per level   plan[2] pixel blocks   plan[3] matrix ints   plan[4] base ints
//   plan[5] bytes of the item buffer
// or returns 0 when the shape is not served (then gngf_encode_bwd's atomics are the way): F not in {1, 2, 4}, more than 8192
// buckets per level or 65536 in all, 2^31 contributions or more.
extern "C" int gngf_encode_bwd_bucketed_plan(int64_t P, int F, int64_t T, int nl, int image_bytes, int64_t* plan) {
  if (!plan || P <= 0 || nl <= 0 || T <= 0 || (F != 1 && F != 2 && F != 4)) return 0;
  if (image_bytes < 1024 || image_bytes > 128 * 1024) return 0;
  const int bshift = bucket_shift_for(F, image_bytes);
  const int64_t B = (T + (1ll << bshift) - 1) >> bshift;
  const int64_t total = P * 4 * nl;
  if (B > kBkMaxBuckets || total >= (1ll << 31) || nl * B > (int64_t)kBkMaxGroups * kBkCols) return 0;
  const int64_t nblk = ceil_div(P, kBkChunk);
  const int64_t ngroups = ceil_div(nl * B, kBkCols);
  // the dynamic LDS the two big kernels will ask for (the scatter's bucket offsets + scanned group sums, the summing
  // kernel's 64-bit image) must be grantable on THIS device, or the caller has to be told to take the atomics (0) — not find out
  // from a failing launch after the answer "1".  Asked of the runtime exactly as the launcher asks (no device visible — the
  // host-side planning tests — counts as granted: the sizing itself is pure arithmetic).
  {
    const int LG = bucket_levels_per_group(nl, B);
    const size_t offs = sizeof(int) * (size_t)(LG * B + ngroups + 1), img = (size_t)8 * F << bshift;
    if (!bucket_lds_grantable(F, LG, offs, img)) return 0;
  }
  plan[0] = bshift; plan[1] = B; plan[2] = nblk; plan[3] = nl * nblk * B;
  plan[4] = (nl * B + 1) + nl * B + ngroups;                  // base | local | group sums
  plan[5] = total * (F == 1 ? 8 : (F == 2 ? 16 : 20));
  return 1;
}

// dtables (L,T,F) fp32: levels [l0, l1) receive the gradient of the batch (accumulate = 1: added to what is there — the caller
// cleared it, or other batches' gradients are in it; 0: every row of those levels is WRITTEN, touched or not — no clear needed).
// matrix / base / items: scratch sized by gngf_encode_bwd_bucketed_plan (same P, F, T, l1 - l0, image_bytes).
// pixel_order (optional, P float4 records {x, y, bits(original index), 0}: the binned pixels of the tiled form's workspace):
// the batch is walked in that order instead of in xy's (same result, bit for bit: the sums are order-free; see bucket_pixel).
extern "C" int gngf_encode_bwd_bucketed(const float* xy, const int32_t* n_ls, const float* genc, float* dtables, int64_t P, int L, int F,
                                        int64_t T, int l0, int l1, int image_bytes, int accumulate, int32_t* matrix, int32_t* base,
                                        void* items, const float* pixel_order, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && T > 0 && l0 >= 0 && l0 <= l1 && l1 <= L);
  if (l0 == l1) return 0;
  int64_t plan[6];
  if (P == 0) {
    if (!accumulate) { GNGF_CHECK_ARG(dtables); return (int)zero_async(dtables + (int64_t)l0 * T * F, sizeof(float) * (size_t)((l1 - l0) * T * F), as_stream(stream)); }
    return 0;
  }
  GNGF_CHECK_ARG(gngf_encode_bwd_bucketed_plan(P, F, T, l1 - l0, image_bytes, plan) == 1);
  GNGF_CHECK_ARG(xy && n_ls && genc && dtables && matrix && base && items && (reinterpret_cast<uintptr_t>(pixel_order) & 15) == 0);
  const float4* order = reinterpret_cast<const float4*>(pixel_order);
  // vector accesses: items as 16-byte records, a table row (F floats) as one store, a gradient vector (F floats) as one load
  GNGF_CHECK_ARG((reinterpret_cast<uintptr_t>(items) & 15) == 0 && (reinterpret_cast<uintptr_t>(dtables) & (4 * F - 1)) == 0 &&
                 (reinterpret_cast<uintptr_t>(genc) & (4 * F - 1)) == 0 && (reinterpret_cast<uintptr_t>(xy) & 7) == 0);
  const int nl = l1 - l0, bshift = (int)plan[0], B = (int)plan[1], nblk = (int)plan[2];
  const int ncols = nl * B, ngroups = (int)ceil_div(ncols, kBkCols);
  int32_t* local = base + ncols + 1;
  int32_t* groupsum = local + ncols;
  const int64_t total = P * 4 * nl;
  const bool pow2 = (T & (T - 1)) == 0;
  hipStream_t s = as_stream(stream);
  const dim3 grid2((unsigned)nblk, (unsigned)nl);
  bucket_count_kernel<<<grid2, dim3(kBkThreads), sizeof(int) * (size_t)B, s>>>(
      reinterpret_cast<const float2*>(xy), order, n_ls, matrix, P, l0, T, pow2, bshift, B, nblk);
  bucket_prefix_kernel<<<dim3((unsigned)ngroups), dim3(kBkCols * kBkSegs), 0, s>>>(matrix, local, groupsum, ncols, B, nblk);
  const size_t img = (size_t)8 * F << bshift;
  const int LG = bucket_levels_per_group(nl, B);
  const size_t offs = sizeof(int) * (size_t)(LG * B + ngroups + 1);
  const dim3 grid3((unsigned)nblk, (unsigned)ceil_div(nl, LG));
#define GNGF_BUCKET_SCATTER(kF, kLG)                                                                                             \
  {                                                                                                                             \
    if (offs > 48 * 1024) {                                                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter_kernel<kF, kLG>),                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)offs);                                \
      if (e != hipSuccess) return (int)e;                                                                                       \
    }                                                                                                                           \
    bucket_scatter_kernel<kF, kLG><<<grid3, dim3(kBkThreads), offs, s>>>(                                                       \
        reinterpret_cast<const float2*>(xy), order, n_ls, genc, matrix, local, groupsum, base, items, total, P, L, l0, nl, T, pow2, \
        bshift, B, nblk);                                                                                                       \
  }
#define GNGF_BUCKET_F(kF)                                                                                                       \
  {                                                                                                                             \
    if (LG == 4) GNGF_BUCKET_SCATTER(kF, 4) else if (LG == 2) GNGF_BUCKET_SCATTER(kF, 2) else GNGF_BUCKET_SCATTER(kF, 1)        \
    if (img > 48 * 1024) {                                                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sum_single_kernel<kF>),                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)img);                                 \
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sum_kernel<kF>),                  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)img);                       \
      if (e != hipSuccess) return (int)e;                                                                                       \
    }                                                                                                                           \
    /* persistent + prefetching workgroups where the buckets are small (<= 2048 items on average: the 8192^2 shape, 552 vs 612 us);   \
       one workgroup per bucket where they are big (the 4096^2 shape: ~4000 items, 130 vs 157 us — the hardware's own dispatch       \
       balances unequal buckets better than a stride does) */                                                                   \
    const int nbk = nl * B, fit = 2 * bucket_compute_units();                                                                  \
    const bool persistent = kF == 4 && total <= (int64_t)2048 * nbk && fit < nbk;                                              \
    if (persistent)                                                                                                             \
      bucket_sum_kernel<kF><<<dim3((unsigned)fit), dim3(kSumThreads), img, s>>>(items, total, base, dtables, l0, T, bshift, B,   \
                                                                               accumulate, nbk);                               \
    else                                                                                                                        \
      bucket_sum_single_kernel<kF><<<dim3((unsigned)nbk), dim3(kSumThreads), img, s>>>(items, total, base, dtables, l0, T, bshift, \
                                                                                      B, accumulate);                          \
  }
  if (F == 1) GNGF_BUCKET_F(1) else if (F == 2) GNGF_BUCKET_F(2) else GNGF_BUCKET_F(4)
#undef GNGF_BUCKET_F
#undef GNGF_BUCKET_SCATTER
  GNGF_RETURN_LAUNCH();
}

// dtables (L,T,F) fp32: zeroes row hash(gx, gy) of level l for every vertex (gx, gy) in [0, N_l + 1]^2 of levels [0, Ls) — every row
// the staged levels of a hash-indexed encoder (models.py:504-528) can touch, whatever the batch.  vtot = sum (N_l + 2)^2.
extern "C" int gngf_clear_hashed_rows(float* dtables, const int32_t* n_ls, int Ls, int F, int64_t T, int64_t vtot, void* stream) {
  GNGF_CHECK_ARG(Ls >= 0 && Ls <= GNGF_MAX_LEVELS && T > 0 && vtot >= 0 && (F == 1 || F == 2 || F == 4 || F == 8));
  if (Ls == 0 || vtot == 0) return 0;
  GNGF_CHECK_ARG(dtables && n_ls && (reinterpret_cast<uintptr_t>(dtables) & (4 * (F > 4 ? 4 : F) - 1)) == 0);
  const dim3 grid((unsigned)ceil_div(vtot, 256));
  const bool pow2 = (T & (T - 1)) == 0;
  hipStream_t s = as_stream(stream);
  switch (F) {
    case 1: clear_hashed_rows_kernel<1><<<grid, dim3(256), 0, s>>>(dtables, n_ls, Ls, T, pow2); break;
    case 2: clear_hashed_rows_kernel<2><<<grid, dim3(256), 0, s>>>(dtables, n_ls, Ls, T, pow2); break;
    case 4: clear_hashed_rows_kernel<4><<<grid, dim3(256), 0, s>>>(dtables, n_ls, Ls, T, pow2); break;
    default: clear_hashed_rows_kernel<8><<<grid, dim3(256), 0, s>>>(dtables, n_ls, Ls, T, pow2); break;
  }
  GNGF_RETURN_LAUNCH();
}
