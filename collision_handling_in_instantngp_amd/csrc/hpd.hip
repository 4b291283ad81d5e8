// HashProbDistribution tail (reference models.py:85,105-123 and DifferentiableTopk models.py:5-42), evaluated
// once per DISTINCT grid vertex instead of once per (pixel, level, corner) instance:
//   softmax over the T slots + nan_to_num + top-K selection, and its backward WITHOUT the dense zero-filled
//   (P,L,4,T) scatter of the reference (models.py:27-35): the K gradients are folded into the softmax
//   backward row by row.
// Also: vertex-coordinate generation, blend (softmax over K) forward/backward on the per-vertex table,
// per-vertex multiplicities for the batch-mean distribution, and the expansion kernels that rebuild the
// reference-shaped (P,L,4,K) outputs from the per-vertex table.
#include "gngf_common.h"
#include <limits.h>

namespace gngf {
using f32x16_t = __attribute__((ext_vector_type(16))) float;

constexpr int kRowBlock = 256;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// (value desc, index asc) ordering: a beats b
__device__ __forceinline__ bool beats(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

// One 256-thread block per row.  z: logits in, probabilities out (in place).
// torch semantics: softmax = exp(x - max) / sum; a NaN logit poisons the whole row (max = NaN), and
// nan_to_num maps the NaNs to 0.  Top-K is taken on the final fp32 probabilities; ties -> lower index.
__global__ void __launch_bounds__(kRowBlock)
softmax_topk_kernel(float* __restrict__ z, float* __restrict__ topv, int32_t* __restrict__ topi, int64_t T, int K,
                    int do_softmax, float* __restrict__ rowstat) {
  extern __shared__ float smem[];            // K*256 values, K*256 indices, then 16 floats scratch
  float* lv = smem;
  int* li = reinterpret_cast<int*>(smem + (size_t)K * kRowBlock);
  float* red = smem + (size_t)2 * K * kRowBlock;
  int* redi = reinterpret_cast<int*>(red + 8);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* p = z + (int64_t)blockIdx.x * T;

  float m = -INFINITY, s = 1.f;
  bool row_nan = false;
  if (do_softmax) {
    bool has_nan = false;
    for (int64_t t = tid; t < T; t += kRowBlock) { const float v = p[t]; has_nan |= (v != v); m = fmaxf(m, v); }
    m = wave_max(m);
    const unsigned long long nanmask = __ballot(has_nan);
    if (lane == 0) { red[wave] = m; redi[wave] = nanmask != 0ull; }
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    row_nan = (redi[0] | redi[1] | redi[2] | redi[3]) != 0;
    __syncthreads();
    s = 0.f;
    for (int64_t t = tid; t < T; t += kRowBlock) s += expf(p[t] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    if (rowstat && tid == 0) {                 // (max, sum) of the row: lets backward rebuild p = exp(z - m) / s from logits
      rowstat[2 * (int64_t)blockIdx.x] = m;
      rowstat[2 * (int64_t)blockIdx.x + 1] = row_nan ? __int_as_float(0x7fc00000) : s;
    }
  }

  for (int k = 0; k < K; ++k) { lv[k * kRowBlock + tid] = -INFINITY; li[k * kRowBlock + tid] = INT_MAX; }
  float thr = -INFINITY;                       // value of this thread's K-th best so far
  for (int64_t t = tid; t < T; t += kRowBlock) {
    float q = p[t];
    if (do_softmax) {
      q = expf(q - m) / s;
      if (row_nan || q != q) q = 0.f;          // nan_to_num (models.py:111)
      else if (q > 3.4028234663852886e38f) q = 3.4028234663852886e38f;
      p[t] = q;
    }
    if (q > thr) {                             // strictly: a later index never displaces an equal value
      int k = K - 1;
      while (k > 0 && lv[(k - 1) * kRowBlock + tid] < q) {
        lv[k * kRowBlock + tid] = lv[(k - 1) * kRowBlock + tid];
        li[k * kRowBlock + tid] = li[(k - 1) * kRowBlock + tid];
        --k;
      }
      lv[k * kRowBlock + tid] = q;
      li[k * kRowBlock + tid] = (int)t;
      thr = lv[(K - 1) * kRowBlock + tid];
    }
  }
  // K rounds of block-wide arg-best over the heads of the per-thread sorted lists
  int head = 0;
  for (int r = 0; r < K; ++r) {
    float cv = head < K ? lv[head * kRowBlock + tid] : -INFINITY;
    int ci = head < K ? li[head * kRowBlock + tid] : INT_MAX;
    int owner = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(cv, o, 64);
      const int oi = __shfl_xor(ci, o, 64);
      const int oo = __shfl_xor(owner, o, 64);
      if (beats(ov, oi, cv, ci)) { cv = ov; ci = oi; owner = oo; }
    }
    if (lane == 0) { red[wave] = cv; redi[wave] = ci; redi[4 + wave] = owner; }
    __syncthreads();
    float bv = red[0]; int bi = redi[0], bo = redi[4];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (beats(red[w], redi[w], bv, bi)) { bv = red[w]; bi = redi[w]; bo = redi[4 + w]; }
    if (tid == bo) ++head;
    if (tid == 0) { topv[(int64_t)blockIdx.x * K + r] = bv; topi[(int64_t)blockIdx.x * K + r] = bi; }
    __syncthreads();
  }
}

// Softmax backward with the top-K gradient folded in.  One block per row.
//   g_t = gdense[row,t] (optional) + sum_l mw[row,l] * G[l,t] (optional)  [+ dq_k at t = topi_k]
//   dz_t = p_t * (g_t - sum_t' p_t' g_t')          (p = post-nan_to_num probabilities: NaN rows have p = 0 -> dz = 0)
// dz may alias p (in place).
__global__ void __launch_bounds__(kRowBlock)
softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dq, const int32_t* __restrict__ topi,
                   const float* __restrict__ gdense, const float* __restrict__ mw, const float* __restrict__ G, int L,
                   float* __restrict__ dZ, int64_t T, int K) {
  __shared__ float red[4];
  __shared__ float pk[GNGF_MAX_TOPK];
  __shared__ float mwl[GNGF_MAX_LEVELS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const float* p = P + row * T;
  float* dz = dZ + row * T;
  const float* gd = gdense ? gdense + row * T : nullptr;
  if (tid < L && mw) mwl[tid] = mw[row * L + tid];
  if (tid < K) pk[tid] = p[topi[row * K + tid]];
  __syncthreads();
  const bool lowrank = (mw != nullptr) && (G != nullptr);
  float dot = 0.f;
  for (int64_t t = tid; t < T; t += kRowBlock) {
    float g = gd ? gd[t] : 0.f;
    if (lowrank) for (int l = 0; l < L; ++l) g += mwl[l] * G[(int64_t)l * T + t];
    dot += p[t] * g;
  }
  if (tid < K && dq) dot += pk[tid] * dq[row * K + tid];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  dot = (red[0] + red[1]) + (red[2] + red[3]);
  for (int64_t t = tid; t < T; t += kRowBlock) {
    float g = gd ? gd[t] : 0.f;
    if (lowrank) for (int l = 0; l < L; ++l) g += mwl[l] * G[(int64_t)l * T + t];
    dz[t] = p[t] * (g - dot);
  }
  __syncthreads();
  if (tid < K && dq) dz[topi[row * K + tid]] += pk[tid] * dq[row * K + tid];
}

// ---------------------------------------------------------------------------------------------- streaming forward
// One read of the logits per row: online (max, sum-exp) and top-K selection on the LOGITS (softmax is monotone, so the
// K winners are the same; exactly tied probabilities between distinct logits are ordered by logit, equal logits by the
// lower index — torch.topk leaves tie order unspecified).  The K probabilities are exp(z_k - max) / sum, the same
// expression the dense softmax evaluates.  Nothing is written back: the (rows, T) distribution never exists.
__global__ void __launch_bounds__(kRowBlock)
logits_stats_topk_kernel(const float* __restrict__ z, float* __restrict__ topv, int32_t* __restrict__ topi,
                         float* __restrict__ rowstat, int64_t T, int K) {
  extern __shared__ float smem[];
  float* lv = smem;
  int* li = reinterpret_cast<int*>(smem + (size_t)K * kRowBlock);
  float* red = smem + (size_t)2 * K * kRowBlock;
  int* redi = reinterpret_cast<int*>(red + 8);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = z + (int64_t)blockIdx.x * T;
  for (int k = 0; k < K; ++k) { lv[k * kRowBlock + tid] = -INFINITY; li[k * kRowBlock + tid] = INT_MAX; }
  float m = -INFINITY, s = 0.f, thr = -INFINITY;
  bool has_nan = false;
  auto consider = [&](float v, int64_t t) {                      // insertion into the thread's sorted top-K list
    if (v > thr) {
      int k = K - 1;
      while (k > 0 && lv[(k - 1) * kRowBlock + tid] < v) {
        lv[k * kRowBlock + tid] = lv[(k - 1) * kRowBlock + tid];
        li[k * kRowBlock + tid] = li[(k - 1) * kRowBlock + tid];
        --k;
      }
      lv[k * kRowBlock + tid] = v;
      li[k * kRowBlock + tid] = (int)t;
      thr = lv[(K - 1) * kRowBlock + tid];
    }
  };
  // The pass is VALU-bound, not HBM-bound (an online-softmax update with expf costs ~25 instructions per logit): a thread
  // takes 8 logits per trip (two 16-byte loads), rescales its running sum ONCE per trip by the new maximum and adds the 8
  // terms with the hardware exp2 (the row sum only normalises; its ~1e-6 relative error is far inside the parity
  // tolerance, and the K probabilities below still use expf on the exact maximum).
  constexpr float kLog2e = 1.4426950408889634f;
  const bool vec = (T % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
  if (vec) {
    for (int64_t t0 = (int64_t)tid * 4; t0 < T; t0 += (int64_t)kRowBlock * 8) {
      const float4 a = *reinterpret_cast<const float4*>(p + t0);
      const int64_t t1 = t0 + (int64_t)kRowBlock * 4;
      const bool two = t1 < T;
      const float4 b = two ? *reinterpret_cast<const float4*>(p + t1) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      float mx = m;
#pragma unroll
      for (int q = 0; q < 8; ++q) { has_nan |= (v[q] != v[q]); mx = fmaxf(mx, v[q]); }
      if (mx > m) { s = (m == -INFINITY) ? 0.f : s * __builtin_amdgcn_exp2f((m - mx) * kLog2e); m = mx; }
      if (m > -INFINITY) {
#pragma unroll
        for (int q = 0; q < 8; ++q) s += __builtin_amdgcn_exp2f((v[q] - m) * kLog2e);     // exp2(-inf) = 0 for the padding
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) consider(v[q], t0 + q);
      if (two) {
#pragma unroll
        for (int q = 0; q < 4; ++q) consider(v[4 + q], t1 + q);
      }
    }
  } else {
    for (int64_t t = tid; t < T; t += kRowBlock) {
      const float v = p[t];
      has_nan |= (v != v);
      if (v > m) { s = s * expf(m - v) + 1.f; m = v; }            // first element: s = 0 * exp(-inf) + 1
      else s += expf(v - m);
      consider(v, t);
    }
  }
  // block combine of the online statistics: M = max m_t, S = sum s_t exp(m_t - M)
  const float wm = wave_max(m);
  const unsigned long long nanmask = __ballot(has_nan);
  if (lane == 0) { red[wave] = wm; redi[wave] = nanmask != 0ull; }
  __syncthreads();
  const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const bool row_nan = (redi[0] | redi[1] | redi[2] | redi[3]) != 0;
  __syncthreads();
  float sc = (m == -INFINITY) ? 0.f : s * expf(m - M);
  sc = wave_sum(sc);
  if (lane == 0) red[wave] = sc;
  __syncthreads();
  const float S = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  if (tid == 0) {
    rowstat[2 * (int64_t)blockIdx.x] = M;
    rowstat[2 * (int64_t)blockIdx.x + 1] = row_nan ? __int_as_float(0x7fc00000) : S;
  }
  int head = 0;
  for (int r = 0; r < K; ++r) {
    float cv = head < K ? lv[head * kRowBlock + tid] : -INFINITY;
    int ci = head < K ? li[head * kRowBlock + tid] : INT_MAX;
    int owner = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(cv, o, 64);
      const int oi = __shfl_xor(ci, o, 64);
      const int oo = __shfl_xor(owner, o, 64);
      if (beats(ov, oi, cv, ci)) { cv = ov; ci = oi; owner = oo; }
    }
    if (lane == 0) { red[wave] = cv; redi[wave] = ci; redi[4 + wave] = owner; }
    __syncthreads();
    float bv = red[0]; int bi = redi[0], bo = redi[4];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (beats(red[w], redi[w], bv, bi)) { bv = red[w]; bi = redi[w]; bo = redi[4 + w]; }
    if (tid == bo) ++head;
    if (tid == 0) {
      float q = expf(bv - M) / S;
      if (row_nan || q != q) q = 0.f;                           // nan_to_num (models.py:111)
      else if (q > 3.4028234663852886e38f) q = 3.4028234663852886e38f;
      topv[(int64_t)blockIdx.x * K + r] = q;
      topi[(int64_t)blockIdx.x * K + r] = row_nan ? r : bi;     // a NaN row is all zeros after nan_to_num: slots 0..K-1
    }
    __syncthreads();
  }
}

// Row statistics and top-K from the PARTIALS the logits GEMM left in its epilogue (linear.hip: gemm128_split_kernel, rowparts:
// (max, sum exp(z - max)) per row and 64-column block) — instead of logits_stats_topk_kernel's pass over the logits themselves
// (8 GiB per 4096-row chunk at T = 2^19; the partials are 1/32 of that).  One 256-thread block per row:
//   M = max_b m_b,  S = sum_b s_b exp(m_b - M)                    (the online-softmax merge, as logits_stats_topk_kernel's)
//   the K largest logits of the row lie inside the K blocks with the largest maxima (if a top-K element sat in another block,
//   K blocks would each hold an element above it): those K x 64 logits are the only ones read, and the selection among them is
//   the same (value desc, index asc) merge.  A NaN anywhere in the row (a NaN partial sum) makes the row all-zero after
//   nan_to_num, slots 0..K-1, as in logits_stats_topk_kernel.
__global__ void __launch_bounds__(kRowBlock)
rowstats_topk_kernel(const float* __restrict__ z, const float2* __restrict__ parts, int nparts, float* __restrict__ topv,
                     int32_t* __restrict__ topi, float* __restrict__ rowstat, int64_t T, int K) {
  extern __shared__ float smem[];
  float* lv = smem;                                              // [K][256] per-thread sorted lists (block maxima, then logits)
  int* li = reinterpret_cast<int*>(smem + (size_t)K * kRowBlock);
  float* red = smem + (size_t)2 * K * kRowBlock;
  int* redi = reinterpret_cast<int*>(red + 8);
  int* cand = redi + 8;                                          // [K] block indices of the K largest maxima
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float2* pr = parts + (int64_t)blockIdx.x * nparts;
  constexpr float kLog2e = 1.4426950408889634f;
  float thr;
  auto reset = [&]() { for (int k = 0; k < K; ++k) { lv[k * kRowBlock + tid] = -INFINITY; li[k * kRowBlock + tid] = INT_MAX; } thr = -INFINITY; };
  auto consider = [&](float v, int t) {
    if (v > thr) {
      int k = K - 1;
      while (k > 0 && lv[(k - 1) * kRowBlock + tid] < v) {
        lv[k * kRowBlock + tid] = lv[(k - 1) * kRowBlock + tid];
        li[k * kRowBlock + tid] = li[(k - 1) * kRowBlock + tid];
        --k;
      }
      lv[k * kRowBlock + tid] = v;
      li[k * kRowBlock + tid] = t;
      thr = lv[(K - 1) * kRowBlock + tid];
    }
  };
  // the r-th best (value, index) over the whole block, r = 0 .. K-1, each round popping the winner's list (as logits_stats_topk_kernel)
  auto select = [&](auto&& emit) {
    int head = 0;
    for (int r = 0; r < K; ++r) {
      float cv = head < K ? lv[head * kRowBlock + tid] : -INFINITY;
      int ci = head < K ? li[head * kRowBlock + tid] : INT_MAX;
      int owner = tid;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(cv, o, 64);
        const int oi = __shfl_xor(ci, o, 64);
        const int oo = __shfl_xor(owner, o, 64);
        if (beats(ov, oi, cv, ci)) { cv = ov; ci = oi; owner = oo; }
      }
      if (lane == 0) { red[wave] = cv; redi[wave] = ci; redi[4 + wave] = owner; }
      __syncthreads();
      float bv = red[0]; int bi = redi[0], bo = redi[4];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (beats(red[w], redi[w], bv, bi)) { bv = red[w]; bi = redi[w]; bo = redi[4 + w]; }
      if (tid == bo) ++head;
      emit(r, bv, bi);
      __syncthreads();
    }
  };
  // pass over the partials: online merge of (max, sum), K largest block maxima
  reset();
  float m = -INFINITY, s = 0.f;
  bool has_nan = false;
  for (int b = tid; b < nparts; b += kRowBlock) {
    const float2 p = pr[b];
    has_nan |= (p.y != p.y) || (p.x != p.x);
    if (p.x > m) { s = (m == -INFINITY) ? 0.f : s * __builtin_amdgcn_exp2f((m - p.x) * kLog2e); m = p.x; }
    if (p.x > -INFINITY && p.y == p.y) s += p.y * __builtin_amdgcn_exp2f((p.x - m) * kLog2e);
    consider(p.x, b);
  }
  const float wm = wave_max(m);
  const unsigned long long nanmask = __ballot(has_nan);
  if (lane == 0) { red[wave] = wm; redi[wave] = nanmask != 0ull; }
  __syncthreads();
  const float M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const bool row_nan = (redi[0] | redi[1] | redi[2] | redi[3]) != 0;
  __syncthreads();
  float sc = (m == -INFINITY) ? 0.f : s * expf(m - M);
  sc = wave_sum(sc);
  if (lane == 0) red[wave] = sc;
  __syncthreads();
  const float S = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  if (tid == 0) {
    rowstat[2 * (int64_t)blockIdx.x] = M;
    rowstat[2 * (int64_t)blockIdx.x + 1] = row_nan ? __int_as_float(0x7fc00000) : S;
  }
  select([&](int r, float, int bi) { if (tid == 0) cand[r] = bi; });
  // the K candidate blocks' logits: K * 64 values, one or a few per thread
  reset();
  const float* zr = z + (int64_t)blockIdx.x * T;
  for (int e = tid; e < K * 64; e += kRowBlock) {
    const int cb = cand[e >> 6];
    if (cb < 0 || cb >= nparts) continue;                        // (fewer than K blocks in the row)
    const int t = cb * 64 + (e & 63);
    const float v = zr[t];
    if (v == v) consider(v, t);                                  // (a NaN row is reported through row_nan)
  }
  select([&](int r, float bv, int bi) {
    if (tid == 0) {
      float q = expf(bv - M) / S;
      if (row_nan || q != q) q = 0.f;                            // nan_to_num (models.py:111)
      else if (q > 3.4028234663852886e38f) q = 3.4028234663852886e38f;
      topv[(int64_t)blockIdx.x * K + r] = q;
      topi[(int64_t)blockIdx.x * K + r] = row_nan ? r : bi;      // a NaN row is all zeros after nan_to_num: slots 0..K-1
    }
  });
}

// pbar[l][t] += sum_r mw[r][l] * exp(z[r][t] - m_r) / s_r  over the rows of one chunk.  A block owns 256 columns for the
// whole chunk (plain read-modify-write, no atomics: launches of successive chunks are stream-ordered); row statistics
// and multiplicity weights are wave-uniform scalar loads.
__device__ __forceinline__ float prob_of_fwd(float z, float m, float s) {
  const float q = expf(z - m) / s;
  return (q != q) ? 0.f : (q > 3.4028234663852886e38f ? 3.4028234663852886e38f : q);
}

// The streaming passes are VALU-bound (one exp and ~16 FMAs per logit): their probabilities use the hardware exp2 and
// a per-row reciprocal of the row sum (3 instructions instead of ~25 for expf and an IEEE division; relative error
// ~1e-6 from the rounding of (z - m) log2 e, far inside the 5e-5 the parity tests allow; nan_to_num semantics kept).
// The top-K probabilities that reach the encoder keep the exact expression.
__device__ __forceinline__ float prob_fast(float z, float m, float rs) {
  const float q = __builtin_amdgcn_exp2f((z - m) * 1.4426950408889634f) * rs;
  return (q != q) ? 0.f : (q > 3.4028234663852886e38f ? 3.4028234663852886e38f : q);
}

template <int LMAX>
__global__ void __launch_bounds__(256)
pbar_accum_kernel(const float* __restrict__ Z, const float* __restrict__ rowstat, const float* __restrict__ mw, int L,
                  float* __restrict__ pbar, int64_t U, int64_t T) {
  // a thread owns CPT consecutive columns (one 16-byte load per row), so the per-row scalar work is shared by 4 logits
  constexpr int CPT = 4;
  const int64_t t0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * CPT;
  const bool vec = (T % CPT == 0) && t0 + CPT <= T;
  float acc[CPT][LMAX];
#pragma unroll
  for (int c = 0; c < CPT; ++c)
#pragma unroll
    for (int l = 0; l < LMAX; ++l) acc[c][l] = 0.f;
  constexpr int RU = 4;                                            // rows per trip: their loads are issued together
  for (int64_t r0 = 0; r0 < U; r0 += RU) {
    float zv[RU][CPT];
#pragma unroll
    for (int q = 0; q < RU; ++q) {
      const int64_t r = r0 + q < U ? r0 + q : U - 1;
      if (vec) {
        const float4 v = *reinterpret_cast<const float4*>(Z + r * T + t0);
        zv[q][0] = v.x; zv[q][1] = v.y; zv[q][2] = v.z; zv[q][3] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < CPT; ++c) zv[q][c] = t0 + c < T ? Z[r * T + t0 + c] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < RU; ++q) {
      if (r0 + q >= U) break;
      const int64_t r = r0 + q;
      const float m = rowstat[2 * r], rs = 1.0f / rowstat[2 * r + 1];     // uniform -> scalar loads, one reciprocal per row
      float mwr[LMAX];
#pragma unroll
      for (int l = 0; l < LMAX; ++l) mwr[l] = (l < L) ? mw[r * L + l] : 0.f;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const float pv = prob_fast(zv[q][c], m, rs);
#pragma unroll
        for (int l = 0; l < LMAX; ++l) acc[c][l] += mwr[l] * pv;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPT; ++c)
    if (t0 + c < T) {
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        if (l < L) pbar[(int64_t)l * T + t0 + c] += acc[c][l];
    }
}

// The same accumulation on the matrix cores (T % 32 == 0): pbar (L x T) = mw^T (L x U) * P (U x T) is a GEMM whose B
// operand is the probability computed on the fly from the logit each lane has just loaded.  One 32x32x2 MFMA consumes two
// rows x 32 columns: A[m = l][k] = mw[row][l], B[k][n] = p[row][col], with lane (i, h) holding k = h.  A wave keeps a
// 32 x 32 tile of p-bar in its accumulator for every row of the chunk; the rows' (max, 1 / sum) and weights are staged
// in LDS per block of 128 rows and read with immediate offsets.  ~6 VALU instructions and one MFMA per logit-lane instead
// of 19 VALU instructions: the pass follows the HBM stream.
constexpr int kPbRows = 128;

__global__ void __launch_bounds__(256)
pbar_mfma_kernel(const float* __restrict__ Z, const float* __restrict__ rowstat, const float* __restrict__ mw, int L,
                 float* __restrict__ pbar, int64_t U, int64_t T) {
  __shared__ float2 s_stat[kPbRows];
  __shared__ float s_mw[kPbRows * 32];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t c0 = ((int64_t)blockIdx.x * 4 + wave) * 32;          // this wave's 32 columns
  const bool live = c0 < T;
  f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned voff = 4u * ((unsigned)h * (unsigned)T + (unsigned)i);
  const unsigned step = 8u * (unsigned)T;                             // two rows, in bytes
  for (int64_t rb = 0; rb < U; rb += kPbRows) {
    const int vrows = (int)(U - rb < kPbRows ? U - rb : kPbRows);
    __syncthreads();                                                  // the previous block's LDS reads are done
    if (tid < kPbRows) {
      const bool ok = tid < vrows;
      s_stat[tid] = make_float2(ok ? rowstat[2 * (rb + tid)] : 0.f, ok ? 1.0f / rowstat[2 * (rb + tid) + 1] : 0.f);
    }
    for (int e = tid; e < kPbRows * 32; e += 256) {
      const int row = e >> 5, l = e & 31;
      s_mw[e] = (row < vrows && l < L) ? mw[(rb + row) * L + l] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z) + rb * T + c0, 0, (int)((int64_t)vrows * T * 4), 0x00020000);
    const int nsteps = (vrows + 1) >> 1;
    constexpr int NB = 16;                                            // steps per batch: their loads are issued together
    for (int j0 = 0; j0 < nsteps; j0 += NB) {
      float zv[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) zv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, voff + (unsigned)(j0 + q) * step, 0, 0));
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int row = 2 * (j0 + q) + h;                             // < 128; rows past the chunk: logit 0 (out of range), (0, 0), weight 0
        const float2 st = s_stat[row];
        const float a = s_mw[row * 32 + i];
        const float qv = __builtin_amdgcn_exp2f((zv[q] - st.x) * 1.4426950408889634f) * st.y;
        const float pv = __builtin_amdgcn_fmed3f(qv, 0.f, 3.4028234663852886e38f);   // nan_to_num
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pv, acc, 0, 0, 0);
      }
    }
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int l = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (l < L) pbar[(int64_t)l * T + c0 + i] += acc[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------- low-rank softmax backward
// Backward of softmax + top-K + batch-mean loss from RECOMPUTED LOGITS, for the chunked per-vertex path:
//   p = exp(z - m) / s (row stats saved by the forward),  g[r,t] = sum_l mw[r,l] G[l,t]  (+ dq_k at the top-K slots)
//   dz[r,t] = p (g - dot_r),  dot_r = sum_t p g,   db[t] += sum_r dz[r,t]
// Tiling: a block owns 64 rows x a range of columns; a thread owns one column at a time, keeps G[:,t] (L values) in
// registers and reads mw / row stats from LDS, so G is read once per 64 rows instead of once per row (the first
// version re-read all of G for every row: 10.6 ms per 2048-row chunk at T = 2^19; this form is bound by the two
// streaming passes over z).
constexpr int kSbRows = 64;
constexpr int kSbCols = 8192;      // columns per block (32 trips of 256)

__device__ __forceinline__ float prob_of(float z, float m, float s) {
  const float q = expf(z - m) / s;
  return (q != q) ? 0.f : (q > 3.4028234663852886e38f ? 3.4028234663852886e38f : q);   // nan_to_num; s = NaN marks a NaN row
}

template <bool APPLY, int LMAX>
__global__ void __launch_bounds__(256)
softmax_bwd_tile_kernel(float* __restrict__ Z, const float* __restrict__ rowstat, const float* __restrict__ mw,
                        const float* __restrict__ G, int L, float* __restrict__ dot, float* __restrict__ db, int64_t U,
                        int64_t T) {
  // Row-wise operands (row max / sum / dot, the L multiplicity weights) are wave-UNIFORM: they are read straight from
  // global memory with uniform indices, which hipcc turns into scalar loads (SGPR operands of the FMAs) — the first
  // tiled version broadcast them out of LDS and was bound by 16+ ds_reads per element.  A thread owns CPT columns.
  constexpr int CPT = 4;
  __shared__ float red[4][kSbRows];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * kSbRows;
  const int nr = (int)((U - r0) < kSbRows ? (U - r0) : kSbRows);
  const int lane = tid & 63, wave = tid >> 6;
  if (!APPLY) {
    red[wave][lane] = 0.f;                       // kSbRows == 64: one entry per lane; each wave owns its row of `red`
    __syncthreads();
  }
  const int64_t c0 = (int64_t)blockIdx.x * kSbCols;
  const int64_t c1 = (c0 + kSbCols < T) ? c0 + kSbCols : T;
  for (int64_t tb = c0; tb < c1; tb += 256 * CPT) {
    int64_t tc[CPT];
    bool okc[CPT];
    float Gc[CPT][LMAX];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
      tc[q] = tb + q * 256 + tid;
      okc[q] = tc[q] < c1;
      const int64_t ts = okc[q] ? tc[q] : c1 - 1;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) Gc[q][l] = l < L ? G[(int64_t)l * T + ts] : 0.f;
    }
    float colsum[CPT] = {0.f, 0.f, 0.f, 0.f};
    constexpr int RU = 4;                        // rows per trip: their logits are loaded together (the loop is latency-bound otherwise)
    for (int rb = 0; rb < nr; rb += RU) {
      float zv[RU][CPT];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int64_t row = r0 + (rb + u < nr ? rb + u : nr - 1);
#pragma unroll
        for (int q = 0; q < CPT; ++q) zv[u][q] = okc[q] ? Z[row * T + tc[q]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int r = rb + u;
        if (r >= nr) break;
        const int64_t row = r0 + r;
        const float m = rowstat[2 * row], rs = 1.0f / rowstat[2 * row + 1];  // uniform -> scalar loads, one reciprocal per row
        const float dr = APPLY ? dot[row] : 0.f;
        float mwr[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) mwr[l] = (l < L) ? mw[row * L + l] : 0.f;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
          if (!okc[q]) continue;
          const float p = prob_fast(zv[u][q], m, rs);
          float g = 0.f;
#pragma unroll
          for (int l = 0; l < LMAX; ++l) g += mwr[l] * Gc[q][l];
          if (APPLY) {
            const float dz = p * (g - dr);
            Z[row * T + tc[q]] = dz;
            colsum[q] += dz;
          } else {
            part += p * g;
          }
        }
        if (!APPLY) {                            // wave-level sum of this trip's 256 columns, kept per (wave, row) in LDS
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
          if (lane == 0) red[wave][r] += part;
        }
      }
    }
    if (APPLY && db) {
#pragma unroll
      for (int q = 0; q < CPT; ++q)
        if (okc[q]) atomicAdd(db + tc[q], colsum[q]);
    }
  }
  if (!APPLY) {
    __syncthreads();
    if (tid < nr) atomicAdd(dot + r0 + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
  }
}

// The same two passes with the low-rank product on the matrix cores (T % 32 == 0).  g = mw (rows x L) * G (L x T) is a
// GEMM with a contraction of L <= 32: eight 32x32x2 MFMAs per 32 x 32 tile instead of 16 VALU FMAs per logit, which
// leaves ~8 VALU instructions per logit (probability, product, sums) and makes both passes streaming-bound.
// A wave owns 32 rows for a range of columns: its A operand (the rows' weights) and the per-row constants of the MFMA
// result layout (lane (i, h), register r <-> row (r & 3) + 8 (r >> 2) + 4 h, column i) stay in registers; logits move
// through buffer descriptors whose record count ends at the last valid row, so rows past U read 0 and are not written.
constexpr int kLrRows = 128;       // 4 waves x 32 rows
constexpr int kLrCols = 4096;      // columns per block: 128 tiles of 32

template <bool APPLY, int LP>
__global__ void __launch_bounds__(256)
softmax_bwd_mfma_kernel(float* __restrict__ Z, const float* __restrict__ rowstat, const float* __restrict__ mw,
                        const float* __restrict__ G, int L, float* __restrict__ dot, float* __restrict__ db, int64_t U,
                        int64_t T) {
  __shared__ float colsum[APPLY ? kLrCols : 1];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.y * kLrRows + wave * 32;
  const int64_t c0 = (int64_t)blockIdx.x * kLrCols;
  const int ntile = (int)(((c0 + kLrCols < T) ? kLrCols : (T - c0)) / 32);
  const bool use_db = APPLY && db != nullptr;
  if (use_db) {
    for (int c = tid; c < kLrCols; c += 256) colsum[c] = 0.f;
    __syncthreads();
  }
  const int64_t vrows = U - r0 < 0 ? 0 : (U - r0 < 32 ? U - r0 : 32);       // valid rows of this wave
  if (vrows > 0 || use_db) {
    // A operand: A[m = i][k = 2 j + h] = mw[r0 + i][2 j + h]
    float aop[LP / 2];
#pragma unroll
    for (int j = 0; j < LP / 2; ++j) aop[j] = (i < vrows && 2 * j + h < L) ? mw[(r0 + i) * L + 2 * j + h] : 0.f;
    float rm[16], rrs[16], rdr[16], dacc[16];
    unsigned voff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = (r & 3) + 8 * (r >> 2) + 4 * h;
      const bool ok = lr < vrows;
      rm[r] = ok ? rowstat[2 * (r0 + lr)] : 0.f;
      rrs[r] = ok ? 1.0f / rowstat[2 * (r0 + lr) + 1] : 0.f;
      rdr[r] = (APPLY && ok) ? dot[r0 + lr] : 0.f;
      dacc[r] = 0.f;
      voff[r] = 4u * ((unsigned)lr * (unsigned)T + (unsigned)i);
    }
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(Z + r0 * T + c0, 0, (int)(vrows * T * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G) + c0, 0, 0x7fffffff, 0x00020000);
    const unsigned goff = 4u * ((unsigned)h * (unsigned)T + (unsigned)i);
    for (int tb = 0; tb < ntile; ++tb) {
      const unsigned so = 128u * (unsigned)tb;
      float zv[16], bop[LP / 2];
#pragma unroll
      for (int r = 0; r < 16; ++r) zv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, voff[r], so, 0));
#pragma unroll
      for (int j = 0; j < LP / 2; ++j)
        bop[j] = (2 * j + h < L) ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, goff + 8u * (unsigned)j * (unsigned)T, so, 0)) : 0.f;
      f32x16_t g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < LP / 2; ++j) g = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[j], bop[j], g, 0, 0, 0);
      float cs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float q = __builtin_amdgcn_exp2f((zv[r] - rm[r]) * 1.4426950408889634f) * rrs[r];
        const float p = __builtin_amdgcn_fmed3f(q, 0.f, 3.4028234663852886e38f);          // nan_to_num: NaN -> 0, inf -> max
        if (APPLY) {
          const float dz = p * (g[r] - rdr[r]);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), rz, voff[r], so, 0);
          cs += dz;
        } else {
          dacc[r] += p * g[r];
        }
      }
      if (use_db) atomicAdd(&colsum[tb * 32 + i], cs);                 // LDS float add; both row halves and all waves
    }
    if (!APPLY) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = dacc[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (i == 0 && lr < vrows) atomicAdd(dot + r0 + lr, v);
      }
    }
  }
  if (use_db) {
    __syncthreads();
    for (int c = tid; c < ntile * 32; c += 256) atomicAdd(db + c0 + c, colsum[c]);
  }
}

// top-K part of the row dots; stashes p at the top-K slots (the logits are overwritten by the apply pass)
__global__ void softmax_bwd_topk_dot_kernel(const float* __restrict__ Z, const float* __restrict__ rowstat,
                                            const float* __restrict__ dq, const int32_t* __restrict__ topi,
                                            float* __restrict__ pk, float* __restrict__ dot, int64_t U, int64_t T, int K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= U * K) return;
  const int64_t r = e / K;
  const float p = prob_of(Z[r * T + topi[e]], rowstat[2 * r], rowstat[2 * r + 1]);
  pk[e] = p;
  atomicAdd(dot + r, p * dq[e]);
}

// dot[r] = sum_k p_k dq_k from the top-K probabilities the FORWARD saved (topk_val: the very values exp(z_k - max) / sum of these
// logits) — written, not added: it also replaces the clear of `dot`, and it does not touch the logits (softmax_bwd_topk_dot_kernel
// re-read them at random: 1.4 ms per chunk beside the GEMM stream, for 16 K numbers)
__global__ void softmax_bwd_dot_init_kernel(const float* __restrict__ pk, const float* __restrict__ dq, float* __restrict__ dot,
                                            int64_t U, int K) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= U) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += pk[r * K + k] * dq[r * K + k];
  dot[r] = s;
}

__global__ void softmax_bwd_topk_fix_kernel(float* __restrict__ dZ, const float* __restrict__ dq, const int32_t* __restrict__ topi,
                                            const float* __restrict__ pk, float* __restrict__ db, int64_t U, int64_t T, int K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= U * K) return;
  const int64_t r = e / K;
  const float add = pk[e] * dq[e];
  dZ[r * T + topi[e]] += add;                  // the K slots of a row are distinct: no race
  if (db) atomicAdd(db + topi[e], add);
}

// verts[u] = (gx, gy) as fp32 with u = gy * vstride + gx   (the HPD input is the raw integer vertex, models.py:416-418)
__global__ void vertex_coords_kernel(float2* __restrict__ verts, int64_t u0, int64_t count, int vstride) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t u = u0 + i;
  verts[i] = make_float2((float)(u % vstride), (float)(u / vstride));
}

template <int KMAX>
__device__ __forceinline__ void blend_w(const float* q, int K, int blend, float* w) {
  if (blend == GNGF_BLEND_RAW) { for (int k = 0; k < K; ++k) w[k] = q[k]; return; }
  if (blend == GNGF_BLEND_SOFTMAX) {
    float m = q[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, q[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) { w[k] = expf(q[k] - m); s += w[k]; }
    for (int k = 0; k < K; ++k) w[k] = w[k] / s;
    return;
  }
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += q[k];
  for (int k = 0; k < K; ++k) w[k] = q[k] / s;
}

__global__ void blend_fwd_kernel(const float* __restrict__ q, float* __restrict__ w, int64_t U, int K, int blend) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  float qq[GNGF_MAX_TOPK], ww[GNGF_MAX_TOPK];
  for (int k = 0; k < K; ++k) qq[k] = q[u * K + k];
  blend_w<GNGF_MAX_TOPK>(qq, K, blend, ww);
  for (int k = 0; k < K; ++k) w[u * K + k] = ww[k];
}

__global__ void blend_bwd_kernel(const float* __restrict__ q, const float* __restrict__ dw, float* __restrict__ dq,
                                 int64_t U, int K, int blend) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  float qq[GNGF_MAX_TOPK], ww[GNGF_MAX_TOPK], d[GNGF_MAX_TOPK];
  for (int k = 0; k < K; ++k) { qq[k] = q[u * K + k]; d[k] = dw[u * K + k]; }
  blend_w<GNGF_MAX_TOPK>(qq, K, blend, ww);
  if (blend == GNGF_BLEND_SOFTMAX) {
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += ww[k] * d[k];
    for (int k = 0; k < K; ++k) d[k] = ww[k] * (d[k] - dot);
  } else if (blend == GNGF_BLEND_NORM) {
    float s = 0.f, dqs = 0.f;
    for (int k = 0; k < K; ++k) { s += qq[k]; dqs += d[k] * qq[k]; }
    const float inv = 1.0f / s;
    for (int k = 0; k < K; ++k) d[k] = d[k] * inv - dqs * inv * inv;
  }
  for (int k = 0; k < K; ++k) dq[u * K + k] = d[k];
}

// counts[l][vid] += 1 for the 4 corners of every (pixel, level): multiplicities of the per-vertex rows inside the
// reference's (P,L,4,T) tensor (for the batch-mean distribution of utils.py:138).
__global__ void __launch_bounds__(256)
vertex_multiplicity_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, int32_t* __restrict__ counts,
                           int64_t total, int L, int vstride, int64_t NV) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t p = gid / L;
  const int l = (int)(gid - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    int64_t vid = (int64_t)(cell.gy + (v >> 1)) * vstride + (cell.gx + (v & 1));
    vid = vid < 0 ? 0 : (vid >= NV ? NV - 1 : vid);
    atomicAdd(counts + (int64_t)l * NV + vid, 1);
  }
}

// counts (L,NV) int32 -> mw (NV,L) fp32 = counts / denom
__global__ void multiplicity_weights_kernel(const int32_t* __restrict__ counts, float* __restrict__ mw, int64_t NV, int L,
                                            float denom) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NV * L) return;
  const int64_t u = i / L;
  const int l = (int)(i - u * L);
  mw[i] = (float)counts[(int64_t)l * NV + u] / denom;
}

// vid (P,L,4) int64 of every instance, and the reference-shaped expansions of per-vertex (NV,K) tables.
__global__ void __launch_bounds__(256)
expand_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, const int32_t* __restrict__ src_i,
              const float* __restrict__ src_f, int64_t* __restrict__ vid_out, int64_t* __restrict__ out_i,
              float* __restrict__ out_f, int64_t total, int L, int K, int vstride, int64_t NV) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over P*L*4
  if (gid >= total) return;
  const int v = (int)(gid & 3);
  const int64_t pl = gid >> 2;
  const int64_t p = pl / L;
  const int l = (int)(pl - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
  int64_t vid = (int64_t)(cell.gy + (v >> 1)) * vstride + (cell.gx + (v & 1));
  vid = vid < 0 ? 0 : (vid >= NV ? NV - 1 : vid);
  if (vid_out) vid_out[gid] = vid;
  for (int k = 0; k < K; ++k) {
    if (out_i) out_i[gid * K + k] = (int64_t)src_i[vid * K + k];
    if (out_f) out_f[gid * K + k] = src_f[vid * K + k];
  }
}

// K = 4, indices only (the reference's (P,L,4,K) int64 return tensor, 2 GiB per step at 2^20 pixels): the four slots of a vertex
// are ONE 16-byte load and leave as two 16-byte non-temporal stores — 32 consecutive bytes per lane, whole lines per wave —
// instead of four 8-byte stores from a runtime loop (0.96 ms -> streaming-store rate).
__global__ void __launch_bounds__(256)
expand_idx4_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, const int32_t* __restrict__ src_i,
                   int64_t* __restrict__ vid_out, int64_t* __restrict__ out_i, int64_t total, int L, int vstride, int64_t NV) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over P*L*4
  if (gid >= total) return;
  const int v = (int)(gid & 3);
  const int64_t pl = gid >> 2;
  const int64_t p = pl / L;
  const int l = (int)(pl - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
  int64_t vid = (int64_t)(cell.gy + (v >> 1)) * vstride + (cell.gx + (v & 1));
  vid = vid < 0 ? 0 : (vid >= NV ? NV - 1 : vid);
  if (vid_out) vid_out[gid] = vid;
  const int4 s4 = reinterpret_cast<const int4*>(src_i)[vid];
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  ll2* o = reinterpret_cast<ll2*>(out_i + gid * 4);
  __builtin_nontemporal_store((ll2){(long long)s4.x, (long long)s4.y}, o);
  __builtin_nontemporal_store((ll2){(long long)s4.z, (long long)s4.w}, o + 1);
}

}  // namespace gngf

using namespace gngf;

extern "C" int gngf_softmax_topk(float* logits_probs, float* topk_val, int32_t* topk_idx, float* rowstat, int64_t U, int64_t T,
                                 int K, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && K > 0 && K <= GNGF_MAX_TOPK && K <= T && T < INT_MAX);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(logits_probs && topk_val && topk_idx);
  const size_t smem = ((size_t)2 * K * kRowBlock + 16) * sizeof(float);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  softmax_topk_kernel<<<dim3((unsigned)U), dim3(kRowBlock), smem, as_stream(stream)>>>(logits_probs, topk_val, topk_idx, T, K, 1,
                                                                                       rowstat);
  GNGF_RETURN_LAUNCH();
}

// Streaming forward of the per-vertex HPD tail: logits (U,T) are only READ.  topk_val/topk_idx (U,K), rowstat (U,2);
// when mw (U,L) is given, pbar (L,T) += mw^T * softmax(logits).
static int launch_pbar(const float* logits, const float* rowstat, const float* mw, int L, float* pbar, int64_t U, int64_t T, hipStream_t s);

extern "C" int gngf_logits_topk_pbar(const float* logits, float* topk_val, int32_t* topk_idx, float* rowstat, const float* mw,
                                     int L, float* pbar, int64_t U, int64_t T, int K, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && K > 0 && K <= GNGF_MAX_TOPK && K <= T && T < INT_MAX && L >= 0 && L <= GNGF_MAX_LEVELS);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(logits && topk_val && topk_idx && rowstat && (L == 0 || (mw && pbar)));
  hipStream_t s = as_stream(stream);
  const size_t smem = ((size_t)2 * K * kRowBlock + 16) * sizeof(float);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(logits_stats_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  logits_stats_topk_kernel<<<dim3((unsigned)U), dim3(kRowBlock), smem, s>>>(logits, topk_val, topk_idx, rowstat, T, K);
  if (L > 0) return launch_pbar(logits, rowstat, mw, L, pbar, U, T, s);
  GNGF_RETURN_LAUNCH();
}

// The two halves of gngf_logits_topk_pbar as calls of their own, for logits whose row partials came out of the GEMM's epilogue
// (gngf_linear_fwd_rowstats): statistics + top-K from the partials (T % 64 == 0: (U, T / 64) pairs), and the batch-mean accumulation.
extern "C" int gngf_rowstats_topk(const float* logits, const float* rowparts, float* topk_val, int32_t* topk_idx, float* rowstat,
                                  int64_t U, int64_t T, int K, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && T % 64 == 0 && K > 0 && K <= GNGF_MAX_TOPK && K <= T && T < INT_MAX && (int64_t)K * 64 <= T);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(logits && rowparts && topk_val && topk_idx && rowstat);
  const size_t smem = ((size_t)2 * K * kRowBlock + 16 + GNGF_MAX_TOPK) * sizeof(float);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowstats_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  rowstats_topk_kernel<<<dim3((unsigned)U), dim3(kRowBlock), smem, as_stream(stream)>>>(
      logits, reinterpret_cast<const float2*>(rowparts), (int)(T / 64), topk_val, topk_idx, rowstat, T, K);
  GNGF_RETURN_LAUNCH();
}

static int launch_pbar(const float* logits, const float* rowstat, const float* mw, int L, float* pbar, int64_t U, int64_t T, hipStream_t s) {
  const dim3 grid((unsigned)ceil_div(T, 1024));
  if (T % 32 == 0 && T < (1 << 22) && L > 4)          // 128 rows * T * 4 B inside 31-bit offsets
    pbar_mfma_kernel<<<dim3((unsigned)ceil_div(T, 128)), dim3(256), 0, s>>>(logits, rowstat, mw, L, pbar, U, T);
  else if (L <= 4) pbar_accum_kernel<4><<<grid, dim3(256), 0, s>>>(logits, rowstat, mw, L, pbar, U, T);
  else if (L <= 16) pbar_accum_kernel<16><<<grid, dim3(256), 0, s>>>(logits, rowstat, mw, L, pbar, U, T);
  else pbar_accum_kernel<32><<<grid, dim3(256), 0, s>>>(logits, rowstat, mw, L, pbar, U, T);
  return (int)hipGetLastError();
}

extern "C" int gngf_pbar_accumulate(const float* logits, const float* rowstat, const float* mw, int L, float* pbar, int64_t U,
                                    int64_t T, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && T < INT_MAX && L > 0 && L <= GNGF_MAX_LEVELS);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(logits && rowstat && mw && pbar);
  return launch_pbar(logits, rowstat, mw, L, pbar, U, T, as_stream(stream));
}

// DifferentiableTopk.forward alone (models.py:11): top-K of arbitrary rows, values sorted descending, ties -> lower index.
extern "C" int gngf_topk(const float* x, float* topk_val, int32_t* topk_idx, int64_t U, int64_t T, int K, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && K > 0 && K <= GNGF_MAX_TOPK && K <= T && T < INT_MAX);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(x && topk_val && topk_idx);
  const size_t smem = ((size_t)2 * K * kRowBlock + 16) * sizeof(float);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  softmax_topk_kernel<<<dim3((unsigned)U), dim3(kRowBlock), smem, as_stream(stream)>>>(const_cast<float*>(x), topk_val, topk_idx,
                                                                                       T, K, 0, nullptr);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_softmax_bwd(const float* probs, const float* dq, const int32_t* topk_idx, const float* gdense,
                                const float* mw, const float* G, int L, float* dlogits, int64_t U, int64_t T, int K,
                                void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && K >= 0 && K <= GNGF_MAX_TOPK && L >= 0 && L <= GNGF_MAX_LEVELS);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(probs && dlogits && (K == 0 || topk_idx) && ((mw == nullptr) == (G == nullptr)));
  softmax_bwd_kernel<<<dim3((unsigned)U), dim3(kRowBlock), 0, as_stream(stream)>>>(probs, dq, topk_idx, gdense, mw, G, L,
                                                                                    dlogits, T, K);
  GNGF_RETURN_LAUNCH();
}

// Softmax(+top-K, + batch-mean loss) backward from recomputed logits, in place: logits_dz (U,T) logits in, d logits out.
// rowstat (U,2) from gngf_softmax_topk; dq/topk_idx (U,K) or K = 0; mw (U,L) and G (L,T) or L = 0;
// db (T) += column sums of d logits (NULL: skipped); scratch: (U + U*K) floats.
extern "C" int gngf_softmax_bwd_lowrank(float* logits_dz, const float* rowstat, const float* dq, const int32_t* topk_idx,
                                        const float* mw, const float* G, int L, float* db, float* scratch, const float* topk_p,
                                        int64_t U, int64_t T, int K, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && K >= 0 && K <= GNGF_MAX_TOPK && L >= 0 && L <= GNGF_MAX_LEVELS);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(logits_dz && rowstat && scratch && (K == 0 || (dq && topk_idx)) && (L == 0 || (mw && G)));
  hipStream_t s = as_stream(stream);
  float* dot = scratch;
  float* pk = scratch + U;
  const bool saved_p = K > 0 && topk_p != nullptr;       // the forward's top-K probabilities: no second look at the logits
  if (saved_p) {
    softmax_bwd_dot_init_kernel<<<dim3((unsigned)ceil_div(U, 256)), dim3(256), 0, s>>>(topk_p, dq, dot, U, K);
  } else {
    hipError_t e = zero_async(dot, sizeof(float) * (size_t)U, s);
    if (e != hipSuccess) return (int)e;
  }
  const bool mfma = (T % 32 == 0) && T < (1 << 24) && L <= 32;      // 32 rows * T * 4 B and L * T * 4 B inside 32-bit offsets
  const dim3 grid(mfma ? (unsigned)ceil_div(T, kLrCols) : (unsigned)ceil_div(T, kSbCols),
                  mfma ? (unsigned)ceil_div(U, kLrRows) : (unsigned)ceil_div(U, kSbRows));
  const bool small = L <= 4;
  if (L > 0) {
    if (mfma) {
      if (small) softmax_bwd_mfma_kernel<false, 4><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, nullptr, U, T);
      else if (L <= 16) softmax_bwd_mfma_kernel<false, 16><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, nullptr, U, T);
      else softmax_bwd_mfma_kernel<false, 32><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, nullptr, U, T);
    } else if (small) softmax_bwd_tile_kernel<false, 4><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, nullptr, U, T);
    else if (L <= 16) softmax_bwd_tile_kernel<false, 16><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, nullptr, U, T);
    else softmax_bwd_tile_kernel<false, 32><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, nullptr, U, T);
  }
  if (K > 0 && !saved_p)
    softmax_bwd_topk_dot_kernel<<<dim3((unsigned)ceil_div(U * K, 256)), dim3(256), 0, s>>>(logits_dz, rowstat, dq, topk_idx, pk,
                                                                                          dot, U, T, K);
  if (mfma) {
    if (small) softmax_bwd_mfma_kernel<true, 4><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, db, U, T);
    else if (L <= 16) softmax_bwd_mfma_kernel<true, 16><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, db, U, T);
    else softmax_bwd_mfma_kernel<true, 32><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, db, U, T);
  } else if (small) softmax_bwd_tile_kernel<true, 4><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, db, U, T);
  else if (L <= 16) softmax_bwd_tile_kernel<true, 16><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, db, U, T);
  else softmax_bwd_tile_kernel<true, 32><<<grid, dim3(256), 0, s>>>(logits_dz, rowstat, mw, G, L, dot, db, U, T);
  if (K > 0)
    softmax_bwd_topk_fix_kernel<<<dim3((unsigned)ceil_div(U * K, 256)), dim3(256), 0, s>>>(logits_dz, dq, topk_idx, saved_p ? topk_p : pk,
                                                                                          db, U, T, K);
  GNGF_RETURN_LAUNCH();
}

// The row dots of that backward alone, for gngf_hpd_bwd_fused (linear.hip), which forms dz in the loaders of the dW / dh GEMMs:
//   dot[r] = sum_k topk_p[r,k] dq[r,k] + sum_t p[r,t] (mw G)[r,t]          one read of the logits, nothing written but dot (U)
extern "C" int gngf_hpd_bwd_dot(const float* logits, const float* rowstat, const float* dq, const float* topk_p, const float* mw,
                                const float* G, int L, float* dot, int64_t U, int64_t T, int K, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && T > 0 && T % 32 == 0 && T < (1 << 24) && K >= 0 && K <= GNGF_MAX_TOPK && L >= 0 && L <= 32);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(logits && rowstat && dot && (K == 0 || (dq && topk_p)) && (L == 0 || (mw && G)));
  hipStream_t s = as_stream(stream);
  if (K > 0) {
    softmax_bwd_dot_init_kernel<<<dim3((unsigned)ceil_div(U, 256)), dim3(256), 0, s>>>(topk_p, dq, dot, U, K);
  } else {
    hipError_t e = zero_async(dot, sizeof(float) * (size_t)U, s);
    if (e != hipSuccess) return (int)e;
  }
  if (L > 0) {
    const dim3 grid((unsigned)ceil_div(T, kLrCols), (unsigned)ceil_div(U, kLrRows));
    float* z = const_cast<float*>(logits);                                         // (the <false> instance only reads)
    if (L <= 4) softmax_bwd_mfma_kernel<false, 4><<<grid, dim3(256), 0, s>>>(z, rowstat, mw, G, L, dot, nullptr, U, T);
    else if (L <= 16) softmax_bwd_mfma_kernel<false, 16><<<grid, dim3(256), 0, s>>>(z, rowstat, mw, G, L, dot, nullptr, U, T);
    else softmax_bwd_mfma_kernel<false, 32><<<grid, dim3(256), 0, s>>>(z, rowstat, mw, G, L, dot, nullptr, U, T);
  }
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_vertex_coords(float* verts, int64_t u0, int64_t count, int vstride, void* stream) {
  GNGF_CHECK_ARG(count >= 0 && u0 >= 0 && vstride > 0);
  if (count == 0) return 0;
  GNGF_CHECK_ARG(verts);
  vertex_coords_kernel<<<dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, as_stream(stream)>>>(
      reinterpret_cast<float2*>(verts), u0, count, vstride);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_blend_fwd(const float* q, float* w, int64_t U, int K, int blend, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && K > 0 && K <= GNGF_MAX_TOPK && blend >= 0 && blend <= 2);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(q && w);
  blend_fwd_kernel<<<dim3((unsigned)ceil_div(U, 256)), dim3(256), 0, as_stream(stream)>>>(q, w, U, K, blend);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_blend_bwd(const float* q, const float* dw, float* dq, int64_t U, int K, int blend, void* stream) {
  GNGF_CHECK_ARG(U >= 0 && K > 0 && K <= GNGF_MAX_TOPK && blend >= 0 && blend <= 2);
  if (U == 0) return 0;
  GNGF_CHECK_ARG(q && dw && dq);
  blend_bwd_kernel<<<dim3((unsigned)ceil_div(U, 256)), dim3(256), 0, as_stream(stream)>>>(q, dw, dq, U, K, blend);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_vertex_multiplicity(const float* xy, const int32_t* n_ls, int32_t* counts, int64_t P, int L, int vstride,
                                        int64_t NV, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && vstride > 0 && NV > 0);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(xy && n_ls && counts);
  const int64_t total = P * L;
  vertex_multiplicity_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream)>>>(
      reinterpret_cast<const float2*>(xy), n_ls, counts, total, L, vstride, NV);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_multiplicity_weights(const int32_t* counts, float* mw, int64_t NV, int L, float denom, void* stream) {
  GNGF_CHECK_ARG(NV >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && denom > 0.f);
  if (NV == 0) return 0;
  GNGF_CHECK_ARG(counts && mw);
  multiplicity_weights_kernel<<<dim3((unsigned)ceil_div(NV * L, 256)), dim3(256), 0, as_stream(stream)>>>(counts, mw, NV, L, denom);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_expand_vertex_table(const float* xy, const int32_t* n_ls, const int32_t* src_idx, const float* src_val,
                                        int64_t* vid_out, int64_t* out_idx, float* out_val, int64_t P, int L, int K,
                                        int vstride, int64_t NV, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && K >= 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(xy && n_ls && (!out_idx || src_idx) && (!out_val || src_val));
  const int64_t total = P * L * 4;
  if (K == 4 && out_idx && !out_val && (reinterpret_cast<uintptr_t>(src_idx) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_idx) & 15) == 0) {
    expand_idx4_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream)>>>(
        reinterpret_cast<const float2*>(xy), n_ls, src_idx, vid_out, out_idx, total, L, vstride, NV);
    GNGF_RETURN_LAUNCH();
  }
  expand_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream)>>>(
      reinterpret_cast<const float2*>(xy), n_ls, src_idx, src_val, vid_out, out_idx, out_val, total, L, K, vstride, NV);
  GNGF_RETURN_LAUNCH();
}
