// Generic fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain) for the
// dense layers that are NOT covered by the fused decoder kernel: the HashProbDistribution MLP
// (reference models.py:80-88,105-106) forward/backward, and decoders with non-default widths
// (models.py:382-392).  One kernel, C (+)= opA(A) * opB(B) with optional bias / activation epilogue and an
// optional activation-backward mask fused into the A loads (dZ = dY * act'(Y)).
//
// Tile: 64x64 per 256-thread block, one 32x32 accumulator per wave, BK = 16 staged through LDS with a
// +1 padded stride (bank = (17*row + k) % 32: conflict-free ds_read_b32 for the one-float A/B operands).
#include "gngf_common.h"
#include <utility>

namespace gngf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SIGMOID = 3 };

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == ACT_RELU) return fmaxf(z, 0.f);
  if (act == ACT_LEAKY) return z > 0.f ? z : z * 0.01f;
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
  return z;
}
// derivative expressed through the OUTPUT y of the activation
__device__ __forceinline__ float act_bwd_from_y(float y, int act) {
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == ACT_LEAKY) return y > 0.f ? 1.f : 0.01f;
  if (act == ACT_SIGMOID) return y * (1.f - y);
  return 1.f;
}

constexpr int BM = 64, BN = 64, BK = 16, LDSS = BK + 1;

// A(m,k) = TA ? A[k*lda + m] : A[m*lda + k]     (optionally * act'(Ymask at the same index))
// B(k,n) = TB ? B[n*ldb + k] : B[k*ldb + n]
// grid = (ceil(N/64), ceil(M/64), splitk); each z-slice handles kchunk of the contraction and, when
// splitk > 1, adds its partial tile with float atomics (C must be zero-filled; bias is added by slice 0).
template <bool TA, bool TB>
__global__ void __launch_bounds__(256)
gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
            int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
            const float* __restrict__ bias, int act, const float* __restrict__ amask, int mask_act,
            int64_t kchunk, int atomic_out) {
  __shared__ float As[BM * LDSS];
  __shared__ float Bs[BN * LDSS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  f32x16 acc = {0};
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    // stage A tile [64][16] and B tile [64][16]; 1024 elements each, 4 per thread
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = tid + e * 256;
      // A: choose the thread->element map so that the global reads are contiguous for either storage
      int am, ak;
      if (TA) { am = li & 63; ak = li >> 6; } else { ak = li & 15; am = li >> 4; }
      const int64_t gm = m0 + am, gk = k0 + ak;
      float av = 0.f;
      if (gm < M && gk < kend) {
        const int64_t ai = TA ? gk * lda + gm : gm * lda + gk;
        av = A[ai];
        if (amask) av *= act_bwd_from_y(amask[ai], mask_act);
      }
      As[am * LDSS + ak] = av;
      int bn, bk;
      if (TB) { bk = li & 15; bn = li >> 4; } else { bn = li & 63; bk = li >> 6; }
      const int64_t gn = n0 + bn, gk2 = k0 + bk;
      float bv = 0.f;
      if (gn < N && gk2 < kend) bv = TB ? B[gn * ldb + gk2] : B[gk2 * ldb + gn];
      Bs[bn * LDSS + bk] = bv;
    }
    __syncthreads();
    const float* ap = As + (wm * 32 + (lane & 31)) * LDSS + (lane >> 5);
    const float* bp = Bs + (wn * 32 + (lane & 31)) * LDSS + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
    __syncthreads();
  }
  const int64_t col = n0 + wn * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row >= M) continue;
    float v = acc[r] + bv;
    if (atomic_out) {
      atomicAdd(C + row * ldc + col, v);
    } else {
      C[row * ldc + col] = act_fwd(v, act);
    }
  }
}

// 128x128 tile variant for the large HPD GEMMs (logits, dW_last, dh_last): 4 waves as 2x2, each wave a 64x64 sub-tile =
// 2x2 MFMA accumulators, so every ds_read feeds two MFMAs; operands are staged k-major ([BK][128+4]) with 16-byte global
// loads along whichever axis is contiguous in memory.
constexpr int BM2 = 128, BN2 = 128, LDS2 = BM2 + 4;

template <bool TRANS>   // TRANS: element (r, k) lives at src[k*ld + r] (contiguous along r); else src[r*ld + k]
__device__ __forceinline__ void stage128(float* S, const float* __restrict__ src, const float* __restrict__ mask, int mask_act,
                                         int64_t r0, int64_t R, int64_t k0, int64_t kend, int64_t ld, int tid) {
  // 128 x 16 elements = 512 float4; 2 per thread
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int li = tid + e * 256;
    if (TRANS) {
      const int kk = li >> 5, rq = (li & 31) * 4;                  // 16 k rows x 32 float4 along r
      const int64_t gr = r0 + rq, gk = k0 + kk;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (gk < kend) {
        const int64_t o = gk * ld + gr;
        if (gr + 3 < R && ((o & 3) == 0)) {
          v = *reinterpret_cast<const float4*>(src + o);
          if (mask) {
            const float4 mv = *reinterpret_cast<const float4*>(mask + o);
            v.x *= act_bwd_from_y(mv.x, mask_act); v.y *= act_bwd_from_y(mv.y, mask_act);
            v.z *= act_bwd_from_y(mv.z, mask_act); v.w *= act_bwd_from_y(mv.w, mask_act);
          }
        } else {
          float t4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            t4[q] = (gr + q < R) ? src[o + q] : 0.f;
            if (mask && gr + q < R) t4[q] *= act_bwd_from_y(mask[o + q], mask_act);
          }
          v = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
      *reinterpret_cast<float4*>(S + kk * LDS2 + rq) = v;
    } else {
      const int rr = li >> 2, kq = (li & 3) * 4;                   // 128 rows x 4 float4 along k
      const int64_t gr = r0 + rr, gk = k0 + kq;
      float t4[4] = {0.f, 0.f, 0.f, 0.f};
      if (gr < R) {
        const int64_t o = gr * ld + gk;
        if (gk + 3 < kend && ((o & 3) == 0)) {
          const float4 v = *reinterpret_cast<const float4*>(src + o);
          t4[0] = v.x; t4[1] = v.y; t4[2] = v.z; t4[3] = v.w;
          if (mask) {
            const float4 mv = *reinterpret_cast<const float4*>(mask + o);
            t4[0] *= act_bwd_from_y(mv.x, mask_act); t4[1] *= act_bwd_from_y(mv.y, mask_act);
            t4[2] *= act_bwd_from_y(mv.z, mask_act); t4[3] *= act_bwd_from_y(mv.w, mask_act);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (gk + q < kend) { t4[q] = src[o + q]; if (mask) t4[q] *= act_bwd_from_y(mask[o + q], mask_act); }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) S[(kq + q) * LDS2 + rr] = t4[q];
    }
  }
}

template <bool TA, bool TB>
__global__ void __launch_bounds__(256)
gemm128_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
               int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
               const float* __restrict__ bias, int act, const float* __restrict__ amask, int mask_act,
               int64_t kchunk, int atomic_out) {
  __shared__ float As[BK * LDS2];
  __shared__ float Bs[BK * LDS2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.y * BM2, n0 = (int64_t)blockIdx.x * BN2;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    // A(m,k): TA -> contiguous along m.   B(k,n): TB -> B[n*ldb + k] contiguous along k, else contiguous along n.
    stage128<TA>(As, A, amask, mask_act, m0, M, k0, kend, lda, tid);
    stage128<!TB>(Bs, B, nullptr, 0, n0, N, k0, kend, ldb, tid);
    __syncthreads();
    const float* ap = As + h * LDS2 + wm * 64 + i;
    const float* bp = Bs + h * LDS2 + wn * 64 + i;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = ap[kk * LDS2], a1 = ap[kk * LDS2 + 32];
      const float b0 = bp[kk * LDS2], b1 = bp[kk * LDS2 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int64_t col = n0 + wn * 64 + tn * 32 + i;
    if (col >= N) continue;
    const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= M) continue;
        const float v = acc[tm][tn][r] + bv;
        if (atomic_out) atomicAdd(C + row * ldc + col, v);
        else C[row * ldc + col] = act_fwd(v, act);
      }
  }
}

// Fast path of the 128x128 kernel for full, aligned tiles without an activation mask (every large HPD GEMM but the ragged
// last row chunk).  A lone VALU instruction costs as much as 1/16 of an MFMA and never overlaps one (decoder.hip, "issue
// model"), and the general staging above spends ~270 of them per K-block on 64-bit address arithmetic, bounds and
// alignment tests: 43 % MFMA utilisation.  Here every thread keeps two running source pointers per operand, the next
// K-block's 16-byte loads are in flight while the current one is multiplied (register double buffer), LDS addresses are
// immediates, and the K loop is MFMAs, LDS traffic and two barriers.
#ifndef GNGF_FAST_BK
#define GNGF_FAST_BK 32
#endif
constexpr int kFastBK = GNGF_FAST_BK;
#ifndef GNGF_FAST_WAVES
#define GNGF_FAST_WAVES 4
#endif
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ unsigned lds_base(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}

template <bool TRANS, int BKF> __device__ __forceinline__ void stash128(float* S, const u32x4 (&v)[BKF / 8], int tid) {
#pragma unroll
  for (int e = 0; e < BKF / 8; ++e) {
    const int li = tid + e * 256;
    if (TRANS) {
      const int kk = li >> 5, rq = (li & 31) * 4;
      *reinterpret_cast<u32x4*>(S + kk * LDS2 + rq) = v[e];
    } else {
      const int rr = li / (BKF / 4), kq = (li % (BKF / 4)) * 4;
      S[(kq + 0) * LDS2 + rr] = __uint_as_float(v[e].x); S[(kq + 1) * LDS2 + rr] = __uint_as_float(v[e].y);
      S[(kq + 2) * LDS2 + rr] = __uint_as_float(v[e].z); S[(kq + 3) * LDS2 + rr] = __uint_as_float(v[e].w);
    }
  }
}

// LDS read with an immediate byte offset (no address VALU in the K loop) and the wait that pairs with it: LDS returns
// data in order, so waiting for "at most PENDING operations outstanding" releases the older reads only.
template <int OFF> __device__ __forceinline__ float ds_ld(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16 bits");
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int PENDING> __device__ __forceinline__ void ds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(PENDING) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <typename F, int... I> __device__ __forceinline__ void unrolled_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void unrolled(F&& f) { unrolled_impl(f, std::make_integer_sequence<int, N>{}); }

template <bool TA, bool TB, int BKF>
__global__ void __launch_bounds__(256, GNGF_FAST_WAVES)
gemm128_fast_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                    int64_t lda, int64_t ldb, int64_t ldc, const float* __restrict__ bias, int act, int64_t K,
                    int64_t kchunk, int atomic_out, int tiles_m, int tiles_n) {
  constexpr int NV4 = BKF / 8;                            // float4 per thread and operand per K-block
  __shared__ float As[BKF * LDS2];
  __shared__ float Bs[BKF * LDS2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
  // Tile order.  Workgroups are dealt round-robin to the 8 XCDs (one L2 each), so XCD x takes the x-th eighth of the tile
  // list, and the list runs fastest along the dimension with fewer tiles: the tiles in flight on one XCD then share
  // their tile of the large operand (read from HBM once, not once per tile row) while the small operand stays in L2.
  int64_t t = blockIdx.x;
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
  const int64_t m0 = (tiles_m <= tiles_n ? t % tiles_m : t / tiles_n) * BM2;
  const int64_t n0 = (tiles_m <= tiles_n ? t / tiles_m : t % tiles_n) * BN2;
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  const int nkb = (int)((kend - kbeg) / BKF);
  // Global operands come through buffer descriptors rebased (scalar ALU) on each K-block's window; the per-thread
  // byte offsets inside a window are loop invariant.
  const float* Aw = TA ? A + kbeg * lda + m0 : A + m0 * lda + kbeg;
  const float* Bw = !TB ? B + kbeg * ldb + n0 : B + n0 * ldb + kbeg;
  const int64_t sa = TA ? (int64_t)BKF * lda : BKF, sb = !TB ? (int64_t)BKF * ldb : BKF;
  unsigned oa[NV4], ob[NV4];
#pragma unroll
  for (int e = 0; e < NV4; ++e) {
    const int li = tid + e * 256;
    oa[e] = 4u * (TA ? (unsigned)(li >> 5) * (unsigned)lda + (li & 31) * 4 : (unsigned)(li / (BKF / 4)) * (unsigned)lda + (li % (BKF / 4)) * 4);
    ob[e] = 4u * (!TB ? (unsigned)(li >> 5) * (unsigned)ldb + (li & 31) * 4 : (unsigned)(li / (BKF / 4)) * (unsigned)ldb + (li % (BKF / 4)) * 4);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;
  u32x4 va[NV4], vb[NV4];
  auto fetch = [&](int kb) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Aw + kb * sa), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bw + kb * sb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < NV4; ++e) {
      va[e] = __builtin_amdgcn_raw_buffer_load_b128(rsa, oa[e], 0, 0);
      vb[e] = __builtin_amdgcn_raw_buffer_load_b128(rsb, ob[e], 0, 0);
    }
  };
  fetch(0);
  const unsigned ra = lds_base(As + h * LDS2 + wm * 64 + i), rb = lds_base(Bs + h * LDS2 + wn * 64 + i);
  for (int kb = 0; kb < nkb; ++kb) {
    stash128<TA, BKF>(As, va, tid);
    stash128<!TB, BKF>(Bs, vb, tid);
    __syncthreads();
    if (kb + 1 < nkb) fetch(kb + 1);                      // next block's loads fly under this block's MFMAs
    // operands of step s+1 are read from LDS while the four MFMAs of step s run
    float a0 = ds_ld<0>(ra), a1 = ds_ld<128>(ra), b0 = ds_ld<0>(rb), b1 = ds_ld<128>(rb);
    unrolled<BKF / 2>([&](auto S) {
      constexpr int s = S.value;
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if constexpr (s + 1 < BKF / 2) {
        constexpr int o = (2 * s + 2) * LDS2 * 4;
        na0 = ds_ld<o>(ra); na1 = ds_ld<o + 128>(ra); nb0 = ds_ld<o>(rb); nb1 = ds_ld<o + 128>(rb);
        ds_wait<4>();
      } else {
        ds_wait<0>();
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    });
    __syncthreads();
  }
  // C tile through one descriptor: lane offset in a VGPR, the (row, column block) offset of each store is scalar
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(C + m0 * ldc + n0, 0, 0x7fffffff, 0x00020000);
  const unsigned oc = 4u * ((unsigned)(wm * 64 + 4 * h) * (unsigned)ldc + wn * 64 + i);
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const float bv = (bias && blockIdx.y == 0) ? bias[n0 + wn * 64 + tn * 32 + i] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned so = 4u * ((unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)ldc + tn * 32);
        const float v = acc[tm][tn][r] + bv;
        if (atomic_out) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rsc, oc, so, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(act_fwd(v, act)), rsc, oc, so, 0);
      }
  }
}

// ---------------------------------------------------------------------------------------------- split-bf16 variant
// The same 128x128 GEMM with every fp32 operand value split EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8
// significant bits), and the product formed from six of the nine cross terms on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation:  a b ~= hi hi + hi mid + mid hi + hi lo + lo hi + mid mid.   The dropped terms (mid lo, lo mid, lo lo) are
// below 2^-24 |a b| each, i.e. the result carries an error of <= ~2e-7 sum |a_k b_k| — at or below the rounding an fp32
// fma chain over the same K accumulates — while the bf16 matrix pipe runs 16x the fp32 one per instruction: six bf16 MFMAs
// cost 3/8 of the fp32 MFMA they replace, and the VALU work of the splitting issues in the bf16 MFMA's free issue slots
// (an fp32 MFMA shares the fp32 ALUs with VALU instructions; a bf16 MFMA does not).  Used for the T-wide last layer of
// the HashProbDistribution (logits, dW, dh: 3 x 2.4e13 FLOP per training step at T = 2^19), switchable
// (gngf_set_gemm_split_bf16) — the exact-fp32 kernel above stays the default for everything else.
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned d;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
struct Split3 { u32x4 hi, mid, lo; };
// exact three-way split of eight fp32 values (one bf16 MFMA fragment: k = 8 h + 0..7 of a row)
__device__ __forceinline__ Split3 split8(const float (&x)[8]) {
  Split3 r;
  unsigned hi[4], mid[4], lo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    const unsigned h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);       // exact
    const unsigned m = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);     // exact
    hi[q] = h; mid[q] = m; lo[q] = cvt_pk_bf16(sa, sb);
  }
  r.hi = u32x4{hi[0], hi[1], hi[2], hi[3]}; r.mid = u32x4{mid[0], mid[1], mid[2], mid[3]}; r.lo = u32x4{lo[0], lo[1], lo[2], lo[3]};
  return r;
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Epilogue of the 128 x 128 split kernels: wave (wm, wn) holds rows wm * 64 .. + 63, columns wn * 64 .. + 63 of the tile as 2 x 2
// MFMA accumulators; bias, activation / atomic accumulation, and (rowparts) the row statistics of the tile.
__device__ __forceinline__ void split_epilogue(f32x16 (&acc)[2][2], float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                               int act, int atomic_out, int64_t m0, int64_t n0, int wm, int wn, int i, int h,
                                               float2* __restrict__ rowparts, int nparts) {
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(C + m0 * ldc + n0, 0, 0x7fffffff, 0x00020000);
  const unsigned oc = 4u * ((unsigned)(wm * 64 + 4 * h) * (unsigned)ldc + wn * 64 + i);
  // the mode (accumulate / plain store / activation) is decided ONCE, outside the 64 stores: tested per value it cost the logits
  // kernel ~900 of its 1600 vector instructions per wave and tile (compares, selects and the never-taken activation code in between)
  const int mode = atomic_out ? 0 : (act == 0 ? 1 : 2);
  auto stores = [&](auto MODE) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const float bv = (bias && blockIdx.y == 0) ? bias[n0 + wn * 64 + tn * 32 + i] : 0.f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned so = 4u * ((unsigned)(tm * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)ldc + tn * 32);
          const float v = acc[tm][tn][r] + bv;
          if constexpr (MODE.value == 0) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rsc, oc, so, 0);
          else if constexpr (MODE.value == 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsc, oc, so, 0);
          else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(act_fwd(v, act)), rsc, oc, so, 0);
          acc[tm][tn][r] = v;                                    // (the statistics below are taken on the stored values)
        }
    }
  };
  if (mode == 0) stores(std::integral_constant<int, 0>{});
  else if (mode == 1) stores(std::integral_constant<int, 1>{});
  else stores(std::integral_constant<int, 2>{});
  if (rowparts) {
    // ROW STATISTICS OF THE TILE IN THE EPILOGUE (the logits product of the chunked HashProbDistribution, reference
    // models.py:85,105-116): for every row of the tile and each 64-column half (this wave's columns) the maximum and
    // sum exp(z - max) — 8 bytes per 256 bytes of logits — so that the row maxima / normalisers and the top-K (which lies inside
    // the K half-tiles with the largest maxima) come out of a merge over these partials instead of a second pass over the logits
    // (hpd.hip: rowstats_topk_kernel).  Reduction over the 32 lanes that share a row, for the 16 row-registers of a lane at once: the
    // first step TRANSPOSES — v_permlane16_swap exchanges the upper 16 lanes of register r with the lower 16 of register r + 8, so one
    // max / add folds two registers into one whose lower 16 lanes carry row-register r and whose upper 16 carry r + 8 — and four DPP
    // steps inside each 16-lane row finish 8 registers instead of 16 (quad_perm, half mirror, mirror: every lane of the row ends up with
    // the result); the same swap of a register with itself hands both maxima back to all 32 lanes for the exponentials.  Lanes 0 / 16 /
    // 32 / 48 store.  A NaN logit makes the half's sum NaN (the maximum ignores it, as fmaxf does).
#define GNGF_DPP_F(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, 0xF, 0xF, false))
#define GNGF_DPP_ADD(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xF, 0xF, false))
    constexpr float kLog2e = 1.4426950408889634f;
    const int64_t part = (n0 >> 6) + wn;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      float w[8], mall[16], sx[16];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float va = fmaxf(acc[tm][0][r], acc[tm][1][r]), vb = fmaxf(acc[tm][0][r + 8], acc[tm][1][r + 8]);
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(va), __float_as_uint(vb), false, false);
        float m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        m = fmaxf(m, GNGF_DPP_F(m, 0xB1));
        m = fmaxf(m, GNGF_DPP_F(m, 0x4E));
        m = fmaxf(m, GNGF_DPP_F(m, 0x141));
        m = fmaxf(m, GNGF_DPP_F(m, 0x140));
        w[r] = m;                                                // rows 0 / 2 of the wave: row-register r; rows 1 / 3: row-register r + 8
        const auto bc = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        mall[r] = __uint_as_float(bc[0]);
        mall[r + 8] = __uint_as_float(bc[1]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = acc[tm][0][r], b = acc[tm][1][r];
        float e = (mall[r] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((a - mall[r]) * kLog2e) + __builtin_amdgcn_exp2f((b - mall[r]) * kLog2e);
        if (a != a || b != b) e = __int_as_float(0x7fc00000);
        sx[r] = e;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(sx[r]), __float_as_uint(sx[r + 8]), false, false);
        float t = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        t += GNGF_DPP_ADD(t, 0xB1);
        t += GNGF_DPP_ADD(t, 0x4E);
        t += GNGF_DPP_ADD(t, 0x141);
        t += GNGF_DPP_ADD(t, 0x140);
        if ((i & 15) == 0) {
          const int reg = r + 8 * (i >> 4);
          const int64_t row = m0 + wm * 64 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          rowparts[row * nparts + part] = make_float2(w[r], t);
        }
      }
    }
#undef GNGF_DPP_F
#undef GNGF_DPP_ADD
  }
}

// Same staging and tile order as gemm128_fast_kernel (fp32 K-blocks in LDS, k-major); every wave splits the fragments it
// reads.  (A variant that splits once per workgroup on the way into LDS — bf16 planes, 16-byte fragment reads — was not
// faster on any of the three shapes, and slower where both operands are staged transposed: the kernel is bound by its
// per-K-block barriers and the short K = 128 of the logits product, not by the splitting.)
#ifndef GNGF_SPLIT_WGS
#define GNGF_SPLIT_WGS 3
#endif
template <bool TA, bool TB>
__global__ void __launch_bounds__(256, GNGF_SPLIT_WGS)
gemm128_split_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                     int64_t lda, int64_t ldb, int64_t ldc, const float* __restrict__ bias, int act, int64_t K,
                     int64_t kchunk, int atomic_out, int tiles_m, int tiles_n, float2* __restrict__ rowparts = nullptr,
                     int nparts = 0) {
  constexpr int BKF = 32;
  constexpr int NV4 = BKF / 8;
  __shared__ float As[BKF * LDS2];
  __shared__ float Bs[BKF * LDS2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
  int64_t t = blockIdx.x;
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
  const int64_t m0 = (tiles_m <= tiles_n ? t % tiles_m : t / tiles_n) * BM2;
  const int64_t n0 = (tiles_m <= tiles_n ? t / tiles_m : t % tiles_n) * BN2;
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  const int nkb = (int)((kend - kbeg) / BKF);
  const float* Aw = TA ? A + kbeg * lda + m0 : A + m0 * lda + kbeg;
  const float* Bw = !TB ? B + kbeg * ldb + n0 : B + n0 * ldb + kbeg;
  const int64_t sa = TA ? (int64_t)BKF * lda : BKF, sb = !TB ? (int64_t)BKF * ldb : BKF;
  unsigned oa[NV4], ob[NV4];
#pragma unroll
  for (int e = 0; e < NV4; ++e) {
    const int li = tid + e * 256;
    oa[e] = 4u * (TA ? (unsigned)(li >> 5) * (unsigned)lda + (li & 31) * 4 : (unsigned)(li / (BKF / 4)) * (unsigned)lda + (li % (BKF / 4)) * 4);
    ob[e] = 4u * (!TB ? (unsigned)(li >> 5) * (unsigned)ldb + (li & 31) * 4 : (unsigned)(li / (BKF / 4)) * (unsigned)ldb + (li % (BKF / 4)) * 4);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;
  u32x4 va[NV4], vb[NV4];
  auto fetch = [&](int kb) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Aw + kb * sa), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bw + kb * sb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < NV4; ++e) {
      va[e] = __builtin_amdgcn_raw_buffer_load_b128(rsa, oa[e], 0, 0);
      vb[e] = __builtin_amdgcn_raw_buffer_load_b128(rsb, ob[e], 0, 0);
    }
  };
  fetch(0);
  // lane (i, h) of the bf16 MFMA holds k = 8 h + j (j = 0..7) of row / column i: eight LDS words 8 h + j rows down the
  // k-major image, read with immediate offsets from one base per operand
  const unsigned ra = lds_base(As + 8 * h * LDS2 + wm * 64 + i), rb = lds_base(Bs + 8 * h * LDS2 + wn * 64 + i);
  for (int kb = 0; kb < nkb; ++kb) {
    stash128<TA, BKF>(As, va, tid);
    stash128<!TB, BKF>(Bs, vb, tid);
    __syncthreads();
    if (kb + 1 < nkb) fetch(kb + 1);
    unrolled<BKF / 16>([&](auto KS) {
      constexpr int ks = KS.value;
      Split3 fa[2], fb[2];
      unrolled<2>([&](auto TT_) {
        constexpr int tt = TT_.value;
        float xa[8], xb[8];
        unrolled<8>([&](auto J) {
          constexpr int j = J.value;
          constexpr int o = ((16 * ks + j) * LDS2 + 32 * tt) * 4;
          xa[j] = ds_ld<o>(ra);
          xb[j] = ds_ld<o>(rb);
        });
        ds_wait<0>();                                          // (with its scheduling barrier: the consumers stay below the wait)
        fa[tt] = split8(xa);
        fb[tt] = split8(xb);
      });
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          f32x16 c = acc[ta][tb];                              // small terms first
          c = mfma_bf16(fa[ta].lo, fb[tb].hi, c);
          c = mfma_bf16(fa[ta].hi, fb[tb].lo, c);
          c = mfma_bf16(fa[ta].mid, fb[tb].mid, c);
          c = mfma_bf16(fa[ta].mid, fb[tb].hi, c);
          c = mfma_bf16(fa[ta].hi, fb[tb].mid, c);
          c = mfma_bf16(fa[ta].hi, fb[tb].hi, c);
          acc[ta][tb] = c;
        }
    });
    __syncthreads();
  }
  split_epilogue(acc, C, ldc, bias, act, atomic_out, m0, n0, wm, wn, i, h, rowparts, nparts);
}

// ---------------------------------------------------------------------------------- split once, bf16 planes in LDS
// The split kernel above splits every fragment a wave reads — each operand value is split twice (two waves share a tile row /
// column) by 5.5 VALU instructions, and the K loop is bound by vector issue (SQ counters, dW shape: vector issue 65 % busy + MFMA
// issue 15 %, matrix pipe 60 %, the chip holding ~1.6 GHz under that load).  Here a value is split ONCE, by the thread that loaded
// it, on the way into LDS: the K-block of an operand is kept as NP planes of bf16 (hi, mid, lo — or hi, lo), 8 KB each, and a
// fragment is one 16-byte read per plane.  Two images, both without padding and conflict-free for their writes and reads:
//  * [row][k] for an operand whose k runs contiguously in memory: 64-byte rows, the four 16-byte chunks of a row XOR-ed with
//    (row >> 2) & 3; lane (i, h) reads chunk 2 s + h of row i with ds_read_b128;
//  * [k][row] for an operand stored k-major: 256-byte rows (four 64-byte segments of 32 rows, segment XOR-ed with k & 3), written
//    as loaded (four rows of one k: 8 bytes per plane) and read through the hardware transposition (ds_read_b64_tr_b16: a group of
//    16 lanes reads 4 k-rows x 16 columns and every lane receives its column's four k).
// NP = 3: x = hi + mid + lo exactly, six products, the numbers of the kernel above bit for bit (same terms, same order).
// NP = 2: x ~= hi + lo (16 significant bits), three products hi hi + hi lo + lo hi: |error| <= 3 * 2^-18 |a b| per product — inside
// what an fp32 dot product of >= 200 terms may lose by its own a-priori bound (n 2^-24), and in the measured error against float64 at
// the level of the exact-fp32 kernel (tools/perf_gemm_split.py); for the two accumulating GEMMs of the backward pass (dW, dh).
using s16x4 = __attribute__((ext_vector_type(4))) short;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;

template <int NP> __device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[NP]) {
  const unsigned h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);          // exact
  out[0] = h;
  const unsigned m = cvt_pk_bf16(ra, rb);
  out[1] = m;
  if constexpr (NP == 3) {
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);      // exact
    out[2] = cvt_pk_bf16(sa, sb);
  }
}

// byte offset (inside one 8 KB plane) of the 8 bytes thread `li` of the staging pass writes: KM = the operand is k-major in memory
// (thread li holds rows 4 (li & 31) .. + 3 of k = li >> 5), else k-contiguous (row li >> 3, k = 4 (li & 7) .. + 3)
template <bool KM> __device__ __forceinline__ unsigned plane_write_offset(int li) {
  if (KM) {
    const int k = li >> 5, mq = li & 31;
    return 256u * k + 64u * ((mq >> 3) ^ (k & 3)) + 8u * (mq & 7);
  }
  const int row = li >> 3, kq = li & 7;
  return 64u * row + 16u * ((kq >> 1) ^ ((row >> 2) & 3)) + 8u * (kq & 1);
}

template <int NP> struct Frag { u32x4 p[NP]; };

// fragment (rows w64 + 32 t .. + 31, k = 16 s + 8 h .. + 7) of one operand: lane-dependent part of the address in a0 / a1
//   k-contiguous image: a_s  = 64 (w64 + i) + 16 ((2 s + h) ^ ((i >> 2) & 3))      (s = 0, 1);  + 2048 t + 8192 plane
//   k-major image:      a_t  = 256 (8 h + q) + 64 ((w64 / 32 + t) ^ q) + 32 ((lane >> 4) & 1) + 8 p  (q = (lane >> 2) & 3, p = lane & 3);
//                              + 4096 s + 1024 (second four k) + 8192 plane
template <bool KM, int NP, int S, int TT>
__device__ __forceinline__ Frag<NP> read_frag(const lds_byte* base, unsigned a0, unsigned a1) {
  Frag<NP> f;
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    if constexpr (KM) {
      const lds_byte* q = base + (TT ? a1 : a0) + 4096 * S + 8192 * pl;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 1024));
      const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
      f.p[pl] = u32x4{l2.x, l2.y, h2.x, h2.y};
    } else {
      const lds_byte* q = base + (S ? a1 : a0) + 2048 * TT + 8192 * pl;
      f.p[pl] = *(const lds_u32x4*)q;
    }
  }
  return f;
}

// k-major image whose 8-byte units are ALSO permuted inside their 64-byte segment (unit ^ ((k >> 1) & 7)): for a writer that holds
// one k per lane and 8-byte runs of rows (the fused dW loader: 16 lanes = 16 k of one unit, which without this are 8-way conflicts on
// every store).  The transposed read's lane address picks up (4 h + (q >> 1)) ^ 2 r2 on its unit: a[TT][r2], r2 = the second four k.
template <int NP, int S, int TT>
__device__ __forceinline__ Frag<NP> read_frag_km8(const lds_byte* base, const unsigned (&a)[2][2]) {
  Frag<NP> f;
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + a[TT][0] + 4096 * S + 8192 * pl));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + a[TT][1] + 4096 * S + 1024 + 8192 * pl));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    f.p[pl] = u32x4{l2.x, l2.y, h2.x, h2.y};
  }
  return f;
}

template <int NP> __device__ __forceinline__ f32x16 split_products(const Frag<NP>& a, const Frag<NP>& b, f32x16 c) {
  if constexpr (NP == 3) {                                     // small terms first (the order of gemm128_split_kernel)
    c = mfma_bf16(a.p[2], b.p[0], c);
    c = mfma_bf16(a.p[0], b.p[2], c);
    c = mfma_bf16(a.p[1], b.p[1], c);
    c = mfma_bf16(a.p[1], b.p[0], c);
    c = mfma_bf16(a.p[0], b.p[1], c);
    c = mfma_bf16(a.p[0], b.p[0], c);
  } else {
    c = mfma_bf16(a.p[1], b.p[0], c);
    c = mfma_bf16(a.p[0], b.p[1], c);
    c = mfma_bf16(a.p[0], b.p[0], c);
  }
  return c;
}

template <bool TA, bool TB, int NP>
__global__ void __launch_bounds__(256, 3)
gemm128_planes_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                      int64_t lda, int64_t ldb, int64_t ldc, const float* __restrict__ bias, int act, int64_t K,
                      int64_t kchunk, int atomic_out, int tiles_m, int tiles_n, float2* __restrict__ rowparts = nullptr,
                      int nparts = 0) {
  constexpr int BKF = 32;
  constexpr int NV4 = BKF / 8;
  constexpr bool AKM = TA, BKM = !TB;                           // which operands are k-major in memory
  __shared__ __attribute__((aligned(16))) unsigned char img[2 * NP * 8192];          // planes of A, then planes of B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
  int64_t t = blockIdx.x;
  const int64_t ntiles = (int64_t)tiles_m * tiles_n;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
  const int64_t m0 = (tiles_m <= tiles_n ? t % tiles_m : t / tiles_n) * BM2;
  const int64_t n0 = (tiles_m <= tiles_n ? t / tiles_m : t % tiles_n) * BN2;
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  const int nkb = (int)((kend - kbeg) / BKF);
  const float* Aw = TA ? A + kbeg * lda + m0 : A + m0 * lda + kbeg;
  const float* Bw = !TB ? B + kbeg * ldb + n0 : B + n0 * ldb + kbeg;
  const int64_t sa = TA ? (int64_t)BKF * lda : BKF, sb = !TB ? (int64_t)BKF * ldb : BKF;
  unsigned oa[NV4], ob[NV4];
#pragma unroll
  for (int e = 0; e < NV4; ++e) {
    const int li = tid + e * 256;
    oa[e] = 4u * (TA ? (unsigned)(li >> 5) * (unsigned)lda + (li & 31) * 4 : (unsigned)(li / (BKF / 4)) * (unsigned)lda + (li % (BKF / 4)) * 4);
    ob[e] = 4u * (!TB ? (unsigned)(li >> 5) * (unsigned)ldb + (li & 31) * 4 : (unsigned)(li / (BKF / 4)) * (unsigned)ldb + (li % (BKF / 4)) * 4);
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;
  u32x4 va[NV4], vb[NV4];
  auto fetch = [&](int kb) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Aw + kb * sa), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bw + kb * sb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < NV4; ++e) {
      va[e] = __builtin_amdgcn_raw_buffer_load_b128(rsa, oa[e], 0, 0);
      vb[e] = __builtin_amdgcn_raw_buffer_load_b128(rsb, ob[e], 0, 0);
    }
  };
  fetch(0);
  lds_byte* const imgA = (lds_byte*)img;
  lds_byte* const imgB = imgA + NP * 8192;
  // this thread's staging offsets are the same every K-block: one base each, the four passes 2048 bytes apart in either image
  const unsigned wa0 = plane_write_offset<AKM>(tid), wb0 = plane_write_offset<BKM>(tid);
  // lane-dependent parts of the fragment addresses
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  const unsigned ra0 = AKM ? 256u * (8 * h + q4) + 64u * ((2 * wm) ^ q4) + 32u * g1 + 8u * p4
                           : 64u * (wm * 64 + i) + 16u * ((unsigned)h ^ ((i >> 2) & 3));
  const unsigned ra1 = AKM ? 256u * (8 * h + q4) + 64u * ((2 * wm + 1) ^ q4) + 32u * g1 + 8u * p4
                           : 64u * (wm * 64 + i) + 16u * ((unsigned)(2 + h) ^ ((i >> 2) & 3));
  const unsigned rb0 = BKM ? 256u * (8 * h + q4) + 64u * ((2 * wn) ^ q4) + 32u * g1 + 8u * p4
                           : 64u * (wn * 64 + i) + 16u * ((unsigned)h ^ ((i >> 2) & 3));
  const unsigned rb1 = BKM ? 256u * (8 * h + q4) + 64u * ((2 * wn + 1) ^ q4) + 32u * g1 + 8u * p4
                           : 64u * (wn * 64 + i) + 16u * ((unsigned)(2 + h) ^ ((i >> 2) & 3));
  for (int kb = 0; kb < nkb; ++kb) {
#pragma unroll
    for (int e = 0; e < NV4; ++e) {
      // pass e: li = tid + 256 e  ->  k + 8 e (k-major: 8 rows of 256 bytes) or row + 32 e (32 rows of 64 bytes): + 2048 e bytes;
      // the XOR terms do not change (k & 3, (row >> 2) & 3 are those of pass 0: 8 e and 32 e leave them alone)
      unsigned pa[2][NP], pb[2][NP];
      split_pair<NP>(__uint_as_float(va[e].x), __uint_as_float(va[e].y), pa[0]);
      split_pair<NP>(__uint_as_float(va[e].z), __uint_as_float(va[e].w), pa[1]);
      split_pair<NP>(__uint_as_float(vb[e].x), __uint_as_float(vb[e].y), pb[0]);
      split_pair<NP>(__uint_as_float(vb[e].z), __uint_as_float(vb[e].w), pb[1]);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        *(lds_u32x2*)(imgA + wa0 + 2048 * e + 8192 * pl) = u32x2{pa[0][pl], pa[1][pl]};
        *(lds_u32x2*)(imgB + wb0 + 2048 * e + 8192 * pl) = u32x2{pb[0][pl], pb[1][pl]};
      }
    }
    __syncthreads();
    if (kb + 1 < nkb) fetch(kb + 1);
    unrolled<2>([&](auto S_) {
      constexpr int ks = S_.value;
      const Frag<NP> fa0 = read_frag<AKM, NP, ks, 0>(imgA, ra0, ra1), fa1 = read_frag<AKM, NP, ks, 1>(imgA, ra0, ra1);
      const Frag<NP> fb0 = read_frag<BKM, NP, ks, 0>(imgB, rb0, rb1), fb1 = read_frag<BKM, NP, ks, 1>(imgB, rb0, rb1);
      acc[0][0] = split_products<NP>(fa0, fb0, acc[0][0]);
      acc[0][1] = split_products<NP>(fa0, fb1, acc[0][1]);
      acc[1][0] = split_products<NP>(fa1, fb0, acc[1][0]);
      acc[1][1] = split_products<NP>(fa1, fb1, acc[1][1]);
    });
    __syncthreads();
  }
  split_epilogue(acc, C, ldc, bias, act, atomic_out, m0, n0, wm, wn, i, h, rowparts, nparts);
}

// ------------------------------------------------------------ softmax backward formed in the operand loaders of dW and dh
// Backward of the T-wide last layer of the HashProbDistribution from the LOGITS (reference models.py:84-85,105-116 and the batch-mean
// loss utils.py:138,159 through autograd), without the d-logits matrix ever existing:
//     dz[r,t] = p[r,t] (g[r,t] - dot_r),   p = exp(z - m_r) / s_r,   g = mw (U x L) G (L x T),   dot_r = <p_r, g_r> + top-K part
//     dW[t,:] += sum_r dz[r,t] h[r,:]      db[t] += sum_r dz[r,t]      dh[r,:] += sum_t dz[r,t] W[t,:]
// (the K top-K slots of a row add p_k dq_k to dz: two sparse side products, hpd_topk_side_kernel).  Both GEMMs read z where they
// read dz before: the apply pass of the softmax backward (a read and a write of the (U, T) matrix) and the dz round trip are gone.
// The loader works in the accumulator layout of the small product g^T = G^T mw^T (one 32x32x16 MFMA per product term, L <= 16):
// lane (i, h) holds ROW r = i of the block — its (m, 1/s, dot) are three registers — and the sixteen t = 8 j + 4 h + 0..3 of it:
// four 16-byte loads of z, four packed 8-byte stores per plane into the image of the GEMM proper (bf16 planes as above; g keeps the
// exact three-way split: six MFMAs per 32 x 32 block, against 12-24 of the GEMM).
template <int NP> __device__ __forceinline__ void fused_dz_block(const u32x4 (&zv)[4], const f32x16& g, float m, float rs, float dot,
                                                                 unsigned (&pk)[4][2][NP], float (*dacc)[16]) {
  // the expression of the streaming passes (hpd.hip: prob_fast), so that the dots of gngf_hpd_bwd_dot and these dz are of one piece:
  //   p = nan_to_num(exp2((z - m) log2 e) / s),  dz = p (g - dot)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float q = __builtin_amdgcn_exp2f((__uint_as_float(zv[j][c]) - m) * 1.4426950408889634f) * rs;
      const float p = __builtin_amdgcn_fmed3f(q, 0.f, 3.4028234663852886e38f);            // nan_to_num: NaN -> 0, inf -> max
      d[c] = p * (g[4 * j + c] - dot);
      if (dacc) (*dacc)[4 * j + c] += d[c];
    }
    split_pair<NP>(d[0], d[1], pk[j][0]);
    split_pair<NP>(d[2], d[3], pk[j][1]);
  }
}

// g^T block (32 t x 32 r) = G^T (t x l) mw^T (l x r), both operands split exactly three ways: fp32-accurate
__device__ __forceinline__ f32x16 fused_g_block(const Split3& Gs, const Split3& Ms) {
  f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  c = mfma_bf16(Gs.lo, Ms.hi, c);
  c = mfma_bf16(Gs.hi, Ms.lo, c);
  c = mfma_bf16(Gs.mid, Ms.mid, c);
  c = mfma_bf16(Gs.mid, Ms.hi, c);
  c = mfma_bf16(Gs.hi, Ms.mid, c);
  c = mfma_bf16(Gs.hi, Ms.hi, c);
  return c;
}

// The small operands of both kernels arrive SPLIT (gngf_hpd_bwd_prepare, once per backward pass): bf16 planes
//   hp  (NP, rows_total, 128)  the last hidden layer  (B operand of dW)         Wp  (NP, T, 128)  the last layer's weight (B operand of dH)
//   mwp (3, rows_total, 16)    multiplicity weights, zero-padded to 16 levels   Gtp (3, T, 16)    G transposed, zero-padded
// so a K-block's B tile is four 16-byte loads and four 16-byte LDS stores per thread and an operand of the small product g is
// three 16-byte loads — no split instruction outside the dz values themselves (per wave and K-block: 213 vector instructions where the
// first form, which split h / W / mw / G from fp32 in the loop, had 313).
// (Measured on the way and dropped: the K loop's loads through inline asm with a hand-counted s_waitcnt and a two-deep register ring
// — hipcc sizes the loop-header wait for its worst predecessor and drains the ring in every other step —: exactly as fast, 3.87 vs 3.88
// ms, and unsafe as written: loop-carried asm outputs must be tied operands.  A select on a freshly loaded value makes hipcc wait for
// that load, and every load in flight, on the spot: nothing is selected in the fetch.)
template <bool KM> __device__ __forceinline__ unsigned plane_tile_write_offset(int tid);
// a [32 k][128 n] bf16 tile of a plane, k-major image: thread -> k = tid >> 4 (+ 16 in the second pass: + 4096 bytes), 16-byte chunk tid & 15
template <> __device__ __forceinline__ unsigned plane_tile_write_offset<true>(int tid) {
  const int k = tid >> 4, c16 = tid & 15;
  return 256u * k + 64u * ((unsigned)(c16 >> 2) ^ (k & 3)) + 16u * (c16 & 3);
}

// A wave's 32 rows x 32 columns of logits are LOADED the way the streaming passes load them — whole 128-byte lines, eight rows per
// instruction (lane -> row 8 e + (lane >> 3), 16-byte chunk lane & 7) — and turned into the layout the loader computes in (lane (i, h)
// -> row i, chunks 2 j + h) through 4 KB of LDS private to the wave (no barrier: a wave's LDS operations execute in order): 128-byte
// rows, chunk c of row r at c ^ ((r >> 1) & 7) — conflict-free for the row-wise stores and for the 16-byte reads of 16 lanes.
// Loading in the computing layout directly (each lane 16 bytes of its own row: 32 rows x 32 bytes per instruction) held both kernels
// at 2.4-2.9 TB/s of logits whatever the prefetch depth, the waits or the split of the work over waves.
__device__ __forceinline__ void wave_block_transpose(lds_byte* scr, int lane, const u32x4 (&in)[4], u32x4 (&out)[4]) {
  const int rsub = lane >> 3, c = lane & 7, i = lane & 31, h = lane >> 5;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = 8 * e + rsub;
    *(lds_u32x4*)(scr + 128 * row + 16 * (c ^ ((row >> 1) & 7))) = in[e];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = *(const lds_u32x4*)(scr + 128 * i + 16 * ((2 * j + h) ^ ((i >> 1) & 7)));
}

// dW (T x 128) += dz^T h, db (T) += column sums of dz.  A workgroup owns 128 columns t for ALL U rows (U % 64 == 0): wave w stages the
// 32 columns t0 + 32 w .. + 31 of every 32-row block (A image k-major: [r][t], 8-byte units permuted: read_frag_km8).
template <int NP>
__global__ void __launch_bounds__(256, 2)
hpd_dw_fused_kernel(const float* __restrict__ Z, const float* __restrict__ rowstat, const float* __restrict__ dot,
                    const unsigned short* __restrict__ mwp, const unsigned short* __restrict__ Gtp, const unsigned short* __restrict__ hp,
                    int64_t rows_total, float* __restrict__ dW, float* __restrict__ db, int64_t U, int64_t T) {
  __shared__ __attribute__((aligned(16))) unsigned char img[2 * NP * 8192 + 4 * 4096];   // planes of A, planes of B, four private 4 KB blocks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
  int64_t tile = blockIdx.x;
  const int64_t ntiles = T / 128;
  if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
  const int64_t t0 = tile * 128;
  const int nkb = (int)(U / 32);
  // G^T fragment of this wave's 32 columns, constant over the K loop: lane (i, h) holds G[8 h .. 8 h + 7][t0 + 32 wave + i]
  Split3 Gs;
  {
    const unsigned short* g0 = Gtp + (t0 + 32 * wave + i) * 16 + 8 * h;
    Gs.hi = *reinterpret_cast<const u32x4*>(g0);
    Gs.mid = *reinterpret_cast<const u32x4*>(g0 + T * 16);
    Gs.lo = *reinterpret_cast<const u32x4*>(g0 + 2 * T * 16);
  }
  // the workgroup's 32 rows x 512 bytes of a K-block are loaded by all four waves together, 512 contiguous bytes per row (thread ->
  // row 8 e + (tid >> 5), 16-byte chunk tid & 31), parked in a shared 16 KB tile between the two barriers of the step before, and
  // read back by each wave in its computing layout (row i, chunks 8 wave + 2 j + h); chunk c of row r at c ^ (r & 15): conflict-free
  const unsigned zo = 4u * ((unsigned)(tid >> 5) * (unsigned)T) + 16u * (tid & 31);        // + e * 8 rows; window: 32 rows x T
  const unsigned z8 = 32u * (unsigned)T;                                                  // eight rows, in bytes
  const unsigned bo = 2u * ((unsigned)(tid >> 4) * 128u + 8u * (tid & 15));               // + 4096 in the second pass; window: 32 rows x 128
  const unsigned mo = 2u * ((unsigned)i * 16u + 8u * h);                                  // window: 32 rows x 16
  // two K-blocks of logits in flight per workgroup (register ring zv0 / zv1, the loop unrolled by two); the small operands (L2 hits)
  // one block ahead, issued BEFORE the logits of the block after it (loads return in order) and unconditionally (the last blocks are
  // fetched again, in bounds: behind a branch the compiler's wait counts would have to hold for the path that issued nothing)
  u32x4 zv0[4], zv1[4], vb[NP][2];
  Split3 Ms;
  u32x2 rc_ms;
  float rc_d;
  auto fetch_z = [&](int kb, u32x4 (&zv)[4]) {
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z + (int64_t)kb * 32 * T + t0), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < 4; ++e) zv[e] = __builtin_amdgcn_raw_buffer_load_b128(rz, zo, e * z8, 0);
  };
  auto fetch_small = [&](int kb) {
    const int64_t r0 = (int64_t)kb * 32;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(hp + (pl * rows_total + r0) * 128), 0, 0x7fffffff, 0x00020000);
      vb[pl][0] = __builtin_amdgcn_raw_buffer_load_b128(rh, bo, 0, 0);
      vb[pl][1] = __builtin_amdgcn_raw_buffer_load_b128(rh, bo, 4096, 0);
    }
    const __amdgpu_buffer_rsrc_t rm0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(mwp + r0 * 16), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(mwp + (rows_total + r0) * 16), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(mwp + (2 * rows_total + r0) * 16), 0, 0x7fffffff, 0x00020000);
    Ms.hi = __builtin_amdgcn_raw_buffer_load_b128(rm0, mo, 0, 0);
    Ms.mid = __builtin_amdgcn_raw_buffer_load_b128(rm1, mo, 0, 0);
    Ms.lo = __builtin_amdgcn_raw_buffer_load_b128(rm2, mo, 0, 0);
    rc_ms = *reinterpret_cast<const u32x2*>(rowstat + 2 * (r0 + i));
    rc_d = dot[r0 + i];
  };
  fetch_z(0, zv0);                                            // (in the loop's own order: the wait counts at the loop header are sized for
  __builtin_amdgcn_sched_barrier(0);                          //  the worse of its two predecessors)
  fetch_small(0);
  __builtin_amdgcn_sched_barrier(0);
  fetch_z(nkb > 1 ? 1 : 0, zv1);
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;
  float dacc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
  lds_byte* const imgA = (lds_byte*)img;
  lds_byte* const imgB = imgA + NP * 8192;
  unsigned wa[4];                                                                         // run j: unit 2 j + h of segment `wave`, row k = i
#pragma unroll
  for (int j = 0; j < 4; ++j) wa[j] = 256u * i + 64u * ((unsigned)wave ^ (i & 3)) + 8u * ((unsigned)(2 * j + h) ^ ((i >> 1) & 7));
  const unsigned wb0 = plane_tile_write_offset<true>(tid);
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  unsigned ra[2][2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
      ra[tt][r2] = 256u * (8 * h + q4) + 64u * ((unsigned)(2 * wm + tt) ^ q4) + 8u * ((unsigned)(4 * g1 + p4) ^ (unsigned)((4 * h + (q4 >> 1)) ^ (2 * r2)));
  const unsigned rb0 = 256u * (8 * h + q4) + 64u * ((2 * wn) ^ q4) + 32u * g1 + 8u * p4;
  const unsigned rb1 = 256u * (8 * h + q4) + 64u * ((2 * wn + 1) ^ q4) + 32u * g1 + 8u * p4;
  lds_byte* const raw = imgB + NP * 8192;                                               // 32 rows x 512 bytes
  auto park = [&](const u32x4 (&zv)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 8 * e + (tid >> 5);
      *(lds_u32x4*)(raw + 512 * row + 16 * ((tid & 31) ^ (row & 15))) = zv[e];
    }
  };
  park(zv0);                                                  // block 0 (its loads are waited for here: once)
  __syncthreads();
  auto step = [&](int kb, u32x4 (&zv)[4]) {                   // zv: block kb + 1 in flight (block kb is in the shared tile)
    {
      unsigned pk[4][2][NP];
      u32x4 zt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) zt[j] = *(const lds_u32x4*)(raw + 512 * i + 16 * ((8 * wave + 2 * j + h) ^ (i & 15)));
      const f32x16 g = fused_g_block(Gs, Ms);
      fused_dz_block<NP>(zt, g, __uint_as_float(rc_ms.x), 1.0f / __uint_as_float(rc_ms.y), rc_d, pk, &dacc);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) *(lds_u32x2*)(imgA + wa[j] + 8192 * pl) = u32x2{pk[j][0][pl], pk[j][1][pl]};
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      *(lds_u32x4*)(imgB + wb0 + 8192 * pl) = vb[pl][0];
      *(lds_u32x4*)(imgB + wb0 + 4096 + 8192 * pl) = vb[pl][1];
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch_small(kb + 1 < nkb ? kb + 1 : nkb - 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    park(zv);                                                 // every wave has read block kb out of the tile: block kb + 1 goes in
    __builtin_amdgcn_sched_barrier(0);
    fetch_z(kb + 2 < nkb ? kb + 2 : nkb - 1, zv);
    __builtin_amdgcn_sched_barrier(0);
    unrolled<2>([&](auto S_) {
      constexpr int ks = S_.value;
      const Frag<NP> fa0 = read_frag_km8<NP, ks, 0>(imgA, ra), fa1 = read_frag_km8<NP, ks, 1>(imgA, ra);
      const Frag<NP> fb0 = read_frag<true, NP, ks, 0>(imgB, rb0, rb1), fb1 = read_frag<true, NP, ks, 1>(imgB, rb0, rb1);
      acc[0][0] = split_products<NP>(fa0, fb0, acc[0][0]);
      acc[0][1] = split_products<NP>(fa0, fb1, acc[0][1]);
      acc[1][0] = split_products<NP>(fa1, fb0, acc[1][0]);
      acc[1][1] = split_products<NP>(fa1, fb1, acc[1][1]);
    });
    __syncthreads();
  };
  for (int kb = 0; kb < nkb; kb += 2) {                       // (nkb = U / 32 is even: U % 128 == 0)
    step(kb, zv1);                                            // zv1 holds block kb + 1; refilled with block kb + 2
    step(kb + 1, zv1);                                        // ... which this step parks; refilled with block kb + 3
  }
  split_epilogue(acc, dW, 128, nullptr, 0, 1, t0, 0, wm, wn, i, h, nullptr, 0);
  if (db) {
    // column sums: register r of lane (i, h) is column 32 wave + (r & 3) + 8 (r >> 2) + 4 h summed over this lane's rows; add the 32 lanes of a half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = dacc[r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (i == 0) atomicAdd(db + t0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h, v);
    }
  }
}

// dH (U x 128) += dz W over the columns [slice * kchunk, + kchunk) (kchunk % 64 == 0), rows 128 * row tile .. + 127: wave w stages
// rows 32 w .. + 31 (A image [r][t]: k contiguous).  Float atomics into dH.  u0 = the chunk's first row inside the prepared planes.
template <int NP>
__global__ void __launch_bounds__(256, 2)
hpd_dh_fused_kernel(const float* __restrict__ Z, const float* __restrict__ rowstat, const float* __restrict__ dot,
                    const unsigned short* __restrict__ mwp, const unsigned short* __restrict__ Gtp, const unsigned short* __restrict__ Wp,
                    int64_t rows_total, float* __restrict__ dH, int64_t U, int64_t T, int64_t kchunk, int rtiles, int slices) {
  __shared__ __attribute__((aligned(16))) unsigned char img[2 * NP * 8192 + 4 * 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, i = lane & 31, h = lane >> 5;
  // Workgroups are dealt round-robin to the 8 XCDs (one L2 each).  All row tiles of a column slice read the same rows of W: they are
  // given to ONE XCD (slice s -> XCD s % 8), so that W's planes come from HBM once per chunk and not once per XCD (268 MB against 2.1 GB
  // beside the 8.6 GB of logits).  The logits are read by one workgroup each wherever it runs.
  int rt = (int)(blockIdx.x % rtiles), sl = (int)(blockIdx.x / rtiles);
  if ((slices & 7) == 0) {
    const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
    sl = (j / rtiles) * 8 + xcd;
    rt = j % rtiles;
  }
  const int64_t r0 = (int64_t)rt * 128;
  const int64_t kbeg = (int64_t)sl * kchunk;
  const int64_t kend = (kbeg + kchunk < T) ? kbeg + kchunk : T;
  const int nkb = (int)((kend - kbeg) / 32);
  const int64_t r = r0 + 32 * wave + i;                                                  // this lane's row, for the whole kernel
  const float rc_m = rowstat[2 * r], rc_rs = 1.0f / rowstat[2 * r + 1], rc_d = dot[r];
  Split3 Ms;
  {
    const unsigned short* m0 = mwp + r * 16 + 8 * h;
    Ms.hi = *reinterpret_cast<const u32x4*>(m0);
    Ms.mid = *reinterpret_cast<const u32x4*>(m0 + rows_total * 16);
    Ms.lo = *reinterpret_cast<const u32x4*>(m0 + 2 * rows_total * 16);
  }
  const unsigned zo = 4u * ((unsigned)(32 * wave + (lane >> 3)) * (unsigned)T) + 16u * (lane & 7);   // + e * 8 rows; window: 128 rows x T
  const unsigned z8 = 32u * (unsigned)T;
  const unsigned bo = 2u * ((unsigned)(tid >> 4) * 128u + 8u * (tid & 15));
  const unsigned go = 2u * ((unsigned)i * 16u + 8u * h);
  u32x4 zv0[4], zv1[4], vb[NP][2];                            // (two K-blocks of logits in flight: see hpd_dw_fused_kernel)
  Split3 Gs;
  auto fetch_z = [&](int kb, u32x4 (&zv)[4]) {
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z + r0 * T + kbeg + (int64_t)kb * 32), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < 4; ++e) zv[e] = __builtin_amdgcn_raw_buffer_load_b128(rz, zo, e * z8, 0);
  };
  auto fetch_small = [&](int kb) {
    const int64_t tb = kbeg + (int64_t)kb * 32;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Wp + (pl * T + tb) * 128), 0, 0x7fffffff, 0x00020000);
      vb[pl][0] = __builtin_amdgcn_raw_buffer_load_b128(rw, bo, 0, 0);
      vb[pl][1] = __builtin_amdgcn_raw_buffer_load_b128(rw, bo, 4096, 0);
    }
    const __amdgpu_buffer_rsrc_t rg0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Gtp + tb * 16), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Gtp + (T + tb) * 16), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(Gtp + (2 * T + tb) * 16), 0, 0x7fffffff, 0x00020000);
    Gs.hi = __builtin_amdgcn_raw_buffer_load_b128(rg0, go, 0, 0);
    Gs.mid = __builtin_amdgcn_raw_buffer_load_b128(rg1, go, 0, 0);
    Gs.lo = __builtin_amdgcn_raw_buffer_load_b128(rg2, go, 0, 0);
  };
  fetch_z(0, zv0);                                            // (in the loop's own order: the wait counts at the loop header are sized for
  __builtin_amdgcn_sched_barrier(0);                          //  the worse of its two predecessors)
  fetch_small(0);
  __builtin_amdgcn_sched_barrier(0);
  fetch_z(nkb > 1 ? 1 : 0, zv1);
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = 0;
  lds_byte* const imgA = (lds_byte*)img;
  lds_byte* const imgB = imgA + NP * 8192;
  const unsigned swz = (i >> 2) & 3;
  unsigned wa[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wa[j] = 64u * (32 * wave + i) + 16u * ((unsigned)j ^ swz) + 8u * h;
  const unsigned wb0 = plane_tile_write_offset<true>(tid);
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  const unsigned ra0 = 64u * (wm * 64 + i) + 16u * ((unsigned)h ^ swz);
  const unsigned ra1 = 64u * (wm * 64 + i) + 16u * ((unsigned)(2 + h) ^ swz);
  const unsigned rb0 = 256u * (8 * h + q4) + 64u * ((2 * wn) ^ q4) + 32u * g1 + 8u * p4;
  const unsigned rb1 = 256u * (8 * h + q4) + 64u * ((2 * wn + 1) ^ q4) + 32u * g1 + 8u * p4;
  lds_byte* const scr = imgB + NP * 8192 + 4096 * wave;
  auto step = [&](int kb, u32x4 (&zv)[4]) {
    {
      unsigned pk[4][2][NP];
      u32x4 zt[4];
      wave_block_transpose(scr, lane, zv, zt);
      const f32x16 g = fused_g_block(Gs, Ms);
      fused_dz_block<NP>(zt, g, rc_m, rc_rs, rc_d, pk, nullptr);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) *(lds_u32x2*)(imgA + wa[j] + 8192 * pl) = u32x2{pk[j][0][pl], pk[j][1][pl]};
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      *(lds_u32x4*)(imgB + wb0 + 8192 * pl) = vb[pl][0];
      *(lds_u32x4*)(imgB + wb0 + 4096 + 8192 * pl) = vb[pl][1];
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch_small(kb + 1 < nkb ? kb + 1 : nkb - 1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_z(kb + 2 < nkb ? kb + 2 : nkb - 1, zv);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    unrolled<2>([&](auto S_) {
      constexpr int ks = S_.value;
      const Frag<NP> fa0 = read_frag<false, NP, ks, 0>(imgA, ra0, ra1), fa1 = read_frag<false, NP, ks, 1>(imgA, ra0, ra1);
      const Frag<NP> fb0 = read_frag<true, NP, ks, 0>(imgB, rb0, rb1), fb1 = read_frag<true, NP, ks, 1>(imgB, rb0, rb1);
      acc[0][0] = split_products<NP>(fa0, fb0, acc[0][0]);
      acc[0][1] = split_products<NP>(fa0, fb1, acc[0][1]);
      acc[1][0] = split_products<NP>(fa1, fb0, acc[1][0]);
      acc[1][1] = split_products<NP>(fa1, fb1, acc[1][1]);
    });
    __syncthreads();
  };
  for (int kb = 0; kb < nkb; kb += 2) {                       // (nkb is even: kchunk % 64 == 0 and T % 64 == 0)
    step(kb, zv0);
    step(kb + 1, zv1);
  }
  split_epilogue(acc, dH, 128, nullptr, 0, 1, r0, 0, wm, wn, i, h, nullptr, 0);
}

// gngf_hpd_bwd_prepare: X (rows, 128) fp32 -> NP planes (NP, rows, 128), and V (rows, Lv) -> three planes (3, rows, 16) zero-padded
// (V = mw as it is; V = G^T through the strides: element (row, l) at V[row * vs_row + l * vs_l])
template <int NP>
__global__ void __launch_bounds__(256)
hpd_prepare_kernel(const float* __restrict__ X, unsigned short* __restrict__ Xp, const float* __restrict__ V, int Lv, int64_t vs_row,
                   int64_t vs_l, unsigned short* __restrict__ Vp, int64_t rows) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (e < rows * 128) {
    const float4 a = *reinterpret_cast<const float4*>(X + e), b = *reinterpret_cast<const float4*>(X + e + 4);
    unsigned o[4][NP];
    split_pair<NP>(a.x, a.y, o[0]); split_pair<NP>(a.z, a.w, o[1]); split_pair<NP>(b.x, b.y, o[2]); split_pair<NP>(b.z, b.w, o[3]);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x4*>(Xp + pl * rows * 128 + e) = u32x4{o[0][pl], o[1][pl], o[2][pl], o[3][pl]};
  }
  if (e < rows * 16) {                                         // the first rows * 2 threads also write eight levels of one row of V's planes
    const int64_t row = e >> 4;
    const int l0 = (int)(e & 15);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (l0 + j < Lv) ? V[row * vs_row + (l0 + j) * vs_l] : 0.f;
    const Split3 s3 = split8(x);
    *reinterpret_cast<u32x4*>(Vp + e) = s3.hi;
    *reinterpret_cast<u32x4*>(Vp + rows * 16 + e) = s3.mid;
    *reinterpret_cast<u32x4*>(Vp + 2 * rows * 16 + e) = s3.lo;
  }
}

// the K top-K slots of a row add a = p_k dq_k to dz[r, topk_idx]: dW[slot,:] += a h[r,:], dH[r,:] += a W[slot,:], db[slot] += a.
// One block of 128 threads per (row, k).
__global__ void __launch_bounds__(128)
hpd_topk_side_kernel(const float* __restrict__ pk, const float* __restrict__ dq, const int32_t* __restrict__ topi,
                     const float* __restrict__ Hh, const float* __restrict__ W, float* __restrict__ dW, float* __restrict__ db,
                     float* __restrict__ dH, int K) {
  const int64_t e = blockIdx.x, r = e / K;
  const float a = pk[e] * dq[e];
  if (a == 0.f) return;
  const int64_t slot = topi[e];
  const int c = threadIdx.x;
  atomicAdd(dW + slot * 128 + c, a * Hh[r * 128 + c]);
  atomicAdd(dH + r * 128 + c, a * W[slot * 128 + c]);
  if (c == 0 && db) atomicAdd(db + slot, a);
}

static int g_split_bf16 = 0;       // see gngf_set_gemm_split_bf16

// column sums of dZ = dY * act'(Y):  db[n] = sum_m dZ[m][n].  grid.x = ceil(N/64), grid.y = row slices; atomics.
__global__ void __launch_bounds__(256)
colsum_kernel(const float* __restrict__ dY, const float* __restrict__ Ymask, int mask_act, float* __restrict__ db,
              int64_t M, int64_t N, int64_t rows_per_block) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, c = tid & 63, rsub = tid >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float s = 0.f;
  if (col < N)
    for (int64_t r = r0 + rsub; r < r1; r += 4) {
      float v = dY[r * N + col];
      if (Ymask) v *= act_bwd_from_y(Ymask[r * N + col], mask_act);
      s += v;
    }
  red[rsub][c] = s;
  __syncthreads();
  if (rsub == 0 && col < N) atomicAdd(db + col, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
}

template <bool TA, bool TB>
static int launch_gemm(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                       int64_t ldc, const float* bias, int act, const float* amask, int mask_act, int splitk,
                       hipStream_t s, bool force_atomic = false, float2* rowparts = nullptr) {
  if (M == 0 || N == 0) return 0;
  if (rowparts) {      // row statistics in the epilogue: the split-bf16 128 x 128 kernel only, whole tiles, one K slice, plain stores
    const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0;
    const bool windows32 = lda < (1 << 22) && ldb < (1 << 22) && ldc < (1 << 22);
    if (!(M % BM2 == 0 && N % BN2 == 0 && K % 32 == 0 && !amask && aligned && windows32 && splitk <= 1 && !force_atomic && act == 0))
      return (int)hipErrorInvalidValue;
    const unsigned tm_ = (unsigned)(M / BM2), tn_ = (unsigned)(N / BN2);
    if (g_split_bf16 == 17)
      gemm128_split_kernel<TA, TB><<<dim3(tm_ * tn_, 1), dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, K, 0, (int)tm_, (int)tn_,
                                                                         rowparts, (int)(N / 64));
    else
      gemm128_planes_kernel<TA, TB, 3><<<dim3(tm_ * tn_, 1), dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, K, 0, (int)tm_, (int)tn_,
                                                                             rowparts, (int)(N / 64));
    return (int)hipGetLastError();
  }
  int64_t kchunk = K;
  if (splitk > 1) {
    kchunk = ceil_div(ceil_div(K, splitk), BK) * BK;
    splitk = (int)ceil_div(K, kchunk);
  } else {
    splitk = 1;
  }
  if (M >= BM2 && N >= BN2) {          // large problems: 128x128 tiles, 2x2 accumulators per wave
    dim3 grid2((unsigned)ceil_div(N, BN2), (unsigned)ceil_div(M, BM2), (unsigned)splitk);
    const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0;
    const bool windows32 = lda < (1 << 22) && ldb < (1 << 22) && ldc < (1 << 22);   // 128 rows * ld * 4 B < 2^31
    if (M % BM2 == 0 && N % BN2 == 0 && !amask && aligned && windows32) {
      const dim3 gridf(grid2.x * grid2.y, (unsigned)splitk);
      if (g_split_bf16 && K % 32 == 0 && kchunk % 32 == 0) {
        const int ato = (splitk > 1 || force_atomic) ? 1 : 0;
        // two planes only where the result is ACCUMULATED over a long contraction (the backward pass's dW and dh: force_atomic)
        if (g_split_bf16 == 2 && force_atomic)
          gemm128_planes_kernel<TA, TB, 2><<<gridf, dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, kchunk, ato, (int)grid2.y, (int)grid2.x);
        else if (g_split_bf16 == 17)
          gemm128_split_kernel<TA, TB><<<gridf, dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, kchunk, ato, (int)grid2.y, (int)grid2.x);
        else
          gemm128_planes_kernel<TA, TB, 3><<<gridf, dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, kchunk, ato, (int)grid2.y, (int)grid2.x);
        return (int)hipGetLastError();
      }
      if (K % kFastBK == 0 && kchunk % kFastBK == 0) {
        gemm128_fast_kernel<TA, TB, kFastBK><<<gridf, dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, kchunk,
                                                                        (splitk > 1 || force_atomic) ? 1 : 0, (int)grid2.y, (int)grid2.x);
        return (int)hipGetLastError();
      }
      if (K % BK == 0 && kchunk % BK == 0) {
        gemm128_fast_kernel<TA, TB, BK><<<gridf, dim3(256), 0, s>>>(A, B, C, lda, ldb, ldc, bias, act, K, kchunk,
                                                                   (splitk > 1 || force_atomic) ? 1 : 0, (int)grid2.y, (int)grid2.x);
        return (int)hipGetLastError();
      }
    }
    gemm128_kernel<TA, TB><<<grid2, dim3(256), 0, s>>>(A, B, C, M, N, K, lda, ldb, ldc, bias, act, amask, mask_act, kchunk,
                                                      (splitk > 1 || force_atomic) ? 1 : 0);
    return (int)hipGetLastError();
  }
  dim3 grid((unsigned)ceil_div(N, BN), (unsigned)ceil_div(M, BM), (unsigned)splitk);
  gemm_kernel<TA, TB><<<grid, dim3(256), 0, s>>>(A, B, C, M, N, K, lda, ldb, ldc, bias, act, amask, mask_act, kchunk,
                                                (splitk > 1 || force_atomic) ? 1 : 0);
  return (int)hipGetLastError();
}

}  // namespace gngf

using namespace gngf;

// Large aligned GEMMs (full 128 x 128 tiles, K % 32 == 0) of the entry points below run on the split-bf16 kernel while this
// is on (process-wide switch; returns the previous setting).  Off by default.
extern "C" int gngf_set_gemm_split_bf16(int on) {
  const int prev = g_split_bf16;
  g_split_bf16 = (on == 2 || on == 17) ? on : (on ? 1 : 0);
  return prev;
}

// Y[M,N] = act(X[M,K] * W[N,K]^T + b)          nn.Linear + activation (models.py:84-85, 386-389)
extern "C" int gngf_linear_fwd(const float* X, const float* W, const float* b, float* Y, int64_t M, int N, int K, int act,
                               void* stream) {
  GNGF_CHECK_ARG(M >= 0 && N > 0 && K > 0 && act >= 0 && act <= 3);
  if (M == 0) return 0;
  GNGF_CHECK_ARG(X && W && Y);
  return launch_gemm<false, true>(X, W, Y, M, N, K, K, K, N, b, act, nullptr, 0, 1, as_stream(stream));
}

// The same product (no activation) with ROW STATISTICS of Y in the epilogue: rowparts (M, N / 64) pairs (max, sum exp(y - max)) over
// each row's 64-column blocks, for gngf_rowstats_topk.  Only for whole 128 x 128 tiles (M % 128 == 0, N % 128 == 0, K % 32 == 0,
// 16-byte aligned operands): anything else is rejected (the caller then takes gngf_linear_fwd + gngf_logits_topk_pbar).  Always on
// the split-bf16 kernel (exact three-way split, fp32 accumulation).
extern "C" int gngf_linear_fwd_rowstats(const float* X, const float* W, const float* b, float* Y, float* rowparts, int64_t M, int N,
                                        int K, void* stream) {
  GNGF_CHECK_ARG(M >= 0 && N > 0 && K > 0);
  if (M == 0) return 0;
  GNGF_CHECK_ARG(X && W && Y && rowparts);
  return launch_gemm<false, true>(X, W, Y, M, N, K, K, K, N, b, 0, nullptr, 0, 1, as_stream(stream), false,
                                  reinterpret_cast<float2*>(rowparts));
}

// dX[M,K] = (dY .* act'(Y))[M,N] * W[N,K]      Y = the layer's activated output (NULL / act 0: no activation)
extern "C" int gngf_linear_bwd_input(const float* dY, const float* Y, const float* W, float* dX, int64_t M, int N, int K,
                                     int act, void* stream) {
  GNGF_CHECK_ARG(M >= 0 && N > 0 && K > 0 && act >= 0 && act <= 3);
  if (M == 0) return 0;
  GNGF_CHECK_ARG(dY && W && dX && (act == 0 || Y));
  return launch_gemm<false, false>(dY, W, dX, M, K, N, N, K, K, nullptr, 0, act ? Y : nullptr, act, 1, as_stream(stream));
}

// dW[N,K] += (dY .* act'(Y))^T * X ;  db[N] += colsum(dY .* act'(Y)).   dW and db must be zero-filled by the caller
// when a fresh gradient is wanted (split over the M contraction, accumulated with float atomics).
extern "C" int gngf_linear_bwd_weight(const float* dY, const float* Y, const float* X, float* dW, float* db, int64_t M, int N,
                                      int K, int act, void* stream) {
  GNGF_CHECK_ARG(M >= 0 && N > 0 && K > 0 && act >= 0 && act <= 3);
  if (M == 0) return 0;
  GNGF_CHECK_ARG(dY && X && dW && (act == 0 || Y));
  const int tsz = (N >= BM2 && K >= BN2) ? BM2 : BM;
  const int64_t tiles = ceil_div(N, tsz) * ceil_div(K, tsz);
  int splitk = (int)((1024 + tiles - 1) / tiles);
  const int64_t max_split = ceil_div(M, 256);
  if (splitk > max_split) splitk = (int)max_split;
  int rc = launch_gemm<true, false>(dY, X, dW, N, K, M, N, K, K, nullptr, 0, act ? Y : nullptr, act, splitk, as_stream(stream),
                                    /*force_atomic=*/true);
  if (rc) return rc;
  if (db) {
    // enough row slices for ~512 blocks (a 2048 x 128 gradient on 2 blocks took 138 us, all load latency)
    const int64_t slices = ceil_div(512, ceil_div(N, 64));
    int64_t rows_per_block = ceil_div(M, slices);
    rows_per_block = rows_per_block < 32 ? 32 : (rows_per_block > 2048 ? 2048 : rows_per_block);
    dim3 grid((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, rows_per_block));
    colsum_kernel<<<grid, dim3(256), 0, as_stream(stream)>>>(dY, act ? Y : nullptr, act, db, M, N, rows_per_block);
  }
  GNGF_RETURN_LAUNCH();
}

// General accumulate-GEMM used by the HPD path:  C[M,N] += opA(A) * opB(B), contraction length Kc split over
// blocks (float atomics; C zero-filled by the caller for a fresh result).
//   ta = 0: A is (M,Kc) row-major, ta = 1: A is (Kc,M) row-major;  tb = 0: B is (Kc,N), tb = 1: B is (N,Kc).
extern "C" int gngf_gemm_acc(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kc, int ta, int tb,
                             void* stream) {
  GNGF_CHECK_ARG(M >= 0 && N >= 0 && Kc >= 0);
  if (M == 0 || N == 0 || Kc == 0) return 0;
  GNGF_CHECK_ARG(A && B && C);
  const int tsz = (M >= BM2 && N >= BN2) ? BM2 : BM;
  const int64_t tiles = ceil_div(M, tsz) * ceil_div(N, tsz);
  int splitk = (int)ceil_div(1024, tiles);
  const int64_t max_split = ceil_div(Kc, 64);
  if (splitk > max_split) splitk = (int)max_split;
  if (splitk < 1) splitk = 1;
  const int64_t lda = ta ? M : Kc, ldb = tb ? Kc : N;
  hipStream_t s = as_stream(stream);
  if (!ta && !tb) return launch_gemm<false, false>(A, B, C, M, N, Kc, lda, ldb, N, nullptr, 0, nullptr, 0, splitk, s, true);
  if (!ta && tb) return launch_gemm<false, true>(A, B, C, M, N, Kc, lda, ldb, N, nullptr, 0, nullptr, 0, splitk, s, true);
  if (ta && !tb) return launch_gemm<true, false>(A, B, C, M, N, Kc, lda, ldb, N, nullptr, 0, nullptr, 0, splitk, s, true);
  return launch_gemm<true, true>(A, B, C, M, N, Kc, lda, ldb, N, nullptr, 0, nullptr, 0, splitk, s, true);
}

// Does gngf_hpd_bwd_fused take this shape?  (whole 128-row / 128-column tiles, last hidden width 128, L <= 16, 32-bit windows)
extern "C" int gngf_hpd_bwd_fused_applies(int64_t U, int64_t T, int L, int K, int hidden) {
  return (U > 0 && U % 128 == 0 && T % 128 == 0 && T < (1 << 22) && hidden == 128 && L >= 0 && L <= 16 && K >= 0 && K <= GNGF_MAX_TOPK) ? 1 : 0;
}

// The small operands of gngf_hpd_bwd_fused, split once per backward pass into bf16 planes (planes = 2 or 3 for h and W; always the
// exact three for the operands of the small product g = mw G):
//   h (rows, 128) -> hp (planes, rows, 128),  mw (rows, L) -> mwp (3, rows, 16) zero-padded      (rows = all vertices of the step)
//   W (T, 128)    -> Wp (planes, T, 128),     G (L, T)     -> Gtp (3, T, 16) transposed, zero-padded
// Either half may be skipped (h == NULL / W == NULL).  L = 0: mw / G may be NULL, their planes are zero-filled.
extern "C" int gngf_hpd_bwd_prepare(const float* h, const float* mw, int64_t rows, void* hp, void* mwp, const float* W, const float* G,
                                    int64_t T, void* Wp, void* Gtp, int L, int hidden, int planes, void* stream) {
  GNGF_CHECK_ARG(rows >= 0 && T >= 0 && hidden == 128 && L >= 0 && L <= 16 && (planes == 2 || planes == 3));
  hipStream_t s = as_stream(stream);
  if (h && rows > 0) {
    GNGF_CHECK_ARG(hp && mwp && (L == 0 || mw));
    const dim3 g((unsigned)ceil_div(rows * 128 / 8, 256));
    if (planes == 2) hpd_prepare_kernel<2><<<g, dim3(256), 0, s>>>(h, static_cast<unsigned short*>(hp), mw, L, L, 1, static_cast<unsigned short*>(mwp), rows);
    else hpd_prepare_kernel<3><<<g, dim3(256), 0, s>>>(h, static_cast<unsigned short*>(hp), mw, L, L, 1, static_cast<unsigned short*>(mwp), rows);
  }
  if (W && T > 0) {
    GNGF_CHECK_ARG(Wp && Gtp && (L == 0 || G));
    const dim3 g((unsigned)ceil_div(T * 128 / 8, 256));
    if (planes == 2) hpd_prepare_kernel<2><<<g, dim3(256), 0, s>>>(W, static_cast<unsigned short*>(Wp), G, L, 1, T, static_cast<unsigned short*>(Gtp), T);
    else hpd_prepare_kernel<3><<<g, dim3(256), 0, s>>>(W, static_cast<unsigned short*>(Wp), G, L, 1, T, static_cast<unsigned short*>(Gtp), T);
  }
  GNGF_RETURN_LAUNCH();
}

// Backward of the HashProbDistribution's last layer from the logits of one chunk (see hpd_dw_fused_kernel):
//   dW (T,128) += dz^T h,  db (T) += colsum dz,  dH (U,128) += dz W      with dz = p (mw G - dot) + [top-K slots] p_k dq_k
// logits (U,T), rowstat (U,2) = (max, sum exp), dot (U) from gngf_hpd_bwd_dot, dq / topk_p / topk_idx (U,K) or K = 0;
// hp / mwp = the planes of gngf_hpd_bwd_prepare AT THE CHUNK'S FIRST ROW (pointer + u0 * 128 resp. + u0 * 16 elements; rows_total = the
// rows they were prepared with: the plane stride), Wp / Gtp as prepared; h (U,128) and W (T,128) in fp32 for the K top-K terms (K = 0: unused).
// planes: as prepared.  + 16 / + 32 / + 48 on `planes` run the dW / the dH / the top-K part alone (diagnostic).
extern "C" int gngf_hpd_bwd_fused(const float* logits, const float* rowstat, const float* dot, const float* dq, const float* topk_p,
                                  const int32_t* topk_idx, const void* hp, const void* mwp, int64_t rows_total, const void* Wp,
                                  const void* Gtp, const float* h, const float* W, float* dW, float* db, float* dH, int64_t U,
                                  int64_t T, int K, int hidden, int planes, void* stream) {
  const int only = planes >> 4;
  planes &= 15;
  GNGF_CHECK_ARG(gngf_hpd_bwd_fused_applies(U, T, 0, K, hidden) && (planes == 2 || planes == 3) && rows_total >= U);
  GNGF_CHECK_ARG(logits && rowstat && dot && hp && mwp && Wp && Gtp && dW && dH && (K == 0 || (dq && topk_p && topk_idx && h && W)));
  hipStream_t s = as_stream(stream);
  const unsigned short *hp_ = static_cast<const unsigned short*>(hp), *mwp_ = static_cast<const unsigned short*>(mwp);
  const unsigned short *Wp_ = static_cast<const unsigned short*>(Wp), *Gtp_ = static_cast<const unsigned short*>(Gtp);
  const dim3 gw((unsigned)(T / 128));
  // dh: enough column slices for ~1024 workgroups, each an even number of 32-column K-blocks
  const int64_t rtiles = U / 128;
  int64_t slices = ceil_div(1024, rtiles);
  int64_t kchunk = ceil_div(ceil_div(T, slices), 64) * 64;
  slices = ceil_div(T, kchunk);
  const dim3 gh((unsigned)(rtiles * slices));
  if (planes == 2) {
    if (only == 0 || only == 1) hpd_dw_fused_kernel<2><<<gw, dim3(256), 0, s>>>(logits, rowstat, dot, mwp_, Gtp_, hp_, rows_total, dW, db, U, T);
    if (only == 0 || only == 2) hpd_dh_fused_kernel<2><<<gh, dim3(256), 0, s>>>(logits, rowstat, dot, mwp_, Gtp_, Wp_, rows_total, dH, U, T, kchunk, (int)rtiles, (int)slices);
  } else {
    if (only == 0 || only == 1) hpd_dw_fused_kernel<3><<<gw, dim3(256), 0, s>>>(logits, rowstat, dot, mwp_, Gtp_, hp_, rows_total, dW, db, U, T);
    if (only == 0 || only == 2) hpd_dh_fused_kernel<3><<<gh, dim3(256), 0, s>>>(logits, rowstat, dot, mwp_, Gtp_, Wp_, rows_total, dH, U, T, kchunk, (int)rtiles, (int)slices);
  }
  if (K > 0 && (only == 0 || only == 3))
    hpd_topk_side_kernel<<<dim3((unsigned)(U * K)), dim3(128), 0, s>>>(topk_p, dq, topk_idx, h, W, dW, db, dH, K);
  GNGF_RETURN_LAUNCH();
}
