// "Direct" form of the encoder hot path: one lane per (pixel, level) [fused kernels] or per
// (pixel, level, corner) [module-boundary kernels]; table rows are gathered straight from HBM / L2 /
// Infinity Cache and gradients are scattered with global float atomics.
//
// This is the general path: any (L, F, T, K), any coordinate distribution, and levels whose vertex grid
// is larger than the table.  The tiled/LDS-privatised path (encode_tiled.hip) is the fast path for
// levels whose vertex grid is small enough to stage.
//
// Replaces (reference file:line):  models.py:486-528 (_scale_to_grid, _fast_hash), models.py:173-229
// (MultiResHashEncoding.forward), models.py:621-655 (_bilinear_interpolate) and the autograd backward of
// those (embedding_dense_backward scatter-add, softmax-over-K backward).
#include "gngf_common.h"

namespace gngf {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// a5 + a6: hashed indices as the module returns them, (P, L, 4) int64.
__global__ void __launch_bounds__(kBlock)
hash_indices_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, int64_t* __restrict__ idx,
                    int64_t total /* P*L */, int L, int64_t T, bool pow2) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  const int64_t p = gid / L;
  const int l = (int)(gid - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
  longlong4 o;
  o.x = spatial_hash(cell.gx, cell.gy, T, pow2);
  o.y = spatial_hash(cell.gx + 1, cell.gy, T, pow2);
  o.z = spatial_hash(cell.gx, cell.gy + 1, T, pow2);
  o.w = spatial_hash(cell.gx + 1, cell.gy + 1, T, pow2);
  reinterpret_cast<longlong4*>(idx)[gid] = o;
}

// ------------------------------------------------------------------------------------------------
// Blend weights of the K looked-up rows (models.py:212-217) and their backward.
template <int KMAX>
__device__ __forceinline__ void blend_weights(const float* q, int K, int blend, float* w) {
  if (blend == GNGF_BLEND_RAW) {
    for (int k = 0; k < K; ++k) w[k] = q[k];
  } else if (blend == GNGF_BLEND_SOFTMAX) {
    float m = q[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, q[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) { w[k] = expf(q[k] - m); s += w[k]; }
    for (int k = 0; k < K; ++k) w[k] = w[k] / s;
  } else {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += q[k];
    for (int k = 0; k < K; ++k) w[k] = q[k] / s;
  }
}

// d (dL/dw_k) -> dq (dL/dq_k), in place.
__device__ __forceinline__ void blend_backward(const float* q, const float* w, int K, int blend, float* d) {
  if (blend == GNGF_BLEND_RAW) return;
  if (blend == GNGF_BLEND_SOFTMAX) {
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += w[k] * d[k];
    for (int k = 0; k < K; ++k) d[k] = w[k] * (d[k] - dot);
  } else {
    float s = 0.f, dq = 0.f;
    for (int k = 0; k < K; ++k) { s += q[k]; dq += d[k] * q[k]; }
    const float inv = 1.0f / s;
    for (int k = 0; k < K; ++k) d[k] = d[k] * inv - dq * inv * inv;
  }
}

// ------------------------------------------------------------------------------------------------
// a10/a11: MultiResHashEncoding.forward at the module boundary.  One lane per (p, l, v).
// out (P,F,L,4): for fixed (p,f) the (l,v) plane is contiguous, so a wave writes 256 contiguous bytes per f.
template <int F, typename TT>
__global__ void __launch_bounds__(kBlock)
mrhe_fwd_kernel(const TT* __restrict__ tables, const int64_t* __restrict__ idx, const float* __restrict__ probs,
                float* __restrict__ out, int64_t total /* P*L*4 */, int L, int64_t T, int K, int blend) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  const int L4 = L * 4;
  const int64_t p = gid / L4;
  const int lv = (int)(gid - p * L4);
  const int l = lv >> 2;
  const TT* tab = tables + (int64_t)l * T * F;
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  if (K == 0) {
    const int64_t row = idx[gid];
    const TT* r = tab + row * F;
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = tload(r + f);
  } else {
    float q[GNGF_MAX_TOPK], w[GNGF_MAX_TOPK];
    for (int k = 0; k < K; ++k) q[k] = probs[gid * K + k];
    blend_weights<GNGF_MAX_TOPK>(q, K, blend, w);
    for (int k = 0; k < K; ++k) {
      const TT* r = tab + idx[gid * K + k] * F;
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] += tload(r + f) * w[k];
    }
  }
  float* o = out + p * (int64_t)F * L4 + lv;
#pragma unroll
  for (int f = 0; f < F; ++f) o[(int64_t)f * L4] = acc[f];
}

template <int F, typename TT>
__global__ void __launch_bounds__(kBlock)
mrhe_bwd_kernel(const TT* __restrict__ tables, const int64_t* __restrict__ idx, const float* __restrict__ probs,
                const float* __restrict__ gout, float* __restrict__ dtables, float* __restrict__ dprobs,
                int64_t total, int L, int64_t T, int K, int blend) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  const int L4 = L * 4;
  const int64_t p = gid / L4;
  const int lv = (int)(gid - p * L4);
  const int l = lv >> 2;
  const TT* tab = tables + (int64_t)l * T * F;
  float* dtab = dtables + (int64_t)l * T * F;
  float g[F];
  const float* gp = gout + p * (int64_t)F * L4 + lv;
#pragma unroll
  for (int f = 0; f < F; ++f) g[f] = gp[(int64_t)f * L4];
  if (K == 0) {
    float* r = dtab + idx[gid] * F;
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(r + f, g[f]);
    return;
  }
  float q[GNGF_MAX_TOPK], w[GNGF_MAX_TOPK], d[GNGF_MAX_TOPK];
  for (int k = 0; k < K; ++k) q[k] = probs[gid * K + k];
  blend_weights<GNGF_MAX_TOPK>(q, K, blend, w);
  for (int k = 0; k < K; ++k) {
    const int64_t row = idx[gid * K + k];
    const TT* r = tab + row * F;
    float dot = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      dot += g[f] * tload(r + f);
      atomicAdd(dtab + row * F + f, g[f] * w[k]);
    }
    d[k] = dot;
  }
  if (dprobs) {
    blend_backward(q, w, K, blend, d);
    for (int k = 0; k < K; ++k) dprobs[gid * K + k] = d[k];
  }
}

// ------------------------------------------------------------------------------------------------
// a12 at the module boundary: feats (P,F,L,4) -> enc (P, L*F).  One lane per (p, l).
template <int F>
__global__ void __launch_bounds__(kBlock)
bilinear_fwd_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, const float* __restrict__ feats,
                    float* __restrict__ enc, int64_t total /* P*L */, int L) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  const int64_t p = gid / L;
  const int l = (int)(gid - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const float4 v = *reinterpret_cast<const float4*>(feats + ((p * F + f) * L + l) * 4);
    // models.py:642-646: weighted = feats * coeffs; sum over the 4 corners
    enc[gid * F + f] = ((v.x * cell.c[0] + v.y * cell.c[1]) + v.z * cell.c[2]) + v.w * cell.c[3];
  }
}

template <int F>
__global__ void __launch_bounds__(kBlock)
bilinear_bwd_kernel(const float2* __restrict__ xy, const int32_t* __restrict__ n_ls, const float* __restrict__ genc,
                    float* __restrict__ dfeats, int64_t total, int L) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  const int64_t p = gid / L;
  const int l = (int)(gid - p * L);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const float g = genc[gid * F + f];
    float4 v = {g * cell.c[0], g * cell.c[1], g * cell.c[2], g * cell.c[3]};
    *reinterpret_cast<float4*>(dfeats + ((p * F + f) * L + l) * 4) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused direct encoder: coords -> (P, L*F).  One lane per (pixel, level): the L lanes of a pixel write one
// contiguous L*F*4-byte row (128 B at L=16, F=2).
template <int F, bool VT, typename TT>
__global__ void __launch_bounds__(kBlock)
encode_fwd_kernel(const float2* __restrict__ xy, const TT* __restrict__ tables,
                  const int32_t* __restrict__ vert_idx, const float* __restrict__ vert_w,
                  const int32_t* __restrict__ n_ls, float* __restrict__ enc,
                  int64_t total, int L, int l0, int nl, int64_t T, int K, int vstride, int64_t NV, bool pow2,
                  const float4* __restrict__ order = nullptr) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  int64_t p = gid / nl;
  const int l = l0 + (int)(gid - p * nl);
  // order (optional): the pixels in TILE order (binned records {x, y, bits(original index), 0} of the tiled form) — neighbouring
  // lanes then gather from the same few aligned blocks of table rows (the spatial hash keeps the low bits of gx: one grid row =
  // one block) instead of from anywhere in the table; the result lands in the pixel's own row of enc either way
  float2 c;
  if (order) { const float4 s = order[p]; c = make_float2(s.x, s.y); p = (int64_t)__float_as_int(s.z); }
  else c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
  const TT* tab = tables + (int64_t)l * T * F;
  float feat[4][F];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int gx = cell.gx + (v & 1), gy = cell.gy + (v >> 1);
    if constexpr (!VT) {
      const TT* r = tab + spatial_hash(gx, gy, T, pow2) * F;
#pragma unroll
      for (int f = 0; f < F; ++f) feat[v][f] = tload(r + f);
    } else {
      int64_t vid = (int64_t)gy * vstride + gx;
      vid = vid < 0 ? 0 : (vid >= NV ? NV - 1 : vid);   // never fault on out-of-domain coordinates
#pragma unroll
      for (int f = 0; f < F; ++f) feat[v][f] = 0.f;
      for (int k = 0; k < K; ++k) {
        const float w = vert_w[vid * K + k];
        const TT* r = tab + (int64_t)vert_idx[vid * K + k] * F;
#pragma unroll
        for (int f = 0; f < F; ++f) feat[v][f] += tload(r + f) * w;
      }
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f)
    enc[(p * L + l) * F + f] = ((feat[0][f] * cell.c[0] + feat[1][f] * cell.c[1]) + feat[2][f] * cell.c[2]) + feat[3][f] * cell.c[3];
}

// Backward, direct form.  One lane per (pixel, level, FEATURE), feature fastest: the F lanes of a corner add to F
// consecutive floats of one table row, so one wave-instruction's atomics fall into 64/F rows instead of 64 — the
// memory-side atomic unit works in 64-byte requests and merges the lanes of a row (F = 4: 4x fewer requests).
template <int F, bool VT, typename TT>
__global__ void __launch_bounds__(kBlock)
encode_bwd_kernel(const float2* __restrict__ xy, const TT* __restrict__ tables,
                  const int32_t* __restrict__ vert_idx, const float* __restrict__ vert_w,
                  const int32_t* __restrict__ n_ls, const float* __restrict__ genc,
                  float* __restrict__ dtables, float* __restrict__ dvert_w,
                  int64_t total /* P*nl*F */, int L, int l0, int nl, int64_t T, int K, int vstride, int64_t NV, bool pow2) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= total) return;
  const int f = (int)(gid % F);
  const int64_t pl = gid / F;
  const int64_t p = pl / nl;
  const int l = l0 + (int)(pl - p * nl);
  const float2 c = xy[p];
  const Cell cell = make_cell(c.x, c.y, n_ls[l]);
  const TT* tab = tables + (int64_t)l * T * F;
  float* dtab = dtables + (int64_t)l * T * F;
  const float g = genc[(p * L + l) * F + f];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int gx = cell.gx + (v & 1), gy = cell.gy + (v >> 1);
    const float cv = cell.c[v];
    if constexpr (!VT) {
      atomicAdd(dtab + spatial_hash(gx, gy, T, pow2) * F + f, g * cv);
    } else {
      int64_t vid = (int64_t)gy * vstride + gx;
      vid = vid < 0 ? 0 : (vid >= NV ? NV - 1 : vid);
      for (int k = 0; k < K; ++k) {
        const float w = vert_w[vid * K + k];
        const int64_t row = vert_idx[vid * K + k];
        atomicAdd(dtab + row * F + f, (g * cv) * w);
        if (dvert_w) {                       // <g, E_l[row]> over the F feature lanes of this corner (adjacent lanes)
          float dot = g * tload(tab + row * F + f);
#pragma unroll
          for (int o = 1; o < F; o <<= 1) dot += __shfl_xor(dot, o, 64);
          if (f == 0) atomicAdd(dvert_w + vid * K + k, dot * cv);
        }
      }
    }
  }
}

}  // namespace gngf

using namespace gngf;

#define DISPATCH_TT(dt, ...)                                                          \
  if ((dt) == GNGF_FEAT_F32) { using TT = float; __VA_ARGS__; }                       \
  else if ((dt) == GNGF_FEAT_F16) { using TT = __half; __VA_ARGS__; }                 \
  else return (int)hipErrorInvalidValue;

#define DISPATCH_F(F, ...)                          \
  switch (F) {                                      \
    case 1: { constexpr int kF = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int kF = 2; __VA_ARGS__; } break; \
    case 4: { constexpr int kF = 4; __VA_ARGS__; } break; \
    case 8: { constexpr int kF = 8; __VA_ARGS__; } break; \
    default: return (int)hipErrorInvalidValue;      \
  }

extern "C" int gngf_abi_version(void) { return GNGF_ABI_VERSION; }

extern "C" int gngf_hash_indices(const float* xy, const int32_t* n_ls, int64_t* idx, int64_t P, int L, int64_t T,
                                 void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && T > 0);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(xy && n_ls && idx);
  const int64_t total = P * L;
  hash_indices_kernel<<<dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, as_stream(stream)>>>(
      reinterpret_cast<const float2*>(xy), n_ls, idx, total, L, T, (T & (T - 1)) == 0);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_mrhe_fwd(const void* tables, int feat_dtype, const int64_t* idx, const float* probs, float* out,
                             int64_t P, int L, int F, int64_t T, int K, int blend, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && T > 0 && K >= 0 && K <= GNGF_MAX_TOPK);
  GNGF_CHECK_ARG(blend >= 0 && blend <= 2);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(tables && idx && out && (K == 0 || probs));
  const int64_t total = P * L * 4;
  DISPATCH_TT(feat_dtype, DISPATCH_F(F, (mrhe_fwd_kernel<kF, TT><<<dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0,
                                                                  as_stream(stream)>>>(
                              static_cast<const TT*>(tables), idx, probs, out, total, L, T, K, blend))));
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_mrhe_bwd(const void* tables, int feat_dtype, const int64_t* idx, const float* probs, const float* gout,
                             float* dtables, float* dprobs, int64_t P, int L, int F, int64_t T, int K, int blend,
                             void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && T > 0 && K >= 0 && K <= GNGF_MAX_TOPK);
  GNGF_CHECK_ARG(blend >= 0 && blend <= 2);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(tables && idx && gout && dtables && (K == 0 || probs));
  const int64_t total = P * L * 4;
  DISPATCH_TT(feat_dtype, DISPATCH_F(F, (mrhe_bwd_kernel<kF, TT><<<dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0,
                                                                  as_stream(stream)>>>(
                              static_cast<const TT*>(tables), idx, probs, gout, dtables, dprobs, total, L, T, K, blend))));
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_bilinear_fwd(const float* xy, const int32_t* n_ls, const float* feats, float* enc,
                                 int64_t P, int L, int F, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(xy && n_ls && feats && enc);
  const int64_t total = P * L;
  DISPATCH_F(F, (bilinear_fwd_kernel<kF><<<dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, as_stream(stream)>>>(
                    reinterpret_cast<const float2*>(xy), n_ls, feats, enc, total, L)));
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_bilinear_bwd(const float* xy, const int32_t* n_ls, const float* genc, float* dfeats,
                                 int64_t P, int L, int F, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(xy && n_ls && genc && dfeats);
  const int64_t total = P * L;
  DISPATCH_F(F, (bilinear_bwd_kernel<kF><<<dim3((unsigned)ceil_div(total, kBlock)), dim3(kBlock), 0, as_stream(stream)>>>(
                    reinterpret_cast<const float2*>(xy), n_ls, genc, dfeats, total, L)));
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_encode_fwd(const float* xy, const void* tables_v, int feat_dtype, const int32_t* vert_idx, const float* vert_w,
                               const int32_t* n_ls, float* enc, int64_t P, int L, int F, int64_t T, int K,
                               int mode, int vstride, int64_t NV, int l0, int l1, const float* pixel_order, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && T > 0 && l0 >= 0 && l0 <= l1 && l1 <= L);
  GNGF_CHECK_ARG(mode == GNGF_MODE_HASH || mode == GNGF_MODE_VERTEX_TABLE);
  if (P == 0 || l0 == l1) return 0;
  GNGF_CHECK_ARG(xy && tables_v && n_ls && enc && (reinterpret_cast<uintptr_t>(pixel_order) & 15) == 0);
  const float4* order = reinterpret_cast<const float4*>(pixel_order);
  const int nl = l1 - l0;
  const int64_t total = P * nl;
  const dim3 grid((unsigned)ceil_div(total, kBlock)), block(kBlock);
  const bool pow2 = (T & (T - 1)) == 0;
  if (mode == GNGF_MODE_HASH) {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (encode_fwd_kernel<kF, false, TT><<<grid, block, 0, as_stream(stream)>>>(
                      reinterpret_cast<const float2*>(xy), static_cast<const TT*>(tables_v), nullptr, nullptr, n_ls, enc, total, L, l0,
                      nl, T, 0, 0, 0, pow2, order))));
  } else {
    GNGF_CHECK_ARG(vert_idx && vert_w && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0);
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (encode_fwd_kernel<kF, true, TT><<<grid, block, 0, as_stream(stream)>>>(
                      reinterpret_cast<const float2*>(xy), static_cast<const TT*>(tables_v), vert_idx, vert_w, n_ls, enc, total, L,
                      l0, nl, T, K, vstride, NV, pow2, order))));
  }
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_encode_bwd(const float* xy, const void* tables_v, int feat_dtype, const int32_t* vert_idx, const float* vert_w,
                               const int32_t* n_ls, const float* genc, float* dtables, float* dvert_w,
                               int64_t P, int L, int F, int64_t T, int K, int mode, int vstride, int64_t NV,
                               int l0, int l1, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && L > 0 && L <= GNGF_MAX_LEVELS && T > 0 && l0 >= 0 && l0 <= l1 && l1 <= L);
  GNGF_CHECK_ARG(mode == GNGF_MODE_HASH || mode == GNGF_MODE_VERTEX_TABLE);
  if (P == 0 || l0 == l1) return 0;
  GNGF_CHECK_ARG(xy && tables_v && n_ls && genc && dtables);
  const int nl = l1 - l0;
  const int64_t total = P * nl * F;                       // one lane per (pixel, level, feature)
  const dim3 grid((unsigned)ceil_div(total, kBlock)), block(kBlock);
  const bool pow2 = (T & (T - 1)) == 0;
  if (mode == GNGF_MODE_HASH) {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (encode_bwd_kernel<kF, false, TT><<<grid, block, 0, as_stream(stream)>>>(
                      reinterpret_cast<const float2*>(xy), static_cast<const TT*>(tables_v), nullptr, nullptr, n_ls, genc, dtables,
                      nullptr, total, L, l0, nl, T, 0, 0, 0, pow2))));
  } else {
    GNGF_CHECK_ARG(vert_idx && vert_w && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0);
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (encode_bwd_kernel<kF, true, TT><<<grid, block, 0, as_stream(stream)>>>(
                      reinterpret_cast<const float2*>(xy), static_cast<const TT*>(tables_v), vert_idx, vert_w, n_ls, genc, dtables,
                      dvert_w, total, L, l0, nl, T, K, vstride, NV, pow2))));
  }
  GNGF_RETURN_LAUNCH();
}
