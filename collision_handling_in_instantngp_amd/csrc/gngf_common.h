// Shared device helpers for the gfx950 kernels.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/gngf.h"

#define GNGF_CHECK_ARG(cond) do { if (!(cond)) return (int)hipErrorInvalidValue; } while (0)
#define GNGF_RETURN_LAUNCH() return (int)hipGetLastError()

namespace gngf {

// Table element type: fp32 (the reference) or fp16 storage (BASELINE config 5); arithmetic is always fp32.
__device__ __forceinline__ float tload(const float* p) { return *p; }
__device__ __forceinline__ float tload(const __half* p) { return __half2float(*p); }

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// _fast_hash, models.py:504-528.  The reference multiplies an int32 tensor by a 0-dim int64 prime, which
// torch keeps in int32: the product wraps in 32 bits; the XOR sign-extends; remainder is non-negative.
__device__ __forceinline__ int64_t spatial_hash(int gx, int gy, int64_t T, bool pow2) {
  const int32_t h = gx ^ (int32_t)((uint32_t)gy * 2654435761u);
  const int64_t h64 = (int64_t)h;
  if (pow2) return h64 & (T - 1);
  int64_t r = h64 % T;
  return r < 0 ? r + T : r;
}

// Per-(pixel, level) cell: _scale_to_grid (models.py:486-502) + the coefficients of _bilinear_interpolate
// (models.py:632-637).  Every operation is a separately rounded fp32 op, as in the reference (compile with
// -ffp-contract=off).
struct Cell {
  int gx, gy;      // floor corner (vertex 0)
  float c[4];      // c0=(xd-x)(yd-y) c1=(x-xa)(yd-y) c2=(xd-x)(y-ya) c3=(x-xa)(y-ya)
};

__device__ __forceinline__ Cell make_cell(float x, float y, int n) {
  Cell r;
  const float fn = (float)n;
  const float sx = x * fn, sy = y * fn;
  const float ax = floorf(sx), ay = floorf(sy);
  const float dx = ax + 1.0f, dy = ay + 1.0f;
  const float wx0 = dx - sx, wx1 = sx - ax;
  const float wy0 = dy - sy, wy1 = sy - ay;
  r.c[0] = wx0 * wy0; r.c[1] = wx1 * wy0; r.c[2] = wx0 * wy1; r.c[3] = wx1 * wy1;
  r.gx = (int)ax; r.gy = (int)ay;
  return r;
}

__device__ __forceinline__ bool is_pow2(int64_t v) { return (v & (v - 1)) == 0; }

// Clearing a buffer from inside an entry point: a plain KERNEL, never hipMemsetAsync.  A memset NODE of a captured hipGraph was seen
// to run out of order on replays after the first (round 4: a table-gradient clear captured as a memset node inside
// gngf_decoder_bwd zeroed, on every replay but the first, part of a gradient buffer that a LATER step of the same graph had
// legitimately placed in the same pool memory) — kernel nodes keep their stream order.  bytes: a multiple of 4.
static __global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* __restrict__ p, int64_t nwords) {
  const int64_t n4 = nwords >> 2;
  uint4* p4 = reinterpret_cast<uint4*>(p);
  const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (aligned) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += stride) p4[e] = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t e = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; e < nwords; e += stride) p[e] = 0u;
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nwords; e += stride) p[e] = 0u;
  }
}
static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const int64_t nwords = (int64_t)(bytes / 4);
  int64_t blocks = (nwords / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  zero_words_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(static_cast<uint32_t*>(p), nwords);
  return hipGetLastError();
}

// torch.nn.MSELoss() value (csrc/loss.hip): block `blk` of `nblocks` 1024-thread blocks adds its share of sum((pred-label)^2)
// to a double and takes a ticket (the atomic's RETURN value feeds the ticket request, so the add is performed before the
// ticket exists — no release fence, which writes the whole L2 back on this chip); the last ticket holder writes
// loss = total / n and resets both words.  Called by mse_fwd_kernel and, as extra workgroups riding on its launch, by
// the tiled encoder backward.
__device__ __forceinline__ void mse_sum_block(int blk, int nblocks, const float* __restrict__ pred, const float* __restrict__ label,
                                              float* __restrict__ loss, double* __restrict__ acc, unsigned* __restrict__ counter,
                                              int64_t n) {
  constexpr int kThreads = 1024;
  __shared__ float mse_red[kThreads / 64];
  float s = 0.f, s2 = 0.f;
  const int64_t n4 = n >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(pred);
  const float4* l4 = reinterpret_cast<const float4*>(label);
  const int64_t stride = (int64_t)nblocks * kThreads;
  int64_t e = (int64_t)blk * kThreads + threadIdx.x;
  for (; e + stride < n4; e += 2 * stride) {              // two independent load pairs in flight per trip
    const float4 a = p4[e], b = l4[e], c = p4[e + stride], d = l4[e + stride];
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    const float ex = c.x - d.x, ey = c.y - d.y, ez = c.z - d.z, ew = c.w - d.w;
    s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    s2 += (ex * ex + ey * ey) + (ez * ez + ew * ew);
  }
  if (e < n4) {
    const float4 a = p4[e], b = l4[e];
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  s += s2;
  if (blk == 0)
    for (int64_t t = (n4 << 2) + threadIdx.x; t < n; t += kThreads) { const float d = pred[t] - label[t]; s += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) mse_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double b = 0.0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) b += (double)mse_red[w];
  const double before = atomicAdd(acc, b);
  const unsigned ticket = atomicAdd(counter, before < 0.0 ? 2u : 1u);      // sums of squares are never negative: always 1
  if (ticket != (unsigned)nblocks - 1u) return;
  const double total = atomicAdd(acc, 0.0);
  *loss = (float)(total / (double)n);
  atomicExch(reinterpret_cast<unsigned long long*>(acc), 0ull);
  atomicExch(counter, 0u);
}

// The fused training decoder (gngf_decoder_train) ran its backward with a PROMISED loss gradient before autograd delivered the
// real one.  The consumers of its results (the slab reduction below, the tiled encoder backward) compare the two on the device
// — no host synchronisation, always on — and poison what they write with NaN when the promise was broken: a wrong promise is
// loud (NaN loss / parameters at the next step), never a silently wrong gradient.  NULL pointers: nothing was promised.
__device__ __forceinline__ bool promise_broken(const float* __restrict__ promised, const float* __restrict__ arrived) {
  if (!promised || !arrived) return false;
  const float a = *promised, b = *arrived;
  return !(fabsf(a - b) <= 1e-6f * fabsf(a));            // (NaN on either side counts as broken)
}

// Decoder backward, second half (csrc/decoder.hip): sums the per-workgroup gradient slabs
//   dW0 [64*in_dim] | dW1 [64*64] | dW2 [out_dim*64] | db0 [64] | db1 [64] | db2 [out_dim] | max |d enc| (bit pattern)
// into the six gradient tensors; block `blk` of 1024 threads owns elements [64 blk, 64 blk + 64) (64 elements x 16
// slab-groups).  The last slot takes the maximum of bit patterns instead of a sum.  Called by decoder_reduce_kernel and,
// as extra workgroups riding on its own launch, by the tiled encoder backward (one kernel launch less on the step's
// critical path: a launch costs ~6 us inside a replayed step).
__device__ __forceinline__ void decoder_reduce_block(int blk, const float* __restrict__ slabs, int nslabs, int nslab, int in_dim,
                                                     int out_dim, float* __restrict__ dW0, float* __restrict__ db0,
                                                     float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
                                                     float* __restrict__ db2, float* __restrict__ absmax,
                                                     const float* __restrict__ promised = nullptr,
                                                     const float* __restrict__ arrived = nullptr) {
  __shared__ float red[16][64];
  const bool broken = promise_broken(promised, arrived);
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blk * 64 + c;
  const bool is_max = e == nslab - 1;                    // last slot: max of bit patterns, not a sum
  float s = 0.f;
  if (e < nslab)
    for (int b = q; b < nslabs; b += 16) {
      const float v = slabs[(int64_t)b * nslab + e];
      s = is_max ? (__float_as_uint(v) > __float_as_uint(s) ? v : s) : s + v;
    }
  red[q][c] = s;
  __syncthreads();
  if (q != 0 || e >= nslab) return;
  if (is_max) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s = __float_as_uint(red[k][c]) > __float_as_uint(s) ? red[k][c] : s;
    if (absmax) *absmax = broken ? __int_as_float(0x7fc00000) : s;
    return;
  }
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][c];
  if (broken) s = __int_as_float(0x7fc00000);
  const int o0 = 64 * in_dim, o1 = o0 + 64 * 64, o2 = o1 + out_dim * 64, o3 = o2 + 64, o4 = o3 + 64;
  if (e < o0) dW0[e] = s;
  else if (e < o1) dW1[e - o0] = s;
  else if (e < o2) dW2[e - o1] = s;
  else if (e < o3) db0[e - o2] = s;
  else if (e < o4) db1[e - o3] = s;
  else db2[e - o4] = s;
}

}  // namespace gngf
