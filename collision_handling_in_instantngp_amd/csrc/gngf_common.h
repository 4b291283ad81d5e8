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

}  // namespace gngf
