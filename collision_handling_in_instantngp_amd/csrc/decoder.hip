// Fused decoder MLP on the matrix cores (reference models.py:382-392, 469-470):
//     rgb = Sigmoid(W2 · act(W1 · act(W0 · enc + b0) + b1) + b2),   act = ReLU | LeakyReLU(0.01), hidden widths 64/64
// forward in ONE kernel and backward (d enc, dW*, db*) in ONE kernel, exact fp32 on v_mfma_f32_32x32x2_f32.
//
// gfx950 mapping ("accumulator tile as the next MFMA's operand", cdna_hip_programming.md §3):
//   every product is computed TRANSPOSED, H^T[feature][pixel] = W[feature][k] · X^T[k][pixel], so the pixel sits on the
//   MFMA lane (col = lane & 31) and the features in the 16 accumulator registers (row = (r&3) + 8(r>>2) + 4(lane>>5)).
//   The f32 MFMA takes ONE VGPR per operand, so accumulator register r of layer n IS the B operand of k-step r of
//   layer n+1 — no LDS round trip, no lane movement between layers; only the weight (A) fragments come from LDS, stored
//   in exactly the k-order the accumulator layout dictates.  One wave owns 32 pixels; a 256-thread workgroup 128.
//   Weight gradients contract over the pixel index, which needs pixel on the k axis: the wave transposes its 32-pixel
//   tiles through a private, +1-padded LDS image ([feature][33]: conflict-free b32 reads and writes), accumulates
//   dW tiles in registers across its whole pixel range, and the workgroup emits ONE partial slab; a tiny second
//   kernel sums the slabs (store pass + sum pass instead of ~10^6 contended float atomics).
//   Hidden activations are recomputed in backward (96 of 324 MFMAs per 32 pixels) instead of being stored
//   (512 B/pixel of HBM traffic each way).
#include "gngf_common.h"

namespace gngf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kH = 64;                 // hidden width (both hidden layers)
constexpr int kDecThreads = 256;

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// k index (feature of a 64-wide activation held as two accumulator tiles) consumed at chained k-step s2 by lane half h
__device__ __forceinline__ int kmapC(int s2, int h) { return 32 * (s2 >> 4) + crow(s2 & 15, h); }

__device__ __forceinline__ float hidden_act(float z, bool leaky) { return z > 0.f ? z : (leaky ? 0.01f * z : 0.f); }
__device__ __forceinline__ float hidden_dact(float y, bool leaky) { return y > 0.f ? 1.f : (leaky ? 0.01f : 0.f); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// LDS fragment images.  frag[(tile * S + s) * 64 + lane].
template <int KIN>
struct FwdFrags {
  static constexpr int S0 = KIN / 2;
  static constexpr int kA0 = 2 * S0 * 64, kA1 = 2 * 32 * 64, kA2 = 32 * 64;
};

template <int KIN>
__device__ __forceinline__ void fill_fwd_frags(float* A0, float* A1, float* A2, const float* __restrict__ W0,
                                               const float* __restrict__ W1, const float* __restrict__ W2, int in_dim,
                                               int out_dim, bool need_a2) {
  constexpr int S0 = KIN / 2;
  for (int e = threadIdx.x; e < 2 * S0 * 64; e += kDecThreads) {
    const int lane = e & 63, s = (e >> 6) % S0, t = (e >> 6) / S0;
    const int i = lane & 31, h = lane >> 5, k = h * S0 + s;
    A0[e] = k < in_dim ? W0[(32 * t + i) * in_dim + k] : 0.f;
  }
  for (int e = threadIdx.x; e < 2 * 32 * 64; e += kDecThreads) {
    const int lane = e & 63, s2 = (e >> 6) & 31, t = e >> 11;
    const int i = lane & 31, h = lane >> 5;
    A1[e] = W1[(32 * t + i) * kH + kmapC(s2, h)];
  }
  if (need_a2)
    for (int e = threadIdx.x; e < 32 * 64; e += kDecThreads) {
      const int lane = e & 63, s2 = e >> 6;
      const int i = lane & 31, h = lane >> 5;
      A2[e] = i < out_dim ? W2[i * kH + kmapC(s2, h)] : 0.f;
    }
}

// Loads the lane's slice of its pixel's input row: xr[s] = X[pix][h*KIN/2 + s].
template <int KIN>
__device__ __forceinline__ void load_x(const float* __restrict__ X, int64_t pix, bool valid, int in_dim, int h, float* xr) {
  constexpr int S0 = KIN / 2;
  if (valid && in_dim == KIN) {
    const float4* src = reinterpret_cast<const float4*>(X + pix * KIN + h * S0);
#pragma unroll
    for (int q = 0; q < S0 / 4; ++q) {
      const float4 v = src[q];
      xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < S0; ++s) {
      const int k = h * S0 + s;
      xr[s] = (valid && k < in_dim) ? X[pix * in_dim + k] : 0.f;
    }
  }
}

// Layers 1 and 2 for one 32-pixel tile of this wave: acc1 = h1^T, acc2 = h2^T (both activated).
template <int KIN>
__device__ __forceinline__ void hidden_layers(const float* A0, const float* A1, const float* b0s, const float* b1s,
                                              const float* xr, int lane, int h, bool leaky, f32x16 (&acc1)[2], f32x16 (&acc2)[2]) {
  constexpr int S0 = KIN / 2;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[t][r] = b0s[32 * t + crow(r, h)]; acc2[t][r] = b1s[32 * t + crow(r, h)]; }
#pragma unroll
  for (int s = 0; s < S0; ++s) {
    acc1[0] = MFMA(A0[(0 * S0 + s) * 64 + lane], xr[s], acc1[0]);
    acc1[1] = MFMA(A0[(1 * S0 + s) * 64 + lane], xr[s], acc1[1]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[t][r] = hidden_act(acc1[t][r], leaky);
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) {
    const float b = acc1[s2 >> 4][s2 & 15];
    acc2[0] = MFMA(A1[(0 * 32 + s2) * 64 + lane], b, acc2[0]);
    acc2[1] = MFMA(A1[(1 * 32 + s2) * 64 + lane], b, acc2[1]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = hidden_act(acc2[t][r], leaky);
}

// ------------------------------------------------------------------------------------------------ forward
template <int KIN>
__global__ void __launch_bounds__(kDecThreads, 1)
decoder_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W0, const float* __restrict__ b0,
                   const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                   const float* __restrict__ b2, float* __restrict__ Y, int64_t P, int in_dim, int out_dim, int leaky_i) {
  using FF = FwdFrags<KIN>;
  __shared__ float A0[FF::kA0];
  __shared__ float A1[FF::kA1];
  __shared__ float bs[2 * kH + 32];
  const bool leaky = leaky_i != 0;
  fill_fwd_frags<KIN>(A0, A1, nullptr, W0, W1, W2, in_dim, out_dim, false);
  for (int e = threadIdx.x; e < 2 * kH + 32; e += kDecThreads)
    bs[e] = e < kH ? b0[e] : (e < 2 * kH ? b1[e - kH] : ((e - 2 * kH) < out_dim ? b2[e - 2 * kH] : 0.f));
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int64_t ntiles = (P + 127) / 128;
  // output layer on the VALU: 3-4 outputs do not fill a 32-row MFMA tile.  Each lane keeps the W2 entries of the 32
  // features it owns (registers, loaded once); the two halves of a pixel are combined with one cross-half shuffle.
  float w2r[4][32];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 32; ++q) w2r[c][q] = c < out_dim ? W2[c * kH + 32 * (q >> 4) + crow(q & 15, h)] : 0.f;
  float xn[KIN / 2];
  {
    const int64_t pix0 = (int64_t)blockIdx.x * 128 + wave * 32 + i;
    load_x<KIN>(X, pix0, pix0 < P && (int64_t)blockIdx.x < ntiles, in_dim, h, xn);
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t pix = tile * 128 + wave * 32 + i;
    const bool valid = pix < P;
    float xr[KIN / 2];
#pragma unroll
    for (int s = 0; s < KIN / 2; ++s) xr[s] = xn[s];
    {                                                     // prefetch the next tile's rows under this tile's MFMAs
      const int64_t npix = (tile + gridDim.x) * 128 + wave * 32 + i;
      load_x<KIN>(X, npix, npix < P && tile + gridDim.x < ntiles, in_dim, h, xn);
    }
    f32x16 acc1[2], acc2[2];
    hidden_layers<KIN>(A0, A1, bs, bs + kH, xr, lane, h, leaky, acc1, acc2);
    float y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) a = fmaf(w2r[c][q], acc2[q >> 4][q & 15], a);
      y[c] = a + __shfl_xor(a, 32, 64) + bs[2 * kH + c];
    }
    if (valid && h == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < out_dim) Y[pix * out_dim + c] = 1.0f / (1.0f + expf(-y[c]));
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// slab layout (floats): dW0 [64*in_dim] | dW1 [64*64] | dW2 [out_dim*64] | db0 [64] | db1 [64] | db2 [out_dim]
__host__ __device__ inline int slab_size(int in_dim, int out_dim) { return kH * in_dim + kH * kH + out_dim * kH + 2 * kH + out_dim; }

constexpr int kImgStride = 33;
constexpr int kImgFloats = 64 * kImgStride;

template <int KIN>
__global__ void __launch_bounds__(kDecThreads, 1)
decoder_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Yout, const float* __restrict__ dY,
                   const float* __restrict__ W0, const float* __restrict__ b0, const float* __restrict__ W1,
                   const float* __restrict__ b1, const float* __restrict__ W2, float* __restrict__ dX,
                   float* __restrict__ slabs, int64_t P, int in_dim, int out_dim, int leaky_i) {
  using FF = FwdFrags<KIN>;
  constexpr int S0 = KIN / 2;
  constexpr int TX = (KIN + 31) / 32;                    // 32-row tiles of the input width
  extern __shared__ float smem[];
  float* A0 = smem;                                      // forward fragments (recompute)
  float* A1 = A0 + FF::kA0;
  float* A2T = A1 + FF::kA1;                             // [t(2)][s(2)][64]   : W2[c = 2s+h][32t+i]
  float* A1T = A2T + 2 * 2 * 64;                         // [t(2)][s2(32)][64] : W1[kmapC(s2,h)][32t+i]
  float* A0T = A1T + 2 * 32 * 64;                        // [t(TX)][s2(32)][64]: W0[kmapC(s2,h)][32t+i]
  float* bs = A0T + TX * 32 * 64;                        // b0 | b1
  float* img = bs + 2 * kH;                              // per wave: imgA [64][33], imgB [64][33]
  float* acc_lds = img;                                  // workgroup slab accumulator: reuses the images after the loop
  const bool leaky = leaky_i != 0;
  const int nslab = slab_size(in_dim, out_dim);

  fill_fwd_frags<KIN>(A0, A1, nullptr, W0, W1, W2, in_dim, out_dim, false);
  for (int e = threadIdx.x; e < 2 * 2 * 64; e += kDecThreads) {
    const int lane = e & 63, s = (e >> 6) & 1, t = e >> 7;
    const int c = 2 * s + (lane >> 5);
    A2T[e] = c < out_dim ? W2[c * kH + 32 * t + (lane & 31)] : 0.f;
  }
  for (int e = threadIdx.x; e < 2 * 32 * 64; e += kDecThreads) {
    const int lane = e & 63, s2 = (e >> 6) & 31, t = e >> 11;
    A1T[e] = W1[kmapC(s2, lane >> 5) * kH + 32 * t + (lane & 31)];
  }
  for (int e = threadIdx.x; e < TX * 32 * 64; e += kDecThreads) {
    const int lane = e & 63, s2 = (e >> 6) & 31, t = e >> 11;
    const int j = 32 * t + (lane & 31);
    A0T[e] = j < in_dim ? W0[kmapC(s2, lane >> 5) * in_dim + j] : 0.f;
  }
  for (int e = threadIdx.x; e < 2 * kH; e += kDecThreads) bs[e] = e < kH ? b0[e] : b1[e - kH];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  float* imgA = img + wave * 2 * kImgFloats;
  float* imgB = imgA + kImgFloats;

  f32x16 dW1acc[2][2], dW0acc[2][TX], dW2acc[2];
  f32x16 db0acc[2], db1acc[2];
  float db2acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    dW2acc[a] = 0; db0acc[a] = 0; db1acc[a] = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) dW1acc[a][b] = 0;
#pragma unroll
    for (int b = 0; b < TX; ++b) dW0acc[a][b] = 0;
  }

  const int64_t ntiles = (P + 127) / 128;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t pix = tile * 128 + wave * 32 + i;
    const bool valid = pix < P;
    float xr[S0];
    load_x<KIN>(X, pix, valid, in_dim, h, xr);
    // dz3 = dY * y (1 - y)   (Sigmoid backward), zero for padding pixels: they then contribute nothing anywhere
    float dz3[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = 0.f;
      if (valid && c < out_dim) { const float y = Yout[pix * out_dim + c]; v = dY[pix * out_dim + c] * (y * (1.f - y)); }
      dz3[c] = v;
      if (h == 0) db2acc[c] += v;
    }
    f32x16 acc1[2], acc2[2];
    hidden_layers<KIN>(A0, A1, bs, bs + kH, xr, lane, h, leaky, acc1, acc2);

    // ---- dW2 += dz3^T h2 : images  dz3T -> imgA rows 0..3,  h2T -> imgB
    if (h == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) imgA[c * kImgStride + i] = dz3[c];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) imgB[(32 * t + crow(r, h)) * kImgStride + i] = acc2[t][r];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = i < 4 ? imgA[i * kImgStride + 2 * s + h] : 0.f;
      dW2acc[0] = MFMA(a, imgB[(i) * kImgStride + 2 * s + h], dW2acc[0]);
      dW2acc[1] = MFMA(a, imgB[(32 + i) * kImgStride + 2 * s + h], dW2acc[1]);
    }
    // ---- dh2^T = W2^T dz3^T  (k = output channel, padded to 4)
    f32x16 d2[2];
    d2[0] = 0; d2[1] = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float b = h == 0 ? dz3[2 * s] : dz3[2 * s + 1];
      d2[0] = MFMA(A2T[(0 * 2 + s) * 64 + lane], b, d2[0]);
      d2[1] = MFMA(A2T[(1 * 2 + s) * 64 + lane], b, d2[1]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { d2[t][r] *= hidden_dact(acc2[t][r], leaky); db1acc[t][r] += d2[t][r]; }
    // ---- dW1 += dz2^T h1 : dz2T -> imgA, h1T -> imgB
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        imgA[(32 * t + crow(r, h)) * kImgStride + i] = d2[t][r];
        imgB[(32 * t + crow(r, h)) * kImgStride + i] = acc1[t][r];
      }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a0 = imgA[i * kImgStride + 2 * s + h], a1 = imgA[(32 + i) * kImgStride + 2 * s + h];
      const float b0v = imgB[i * kImgStride + 2 * s + h], b1v = imgB[(32 + i) * kImgStride + 2 * s + h];
      dW1acc[0][0] = MFMA(a0, b0v, dW1acc[0][0]);
      dW1acc[0][1] = MFMA(a0, b1v, dW1acc[0][1]);
      dW1acc[1][0] = MFMA(a1, b0v, dW1acc[1][0]);
      dW1acc[1][1] = MFMA(a1, b1v, dW1acc[1][1]);
    }
    // ---- dh1^T = W1^T dz2^T
    f32x16 d1[2];
    d1[0] = 0; d1[1] = 0;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
      const float b = d2[s2 >> 4][s2 & 15];
      d1[0] = MFMA(A1T[(0 * 32 + s2) * 64 + lane], b, d1[0]);
      d1[1] = MFMA(A1T[(1 * 32 + s2) * 64 + lane], b, d1[1]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { d1[t][r] *= hidden_dact(acc1[t][r], leaky); db0acc[t][r] += d1[t][r]; }
    // ---- dW0 += dz1^T x : dz1T -> imgA, xT -> imgB (rows = input features)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) imgA[(32 * t + crow(r, h)) * kImgStride + i] = d1[t][r];
#pragma unroll
    for (int s = 0; s < S0; ++s) imgB[(h * S0 + s) * kImgStride + i] = xr[s];
    if (KIN < 32) {
      for (int rr = KIN + h; rr < 32; rr += 2) imgB[rr * kImgStride + i] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a0 = imgA[i * kImgStride + 2 * s + h], a1 = imgA[(32 + i) * kImgStride + 2 * s + h];
#pragma unroll
      for (int tx = 0; tx < TX; ++tx) {
        const float bv = imgB[(32 * tx + i) * kImgStride + 2 * s + h];
        dW0acc[0][tx] = MFMA(a0, bv, dW0acc[0][tx]);
        dW0acc[1][tx] = MFMA(a1, bv, dW0acc[1][tx]);
      }
    }
    // ---- d enc^T = W0^T dz1^T ; regs 4g..4g+3 are 4 consecutive input features
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
      f32x16 dx;
      dx = 0;
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) dx = MFMA(A0T[(tx * 32 + s2) * 64 + lane], d1[s2 >> 4][s2 & 15], dx);
      if (valid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * tx + 8 * g + 4 * h;
          if (in_dim == KIN) {
            if (32 * tx + 8 * g + 8 <= KIN)            // KIN = 16: only g = 0, 1 are real input features
              *reinterpret_cast<float4*>(dX + pix * KIN + col) = make_float4(dx[4 * g], dx[4 * g + 1], dx[4 * g + 2], dx[4 * g + 3]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (col + q < in_dim) dX[pix * in_dim + col + q] = dx[4 * g + q];
          }
        }
      }
    }
  }

  // ---- wave accumulators -> workgroup slab (LDS atomics), then one plain-store slab per workgroup
  __syncthreads();                                       // every wave is done with its transposition images
  for (int e = threadIdx.x; e < nslab; e += kDecThreads) acc_lds[e] = 0.f;
  __syncthreads();
  float* sW0 = acc_lds;
  float* sW1 = sW0 + kH * in_dim;
  float* sW2 = sW1 + kH * kH;
  float* sb0 = sW2 + out_dim * kH;
  float* sb1 = sb0 + kH;
  float* sb2 = sb1 + kH;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * ti + crow(r, h);
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) atomicAdd(sW1 + row * kH + 32 * tj + i, dW1acc[ti][tj][r]);
#pragma unroll
      for (int tx = 0; tx < TX; ++tx)
        if (32 * tx + i < in_dim) atomicAdd(sW0 + row * in_dim + 32 * tx + i, dW0acc[ti][tx][r]);
      // bias partials: every lane of a half holds a different pixel's share of the same feature
      atomicAdd(sb0 + row, db0acc[ti][r]);
      atomicAdd(sb1 + row, db1acc[ti][r]);
    }
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = crow(r, h);
      if (row < out_dim) atomicAdd(sW2 + row * kH + 32 * tj + i, dW2acc[tj][r]);
    }
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < out_dim) atomicAdd(sb2 + c, db2acc[c]);
  }
  __syncthreads();
  float* out = slabs + (int64_t)blockIdx.x * nslab;
  for (int e = threadIdx.x; e < nslab; e += kDecThreads) out[e] = acc_lds[e];
}

// sums the per-workgroup slabs and writes the six gradient tensors.  64 elements x 16 slab-groups per block.
__global__ void __launch_bounds__(1024)
decoder_reduce_kernel(const float* __restrict__ slabs, int nslabs, int nslab, int in_dim, int out_dim,
                      float* __restrict__ dW0, float* __restrict__ db0, float* __restrict__ dW1, float* __restrict__ db1,
                      float* __restrict__ dW2, float* __restrict__ db2) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + c;
  float s = 0.f;
  if (e < nslab)
    for (int b = q; b < nslabs; b += 16) s += slabs[(int64_t)b * nslab + e];
  red[q][c] = s;
  __syncthreads();
  if (q != 0 || e >= nslab) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][c];
  const int o0 = kH * in_dim, o1 = o0 + kH * kH, o2 = o1 + out_dim * kH, o3 = o2 + kH, o4 = o3 + kH;
  if (e < o0) dW0[e] = s;
  else if (e < o1) dW1[e - o0] = s;
  else if (e < o2) dW2[e - o1] = s;
  else if (e < o3) db0[e - o2] = s;
  else if (e < o4) db1[e - o3] = s;
  else db2[e - o4] = s;
}

template <int KIN>
static size_t bwd_smem_bytes(int in_dim, int out_dim) {
  using FF = FwdFrags<KIN>;
  constexpr int TX = (KIN + 31) / 32;
  const int img = 4 * 2 * kImgFloats, slab = slab_size(in_dim, out_dim);
  return sizeof(float) * (size_t)(FF::kA0 + FF::kA1 + 2 * 2 * 64 + 2 * 32 * 64 + TX * 32 * 64 + 2 * kH + (img > slab ? img : slab));
}

}  // namespace gngf

using namespace gngf;

#define DISPATCH_KIN(in_dim, ...)                                         \
  if ((in_dim) <= 16) { constexpr int kKIN = 16; __VA_ARGS__; }           \
  else if ((in_dim) <= 32) { constexpr int kKIN = 32; __VA_ARGS__; }      \
  else { constexpr int kKIN = 64; __VA_ARGS__; }

// number of workgroups (= partial slabs) gngf_decoder_bwd launches for P pixels: size the slab workspace with it
extern "C" int gngf_decoder_bwd_slabs(int64_t P) {
  const int64_t tiles = (P + 127) / 128;
  return (int)(tiles < 256 ? (tiles > 0 ? tiles : 1) : 256);
}
extern "C" int gngf_decoder_slab_floats(int in_dim, int out_dim) { return slab_size(in_dim, out_dim); }

// rgb (P,out_dim) = decoder(enc (P,in_dim)); hidden widths fixed at 64/64, in_dim <= 64, out_dim <= 4.
extern "C" int gngf_decoder_fwd(const float* enc, const float* W0, const float* b0, const float* W1, const float* b1,
                                const float* W2, const float* b2, float* rgb, int64_t P, int in_dim, int out_dim, int leaky,
                                void* stream) {
  GNGF_CHECK_ARG(P >= 0 && in_dim > 0 && in_dim <= 64 && out_dim > 0 && out_dim <= 4);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(enc && W0 && b0 && W1 && b1 && W2 && b2 && rgb);
  const int64_t tiles = (P + 127) / 128;
  const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);
  DISPATCH_KIN(in_dim, (decoder_fwd_kernel<kKIN><<<dim3(grid), dim3(kDecThreads), 0, as_stream(stream)>>>(
                           enc, W0, b0, W1, b1, W2, b2, rgb, P, in_dim, out_dim, leaky)));
  GNGF_RETURN_LAUNCH();
}

// d enc (P,in_dim) and the six parameter gradients (each WRITTEN, not accumulated).  rgb = the forward output.
// slabs: workspace of gngf_decoder_bwd_slabs(P) * gngf_decoder_slab_floats(in_dim, out_dim) floats.
extern "C" int gngf_decoder_bwd(const float* enc, const float* rgb, const float* drgb, const float* W0, const float* b0,
                                const float* W1, const float* b1, const float* W2, float* denc, float* dW0, float* db0,
                                float* dW1, float* db1, float* dW2, float* db2, float* slabs, int64_t P, int in_dim,
                                int out_dim, int leaky, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && in_dim > 0 && in_dim <= 64 && out_dim > 0 && out_dim <= 4);
  GNGF_CHECK_ARG(dW0 && db0 && dW1 && db1 && dW2 && db2 && slabs);
  const int nslab = slab_size(in_dim, out_dim);
  hipStream_t s = as_stream(stream);
  const int nslabs = gngf_decoder_bwd_slabs(P);
  if (P == 0) {
    hipError_t e = hipMemsetAsync(slabs, 0, sizeof(float) * (size_t)nslab, s);
    if (e != hipSuccess) return (int)e;
  } else {
    GNGF_CHECK_ARG(enc && rgb && drgb && W0 && b0 && W1 && b1 && W2 && denc);
    DISPATCH_KIN(in_dim, {
      const size_t smem = bwd_smem_bytes<kKIN>(in_dim, out_dim);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decoder_bwd_kernel<kKIN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      decoder_bwd_kernel<kKIN><<<dim3((unsigned)nslabs), dim3(kDecThreads), smem, s>>>(enc, rgb, drgb, W0, b0, W1, b1, W2, denc,
                                                                                      slabs, P, in_dim, out_dim, leaky);
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  decoder_reduce_kernel<<<dim3((unsigned)((nslab + 63) / 64)), dim3(1024), 0, s>>>(slabs, nslabs, nslab, in_dim, out_dim, dW0,
                                                                                   db0, dW1, db1, dW2, db2);
  GNGF_RETURN_LAUNCH();
}
