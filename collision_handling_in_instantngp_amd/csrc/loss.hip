// Pixel loss of the training step (reference utils.py:99, `torch.nn.MSELoss()` on (P,3) outputs): small dedicated
// kernels instead of the three framework kernels (square-difference, mean, backward) that cost ~3 % of a cfg2 step.
#include "gngf_common.h"

namespace gngf {

constexpr int kLossThreads = 1024;                      // 16 waves per CU: the value kernel is a latency-bound stream
constexpr int kLossBlocks = 64;                       // few workgroups: their two same-address atomics each are the serial part

// loss = sum((pred - label)^2) / n in ONE launch (a launch costs ~6 us inside a replayed step; a last-block-reduces
// scheme built on an agent-scope release FENCE is worse: on this chip the fence writes the whole L2 back, 15 us
// measured).  Cross-workgroup traffic goes through device-scope atomics only: every workgroup adds its partial to a
// double (order-dependent only in the 53-bit sum: the rounded float result is reproducible in practice) and takes a
// ticket; the atomic's RETURN value feeds the ticket request, so the add is performed before the ticket exists.  The
// last ticket holder reads the total with an atomic, writes the loss and resets both words for the next launch.
__global__ void __launch_bounds__(kLossThreads)
mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ label, float* __restrict__ loss,
               double* __restrict__ acc, unsigned* __restrict__ counter, int64_t n) {
  __shared__ float red[kLossThreads / 64];
  float s = 0.f, s2 = 0.f;
  const int64_t n4 = n >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(pred);
  const float4* l4 = reinterpret_cast<const float4*>(label);
  const int64_t stride = (int64_t)gridDim.x * kLossThreads;
  int64_t e = (int64_t)blockIdx.x * kLossThreads + threadIdx.x;
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
  if (blockIdx.x == 0)
    for (int64_t t = (n4 << 2) + threadIdx.x; t < n; t += kLossThreads) { const float d = pred[t] - label[t]; s += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  double b = 0.0;
#pragma unroll
  for (int w = 0; w < kLossThreads / 64; ++w) b += (double)red[w];
  const double before = atomicAdd(acc, b);
  const unsigned ticket = atomicAdd(counter, before < 0.0 ? 2u : 1u);      // sums of squares are never negative: always 1
  if (ticket != gridDim.x - 1) return;
  const double total = atomicAdd(acc, 0.0);
  *loss = (float)(total / (double)n);
  atomicExch(reinterpret_cast<unsigned long long*>(acc), 0ull);
  atomicExch(counter, 0u);
}

// dpred = gout * 2 (pred - label) / n
__global__ void __launch_bounds__(kLossThreads)
mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ label, const float* __restrict__ gout,
               float* __restrict__ dpred, int64_t n) {
  const float k = *gout * (2.0f / (float)n);
  const int64_t n4 = n >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(pred);
  const float4* l4 = reinterpret_cast<const float4*>(label);
  float4* d4 = reinterpret_cast<float4*>(dpred);
  for (int64_t e = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; e < n4; e += (int64_t)gridDim.x * kLossThreads) {
    const float4 a = p4[e], b = l4[e];
    d4[e] = make_float4(k * (a.x - b.x), k * (a.y - b.y), k * (a.z - b.z), k * (a.w - b.w));
  }
  if (blockIdx.x == 0)
    for (int64_t e = (n4 << 2) + threadIdx.x; e < n; e += kLossThreads) dpred[e] = k * (pred[e] - label[e]);
}

}  // namespace gngf

using namespace gngf;

extern "C" int gngf_mse_workspace_floats(void) { return 4; }

// loss (1) = mean((pred - label)^2) over n elements.  workspace: gngf_mse_workspace_floats() floats, 8-byte aligned and
// zero-filled once before the first call (an accumulator and a ticket counter the kernel itself resets).
extern "C" int gngf_mse_fwd(const float* pred, const float* label, float* loss, float* workspace, int64_t n, void* stream) {
  GNGF_CHECK_ARG(n > 0 && pred && label && loss && workspace);
  GNGF_CHECK_ARG((reinterpret_cast<uintptr_t>(pred) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(workspace) & 7) == 0);
  const int64_t want = (n / 4 + kLossThreads - 1) / kLossThreads;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > kLossBlocks ? kLossBlocks : want));
  mse_fwd_kernel<<<dim3(grid), dim3(kLossThreads), 0, as_stream(stream)>>>(pred, label, loss, reinterpret_cast<double*>(workspace),
                                                                         reinterpret_cast<unsigned*>(workspace + 2), n);
  GNGF_RETURN_LAUNCH();
}

// dpred (n) = gout[0] * 2 (pred - label) / n
extern "C" int gngf_mse_bwd(const float* pred, const float* label, const float* gout, float* dpred, int64_t n, void* stream) {
  GNGF_CHECK_ARG(n > 0 && pred && label && gout && dpred);
  GNGF_CHECK_ARG((reinterpret_cast<uintptr_t>(pred) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(dpred) & 15) == 0);
  const int64_t want = (n / 4 + kLossThreads - 1) / kLossThreads;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  mse_bwd_kernel<<<dim3(grid), dim3(kLossThreads), 0, as_stream(stream)>>>(pred, label, gout, dpred, n);
  GNGF_RETURN_LAUNCH();
}
