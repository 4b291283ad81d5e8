// Pixel loss of the training step (reference utils.py:99, `torch.nn.MSELoss()` on (P,3) outputs) as two launches instead
// of the three framework kernels (square-difference, mean, backward) whose launch latency is ~3 % of a cfg2 step.
#include "gngf_common.h"

namespace gngf {

constexpr int kLossThreads = 256;
constexpr int kLossBlocks = 256;

// loss = sum((pred - label)^2) / n.  Deterministic: per-block partials, the last block to finish adds them in index order.
__global__ void __launch_bounds__(kLossThreads)
mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ label, float* __restrict__ loss,
               float* __restrict__ partials, unsigned* __restrict__ counter, int64_t n) {
  __shared__ float red[kLossThreads / 64];
  __shared__ bool last;
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(pred);
  const float4* l4 = reinterpret_cast<const float4*>(label);
  for (int64_t e = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; e < n4; e += (int64_t)gridDim.x * kLossThreads) {
    const float4 a = p4[e], b = l4[e];
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    s += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  if (blockIdx.x == 0)
    for (int64_t e = (n4 << 2) + threadIdx.x; e < n; e += kLossThreads) { const float d = pred[e] - label[e]; s += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();
    last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = threadIdx.x < gridDim.x ? __builtin_nontemporal_load(partials + threadIdx.x) : 0.f;   // gridDim.x <= kLossThreads
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    *loss = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
    *counter = 0u;                                       // ready for the next launch (hipGraph replay included)
  }
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

extern "C" int gngf_mse_workspace_floats(void) { return kLossBlocks + 1; }

// loss (1) = mean((pred - label)^2) over n elements.  workspace: gngf_mse_workspace_floats() floats whose LAST word is a
// counter that must be zero before the first call (the kernel leaves it zero).
extern "C" int gngf_mse_fwd(const float* pred, const float* label, float* loss, float* workspace, int64_t n, void* stream) {
  GNGF_CHECK_ARG(n > 0 && pred && label && loss && workspace);
  GNGF_CHECK_ARG((reinterpret_cast<uintptr_t>(pred) & 15) == 0 && (reinterpret_cast<uintptr_t>(label) & 15) == 0);
  const int64_t want = (n / 4 + kLossThreads - 1) / kLossThreads;
  const unsigned grid = (unsigned)(want < 1 ? 1 : (want > kLossBlocks ? kLossBlocks : want));
  mse_fwd_kernel<<<dim3(grid), dim3(kLossThreads), 0, as_stream(stream)>>>(pred, label, loss, workspace,
                                                                         reinterpret_cast<unsigned*>(workspace + kLossBlocks), n);
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
