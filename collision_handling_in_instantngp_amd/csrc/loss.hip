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
  static_assert(kLossThreads == 1024, "mse_sum_block is written for 1024-thread workgroups");
  mse_sum_block((int)blockIdx.x, (int)gridDim.x, pred, label, loss, acc, counter, n);     // body: gngf_common.h
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
extern "C" int gngf_mse_blocks(int64_t n) {            // workgroups gngf_mse_fwd uses for n elements (riders use the same)
  const int64_t want = (n / 4 + kLossThreads - 1) / kLossThreads;
  return (int)(want < 1 ? 1 : (want > kLossBlocks ? kLossBlocks : want));
}

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

// ---------------------------------------------------------------------------------------------- JS / KL term on p-bar
// Loss.forward's distribution term (utils.py:122-174) per level l on the batch-mean distribution p = pbar[l,:] (T slots),
// against the uniform q = 1/T, with torch.nn.KLDivLoss(reduction='batchmean') applied to 1-D rows (so "batch" = T):
//   kl  = (1/T) sum_t (q log q - q log p_t)
//   js  = ( (1/T) sum_t (m log m - m log p_t) + (1/T) sum_t (m log m - m log q) ) / 2,    m = (p_t + q) / 2
//   out = -(gamma + eps) js + eps kl
// Every per-slot expression is evaluated in fp32 as torch's elementwise ops do; the sums over 2^19 slots run in double
// (torch: fp32 tree sums), one partial triple per workgroup, finished by a second small launch — no fences, no atomics.
constexpr int kJsThreads = 256;
constexpr int kJsSlices = 64;       // workgroups per level

__device__ __forceinline__ float xlogx(float v) { return v == 0.f ? 0.f : v * logf(v); }      // torch.xlogy(v, v)

__global__ void __launch_bounds__(kJsThreads)
js_kl_partial_kernel(const float* __restrict__ pbar, double* __restrict__ partial, int64_t T) {
  __shared__ double red[3][kJsThreads / 64];
  const int l = blockIdx.y;
  const float* p = pbar + (int64_t)l * T;
  const float q = 1.0f / (float)T, lq = logf(q), qlq = xlogx(q);
  double s_kl = 0.0, s_1 = 0.0, s_2 = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * kJsThreads + threadIdx.x; t < T; t += (int64_t)gridDim.x * kJsThreads) {
    const float pv = p[t], lp = logf(pv);
    const float m = (pv + q) / 2.f, mlm = xlogx(m);
    s_kl += (double)(qlq - q * lp);
    s_1 += (double)(mlm - m * lp);
    s_2 += (double)(mlm - m * lq);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s_kl += __shfl_xor(s_kl, o, 64); s_1 += __shfl_xor(s_1, o, 64); s_2 += __shfl_xor(s_2, o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s_kl; red[1][wave] = s_1; red[2][wave] = s_2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int w = 0; w < kJsThreads / 64; ++w) s += red[threadIdx.x][w];
    partial[((int64_t)l * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = s;
  }
}

__global__ void js_kl_finish_kernel(const double* __restrict__ partial, float* __restrict__ out, int L, int slices, int64_t T,
                                    float gamma, float eps) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= L) return;
  double s[3] = {0.0, 0.0, 0.0};
  for (int b = 0; b < slices; ++b)
    for (int k = 0; k < 3; ++k) s[k] += partial[((int64_t)l * slices + b) * 3 + k];
  const float kl = (float)(s[0] / (double)T);
  const float js = ((float)(s[1] / (double)T) + (float)(s[2] / (double)T)) / 2.f;
  out[l] = -(gamma + eps) * js + eps * kl;
}

// d out_l / d p_t, times the incoming gradient of out_l
__global__ void __launch_bounds__(kJsThreads)
js_kl_bwd_kernel(const float* __restrict__ pbar, const float* __restrict__ gout, float* __restrict__ dpbar, int64_t T,
                 float gamma, float eps) {
  const int l = blockIdx.y;
  const float* p = pbar + (int64_t)l * T;
  float* d = dpbar + (int64_t)l * T;
  const float q = 1.0f / (float)T, lq = logf(q), invT = 1.0f / (float)T;
  const float g = gout[l];
  const float cj = -(gamma + eps) * 0.5f * invT, ck = eps * invT;
  for (int64_t t = (int64_t)blockIdx.x * kJsThreads + threadIdx.x; t < T; t += (int64_t)gridDim.x * kJsThreads) {
    const float pv = p[t], lp = logf(pv);
    const float m = (pv + q) / 2.f, lm1 = logf(m) + 1.0f;
    // js1' = (log m + 1)/2 - (log p)/2 - m/p,   js2' = (log m + 1)/2 - (log q)/2,   kl' = -q/p       (each times 1/T)
    const float djs = (0.5f * lm1 - 0.5f * lp - m / pv) + (0.5f * lm1 - 0.5f * lq);
    d[t] = g * (cj * djs + ck * (-q / pv));
  }
}

extern "C" int gngf_js_kl_workspace_doubles(int L) { return L * kJsSlices * 3; }

// out (L) = -(gamma + eps) JS(p_l, uniform) + eps KL(uniform || p_l) for the L rows of pbar (L,T); workspace:
// gngf_js_kl_workspace_doubles(L) doubles, 8-byte aligned.
extern "C" int gngf_js_kl_fwd(const float* pbar, float* out, double* workspace, int L, int64_t T, float gamma, float eps,
                              void* stream) {
  GNGF_CHECK_ARG(L > 0 && T > 0 && pbar && out && workspace && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0);
  hipStream_t s = as_stream(stream);
  js_kl_partial_kernel<<<dim3(kJsSlices, (unsigned)L), dim3(kJsThreads), 0, s>>>(pbar, workspace, T);
  js_kl_finish_kernel<<<dim3((unsigned)((L + 63) / 64)), dim3(64), 0, s>>>(workspace, out, L, kJsSlices, T, gamma, eps);
  GNGF_RETURN_LAUNCH();
}

// dpbar (L,T) = gout[l] * d out_l / d pbar[l,t]
extern "C" int gngf_js_kl_bwd(const float* pbar, const float* gout, float* dpbar, int L, int64_t T, float gamma, float eps,
                              void* stream) {
  GNGF_CHECK_ARG(L > 0 && T > 0 && pbar && gout && dpbar);
  const int64_t want = (T + kJsThreads - 1) / kJsThreads;
  const unsigned gx = (unsigned)(want > 256 ? 256 : want);
  js_kl_bwd_kernel<<<dim3(gx, (unsigned)L), dim3(kJsThreads), 0, as_stream(stream)>>>(pbar, gout, dpbar, T, gamma, eps);
  GNGF_RETURN_LAUNCH();
}
