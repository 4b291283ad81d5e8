// Dense Adam for the whole model in one launch: the reference's optimizer is torch.optim.Adam over
// three parameter groups with betas (0.9, 0.99), eps 1e-15, L2-style weight decay and dense moments (functions.py:96-127).
// All tensors of all groups are described by one segment list on the device; a block owns 2048 consecutive elements of
// one segment (found by bisection on the segments' first block) and streams param / grad / exp_avg / exp_avg_sq once:
// 28 bytes per element, HBM-bound.  The step count lives on the device and is advanced by a one-thread launch in front,
// so the pair is capturable in a hipGraph and needs no host synchronisation; the bias corrections are evaluated in
// double precision once per block (torch evaluates them in double on the host).
#include "gngf_common.h"

namespace gngf {

struct AdamSegment {        // mirrored by the host packer (train.py); 64 bytes
  void* param;              // fp32, or fp16 storage when flags & 1 (BASELINE config 5: fp16 level tables)
  const void* grad;         // same type as param
  float* exp_avg;
  float* exp_avg_sq;
  float* master;            // fp16 storage only: the fp32 master copy the update is applied to (param = round(master))
  int64_t n;
  int64_t first_block;
  int32_t group;
  int32_t flags;            // bit 0: param (and grad) are __half; bit 1 (with bit 0): grad is fp32 all the same
};
static_assert(sizeof(AdamSegment) == 64, "host packer layout");

struct AdamHyper {
  float lr[GNGF_ADAM_MAX_GROUPS];
  float weight_decay[GNGF_ADAM_MAX_GROUPS];
  float beta1, beta2, eps;
  float inv_grad_scale;     // gradients are multiplied by this first (1 / loss scale of fp16 training; 1 = off)
};

constexpr int kAdamBlock = 2048;     // elements per block: 256 threads x 2 x float4

__global__ void adam_tick_kernel(float* __restrict__ step) { *step += 1.0f; }

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float wd, float beta1, float beta2, float eps,
                                         float step_size, float bc2_sqrt) {
  if (wd != 0.f) g += wd * p;
  m = m + (1.0f - beta1) * (g - m);                       // lerp, as torch's fused kernel
  v = beta2 * v + (1.0f - beta2) * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p -= step_size * m / denom;
}

__global__ void __launch_bounds__(256)
adam_multi_kernel(const AdamSegment* __restrict__ segs, int nseg, const float* __restrict__ step, AdamHyper h) {
  __shared__ float s_corr[2];
  __shared__ int s_seg;
  const int64_t blk = blockIdx.x;
  if (threadIdx.x == 0) {
    int lo = 0, hi = nseg - 1;                            // last segment with first_block <= blk
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (segs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    s_seg = lo;
    const double t = (double)*step;
    s_corr[0] = (float)((double)h.lr[segs[lo].group] / (1.0 - pow((double)h.beta1, t)));      // step size
    s_corr[1] = (float)sqrt(1.0 - pow((double)h.beta2, t));
  }
  __syncthreads();
  const AdamSegment sg = segs[s_seg];
  const float step_size = s_corr[0], bc2_sqrt = s_corr[1];
  const float wd = h.weight_decay[sg.group];
  const float gs = h.inv_grad_scale;
  const int64_t e0 = (blk - sg.first_block) * kAdamBlock;
  if (sg.flags & 1) {
    // fp16 storage: fp16 gradient in, fp32 master weight + fp32 moments updated, fp16 parameter = round(master)
    __half* ph = static_cast<__half*>(sg.param);
    const __half* gh = static_cast<const __half*>(sg.grad);
    const float* g32 = static_cast<const float*>(sg.grad);
    const bool gf = (sg.flags & 2) != 0;       // the fp32 buffer the table gradient was accumulated in, not an fp16 copy of it
    const bool vec = ((reinterpret_cast<uintptr_t>(ph) | reinterpret_cast<uintptr_t>(gh)) & (gf ? 15 : 7)) == 0 &&
                     ((reinterpret_cast<uintptr_t>(sg.master) | reinterpret_cast<uintptr_t>(sg.exp_avg) |
                       reinterpret_cast<uintptr_t>(sg.exp_avg_sq)) & 15) == 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int64_t e = e0 + (int64_t)(half * 256 + threadIdx.x) * 4;
      if (e >= sg.n) continue;
      if (vec && e + 4 <= sg.n) {
        float4 p = *reinterpret_cast<float4*>(sg.master + e);
        float4 m = *reinterpret_cast<float4*>(sg.exp_avg + e);
        float4 v = *reinterpret_cast<float4*>(sg.exp_avg_sq + e);
        float4 g;
        if (gf) g = *reinterpret_cast<const float4*>(g32 + e);
        else {
          const __half2 g01 = *reinterpret_cast<const __half2*>(gh + e), g23 = *reinterpret_cast<const __half2*>(gh + e + 2);
          g = float4{__low2float(g01), __high2float(g01), __low2float(g23), __high2float(g23)};
        }
        adam_one(p.x, g.x * gs, m.x, v.x, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
        adam_one(p.y, g.y * gs, m.y, v.y, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
        adam_one(p.z, g.z * gs, m.z, v.z, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
        adam_one(p.w, g.w * gs, m.w, v.w, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
        *reinterpret_cast<float4*>(sg.master + e) = p;
        *reinterpret_cast<float4*>(sg.exp_avg + e) = m;
        *reinterpret_cast<float4*>(sg.exp_avg_sq + e) = v;
        *reinterpret_cast<__half2*>(ph + e) = __floats2half2_rn(p.x, p.y);
        *reinterpret_cast<__half2*>(ph + e + 2) = __floats2half2_rn(p.z, p.w);
      } else {
        for (int64_t q = e; q < e + 4 && q < sg.n; ++q) {
          float p = sg.master[q], m = sg.exp_avg[q], v = sg.exp_avg_sq[q];
          adam_one(p, (gf ? g32[q] : __half2float(gh[q])) * gs, m, v, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
          sg.master[q] = p; sg.exp_avg[q] = m; sg.exp_avg_sq[q] = v; ph[q] = __float2half_rn(p);
        }
      }
    }
    return;
  }
  float* pf = static_cast<float*>(sg.param);
  const float* gf = static_cast<const float*>(sg.grad);
  const bool vec = ((reinterpret_cast<uintptr_t>(pf) | reinterpret_cast<uintptr_t>(gf) |
                     reinterpret_cast<uintptr_t>(sg.exp_avg) | reinterpret_cast<uintptr_t>(sg.exp_avg_sq)) & 15) == 0;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int64_t e = e0 + (int64_t)(half * 256 + threadIdx.x) * 4;
    if (e >= sg.n) continue;
    if (vec && e + 4 <= sg.n) {
      float4 p = *reinterpret_cast<float4*>(pf + e);
      const float4 g = *reinterpret_cast<const float4*>(gf + e);
      float4 m = *reinterpret_cast<float4*>(sg.exp_avg + e);
      float4 v = *reinterpret_cast<float4*>(sg.exp_avg_sq + e);
      adam_one(p.x, g.x * gs, m.x, v.x, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
      adam_one(p.y, g.y * gs, m.y, v.y, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
      adam_one(p.z, g.z * gs, m.z, v.z, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
      adam_one(p.w, g.w * gs, m.w, v.w, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
      *reinterpret_cast<float4*>(pf + e) = p;
      *reinterpret_cast<float4*>(sg.exp_avg + e) = m;
      *reinterpret_cast<float4*>(sg.exp_avg_sq + e) = v;
    } else {
      for (int64_t q = e; q < e + 4 && q < sg.n; ++q) {
        float p = pf[q], m = sg.exp_avg[q], v = sg.exp_avg_sq[q];
        adam_one(p, gf[q] * gs, m, v, wd, h.beta1, h.beta2, h.eps, step_size, bc2_sqrt);
        pf[q] = p; sg.exp_avg[q] = m; sg.exp_avg_sq[q] = v;
      }
    }
  }
}

}  // namespace gngf

using namespace gngf;

extern "C" int gngf_adam_block_elems(void) { return kAdamBlock; }

// One Adam step over `nseg` tensors.  segments: device array of 64-byte records {param, grad, exp_avg, exp_avg_sq, master,
// n, first_block, group, flags} with first_block = running sum of ceil(n / gngf_adam_block_elems()); total_blocks = that
// sum.  flags bit 0: param and grad are fp16 and `master` is the fp32 master copy (else master is ignored); bit 1 (with
// bit 0): the gradient is fp32 all the same (the buffer an fp16 table's gradient was accumulated in).
// step: device float, incremented by this call before it is used (t = 1 on the first step).  lr / weight_decay: host
// arrays of ngroups <= GNGF_ADAM_MAX_GROUPS values.  inv_grad_scale multiplies every gradient first (1 / loss scale).
extern "C" int gngf_adam_step(const void* segments, int nseg, int64_t total_blocks, float* step, const float* lr,
                              const float* weight_decay, int ngroups, float beta1, float beta2, float eps,
                              float inv_grad_scale, void* stream) {
  GNGF_CHECK_ARG(nseg >= 0 && total_blocks >= 0 && ngroups > 0 && ngroups <= GNGF_ADAM_MAX_GROUPS && total_blocks < INT32_MAX);
  GNGF_CHECK_ARG(step && lr && weight_decay);
  hipStream_t s = as_stream(stream);
  adam_tick_kernel<<<dim3(1), dim3(1), 0, s>>>(step);
  if (nseg == 0 || total_blocks == 0) GNGF_RETURN_LAUNCH();
  GNGF_CHECK_ARG(segments);
  AdamHyper h;
  for (int g = 0; g < GNGF_ADAM_MAX_GROUPS; ++g) {
    h.lr[g] = g < ngroups ? lr[g] : 0.f;
    h.weight_decay[g] = g < ngroups ? weight_decay[g] : 0.f;
  }
  h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.inv_grad_scale = inv_grad_scale;
  adam_multi_kernel<<<dim3((unsigned)total_blocks), dim3(256), 0, s>>>(static_cast<const AdamSegment*>(segments), nseg, step, h);
  GNGF_RETURN_LAUNCH();
}
