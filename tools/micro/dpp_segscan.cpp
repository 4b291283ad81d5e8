#include <hip/hip_runtime.h>
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dppf(float old, float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dppi(int old, int x) {
  return __builtin_amdgcn_update_dpp(old, x, CTRL, ROWMASK, 0xF, false);
}
__global__ void k(const int* slots, const float* vals, float* out) {
  const int lane = threadIdx.x;
  const int slot = slots[lane];
  float x = vals[lane];
  const bool p1 = dppi<0x111, 0xF>(-1, slot) == slot, p2 = dppi<0x112, 0xF>(-1, slot) == slot, p4 = dppi<0x114, 0xF>(-1, slot) == slot,
             p8 = dppi<0x118, 0xF>(-1, slot) == slot, pA = dppi<0x142, 0xA>(-1, slot) == slot, pB = dppi<0x143, 0xC>(-1, slot) == slot;
  float y;
  y = dppf<0x111, 0xF>(0.f, x); x += p1 ? y : 0.f;
  y = dppf<0x112, 0xF>(0.f, x); x += p2 ? y : 0.f;
  y = dppf<0x114, 0xF>(0.f, x); x += p4 ? y : 0.f;
  y = dppf<0x118, 0xF>(0.f, x); x += p8 ? y : 0.f;
  y = dppf<0x142, 0xA>(0.f, x); x += pA ? y : 0.f;
  y = dppf<0x143, 0xC>(0.f, x); x += pB ? y : 0.f;
  out[lane] = x;
}
int main() {
  int hs[64]; float hv[64], ho[64];
  int s = 0;
  unsigned rng = 12345;
  for (int i = 0; i < 64; ++i) { rng = rng * 1664525u + 1013904223u; if ((rng >> 28) < 3) ++s; hs[i] = s; hv[i] = (float)(i + 1); }
  int* ds; float *dv, *dout;
  hipMalloc(&ds, 256); hipMalloc(&dv, 256); hipMalloc(&dout, 256);
  hipMemcpy(ds, hs, 256, hipMemcpyHostToDevice); hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(ds, dv, dout);
  hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
  int bad = 0; float run = 0;
  for (int i = 0; i < 64; ++i) { run = (i > 0 && hs[i] == hs[i - 1]) ? run + hv[i] : hv[i]; if (run != ho[i]) { ++bad; printf("lane %d slot %d want %g got %g\n", i, hs[i], run, ho[i]); } }
  printf("segmented scan via DPP: %d mismatches\n", bad);
  return bad != 0;
}
