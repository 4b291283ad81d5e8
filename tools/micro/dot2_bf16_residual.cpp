// micro-test: the residual v - trunc_bf16(v) in ONE instruction, v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) packed bf16 constant
// against the packed upper halves of a value pair (instead of v_and_b32 + v_sub_f32).  Exactness check on a few magnitudes.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/dot2_bf16_residual.cpp -o /tmp/dot2
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out) {
  float v = in[threadIdx.x], w = in[threadIdx.x + 64];
  unsigned pk = __builtin_amdgcn_perm(__float_as_uint(w), __float_as_uint(v), 0x07060302u);   // (hi16 of w) : (hi16 of v)
  bf16x2 a = __builtin_bit_cast(bf16x2, pk);
  unsigned c_lo = 0x0000bf80u;                               // lower half = -1.0 (kept opaque: as an INLINE constant the compiler
  asm volatile("" : "+s"(c_lo));                            // emits -1.0, which the hardware applies to BOTH halves)
  bf16x2 m_lo = __builtin_bit_cast(bf16x2, c_lo);
  bf16x2 m_hi = __builtin_bit_cast(bf16x2, 0xbf800000u);   // (-1.0, 0): upper half = -1.0
  float r0 = __builtin_amdgcn_fdot2_f32_bf16(a, m_lo, v, false);
  float r1 = __builtin_amdgcn_fdot2_f32_bf16(a, m_hi, w, false);
  out[threadIdx.x] = r0; out[threadIdx.x + 64] = r1;
}
int main() {
  float h[128], o[128]; float *d, *e;
  for (int i = 0; i < 128; ++i) h[i] = (i % 2 ? -1.f : 1.f) * (1.2345678f + i * 0.0137f) * (i < 8 ? 1e-38f : (i > 120 ? 1e30f : 1.f));
  hipMalloc(&d, 512); hipMalloc(&e, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, e); hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) { unsigned b; memcpy(&b, &h[i], 4); b &= 0xffff0000u; float t; memcpy(&t, &b, 4); float want = h[i] - t; if (want != o[i]) { ++bad; printf("%d %g want %g got %g\n", i, h[i], want, o[i]); } }
  printf("bad %d\n", bad);
}
