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
//   tiles through private LDS images ([feature][34]: conflict-free b32 stores and two-pixel b64 loads), accumulates dW
//   tiles in accumulation registers across its whole pixel range, and the workgroup emits ONE partial slab; a tiny second
//   kernel sums the slabs (store pass + sum pass instead of ~10^6 contended float atomics).  The 3-wide dW2 runs on
//   v_mfma_f32_4x4x1_16b blocks, bias gradients fall out of the transposed operands (2 registers instead of 32).
//   Hidden activations travel from forward to backward through HBM (512 B/pixel each way, register-layout tiles of
//   1 KB per instruction): the stores and loads are issued between MFMAs and cost no issue time, whereas recomputing
//   the two layers cost 96 of 292 MFMAs per 32 pixels (the recompute variant is kept for callers without the buffer).
//   The extra traffic is not free either — the chip is power-limited and clocks the cores ~10 % lower under it
//   (tools/perf_decoder.py: same cycle counts, longer wall time) — but the trade wins 40 us per step.
//
// Issue model the schedules are built on (measured, tools/micro/gen_mfma_mix.py, mfma_2wave.cpp; one wave per SIMD):
//   * v_mfma_f32_32x32x2_f32 issues back to back at exactly 64 cycles, dependent accumulator chains included;
//   * a VALU instruction NEVER overlaps the wave's (or a sibling wave's) MFMAs: an MFMA followed by n VALU instructions
//     costs 64 + ~10 + 4n cycles — v_accvgpr_read/write and address arithmetic included;
//   * LDS, VMEM and scalar instructions placed between two MFMAs are free (up to ~15 per MFMA).
//   Hence: as few VALU instructions as possible (one-instruction ReLU, selects instead of mask multiplies, immediate
//   LDS offsets instead of address adds, buffer loads/stores whose range check replaces the tail masks, loop-carried
//   tiles pinned to AGPRs so hipcc does not shuttle them), gathered in a few bursts, and all LDS traffic of a phase
//   issued underneath the MFMA run of the phase before (a scheduling barrier per k-step pins the interleave).
#include "gngf_common.h"
#include <type_traits>
#include <utility>

namespace gngf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kH = 64;                 // hidden width (both hidden layers)
constexpr int kDecThreads = 256;

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// k index (feature of a 64-wide activation held as two accumulator tiles) consumed at chained k-step s2 by lane half h
__device__ __forceinline__ int kmapC(int s2, int h) { return 32 * (s2 >> 4) + crow(s2 & 15, h); }

// One VALU instruction for ReLU (fmaxf compiles to a canonicalising v_max pair), two for LeakyReLU (max(z, 0.01 z)).
// On gfx950 a wave's VALU instructions do NOT overlap its MFMAs (measured: 64 + 10 + 4n cycles for an MFMA followed by
// n VALU instructions, tools/micro/gen_mfma_mix.py), so every VALU instruction in the tile loop costs 4 cycles flat.
__device__ __forceinline__ float vmax0(float z) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(z)); return r; }
template <bool LEAKY> __device__ __forceinline__ float hidden_act(float z) {
  if (LEAKY) { float r; const float t = 0.01f * z; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(z), "v"(t)); return r; }
  return vmax0(z);
}
// d * act'(y) as a select (two VALU instructions; the multiply by a 0/1 mask would be a third)
template <bool LEAKY> __device__ __forceinline__ float hidden_dsel(float y, float d) { return y > 0.f ? d : (LEAKY ? 0.01f * d : 0.f); }

using f32x2 = __attribute__((ext_vector_type(2))) float;
// LDS access with an immediate byte offset from a per-lane base address (see decoder_bwd_kernel)
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}
template <int OFF> __device__ __forceinline__ void lds_store(unsigned addr, float v) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16 bits");
  asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ f32x2 lds_load2(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16 bits");
  f32x2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF> __device__ __forceinline__ float lds_load1(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16 bits");
  float r;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
// The consumers of lds_load* results depend on the loads, not on this wait: the scheduling barrier keeps them below it.
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// The same MFMA with its accumulator tile in ARCHITECTURAL VGPRs (hipcc's builtin always selects the AGPR form in a
// 512-register kernel, and every VALU touch of an AGPR result costs a v_accvgpr_read): used for the tiles whose results
// go straight through VALU (activations, masks).  Inline asm is invisible to the hazard recogniser, so the one hazard
// these chains have is handled by hand: MFMA_DRAIN (18 wait states, ISA "XDL write VGPR -> VALU / memory read" for a
// 16-pass MFMA) sits between the last MFMA of a chain and the first consumer of its result.  Dependent accumulation
// into the same registers issues back to back (measured: tools/micro/gen_mfma_regs.py), operands are read at issue.
#define MFMA_V(acc, a, b) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b))
#define MFMA_V_INIT(acc, a, b, c) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "v"(b), "v"(c))
#define MFMA_VV(acc, a, b) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_VV_ZERO(acc, a, b) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b))
#define MFMA_DRAIN(x, y) asm volatile("s_nop 15\n\ts_nop 2" : "+v"(x), "+v"(y))
#define MFMA_DRAIN1(x) asm volatile("s_nop 15\n\ts_nop 2" : "+v"(x))
// scheduling groups (LLVM SchedGroupMask): the next N instructions of that class, in the order the groups are written
#define SG_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, (n), 0)
#define SG_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, (n), 0)
#define SG_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, (n), 0)
#define SG_DSW(n) __builtin_amdgcn_sched_group_barrier(0x200, (n), 0)

// Coalesced copy of the raw weights into LDS with a +1 padded row stride (conflict-free strided reads afterwards):
//   W0s [64][in_dim+1] | W1s [64][65] | W2s [4][65] | b0 [64] | b1 [64] | b2 [4]
struct RawOff { int w0, w1, w2, b0, b1, b2, total; };
__host__ __device__ inline RawOff raw_offsets(int in_dim) {
  RawOff o;
  o.w0 = 0; o.w1 = o.w0 + kH * (in_dim + 1); o.w2 = o.w1 + kH * 65; o.b0 = o.w2 + 4 * 65; o.b1 = o.b0 + kH; o.b2 = o.b1 + kH;
  o.total = o.b2 + 4;
  return o;
}
__device__ __forceinline__ void stage_raw(float* raw, const float* __restrict__ W0, const float* __restrict__ b0,
                                          const float* __restrict__ W1, const float* __restrict__ b1,
                                          const float* __restrict__ W2, const float* __restrict__ b2, int in_dim, int out_dim) {
  const RawOff o = raw_offsets(in_dim);
  for (int e = threadIdx.x; e < kH * in_dim; e += kDecThreads) raw[o.w0 + (e / in_dim) * (in_dim + 1) + e % in_dim] = W0[e];
  for (int e = threadIdx.x; e < kH * kH; e += kDecThreads) raw[o.w1 + (e >> 6) * 65 + (e & 63)] = W1[e];
  for (int e = threadIdx.x; e < 4 * kH; e += kDecThreads) raw[o.w2 + (e >> 6) * 65 + (e & 63)] = (e >> 6) < out_dim ? W2[e] : 0.f;
  for (int e = threadIdx.x; e < kH; e += kDecThreads) { raw[o.b0 + e] = b0[e]; raw[o.b1 + e] = b1[e]; }
  if (threadIdx.x < 4) raw[o.b2 + threadIdx.x] = (b2 && (int)threadIdx.x < out_dim) ? b2[threadIdx.x] : 0.f;
  __syncthreads();
}

// LDS fragment images.  frag[(tile * S + s) * 64 + lane].
template <int KIN>
struct FwdFrags {
  static constexpr int S0 = KIN / 2;
  static constexpr int kA0 = 2 * S0 * 64, kA1 = 2 * 32 * 64, kA2 = 32 * 64;
};

template <int KIN>
__device__ __forceinline__ void fill_fwd_frags(float* A0, float* A1, const float* raw, int in_dim) {
  constexpr int S0 = KIN / 2;
  const RawOff o = raw_offsets(in_dim);
  for (int e = threadIdx.x; e < 2 * S0 * 64; e += kDecThreads) {
    const int lane = e & 63, s = (e >> 6) % S0, t = (e >> 6) / S0;
    const int i = lane & 31, h = lane >> 5, k = h * S0 + s;
    A0[e] = k < in_dim ? raw[o.w0 + (32 * t + i) * (in_dim + 1) + k] : 0.f;
  }
  for (int e = threadIdx.x; e < 2 * 32 * 64; e += kDecThreads) {
    const int lane = e & 63, s2 = (e >> 6) & 31, t = e >> 11;
    const int i = lane & 31, h = lane >> 5;
    A1[e] = raw[o.w1 + (32 * t + i) * 65 + kmapC(s2, h)];
  }
}

// ------------------------------------------------------------------------------------------------ forward
// One wave per SIMD (the whole 512-entry register file): every weight fragment and bias lives in registers for the
// lifetime of the persistent workgroup, so a tile is: 4 x 16-byte loads (prefetched one tile ahead), 32 + 64 f32 MFMAs
// chained through the accumulators, and the 3-4 wide output layer on v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 pixels:
// the lane's own accumulator register is the B operand, the lane's (channel = lane%4) weight the A operand; the two
// lane halves hold different features of the same pixel and are combined with one cross-half shuffle).
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// Device-clock span of the last decoder_bwd launch: [0] = constant 100 MHz counter when workgroup 0 started, [1] = the latest
// workgroup end.  One plain store and one result-less atomic per workgroup: this is how bench.py sees the kernel's duration
// INSIDE a replayed hipGraph, where HIP events cannot be recorded (gngf_decoder_bwd_last_span_ns).
__device__ unsigned long long g_bwd_span[2];

#if defined(GNGF_STAMPS)   // diagnostic build only (tools/perf_decoder.py --stamps): per-phase cycle shares of the backward loop
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long g_fstamps[16];
__device__ unsigned long long g_blocktime[2][256][2];   // [fwd/bwd][workgroup][start, end] (s_memtime)
#define STAMP(k) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    ph[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// Saved hidden activations: per 128-pixel tile 64 KB = [layer 2][wave 4][register group 8][lane 64][4 floats]; a lane's
// 16-byte pieces are the register quadruples 4g..4g+3 of its accumulator tile pair, so that every store / load instruction
// of a wave moves 1 KB of consecutive memory and the backward kernel gets the tiles back in the register layout it needs.
constexpr int kHiddenTileFloats = 2 * 4 * 8 * 64 * 4;
#ifndef GNGF_BWD_X_AUX
#define GNGF_BWD_X_AUX 0       // cache policy of the backward kernel's (last) read of enc
#endif
#ifndef GNGF_HIDDEN_AUX
#define GNGF_HIDDEN_AUX 2      // cache policy of the hidden-layer traffic: nt (streaming: written once, read once ~0.3 ms later; -15 us per step vs default)
#endif
#ifndef GNGF_HIDDEN_ST_AUX
#define GNGF_HIDDEN_ST_AUX GNGF_HIDDEN_AUX
#endif
#ifndef GNGF_HIDDEN_LD_AUX
#define GNGF_HIDDEN_LD_AUX GNGF_HIDDEN_AUX
#endif
__device__ __forceinline__ void hidden_store(float* hidden, int64_t tile, int which, unsigned hoff, const f32x16 (&v)[2]) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(hidden + tile * kHiddenTileFloats, 0, kHiddenTileFloats * 4, 0x00020000);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 q = {__float_as_uint(v[t][4 * g]), __float_as_uint(v[t][4 * g + 1]), __float_as_uint(v[t][4 * g + 2]),
                       __float_as_uint(v[t][4 * g + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(q, rs, hoff, which * 32768 + (t * 4 + g) * 1024, GNGF_HIDDEN_ST_AUX);
    }
}
#ifndef GNGF_BWD_REVERSE
#define GNGF_BWD_REVERSE 0
#endif
// The backward kernel can walk the tiles in the opposite order to the forward kernel, so that it starts on the hidden
// layers written last (the part of the 512 MiB buffer that may still sit in the 256 MiB Infinity Cache).  Measured: no
// effect with any cache policy of the hidden-layer traffic (249 us either way: its reads are prefetched a tile ahead and
// the kernel is bound by MFMA issue), so the default keeps the forward order.
__device__ __forceinline__ int64_t bwd_tile(int64_t t, int64_t ntiles) { return GNGF_BWD_REVERSE ? ntiles - 1 - t : t; }
__device__ __forceinline__ void hidden_load(const float* hidden, int64_t tile, int64_t ntiles, int which, unsigned hoff, f32x16 (&v)[2]) {
  const int64_t tt = tile < ntiles ? bwd_tile(tile, ntiles) : ntiles - 1;   // past the end: any valid tile (the values are never used)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hidden) + tt * kHiddenTileFloats, 0, kHiddenTileFloats * 4, 0x00020000);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, hoff, which * 32768 + (t * 4 + g) * 1024, GNGF_HIDDEN_LD_AUX);
      v[t][4 * g] = __uint_as_float(q.x); v[t][4 * g + 1] = __uint_as_float(q.y);
      v[t][4 * g + 2] = __uint_as_float(q.z); v[t][4 * g + 3] = __uint_as_float(q.w);
    }
}

// SAVE: the activated hidden tiles are also written to `hidden` for the backward kernel (layout: hidden_offset below).
// Stores are issued between MFMAs and cost no issue time; the 512 B/pixel ride on HBM bandwidth the kernel does not use.
template <int KIN, bool LEAKY, bool EXACT, bool SAVE>
__global__ void __launch_bounds__(kDecThreads, 1)
decoder_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W0, const float* __restrict__ b0,
                   const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                   const float* __restrict__ b2, float* __restrict__ Y, float* __restrict__ hidden, int64_t P, int in_dim,
                   int out_dim) {
  constexpr int S0 = KIN / 2;
  if (EXACT) in_dim = KIN;
#if defined(GNGF_STAMPS)
  if (threadIdx.x == 0) g_blocktime[0][blockIdx.x & 255][0] = __builtin_readcyclecounter();
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int64_t ntiles = (P + 127) / 128;
  // register-resident operands, gathered from a coalesced LDS copy of the raw weights
  extern __shared__ float raw[];
  stage_raw(raw, W0, b0, W1, b1, W2, b2, in_dim, out_dim);
  const RawOff o = raw_offsets(in_dim);
  float a0r[2][S0], a1r[2][32], w2a[32];
  f32x16 b0v[2], b1v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int s = 0; s < S0; ++s) { const int k = h * S0 + s; a0r[t][s] = k < in_dim ? raw[o.w0 + (32 * t + i) * (in_dim + 1) + k] : 0.f; }
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) a1r[t][s2] = raw[o.w1 + (32 * t + i) * 65 + kmapC(s2, h)];
#pragma unroll
    for (int r = 0; r < 16; ++r) { b0v[t][r] = raw[o.b0 + 32 * t + crow(r, h)]; b1v[t][r] = raw[o.b1 + 32 * t + crow(r, h)]; }
  }
  const int ch = lane & 3;
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) w2a[s2] = raw[o.w2 + ch * 65 + kmapC(s2, h)];
  float b2v[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) b2v[c] = raw[o.b2 + c];
  // The weight fragments are only ever MFMA operands: park them in the accumulation half of the register file so that
  // the activations (touched by VALU between the layers) get the architectural VGPRs and need no v_accvgpr moves.
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int s = 0; s < S0; ++s) asm volatile("" : "+a"(a0r[t][s]));
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) asm volatile("" : "+a"(a1r[t][s2]));
  }
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) asm volatile("" : "+a"(w2a[s2]));

  // Rows are fetched and results stored through per-tile buffer descriptors: out-of-range lanes (the ragged last tile,
  // tiles past the end, the idle lane half of a store, channels >= out_dim) read zeros / are dropped by the hardware
  // range check, so the loop body has no branches and stays ONE scheduling region per phase.
  const unsigned xoff = (unsigned)(((wave * 32 + i) * in_dim + (EXACT ? h * S0 : 0)) * 4);
  auto fetch = [&](int64_t t, float* dst) {
    const int64_t tt = t < ntiles ? t : ntiles;          // past the end: an empty window, every lane reads zeros
    int64_t rem = (P - tt * 128) * in_dim * 4;
    rem = rem < 0 ? 0 : (rem > 128 * in_dim * 4 ? 128 * in_dim * 4 : rem);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + tt * 128 * in_dim, 0, (int)rem, 0x00020000);
    if (EXACT) {                                          // asm loads: see the vmcnt note below
      static_for<S0 / 4>([&](auto K) {
        f32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(v) : "v"(xoff), "s"(rs), "n"(16 * K.value) : "memory");
        dst[4 * K.value] = v.x; dst[4 * K.value + 1] = v.y; dst[4 * K.value + 2] = v.z; dst[4 * K.value + 3] = v.w;
      });
    } else {
#pragma unroll
      for (int sx = 0; sx < S0; ++sx) {
        const int k = h * S0 + sx;
        dst[sx] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, k < in_dim ? xoff + 4 * k : 0x40000000u, 0, 0));
      }
    }
  };
  const unsigned yoff = (unsigned)(((wave * 32 + i) * out_dim) * 4);
  const unsigned hoff = (unsigned)(wave * 8192 + lane * 16);
  // x of the next tile is loaded straight into xr as soon as layer 1 has consumed it: it lands under layer 2
  float xr[S0];
  fetch(blockIdx.x, xr);
  // hipcc sizes the vmcnt wait in front of the first use of a prefetched register for the WORST predecessor of the loop
  // header: coming from the prologue nothing follows the loads, so it waits for "all but 3" memory operations on every
  // iteration — including the 12-20 stores the previous tile issued after its loads, i.e. for the full write latency at
  // the top of every tile (+2.8 k cycles per tile with the hidden-layer stores).  The exact-width path therefore issues
  // the row loads through inline asm and waits for them itself: kStoresPerTile memory operations follow them in a tile.
  constexpr int kStoresPerTile = (SAVE ? 16 : 0) + 4;
  if (EXACT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(GNGF_STAMPS)
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast) :: "memory");
#endif
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    STAMP(0);
    if (EXACT) {                                           // the row loads of this tile (issued a tile ago) have landed
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kStoresPerTile) : "memory");
#pragma unroll
      for (int sx = 0; sx < S0; ++sx) asm volatile("" : "+v"(xr[sx]));
    }
    STAMP(1);
    // MFMA runs and VALU bursts strictly alternate (every switch costs ~10 cycles on top of 4 per VALU instruction)
    f32x16 acc1[2], acc2[2];
    MFMA_V_INIT(acc1[0], a0r[0][0], xr[0], b0v[0]);
    MFMA_V_INIT(acc1[1], a0r[1][0], xr[0], b0v[1]);
#pragma unroll
    for (int sx = 1; sx < S0; ++sx) {
      MFMA_V(acc1[0], a0r[0][sx], xr[sx]);
      MFMA_V(acc1[1], a0r[1][sx], xr[sx]);
    }
    MFMA_DRAIN(acc1[0], acc1[1]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(tile + gridDim.x, xr);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[t][r] = hidden_act<LEAKY>(acc1[t][r]);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(2);
    if (SAVE) hidden_store(hidden, tile, 0, hoff, acc1);
    MFMA_V_INIT(acc2[0], a1r[0][0], acc1[0][0], b1v[0]);
    MFMA_V_INIT(acc2[1], a1r[1][0], acc1[0][0], b1v[1]);
#pragma unroll
    for (int s2 = 1; s2 < 32; ++s2) {
      const float b = acc1[s2 >> 4][s2 & 15];
      MFMA_V(acc2[0], a1r[0][s2], b);
      MFMA_V(acc2[1], a1r[1][s2], b);
    }
    MFMA_DRAIN(acc2[0], acc2[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = hidden_act<LEAKY>(acc2[t][r]);
    __builtin_amdgcn_sched_barrier(0);
    STAMP(3);
    if (SAVE) hidden_store(hidden, tile, 1, hoff, acc2);
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 32; s2 += 2) {
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w2a[s2], acc2[s2 >> 4][s2 & 15], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w2a[s2 + 1], acc2[(s2 + 1) >> 4][(s2 + 1) & 15], d1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      int64_t rem = (P - tile * 128) * out_dim * 4;
      rem = rem > 128 * out_dim * 4 ? 128 * out_dim * 4 : rem;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Y + tile * 128 * out_dim, 0, (int)rem, 0x00020000);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float dc = d0[c] + d1[c];
        const float z = dc + __shfl_xor(dc, 32, 64) + b2v[c];
        // Sigmoid on the hardware exp2 / rcp units (1 ulp each; |error| < 3e-7 absolute, tests hold 1e-6)
        const float y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
        const unsigned off = (h == 0 && c < out_dim) ? yoff + 4u * c : 0x40000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs, off, 0, 0);
      }
    }
    STAMP(4);
  }
#if defined(GNGF_STAMPS)
  if (blockIdx.x == 7 && threadIdx.x == 0)
    for (int k = 0; k < 10; ++k) g_fstamps[k] = ph[k];
  if (threadIdx.x == 0) g_blocktime[0][blockIdx.x & 255][1] = __builtin_readcyclecounter();
#endif
}

// ------------------------------------------------------------------------------------------------ backward
// slab layout (floats): dW0 [64*in_dim] | dW1 [64*64] | dW2 [out_dim*64] | db0 [64] | db1 [64] | db2 [out_dim]
// ... | max |d enc| of the workgroup (bit pattern; the slab reduction takes the maximum of this slot instead of the sum —
// no atomics and no memset for the encoder's fixed-point bound)
__host__ __device__ inline int slab_sums(int in_dim, int out_dim) { return kH * in_dim + kH * kH + out_dim * kH + 2 * kH + out_dim; }
__host__ __device__ inline int slab_size(int in_dim, int out_dim) { return slab_sums(in_dim, out_dim) + 1; }

constexpr int kImgStride = 34;   // 8-byte aligned rows: ds_read_b64 of two consecutive pixels, conflict-free (34*i mod 64)
constexpr int kImgFloats = 64 * kImgStride;
#include "decoder_split.inc"
// per-wave transposition images of the backward kernel: imgA | imgB | imgZ (dz3, 4 rows) | imgX (input rows; aliased to
// imgB when the input is 64 wide — the four dedicated images would not fit next to the 64-wide fragments in 160 KB)
template <int KIN> struct BwdLds {
  static constexpr bool kDedicatedX = KIN <= 32;
  static constexpr int kWaveFloats = 2 * kImgFloats + 4 * kImgStride + (kDedicatedX ? KIN * kImgStride : 0);
};

// EXACT: in_dim == KIN (no per-feature predicates anywhere in the loop)
// RECOMPUTE: the hidden activations are recomputed from enc (96 more MFMAs per tile); otherwise they are read back from
// the buffer the forward kernel saved them to (loads are free between MFMAs: 22.0 k -> ~14 k cycles per tile).
// HYB (KIN = 32, exact width, saved hidden layers): the two products with the PIXEL on the lane — dh1 = W1^T dz2 and
// d enc = W0^T dz1, 96 of the 196 fp32 MFMAs of a tile — run on v_mfma_f32_32x32x16_bf16 with the exact three-way bf16 split of
// decoder_split.inc (operands straight from the accumulator registers, W^T as bf16 planes in LDS): 72 MFMAs of 32 cycles
// instead of 96 of 64, and the splitting issues underneath them.  The weight-gradient products (pixel on the k axis, fp32
// transposition images) stay on the fp32 pipe, which keeps the kernel below the power limit of an all-bf16 one.
// TRAIN (with HYB): forward AND backward of the decoder for the training step whose loss is MSELoss(rgb, target) with a KNOWN
// upstream gradient (gloss): the tile's hidden layers are computed here (both layers on the bf16 pipe with the exact split, as
// decoder_fwd_split_kernel), rgb is WRITTEN to Yout, d rgb is formed from it and the target, and the backward phases follow on
// the registers that hold h1 / h2 — the hidden layers never travel through HBM (1 GB per step at 2^20 pixels) and the forward
// kernel's launch disappears.
template <int KIN, bool LEAKY, bool EXACT, bool RECOMPUTE, bool HYB = false, bool TRAIN = false>
__global__ void __launch_bounds__(kDecThreads, 1)
decoder_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Yout, const float* __restrict__ dY,
                   const float* __restrict__ W0, const float* __restrict__ b0, const float* __restrict__ W1,
                   const float* __restrict__ b1, const float* __restrict__ W2, float* __restrict__ dX,
                   float* __restrict__ slabs, const float* __restrict__ hidden, int64_t P, int in_dim, int out_dim,
                   const float* __restrict__ target, const float* __restrict__ gloss, const float* __restrict__ b2 = nullptr,
                   float4* __restrict__ zero_fill = nullptr, int64_t zero_vecs = 0) {
  using FF = FwdFrags<KIN>;
  constexpr int S0 = KIN / 2;
  constexpr int TX = (KIN + 31) / 32;                    // 32-row tiles of the input width
  if (EXACT) in_dim = KIN;
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_bwd_span[1] = 0ull; g_bwd_span[0] = wall_clock64(); }
#if defined(GNGF_STAMPS)
  if (threadIdx.x == 0) g_blocktime[1][blockIdx.x & 255][0] = __builtin_readcyclecounter();
#endif
  extern __shared__ float smem[];
  float* A0 = smem;                                      // forward fragments (recompute)
  float* A1 = A0 + FF::kA0;
  float* A2T = A1 + FF::kA1;                             // [t(2)][s(2)][64]   : W2[c = 2s+h][32t+i]
  static_assert(!HYB || (KIN == 32 && EXACT && !RECOMPUTE), "hybrid variant: 32 input features, saved hidden layers");
  static_assert(!TRAIN || HYB, "the fused training kernel builds on the hybrid backward");
  constexpr int kA1T = HYB ? 8 * 3 * 64 * 4 : 2 * 32 * 64, kA0T = HYB ? 4 * 3 * 64 * 4 : TX * 32 * 64;
  float* A1T = A2T + 2 * 2 * 64;                         // [t(2)][s2(32)][64] : W1[kmapC(s2,h)][32t+i]   (HYB: bf16 planes of W1^T)
  float* A0T = A1T + kA1T;                               // [t(TX)][s2(32)][64]: W0[kmapC(s2,h)][32t+i]   (HYB: bf16 planes of W0^T)
  float* bs = A0T + kA0T;                                // b0 | b1
  float* img = bs + 2 * kH;                              // per wave: imgA [64][34] | imgB [64][34] | imgZ [4][34] | imgX [KIN][34]
  const int nslab = slab_size(in_dim, out_dim);

  float* raw = img;                                      // the image area is free until the main loop starts
  stage_raw(raw, W0, b0, W1, b1, W2, TRAIN ? b2 : nullptr, in_dim, out_dim);
  const RawOff ro = raw_offsets(in_dim);
  // TRAIN: the area of the fp32 recompute fragments holds the forward planes of W0 (fragments 2 c + t: rows 32 t + i,
  // k = 16 h + 8 c + j; 12 KB) and the output layer's weights as [s2 / 4][lane][4] (8 KB)
  u32x4* w0f = reinterpret_cast<u32x4*>(A0);
  float* w2L = A0 + 4 * 3 * 64 * 4;
  if constexpr (TRAIN) {
    const int ln = threadIdx.x & 63, li = ln & 31, lh = ln >> 5;
    {
      const int f = threadIdx.x >> 6, c = f >> 1, t = f & 1;
      const float* r0 = raw + ro.w0 + (32 * t + li) * (in_dim + 1) + lh * S0 + 8 * c;
      store_planes(w0f + f * 3 * 64 + ln, split8(r0[0], r0[1], r0[2], r0[3], r0[4], r0[5], r0[6], r0[7]));
    }
    for (int e = threadIdx.x; e < 32 * 64; e += kDecThreads) {
      const int s2 = e >> 6, l2 = e & 63;
      w2L[((s2 >> 2) * 64 + l2) * 4 + (s2 & 3)] = raw[ro.w2 + (l2 & 3) * 65 + kmapC(s2, l2 >> 5)];
    }
  } else
  fill_fwd_frags<KIN>(A0, A1, raw, in_dim);
  for (int e = threadIdx.x; e < 2 * 2 * 64; e += kDecThreads) {
    const int lane = e & 63, s = (e >> 6) & 1, t = e >> 7;
    const int c = 2 * s + (lane >> 5);
    A2T[e] = raw[ro.w2 + c * 65 + 32 * t + (lane & 31)];
  }
  if constexpr (HYB) {
    // bf16 planes of W1^T (fragments cc * 2 + t: rows 32 t + i, k = kmapS(cc, h, j)) and W0^T (fragments cc: rows i)
    const int ln = threadIdx.x & 63, li = ln & 31, lh = ln >> 5;
    for (int f = threadIdx.x >> 6; f < 12; f += 4) {
      float v[8];
      if (f < 8) {
        const int cc = f >> 1, t = f & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = raw[ro.w1 + kmapS(cc, lh, j) * 65 + 32 * t + li];
        store_planes(reinterpret_cast<u32x4*>(A1T) + f * 3 * 64 + ln, split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]));
      } else {
        const int cc = f - 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = raw[ro.w0 + kmapS(cc, lh, j) * (in_dim + 1) + li];
        store_planes(reinterpret_cast<u32x4*>(A0T) + cc * 3 * 64 + ln, split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]));
      }
    }
  } else {
  for (int e = threadIdx.x; e < 2 * 32 * 64; e += kDecThreads) {
    const int lane = e & 63, s2 = (e >> 6) & 31, t = e >> 11;
    A1T[e] = raw[ro.w1 + kmapC(s2, lane >> 5) * 65 + 32 * t + (lane & 31)];
  }
  for (int e = threadIdx.x; e < TX * 32 * 64; e += kDecThreads) {
    const int lane = e & 63, s2 = (e >> 6) & 31, t = e >> 11;
    const int j = 32 * t + (lane & 31);
    A0T[e] = j < in_dim ? raw[ro.w0 + kmapC(s2, lane >> 5) * (in_dim + 1) + j] : 0.f;
  }
  }
  for (int e = threadIdx.x; e < 2 * kH; e += kDecThreads) bs[e] = e < kH ? raw[ro.b0 + e] : raw[ro.b1 + e - kH];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  using BL = BwdLds<KIN>;
  float* imgA = img + wave * BL::kWaveFloats;
  // Per-lane LDS byte addresses of the transposition images.  All image traffic goes through ds ops with IMMEDIATE
  // offsets from these few bases (inline asm): left to itself hipcc pairs the stores into ds_write2_b32, whose 8-bit
  // offsets force a v_add per pair — and a VALU instruction costs 4 issue cycles that nothing overlaps (see above).
  const unsigned aImg = lds_addr(imgA);
  const unsigned wTile = aImg + (unsigned)(4 * h * kImgStride + i) * 4u;        // accumulator tile -> rows crow(r, h), column i
  constexpr int kImgB = kImgFloats * 4;                                         // byte offsets of images B, Z (dz3), X (input rows)
  constexpr int kImgZ = 2 * kImgFloats * 4;
  constexpr int kImgX = BL::kDedicatedX ? (2 * kImgFloats + 4 * kImgStride) * 4 : kImgB;
  const unsigned wDz3 = aImg + (unsigned)i * 4u;                                // dz3 -> rows c of imgZ
  const unsigned wX = aImg + (unsigned)(h * S0 * kImgStride + i) * 4u;          // x slice -> rows h*S0 + s of imgX
  const unsigned rOp = aImg + (unsigned)(i * kImgStride + 16 * h) * 4u;         // operand rows i (+32), pixels 16h + 2q, 2q+1
  const unsigned rZ2 = aImg + (unsigned)((lane & 3) * kImgStride) * 4u;         // dW2: A rows lane & 3 of imgZ, all 32 pixels
  const unsigned rH2 = aImg + (unsigned)(lane * kImgStride) * 4u;               //      B rows lane of imgA
  f32x16 dW1acc[2][2], dW0acc[2][TX];
  f32x4 dW2acc = {0.f, 0.f, 0.f, 0.f};                   // lane j: dW2[c][j], c = 0..3
  float db0acc[2] = {0.f, 0.f}, db1acc[2] = {0.f, 0.f};  // lane (i,h): partial of db[32a + i] over the half's 16 pixels
  float db2acc[4] = {0.f, 0.f, 0.f, 0.f};                // both lane halves accumulate the same pixel; half 0 is used
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) dW1acc[a][b] = 0;
#pragma unroll
    for (int b = 0; b < TX; ++b) dW0acc[a][b] = 0;
  }

  // The loop-carried weight-gradient tiles must STAY in accumulation registers: left alone hipcc assigns them to VGPRs
  // and copies 100 registers into AGPRs and back around the MFMA runs of every tile.  Pinned at every phase boundary.
  auto pin_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) asm volatile("" : "+a"(dW1acc[a][b]));
#pragma unroll
      for (int b = 0; b < TX; ++b) asm volatile("" : "+a"(dW0acc[a][b]));
    }
    asm volatile("" : "+a"(dW2acc));
  };
  const int64_t ntiles = (P + 127) / 128;
  // TRAIN: a buffer the REST of the step needs zero-filled (the table gradient the encoder backward accumulates into: 64 MiB at
  // T = 2^19) is cleared from here — a few 16-byte stores per lane and tile between the MFMAs, where memory instructions cost
  // no issue time and the HBM is idle — instead of in rider workgroups of the binning launch (8 us of the step's critical path).
  // Workgroup b owns vectors [b per, (b + 1) per); every tile of its loop clears zper vectors per thread.
  int64_t zlo = 0, zhi = 0;
  int zper = 0;
  // (besides the training kernel, the exact 64-feature backward — no training kernel exists at that width — clears the buffer:
  // gngf_decoder_bwd; the other variants stay as they were: the 32-feature hybrid backward spilled 16 registers with it)
  constexpr bool kClears = TRAIN || (KIN == 64 && EXACT && !HYB);
  if constexpr (kClears) {
    if (zero_fill && zero_vecs > 0) {
      const int64_t per = (zero_vecs + gridDim.x - 1) / gridDim.x;
      zlo = (int64_t)blockIdx.x * per + threadIdx.x;
      zhi = (int64_t)(blockIdx.x + 1) * per < zero_vecs ? (int64_t)(blockIdx.x + 1) * per : zero_vecs;
      const int64_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
      zper = (int)((per + (int64_t)kDecThreads * my_tiles - 1) / ((int64_t)kDecThreads * my_tiles));
    }
  }
  // Rows are fetched / stored through per-tile buffer descriptors (as in the forward kernel): lanes past the end of
  // the batch read zeros (so dz3 = 0 and they contribute nothing anywhere) and their stores are dropped — no masks.
  const unsigned xoff = (unsigned)(((wave * 32 + i) * in_dim + (EXACT ? h * S0 : 0)) * 4);
  const unsigned yoff = (unsigned)(((wave * 32 + i) * out_dim) * 4);
  // Pipeline: x of tile t+1 is loaded straight into xr as soon as tile t has consumed it (after layer 1 / after the x
  // image is written), y and dy of tile t+1 once dz3 of tile t is dead; both land long before the next tile starts, and
  // the d-enc stores at the end of a tile are issued after them, so no wait ever covers a store (vmcnt counts in order).
  float xr[S0], yn[4], dyn[4], dz3[4];
  // fused pixel loss (torch.nn.MSELoss backward, csrc/loss.hip::mse_bwd_kernel): d rgb = gloss * 2/n * (rgb - target) is formed
  // here from the target instead of being read from a tensor a separate kernel wrote — the same expression, the same bits
  const bool fused_loss = target != nullptr;              // wave-uniform
  const float* dsrc = fused_loss ? target : dY;
  const float kloss = fused_loss ? gloss[0] * (2.0f / (float)(P * out_dim)) : 0.f;
  auto tile_window = [&](int64_t t, int64_t& tt, int& rows) {
    tt = t < ntiles ? bwd_tile(t, ntiles) : ntiles;      // past the end: an empty window, every lane reads zeros
    int64_t rem = P - tt * 128;
    rows = (int)(rem < 0 ? 0 : (rem > 128 ? 128 : rem));
  };
  auto fetch_x = [&](int64_t t) {
    int64_t tt; int rows;
    tile_window(t, tt, rows);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + tt * 128 * in_dim, 0, rows * in_dim * 4, 0x00020000);
    if (EXACT) {
#pragma unroll
      for (int k = 0; k < S0 / 4; ++k) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + 16 * k, 0, GNGF_BWD_X_AUX);
        xr[4 * k] = __uint_as_float(v.x); xr[4 * k + 1] = __uint_as_float(v.y);
        xr[4 * k + 2] = __uint_as_float(v.z); xr[4 * k + 3] = __uint_as_float(v.w);
      }
    } else {
#pragma unroll
      for (int sx = 0; sx < S0; ++sx) {
        const int k = h * S0 + sx;
        xr[sx] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, k < in_dim ? xoff + 4 * k : 0x40000000u, 0, 0));
      }
    }
  };
  auto fetch_y = [&](int64_t t) {
    int64_t tt; int rows;
    tile_window(t, tt, rows);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Yout) + tt * 128 * out_dim, 0, rows * out_dim * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dsrc) + tt * 128 * out_dim, 0, rows * out_dim * 4, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned off = c < out_dim ? yoff + 4u * c : 0x40000000u;
      if constexpr (!TRAIN) yn[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
      dyn[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, off, 0, 0));
    }
  };
  auto make_dz3 = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float dy = fused_loss ? kloss * (yn[c] - dyn[c]) : dyn[c];
      dz3[c] = dy * (yn[c] * (1.f - yn[c]));              // Sigmoid backward; 0 for padding pixels and channels
      asm volatile("" : "+v"(dz3[c]));                   // (keeps dz3 in registers: it is selected by lane half below)
      db2acc[c] += dz3[c];
    }
  };
  fetch_x(blockIdx.x);
  fetch_y(blockIdx.x);
  if constexpr (!TRAIN) make_dz3();
  float b2v[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (TRAIN) {
#pragma unroll
    for (int c = 0; c < 4; ++c) b2v[c] = raw[ro.b2 + c];
  }
  // biases: accumulator-file residents, the C operand of the first MFMA of each recompute chain
  f32x16 b0v[2], b1v[2];
  if (RECOMPUTE) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { b0v[t][r] = bs[32 * t + crow(r, h)]; b1v[t][r] = bs[kH + 32 * t + crow(r, h)]; }
      asm volatile("" : "+a"(b0v[t]), "+a"(b1v[t]));
    }
  }
  unsigned dxmax = 0u;                                   // bits of the largest |d enc| this lane produced (hint for the encoder backward)
#if defined(GNGF_STAMPS)
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast) :: "memory");
#endif
  // first tile's layer-1 fragments; later tiles load theirs under the dW0 run of the tile before.  (64-wide input: the
  // register file cannot carry them across the tile as well — they are loaded at the top of every tile instead, and the
  // second half of the W0^T fragments takes their place under dW0.)
  constexpr bool kCarryF0 = KIN <= 32;
  float f0[2][S0];
  // hidden tiles (h1 = acc1, h2 = acc2, activated) and the W1^T fragments of dh1: recomputed / loaded inside the tile when
  // RECOMPUTE, else carried across the loop — read from the forward kernel's buffer one tile ahead
  f32x16 acc1[2], acc2[2];
  float ft[2][32];
  const unsigned hoff = (unsigned)(wave * 8192 + lane * 16);
  if (!RECOMPUTE && !TRAIN) {
    hidden_load(hidden, blockIdx.x, ntiles, 0, hoff, acc1);
    hidden_load(hidden, blockIdx.x, ntiles, 1, hoff, acc2);
    if constexpr (!HYB) {
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) { ft[0][s2] = A1T[(0 * 32 + s2) * 64 + lane]; ft[1][s2] = A1T[(1 * 32 + s2) * 64 + lane]; }
    }
  }
  // HYB: h2 -> image A and h1 -> image B, for the tile whose hidden layers sit in acc2 / acc1 (64 stores; E = 0 .. 15)
  auto store_hidden_images = [&](auto E) {
    constexpr int e0 = 2 * E.value;
    static_for<2>([&](auto J) {
      constexpr int e = e0 + J.value, t = e >> 4, r = e & 15;
      lds_store<(32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc2[t][r]);
      lds_store<kImgB + (32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc1[t][r]);
    });
  };
  if constexpr (HYB && !TRAIN) static_for<16>([&](auto E) { store_hidden_images(E); });      // first tile
  const u32x4* w1t = reinterpret_cast<const u32x4*>(A1T) + lane;      // HYB: plane p of fragment f at [(f * 3 + p) * 64]
  const u32x4* w0t = reinterpret_cast<const u32x4*>(A0T) + lane;
  // HYB: the W1^T planes are MFMA operands only and stay in accumulation registers for the kernel's lifetime (96 of them); the
  // W0^T planes (48) are loaded under the dW0 run of every tile
  Planes w1p[HYB ? 8 : 1], w0p[HYB ? 4 : 1];
  if constexpr (HYB) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      w1p[f] = load_planes(w1t + f * 3 * 64);
      asm volatile("" : "+a"(w1p[f].hi), "+a"(w1p[f].mid), "+a"(w1p[f].lo));
    }
  }
  // TRAIN: the W1^T planes live in registers now; their LDS area takes the FORWARD planes of W1 (fragments 2 cc + t: rows
  // 32 t + i, k = kmapS(cc, h, j)).  `raw` (the image area) is still intact: the first image store comes after the barrier.
  u32x4* w1f = reinterpret_cast<u32x4*>(A1T);
  if constexpr (TRAIN) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int f = threadIdx.x >> 6; f < 8; f += 4) {
      const int cc = f >> 1, t = f & 1;
      const float* r1 = raw + ro.w1 + (32 * t + i) * 65;
      store_planes(w1f + f * 3 * 64 + lane, split8(r1[kmapS(cc, h, 0)], r1[kmapS(cc, h, 1)], r1[kmapS(cc, h, 2)], r1[kmapS(cc, h, 3)],
                                                  r1[kmapS(cc, h, 4)], r1[kmapS(cc, h, 5)], r1[kmapS(cc, h, 6)], r1[kmapS(cc, h, 7)]));
    }
    __syncthreads();
  }
  if (RECOMPUTE && kCarryF0) {
#pragma unroll
    for (int sx = 0; sx < S0; ++sx) { f0[0][sx] = A0[(0 * S0 + sx) * 64 + lane]; f0[1][sx] = A0[(1 * S0 + sx) * 64 + lane]; }
  }
  // Schedule of one 32-pixel tile.  A lone wave per SIMD issues in order: VALU instructions never overlap its MFMAs
  // (4 cycles each, flat), but LDS / VMEM / scalar instructions issued between two MFMAs disappear under the 64-cycle
  // matrix pass.  So every MFMA run CARRIES the LDS traffic of the steps that follow it (image stores, operand and
  // fragment loads; a scheduling barrier per k-step pins the interleave), and the VALU work is gathered in five bursts:
  //   R1  [x image, W1 frags]  | relu | R2 [h1 image, dz3 image, W1^T frags] + dh2 | relu, mask |
  //   dh1 [h2 image, dW2 operands -> 32 4x4 MFMAs; dz2 image; dW1 operands; h1 read-back] | mask, db1 |
  //   dW1 [dz1 image, dW0 operands, W0^T frags] | dW0 [next tile's W0 frags] | dX | |d enc| max, db0, next dz3 | stores
#define STEP_END() __builtin_amdgcn_sched_barrier(0)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    pin_acc();
    STAMP(0);
    float f1[2][32];
    if constexpr (TRAIN) {
      // ---- forward of the tile on the bf16 pipe (exact three-way split): h1 = act(W0 x + b0), h2 = act(W1 h1 + b1).
      // Software pipeline: the planes of the next k-chunk are requested and the next chunk's operand is split while the six
      // cross products of the current chunk (x 2 output tiles) issue; the h1 image rides under layer 2.
      Planes pa0 = load_planes(w0f + lane), pa1 = load_planes(w0f + 3 * 64 + lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 q0 = *reinterpret_cast<const f32x4*>(bs + 32 * t + 8 * g + 4 * h);
          const f32x4 q1 = *reinterpret_cast<const f32x4*>(bs + kH + 32 * t + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc1[t][4 * g + e] = q0[e]; acc2[t][4 * g + e] = q1[e]; }
        }
      Planes zc = split8(xr[0], xr[1], xr[2], xr[3], xr[4], xr[5], xr[6], xr[7]);
      // one pair of products (both output tiles) + a quarter of the next operand's split (values nv[2 q], nv[2 q + 1])
      unsigned nh[4], nm[4], nl[4];
      auto quarter = [&](int q, float a, float b) {
        nh[q] = pack_hi16(b, a);
        const float ra = trunc_residual(a), rb = trunc_residual(b);
        nm[q] = pack_hi16(rb, ra);
        nl[q] = pack_hi16(trunc_residual(rb), trunc_residual(ra));
      };
      auto take_next = [&]() {
        zc.hi = u32x4{nh[0], nh[1], nh[2], nh[3]}; zc.mid = u32x4{nm[0], nm[1], nm[2], nm[3]}; zc.lo = u32x4{nl[0], nl[1], nl[2], nl[3]};
      };
      {   // layer 1, chunk 0 (carries: planes of chunk 1, split of x[8 .. 15])
        const Planes na0 = load_planes(w0f + 2 * 3 * 64 + lane), na1 = load_planes(w0f + 3 * 3 * 64 + lane);
        acc1[0] = mfma_b(pa0.hi, zc.lo, acc1[0]);  acc1[1] = mfma_b(pa1.hi, zc.lo, acc1[1]);  quarter(0, xr[8], xr[9]);   STEP_END();
        acc1[0] = mfma_b(pa0.lo, zc.hi, acc1[0]);  acc1[1] = mfma_b(pa1.lo, zc.hi, acc1[1]);  quarter(1, xr[10], xr[11]); STEP_END();
        acc1[0] = mfma_b(pa0.mid, zc.mid, acc1[0]); acc1[1] = mfma_b(pa1.mid, zc.mid, acc1[1]); quarter(2, xr[12], xr[13]); STEP_END();
        acc1[0] = mfma_b(pa0.hi, zc.mid, acc1[0]); acc1[1] = mfma_b(pa1.hi, zc.mid, acc1[1]); quarter(3, xr[14], xr[15]); STEP_END();
        acc1[0] = mfma_b(pa0.mid, zc.hi, acc1[0]); acc1[1] = mfma_b(pa1.mid, zc.hi, acc1[1]);
        acc1[0] = mfma_b(pa0.hi, zc.hi, acc1[0]);  acc1[1] = mfma_b(pa1.hi, zc.hi, acc1[1]);  STEP_END();
        take_next(); pa0 = na0; pa1 = na1;
      }
      {   // layer 1, chunk 1 (carries: planes of layer 2's chunk 0)
        const Planes na0 = load_planes(w1f + lane), na1 = load_planes(w1f + 3 * 64 + lane);
        mfma_split2(pa0, pa1, zc, acc1[0], acc1[1]);
        STEP_END();
        pa0 = na0; pa1 = na1;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[t][r] = hidden_act<LEAKY>(acc1[t][r]);      // (one instruction; its readers are VALU / LDS stores)
      zc = split8(acc1[0][0], acc1[0][1], acc1[0][2], acc1[0][3], acc1[0][4], acc1[0][5], acc1[0][6], acc1[0][7]);
      STEP_END();
      static_for<4>([&](auto CC) {
        constexpr int cc = CC.value, nc = cc + 1;
        Planes na0, na1;
        if constexpr (nc < 4) { na0 = load_planes(w1f + (2 * nc) * 3 * 64 + lane); na1 = load_planes(w1f + (2 * nc + 1) * 3 * 64 + lane); }
        auto step = [&](auto PP, u32x4 a0, u32x4 a1, u32x4 b) {
          constexpr int pp = PP.value;
          acc2[0] = mfma_b(a0, b, acc2[0]);
          acc2[1] = mfma_b(a1, b, acc2[1]);
          if constexpr (nc < 4 && pp < 4) {
            constexpr int t = nc >> 1, r0 = 8 * (nc & 1) + 2 * pp;
            quarter(pp, acc1[t][r0], acc1[t][r0 + 1]);
          }
          if constexpr (pp < 4) {                           // h1 image (image B): 8 stores per chunk
            static_for<2>([&](auto E) {
              constexpr int e = 8 * cc + 2 * pp + E.value, tt = e >> 4, r = e & 15;
              lds_store<kImgB + (32 * tt + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc1[tt][r]);
            });
          }
          STEP_END();
        };
        step(std::integral_constant<int, 0>{}, pa0.hi, pa1.hi, zc.lo);
        step(std::integral_constant<int, 1>{}, pa0.lo, pa1.lo, zc.hi);
        step(std::integral_constant<int, 2>{}, pa0.mid, pa1.mid, zc.mid);
        step(std::integral_constant<int, 3>{}, pa0.hi, pa1.hi, zc.mid);
        step(std::integral_constant<int, 4>{}, pa0.mid, pa1.mid, zc.hi);
        step(std::integral_constant<int, 5>{}, pa0.hi, pa1.hi, zc.hi);
        if constexpr (nc < 4) { take_next(); pa0 = na0; pa1 = na1; }
      });
      f32x4 w2v[8];                                        // output-layer weights: requested before the activation burst
#pragma unroll
      for (int g = 0; g < 8; ++g) w2v[g] = *reinterpret_cast<const f32x4*>(w2L + (g * 64 + lane) * 4);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = hidden_act<LEAKY>(acc2[t][r]);
      STEP_END();                                          // (the 4x4 MFMAs below must not follow their operand's v_max directly)
      // ---- output layer on v_mfma_f32_4x4x1 (as decoder_fwd_kernel), rgb out, d rgb from it and the target
      {
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const f32x4 w = w2v[g];
          o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, acc2[(4 * g) >> 4][(4 * g) & 15], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, acc2[(4 * g + 1) >> 4][(4 * g + 1) & 15], o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.z, acc2[(4 * g + 2) >> 4][(4 * g + 2) & 15], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w, acc2[(4 * g + 3) >> 4][(4 * g + 3) & 15], o1, 0, 0, 0);
        }
        int64_t rem = (P - tile * 128) * out_dim * 4;
        rem = rem > 128 * out_dim * 4 ? 128 * out_dim * 4 : rem;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Yout) + tile * 128 * out_dim, 0, (int)rem, 0x00020000);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float dc = o0[c] + o1[c];
          const float z = dc + __shfl_xor(dc, 32, 64) + b2v[c];
          const float y = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
          const unsigned off = (h == 0 && c < out_dim) ? yoff + 4u * c : 0x40000000u;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs, off, 0, 0);
          // pixels past the end of the batch read x = 0 but still produce an output: they must not produce a gradient
          yn[c] = (c < out_dim && tile * 128 + wave * 32 + i < P) ? y : 0.f;
        }
      }
      make_dz3();
      STEP_END();
    }
    if constexpr (RECOMPUTE) {
    if (!kCarryF0) {
#pragma unroll
      for (int sx = 0; sx < S0; ++sx) { f0[0][sx] = A0[(0 * S0 + sx) * 64 + lane]; f0[1][sx] = A0[(1 * S0 + sx) * 64 + lane]; }
      STEP_END();
    }
    // ---- R1: h1^T = W0 x^T + b0
    static_for<S0>([&](auto SX) {
      constexpr int sx = SX.value;
      if constexpr (sx == 0) { acc1[0] = MFMA(f0[0][0], xr[0], b0v[0]); acc1[1] = MFMA(f0[1][0], xr[0], b0v[1]); }
      else { acc1[0] = MFMA(f0[0][sx], xr[sx], acc1[0]); acc1[1] = MFMA(f0[1][sx], xr[sx], acc1[1]); }
      if (BL::kDedicatedX) lds_store<kImgX + sx * kImgStride * 4>(wX, xr[sx]);
#pragma unroll
      for (int j = 0; j < 64 / S0; ++j) {                // W1 fragments of R2
        const int e = sx * (64 / S0) + j;
        f1[e >> 5][e & 31] = A1[e * 64 + lane];
      }
      STEP_END();
    });
    if (BL::kDedicatedX) fetch_x(tile + gridDim.x);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[t][r] = hidden_act<LEAKY>(acc1[t][r]);
    STEP_END();
    // ---- R2: h2^T = W1 h1^T + b1
    static_for<32>([&](auto S2) {
      constexpr int s2 = S2.value, t = s2 >> 4, r = s2 & 15;
      if constexpr (s2 == 0) { acc2[0] = MFMA(f1[0][0], acc1[0][0], b1v[0]); acc2[1] = MFMA(f1[1][0], acc1[0][0], b1v[1]); }
      else { acc2[0] = MFMA(f1[0][s2], acc1[t][r], acc2[0]); acc2[1] = MFMA(f1[1][s2], acc1[t][r], acc2[1]); }
      lds_store<kImgB + (32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc1[t][r]);          // h1 image
      if constexpr (s2 < 4) lds_store<kImgZ + s2 * kImgStride * 4>(wDz3, dz3[s2]);                           // dz3 image
      ft[0][s2] = A1T[(0 * 32 + s2) * 64 + lane];         // W1^T fragments of dh1
      ft[1][s2] = A1T[(1 * 32 + s2) * 64 + lane];
      STEP_END();
    });
    }   // RECOMPUTE
    pin_acc();
    STAMP(1);
    // ---- dh2^T = W2^T dz3^T  (k = output channel, padded to 4): needs no activation, so it runs straight behind R2 and
    //      the ReLU of h2 and the mask of dh2 share one VALU burst
    f32x16 d2[2];
    {
      float fa[2][2], bsel[2];
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        fa[0][sk] = A2T[(0 * 2 + sk) * 64 + lane]; fa[1][sk] = A2T[(1 * 2 + sk) * 64 + lane];
        float lo = dz3[2 * sk], hi = dz3[2 * sk + 1];
        asm volatile("" : "+v"(lo), "+v"(hi));             // a select of two registers, not a load from a private array
        bsel[sk] = h == 0 ? lo : hi;
      }
      STEP_END();
      MFMA_VV_ZERO(d2[0], fa[0][0], bsel[0]); MFMA_VV_ZERO(d2[1], fa[1][0], bsel[0]);
      MFMA_VV(d2[0], fa[0][1], bsel[1]); MFMA_VV(d2[1], fa[1][1], bsel[1]);
      if constexpr (!RECOMPUTE) {                          // the images R1 / R2 would have carried
        if (BL::kDedicatedX) static_for<S0>([&](auto SX) { lds_store<kImgX + SX.value * kImgStride * 4>(wX, xr[SX.value]); });
        static_for<4>([&](auto C) { lds_store<kImgZ + C.value * kImgStride * 4>(wDz3, dz3[C.value]); });
      }
    }
    if (!RECOMPUTE && BL::kDedicatedX) fetch_x(tile + gridDim.x);
    fetch_y(tile + gridDim.x);
    MFMA_DRAIN(d2[0], d2[1]);
    STEP_END();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (RECOMPUTE) acc2[t][r] = hidden_act<LEAKY>(acc2[t][r]);
        d2[t][r] = hidden_dsel<LEAKY>(acc2[t][r], d2[t][r]);
      }
    pin_acc();
    STEP_END();
    STAMP(3);
    // ---- dh1^T = W1^T dz2^T.  Underneath: the h2 image and the dW2 operands (then the 32 4x4 MFMAs of dW2 in mid-run),
    //      the dz2 image (over h2, whose reads were issued first), the dW1 operands and h1 for the mask.
    f32x16 d1[2];
    f32x2 a0p[8], a1p[8], b0p[8], b1p[8];
    float h1v[2][16];
    if constexpr (HYB) {
      // dh1^T = W1^T dz2^T on the bf16 pipe: k-chunk cc = registers 8 (cc & 1) .. + 7 of d2[cc >> 1] in both lane halves.
      // A chunk is 6 cross products x 2 output tiles; it carries the LDS traffic of eight steps of the fp32 schedule
      // (0-15 dW2 operands | 16-23 dz2 image | 24-31 dW1 operands) and the split of the next chunk.  The h2 / h1 images the
      // operands come from were stored under the dW0 run of the PREVIOUS tile (this phase would be bound by LDS bandwidth with
      // them: its matrix run is 2.7x shorter than on the fp32 pipe).
      f32x2 av[16], bv[16];
      auto lds_step = [&](auto S) {
        constexpr int s2 = S.value;
        if constexpr (s2 < 16) {                          // (the h2 / h1 images of this tile were stored a tile ago, see dW0)
          if constexpr (TRAIN) {                           // h2 was computed in this tile: its image first, then the operands
            if constexpr (s2 < 8) {
              static_for<4>([&](auto E) {
                constexpr int e = 4 * s2 + E.value, t = e >> 4, r = e & 15;
                lds_store<(32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc2[t][r]);
              });
            } else {
              static_for<2>([&](auto E) {
                constexpr int q = 2 * (s2 - 8) + E.value;
                av[q] = lds_load2<kImgZ + 8 * q>(rZ2);
                bv[q] = lds_load2<8 * q>(rH2);
              });
            }
          } else {
            av[s2] = lds_load2<kImgZ + 8 * s2>(rZ2);
            bv[s2] = lds_load2<8 * s2>(rH2);
          }
        } else if constexpr (s2 < 24) {
          static_for<4>([&](auto E) {
            constexpr int e = 4 * (s2 - 16) + E.value, t = e >> 4, r = e & 15;
            lds_store<(32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, d2[t][r]);
          });
        } else {
          constexpr int q = s2 - 24;
          a0p[q] = lds_load2<(2 * q) * 4>(rOp);
          a1p[q] = lds_load2<(32 * kImgStride + 2 * q) * 4>(rOp);
          b0p[q] = lds_load2<kImgB + (2 * q) * 4>(rOp);
          b1p[q] = lds_load2<kImgB + (32 * kImgStride + 2 * q) * 4>(rOp);
        }
      };
      Planes zc = split8(d2[0][0], d2[0][1], d2[0][2], d2[0][3], d2[0][4], d2[0][5], d2[0][6], d2[0][7]);
      d1[0] = 0; d1[1] = 0;
      static_for<4>([&](auto CC) {
        constexpr int cc = CC.value, nc = cc + 1;
        unsigned nh[4], nm[4], nl[4];
        const Planes& fa0 = w1p[2 * cc];
        const Planes& fa1 = w1p[2 * cc + 1];
        auto pair = [&](auto PP, u32x4 a0, u32x4 a1, u32x4 b) {
          constexpr int pp = PP.value;
          d1[0] = mfma_b(a0, b, d1[0]);
          d1[1] = mfma_b(a1, b, d1[1]);
          constexpr int first = 8 * cc + (pp < 1 ? 0 : pp < 4 ? pp + 1 : pp + 2), count = (pp == 0 || pp == 3) ? 2 : 1;
          static_for<count>([&](auto J) { lds_step(std::integral_constant<int, first + J.value>{}); });
          if constexpr (nc < 4) {
            if constexpr (pp < 4) {                       // a quarter of the next chunk's split
              constexpr int t = nc >> 1, r0 = 8 * (nc & 1) + 2 * pp;
              const float a = d2[t][r0], b2 = d2[t][r0 + 1];
              nh[pp] = pack_hi16(b2, a);
              const float ra = trunc_residual(a), rb = trunc_residual(b2);
              nm[pp] = pack_hi16(rb, ra);
              nl[pp] = pack_hi16(trunc_residual(rb), trunc_residual(ra));
            }
          }
          STEP_END();
        };
        pair(std::integral_constant<int, 0>{}, fa0.hi, fa1.hi, zc.lo);
        pair(std::integral_constant<int, 1>{}, fa0.lo, fa1.lo, zc.hi);
        pair(std::integral_constant<int, 2>{}, fa0.mid, fa1.mid, zc.mid);
        pair(std::integral_constant<int, 3>{}, fa0.hi, fa1.hi, zc.mid);
        pair(std::integral_constant<int, 4>{}, fa0.mid, fa1.mid, zc.hi);
        pair(std::integral_constant<int, 5>{}, fa0.hi, fa1.hi, zc.hi);
        if constexpr (nc < 4) {
          zc.hi = u32x4{nh[0], nh[1], nh[2], nh[3]}; zc.mid = u32x4{nm[0], nm[1], nm[2], nm[3]}; zc.lo = u32x4{nl[0], nl[1], nl[2], nl[3]};
        }
        if constexpr (cc == 1) {                           // dW2 (see the fp32 schedule below)
          lds_wait();
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            dW2acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].x, bv[q].x, dW2acc, 0, 0, 0);
            dW2acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].y, bv[q].y, dW2acc, 0, 0, 0);
          }
          STEP_END();
        }
      });
    } else {
    {
      f32x2 av[16], bv[16];
      static_for<16>([&](auto S2) {
        constexpr int s2 = S2.value;
        if constexpr (s2 == 0) { MFMA_VV_ZERO(d1[0], ft[0][0], d2[0][0]); MFMA_VV_ZERO(d1[1], ft[1][0], d2[0][0]); }
        else { MFMA_VV(d1[0], ft[0][s2], d2[0][s2]); MFMA_VV(d1[1], ft[1][s2], d2[0][s2]); }
        if constexpr (s2 < 8) {                           // h2 image (imgA), 4 stores per step
          static_for<4>([&](auto E) {
            constexpr int e = 4 * s2 + E.value, t = e >> 4, r = e & 15;
            lds_store<(32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc2[t][r]);
            if constexpr (!RECOMPUTE)                       // and the h1 image (imgB) that R2 would have carried
              lds_store<kImgB + (32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, acc1[t][r]);
          });
        } else {                                          // then the dW2 operands
          static_for<2>([&](auto E) {
            constexpr int q = 2 * (s2 - 8) + E.value;
            av[q] = lds_load2<kImgZ + 8 * q>(rZ2);
            bv[q] = lds_load2<8 * q>(rH2);
          });
        }
        STEP_END();
      });
      lds_wait();
      // dW2 (3-4 x 64 outputs) on v_mfma_f32_4x4x1_16b: 16 blocks of 4x4, one pixel per instruction, 8 cycles — instead
      // of two 32x32 tiles that would be 90 % zeros.  Lane j: B = h2[j][px] (its own image row), A = dz3[j & 3][px].
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dW2acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].x, bv[q].x, dW2acc, 0, 0, 0);
        dW2acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[q].y, bv[q].y, dW2acc, 0, 0, 0);
      }
      STEP_END();
    }
    static_for<16>([&](auto S2) {
      constexpr int s2 = 16 + S2.value, k = S2.value;
      MFMA_VV(d1[0], ft[0][s2], d2[1][k]);
      MFMA_VV(d1[1], ft[1][s2], d2[1][k]);
      if constexpr (k < 8) {                              // dz2 image: 4 stores per step
        static_for<4>([&](auto E) {
          constexpr int e = 4 * k + E.value, t = e >> 4, r = e & 15;
          lds_store<(32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, d2[t][r]);
        });
      } else {                                            // dW1 operands (after ALL dz2 stores) and h1 read-back
        constexpr int q = k - 8;
        a0p[q] = lds_load2<(2 * q) * 4>(rOp);
        a1p[q] = lds_load2<(32 * kImgStride + 2 * q) * 4>(rOp);
        b0p[q] = lds_load2<kImgB + (2 * q) * 4>(rOp);
        b1p[q] = lds_load2<kImgB + (32 * kImgStride + 2 * q) * 4>(rOp);
        if constexpr (RECOMPUTE)
          static_for<4>([&](auto E) {
            constexpr int e = 4 * q + E.value, t = e >> 4, r = e & 15;
            h1v[t][r] = lds_load1<kImgB + (32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile);
          });
      }
      STEP_END();
    });
    MFMA_DRAIN(d1[0], d1[1]);
    }
    lds_wait();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) d1[t][r] = hidden_dsel<LEAKY>(RECOMPUTE ? h1v[t][r] : acc1[t][r], d1[t][r]);
    {   // db1[32a + i] += sum over this half's 16 pixels of dz2 (the A operands ARE dz2^T): 2 registers instead of 32
      f32x2 s0 = a0p[0], s1 = a1p[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) { s0 += a0p[q]; s1 += a1p[q]; }
      db1acc[0] += s0.x + s0.y; db1acc[1] += s1.x + s1.y;
    }
    if (!RECOMPUTE && !TRAIN) {                            // h2 is dead: fetch the next tile's (needed first, at its mask)
      STEP_END();
      hidden_load(hidden, tile + gridDim.x, ntiles, 1, hoff, acc2);
    }
    pin_acc();
    STEP_END();
    STAMP(5);
    // ---- dW1 += dz2^T h1.  Underneath: the dz1 image (over dz2, already in registers), dW0 operands, W0^T fragments
    f32x2 c0p[8], c1p[8], xp[TX][8];
    float fx[TX][32];
    static_for<8>([&](auto Q) {
      constexpr int q = Q.value;
      if constexpr (HYB && !TRAIN && q == 0) hidden_load(hidden, tile + gridDim.x, ntiles, 0, hoff, acc1);   // h1 is dead: the next tile's
      dW1acc[0][0] = MFMA(a0p[q].x, b0p[q].x, dW1acc[0][0]);
      dW1acc[0][1] = MFMA(a0p[q].x, b1p[q].x, dW1acc[0][1]);
      dW1acc[1][0] = MFMA(a1p[q].x, b0p[q].x, dW1acc[1][0]);
      dW1acc[1][1] = MFMA(a1p[q].x, b1p[q].x, dW1acc[1][1]);
      dW1acc[0][0] = MFMA(a0p[q].y, b0p[q].y, dW1acc[0][0]);
      dW1acc[0][1] = MFMA(a0p[q].y, b1p[q].y, dW1acc[0][1]);
      dW1acc[1][0] = MFMA(a1p[q].y, b0p[q].y, dW1acc[1][0]);
      dW1acc[1][1] = MFMA(a1p[q].y, b1p[q].y, dW1acc[1][1]);
      if constexpr (q < 4) {
        static_for<8>([&](auto E) {
          constexpr int e = 8 * q + E.value, t = e >> 4, r = e & 15;
          lds_store<(32 * t + (r & 3) + 8 * (r >> 2)) * kImgStride * 4>(wTile, d1[t][r]);                 // dz1 image (imgA)
        });
        if constexpr (!BL::kDedicatedX)                    // 64-wide input: its image shares imgB (h1 is dead by now)
          static_for<S0 / 4>([&](auto E) { constexpr int sx = q * (S0 / 4) + E.value; lds_store<kImgX + sx * kImgStride * 4>(wX, xr[sx]); });
      } else {
        static_for<2>([&](auto E) {
          constexpr int qq = 2 * (q - 4) + E.value;
          c0p[qq] = lds_load2<(2 * qq) * 4>(rOp);
          c1p[qq] = lds_load2<(32 * kImgStride + 2 * qq) * 4>(rOp);
          xp[0][qq] = lds_load2<kImgX + (2 * qq) * 4>(rOp);
          if constexpr (TX > 1) xp[TX - 1][qq] = lds_load2<kImgX + (32 * kImgStride + 2 * qq) * 4>(rOp);
        });
        if constexpr (!HYB) {
#pragma unroll
          for (int j = 0; j < 8; ++j) fx[0][8 * (q - 4) + j] = A0T[(8 * (q - 4) + j) * 64 + lane];
        }
        if (kCarryF0 && TX > 1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) fx[TX - 1][8 * (q - 4) + j] = A0T[((TX - 1) * 32 + 8 * (q - 4) + j) * 64 + lane];
        }
      }
      STEP_END();
    });
    if (!BL::kDedicatedX) fetch_x(tile + gridDim.x);
    lds_wait();
    pin_acc();
    STAMP(4);
    // ---- dW0 += dz1^T x, with the next tile's W0 fragments underneath
    static_for<8>([&](auto Q) {
      constexpr int q = Q.value;
#pragma unroll
      for (int tx = 0; tx < TX; ++tx) {
        dW0acc[0][tx] = MFMA(c0p[q].x, xp[tx][q].x, dW0acc[0][tx]);
        dW0acc[1][tx] = MFMA(c1p[q].x, xp[tx][q].x, dW0acc[1][tx]);
        dW0acc[0][tx] = MFMA(c0p[q].y, xp[tx][q].y, dW0acc[0][tx]);
        dW0acc[1][tx] = MFMA(c1p[q].y, xp[tx][q].y, dW0acc[1][tx]);
      }
      if constexpr (!RECOMPUTE && !HYB && q == 0) hidden_load(hidden, tile + gridDim.x, ntiles, 0, hoff, acc1);   // next tile's h1
      if constexpr (HYB) {
        // the NEXT tile's h2 -> image A and h1 -> image B (both idle from here on; loaded under d1 / at the top of dW1): their
        // 64 stores need a long fp32 run above them — under the bf16 runs of d1 or dX they cost 760 cycles per tile
        if constexpr (!TRAIN) {
          store_hidden_images(std::integral_constant<int, 2 * q>{});
          store_hidden_images(std::integral_constant<int, 2 * q + 1>{});
        }
        if constexpr (q >= 2 && q < 6) w0p[q - 2] = load_planes(w0t + (q - 2) * 3 * 64);
      }
      if (RECOMPUTE && kCarryF0) {
#pragma unroll
        for (int j = 0; j < 2 * S0 / 8; ++j) {
          const int e = q * (2 * S0 / 8) + j;
          f0[e / S0][e % S0] = A0[e * 64 + lane];
        }
      } else if (!kCarryF0 && TX > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fx[TX - 1][4 * q + j] = A0T[((TX - 1) * 32 + 4 * q + j) * 64 + lane];
      }
      STEP_END();
    });
    pin_acc();
    STAMP(6);
    // ---- d enc^T = W0^T dz1^T ; regs 4g..4g+3 are 4 consecutive input features
    f32x16 dxv[TX];
    if constexpr (HYB) {
      // d enc^T = W0^T dz1^T on the bf16 pipe (k-chunks as above); the split of chunk cc + 1 issues under the products of chunk cc
      f32x16 acc;
      acc = 0;
      Planes zc = split8(d1[0][0], d1[0][1], d1[0][2], d1[0][3], d1[0][4], d1[0][5], d1[0][6], d1[0][7]);
      static_for<4>([&](auto CC) {
        constexpr int cc = CC.value, nc = cc + 1;
        Planes nz;
        const Planes& fa = w0p[cc];
        if constexpr (nc < 4) {
          constexpr int t = nc >> 1, r0 = 8 * (nc & 1);
          nz = split8(d1[t][r0], d1[t][r0 + 1], d1[t][r0 + 2], d1[t][r0 + 3], d1[t][r0 + 4], d1[t][r0 + 5], d1[t][r0 + 6], d1[t][r0 + 7]);
        }
        // six cross terms, smallest first
        acc = mfma_b(fa.hi, zc.lo, acc);
        acc = mfma_b(fa.lo, zc.hi, acc);
        acc = mfma_b(fa.mid, zc.mid, acc);
        acc = mfma_b(fa.hi, zc.mid, acc);
        acc = mfma_b(fa.mid, zc.hi, acc);
        acc = mfma_b(fa.hi, zc.hi, acc);
        STEP_END();
        if constexpr (nc < 4) zc = nz;
      });
      dxv[0] = acc;
    } else {
#pragma unroll
    for (int tx = 0; tx < TX; ++tx) {
      MFMA_VV_ZERO(dxv[tx], fx[tx][0], d1[0][0]);
#pragma unroll
      for (int s2 = 1; s2 < 32; ++s2) {
        MFMA_VV(dxv[tx], fx[tx][s2], d1[s2 >> 4][s2 & 15]);
        if (!RECOMPUTE && tx == TX - 1) {                  // the next tile's W1^T fragments ride under the last dX chain
          ft[0][s2] = A1T[(0 * 32 + s2) * 64 + lane];
          ft[1][s2] = A1T[(1 * 32 + s2) * 64 + lane];
          if (s2 == 1) { ft[0][0] = A1T[lane]; ft[1][0] = A1T[32 * 64 + lane]; }
        }
      }
    }
    if constexpr (TX > 1) MFMA_DRAIN(dxv[0], dxv[TX - 1]);
    else MFMA_DRAIN1(dxv[0]);
    }
    STEP_END();
#pragma unroll
    for (int tx = 0; tx < TX; ++tx)
#pragma unroll
      for (int r = 0; r < 16; ++r) {      // |x| bit patterns order like the values, NaN above inf: it sticks
        const unsigned a = __float_as_uint(dxv[tx][r]) & 0x7fffffffu;
        dxmax = a > dxmax ? a : dxmax;
      }
    {
      f32x2 s0 = c0p[0], s1 = c1p[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) { s0 += c0p[q]; s1 += c1p[q]; }
      db0acc[0] += s0.x + s0.y; db0acc[1] += s1.x + s1.y;
    }
    if constexpr (!TRAIN) make_dz3();                    // of tile t+1 (y, dy were requested half a tile ago)
    STEP_END();
    {
      const int64_t pt = bwd_tile(tile, ntiles);
      int64_t rem = P - pt * 128;
      rem = rem > 128 ? 128 : rem;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dX + pt * 128 * in_dim, 0, (int)rem * in_dim * 4, 0x00020000);
      const unsigned base = (unsigned)((wave * 32 + i) * in_dim) * 4u;
#pragma unroll
      for (int tx = 0; tx < TX; ++tx)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * tx + 8 * g + 4 * h;
          if (EXACT) {
            if (32 * tx + 8 * g + 8 <= KIN) {          // KIN = 16: only g = 0, 1 are real input features
              u32x4 v = {__float_as_uint(dxv[tx][4 * g]), __float_as_uint(dxv[tx][4 * g + 1]), __float_as_uint(dxv[tx][4 * g + 2]),
                         __float_as_uint(dxv[tx][4 * g + 3])};
              __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + 4u * col, 0, 0);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dxv[tx][4 * g + q]), rs, col + q < in_dim ? base + 4u * (col + q) : 0x40000000u, 0, 0);
          }
        }
    }
    if constexpr (kClears) {
      for (int z = 0; z < zper; ++z, zlo += kDecThreads)
        if (zlo < zhi) zero_fill[zlo] = float4{0.f, 0.f, 0.f, 0.f};
    }
    pin_acc();
    STEP_END();
    STAMP(7);
  }

#if defined(GNGF_STAMPS)
  if (blockIdx.x == 7 && threadIdx.x == 0)
    for (int k = 0; k < 10; ++k) g_stamps[k] = ph[k];
  if (threadIdx.x == 0) g_blocktime[1][blockIdx.x & 255][1] = __builtin_readcyclecounter();
#endif
  // non-negative floats (and NaN > inf) order like their bit patterns
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned ov = (unsigned)__shfl_xor((int)dxmax, o, 64); dxmax = ov > dxmax ? ov : dxmax; }
  // ---- wave accumulators -> workgroup slab.  No LDS float atomics (ds_add_f32 costs ~190 cycles per wave-instruction
  // on gfx950): every wave stores its tiles into its own LDS region, bias partials are reduced over the 32 pixel lanes
  // with shuffles, then the four regions are summed into ONE plain-store slab per workgroup.
  __syncthreads();                                       // every wave is done with the images and fragments
  float* region = smem + wave * nslab;                   // 4 regions; the fragment + image areas are dead now
  float* sW0 = region;
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
      for (int tj = 0; tj < 2; ++tj) sW1[row * kH + 32 * tj + i] = dW1acc[ti][tj][r];
#pragma unroll
      for (int tx = 0; tx < TX; ++tx)
        if (32 * tx + i < in_dim) sW0[row * in_dim + 32 * tx + i] = dW0acc[ti][tx][r];
    }
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {                       // the two lane halves hold the two 16-pixel halves of every tile
    const float v0 = db0acc[ti] + __shfl_xor(db0acc[ti], 32, 64), v1 = db1acc[ti] + __shfl_xor(db1acc[ti], 32, 64);
    if (h == 0) { sb0[32 * ti + i] = v0; sb1[32 * ti + i] = v1; }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (c < out_dim) sW2[c * kH + lane] = dW2acc[c];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = h == 0 ? db2acc[c] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0 && c < out_dim) sb2[c] = v;
  }
  if (lane == 0) region[nslab - 1] = __uint_as_float(dxmax);
  __syncthreads();
  float* out = slabs + (int64_t)blockIdx.x * nslab;
  for (int e = threadIdx.x; e < nslab - 1; e += kDecThreads)
    out[e] = (smem[e] + smem[nslab + e]) + (smem[2 * nslab + e] + smem[3 * nslab + e]);
  if (threadIdx.x == 0) {
    unsigned mx = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const unsigned v = __float_as_uint(smem[w * nslab + nslab - 1]); mx = v > mx ? v : mx; }
    out[nslab - 1] = __uint_as_float(mx);
    atomicMax(&g_bwd_span[1], (unsigned long long)wall_clock64());
  }
}

// sums the per-workgroup slabs and writes the six gradient tensors.  64 elements x 16 slab-groups per block
// (body: decoder_reduce_block in gngf_common.h — the tiled encoder backward can run it in extra workgroups of its own launch).
__global__ void __launch_bounds__(1024)
decoder_reduce_kernel(const float* __restrict__ slabs, int nslabs, int nslab, int in_dim, int out_dim,
                      float* __restrict__ dW0, float* __restrict__ db0, float* __restrict__ dW1, float* __restrict__ db1,
                      float* __restrict__ dW2, float* __restrict__ db2, float* __restrict__ absmax,
                      const float* __restrict__ promised, const float* __restrict__ arrived) {
  decoder_reduce_block(blockIdx.x, slabs, nslabs, nslab, in_dim, out_dim, dW0, db0, dW1, db1, dW2, db2, absmax, promised, arrived);
}

template <int KIN>
static size_t bwd_smem_bytes(int in_dim, int out_dim) {
  using FF = FwdFrags<KIN>;
  constexpr int TX = (KIN + 31) / 32;
  const int img = 4 * BwdLds<KIN>::kWaveFloats, slab = slab_size(in_dim, out_dim);
  const size_t main_loop = (size_t)(FF::kA0 + FF::kA1 + 2 * 2 * 64 + 2 * 32 * 64 + TX * 32 * 64 + 2 * kH + img);
  const size_t epilogue = (size_t)4 * slab;            // four per-wave regions over the dead fragment/image areas
  return sizeof(float) * (main_loop > epilogue ? main_loop : epilogue);
}
// the hybrid variant keeps W1^T / W0^T as bf16 planes (8 + 4 fragments of 3 KB) instead of the fp32 fragment images
static size_t bwd_smem_bytes_hybrid(int out_dim) {
  using FF = FwdFrags<32>;
  const int img = 4 * BwdLds<32>::kWaveFloats, slab = slab_size(32, out_dim);
  const size_t main_loop = (size_t)(FF::kA0 + FF::kA1 + 2 * 2 * 64 + 12 * 3 * 64 * 4 + 2 * kH + img);
  const size_t epilogue = (size_t)4 * slab;
  return sizeof(float) * (main_loop > epilogue ? main_loop : epilogue);
}
// 1 (default): gngf_decoder_bwd at in_dim == 32 with the saved hidden layers runs the hybrid kernel (dh1 and d enc on the
// bf16 pipe with the exact three-way split, weight gradients on the fp32 pipe); 0: all products on the fp32 pipe
static int g_decoder_bwd_hybrid = 1;

// 1: in_dim 32 / 64 run on the split-bf16 kernels (decoder_split.inc); 0: everything on the fp32 matrix pipe
static int g_decoder_split = 0;

}  // namespace gngf

using namespace gngf;

// Switches the decoder between the fp32-MFMA kernels and the split-bf16 ones (same results to fp32 rounding); returns the
// previous setting.  With the split kernels the hidden-layer buffer is neither written nor read.
// Returns -1 (and changes nothing) when the library was built without them (the default: make SPLIT=1 builds them).
extern "C" int gngf_set_decoder_split_bf16(int on) {
#if defined(GNGF_DECODER_SPLIT_KERNELS)
  const int prev = g_decoder_split;
  g_decoder_split = on ? 1 : 0;
  return prev;
#else
  (void)on;
  return -1;
#endif
}
// Switch for the hybrid backward kernel (see g_decoder_bwd_hybrid); returns the previous setting.
extern "C" int gngf_set_decoder_bwd_hybrid(int on) {
  const int prev = g_decoder_bwd_hybrid;
  g_decoder_bwd_hybrid = on ? 1 : 0;
  return prev;
}
static bool decoder_split_applies(int in_dim) { return g_decoder_split && in_dim == 32; }

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

// floats of the saved-activation buffer for P pixels (whole 128-pixel tiles)
extern "C" int64_t gngf_decoder_hidden_floats(int64_t P) { return ((P + 127) / 128) * (int64_t)kHiddenTileFloats; }

// rgb (P,out_dim) = decoder(enc (P,in_dim)); hidden widths fixed at 64/64, in_dim <= 64, out_dim <= 4.
// hidden (optional): gngf_decoder_hidden_floats(P) floats that receive the activated hidden layers for gngf_decoder_bwd.
extern "C" int gngf_decoder_fwd(const float* enc, const float* W0, const float* b0, const float* W1, const float* b1,
                                const float* W2, const float* b2, float* rgb, float* hidden, int64_t P, int in_dim, int out_dim,
                                int leaky, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && in_dim > 0 && in_dim <= 64 && out_dim > 0 && out_dim <= 4);
  if (P == 0) return 0;
  GNGF_CHECK_ARG(enc && W0 && b0 && W1 && b1 && W2 && b2 && rgb);
  const int64_t tiles = (P + 127) / 128;
  const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);       // one persistent workgroup per CU
  const size_t smem = sizeof(float) * (size_t)raw_offsets(in_dim).total;
#if defined(GNGF_DECODER_SPLIT_KERNELS)
  if (decoder_split_applies(in_dim)) {
    hipStream_t s = as_stream(stream);
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);     // two persistent workgroups per CU
    const size_t smem = sizeof(float) * (size_t)((in_dim == 32 ? SplitFwd<32>::kFragWords : SplitFwd<64>::kFragWords) + SplitFwd<32>::kBiasWords + raw_offsets(in_dim).total);
    if (in_dim == 32) {
      if (leaky) decoder_fwd_split_kernel<32, true><<<dim3(grid), dim3(kDecThreads), smem, s>>>(enc, W0, b0, W1, b1, W2, b2, rgb, P, out_dim);
      else decoder_fwd_split_kernel<32, false><<<dim3(grid), dim3(kDecThreads), smem, s>>>(enc, W0, b0, W1, b1, W2, b2, rgb, P, out_dim);
    } else {
      if (leaky) decoder_fwd_split_kernel<64, true><<<dim3(grid), dim3(kDecThreads), smem, s>>>(enc, W0, b0, W1, b1, W2, b2, rgb, P, out_dim);
      else decoder_fwd_split_kernel<64, false><<<dim3(grid), dim3(kDecThreads), smem, s>>>(enc, W0, b0, W1, b1, W2, b2, rgb, P, out_dim);
    }
    GNGF_RETURN_LAUNCH();
  }
#endif
  DISPATCH_KIN(in_dim, {
    using Kern = void (*)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                          float*, int64_t, int, int);
    const bool exact = in_dim == kKIN;
    Kern fn;
    if (hidden)
      fn = leaky ? (exact ? decoder_fwd_kernel<kKIN, true, true, true> : decoder_fwd_kernel<kKIN, true, false, true>)
                 : (exact ? decoder_fwd_kernel<kKIN, false, true, true> : decoder_fwd_kernel<kKIN, false, false, true>);
    else
      fn = leaky ? (exact ? decoder_fwd_kernel<kKIN, true, true, false> : decoder_fwd_kernel<kKIN, true, false, false>)
                 : (exact ? decoder_fwd_kernel<kKIN, false, true, false> : decoder_fwd_kernel<kKIN, false, false, false>);
    fn<<<dim3(grid), dim3(kDecThreads), smem, as_stream(stream)>>>(enc, W0, b0, W1, b1, W2, b2, rgb, hidden, P, in_dim, out_dim);
  });
  GNGF_RETURN_LAUNCH();
}

// d enc (P,in_dim) and the six parameter gradients (each WRITTEN, not accumulated).  rgb = the forward output.
// slabs: workspace of gngf_decoder_bwd_slabs(P) * gngf_decoder_slab_floats(in_dim, out_dim) floats.
// hidden (optional): the buffer gngf_decoder_fwd filled for the SAME enc / weights; NULL: the hidden layers are recomputed.
extern "C" int gngf_decoder_bwd(const float* enc, const float* rgb, const float* drgb, const float* target, const float* gloss,
                                const float* W0, const float* b0,
                                const float* W1, const float* b1, const float* W2, float* denc, float* dW0, float* db0,
                                float* dW1, float* db1, float* dW2, float* db2, float* slabs, float* denc_absmax,
                                const float* hidden, float* zero_fill, int64_t zero_floats, int64_t P, int in_dim, int out_dim,
                                int leaky, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && in_dim > 0 && in_dim <= 64 && out_dim > 0 && out_dim <= 4);
  GNGF_CHECK_ARG(!zero_fill || (zero_floats >= 0 && (zero_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_fill) & 15) == 0));
  float4* zf = reinterpret_cast<float4*>(zero_fill);
  const int64_t zv = zero_fill ? zero_floats / 4 : 0;
  const bool reduce_here = dW0 || db0 || dW1 || db1 || dW2 || db2;       // all NULL: the caller runs gngf_decoder_reduce
  GNGF_CHECK_ARG(slabs && (!reduce_here || (dW0 && db0 && dW1 && db1 && dW2 && db2)));
  const int nslab = slab_size(in_dim, out_dim);
  hipStream_t s = as_stream(stream);
  const int nslabs = gngf_decoder_bwd_slabs(P);
  if (P == 0) {
    hipError_t e = zero_async(slabs, sizeof(float) * (size_t)nslab, s);
    if (e != hipSuccess) return (int)e;
    if (zero_fill && zero_floats > 0) {
      e = zero_async(zero_fill, sizeof(float) * (size_t)zero_floats, s);
      if (e != hipSuccess) return (int)e;
    }
  } else {
    GNGF_CHECK_ARG(enc && rgb && (target ? gloss != nullptr : drgb != nullptr) && W0 && b0 && W1 && b1 && W2 && denc);
    // only decoder_bwd_kernel<64, ., EXACT = true, ...> clears on the way (kClears); every other kernel of this entry gets a memset
    if (zero_fill && zero_floats > 0 && in_dim != 64) {
      hipError_t e = zero_async(zero_fill, sizeof(float) * (size_t)zero_floats, s);
      if (e != hipSuccess) return (int)e;
    }
#if defined(GNGF_DECODER_SPLIT_KERNELS)
    if (decoder_split_applies(in_dim) && in_dim == 32) {
      const size_t main_loop = SplitBwd<32>::kMainBytes, epilogue = sizeof(float) * 4 * (size_t)nslab;
      const size_t smem = main_loop > epilogue ? main_loop : epilogue;
      auto fn = leaky ? decoder_bwd_split_kernel<32, true> : decoder_bwd_split_kernel<32, false>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      fn<<<dim3((unsigned)nslabs), dim3(kDecThreads), smem, s>>>(enc, rgb, drgb, W0, b0, W1, b1, W2, denc, slabs, P, out_dim, target, gloss);
    } else
#endif
    if (g_decoder_bwd_hybrid && in_dim == 32 && hidden) {
      const size_t smem = bwd_smem_bytes_hybrid(out_dim);
      auto fn = leaky ? decoder_bwd_kernel<32, true, true, false, true> : decoder_bwd_kernel<32, false, true, false, true>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      fn<<<dim3((unsigned)nslabs), dim3(kDecThreads), smem, s>>>(enc, rgb, drgb, W0, b0, W1, b1, W2, denc, slabs, hidden, P, in_dim, out_dim,
                                                                 target, gloss, nullptr, zf, zv);
    } else
    DISPATCH_KIN(in_dim, {
      const size_t smem = bwd_smem_bytes<kKIN>(in_dim, out_dim);
      using Kern = void (*)(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                            const float*, float*, float*, const float*, int64_t, int, int, const float*, const float*, const float*,
                            float4*, int64_t);
      const bool exact = in_dim == kKIN;
      Kern fn;
      if (hidden)
        fn = leaky ? (exact ? decoder_bwd_kernel<kKIN, true, true, false> : decoder_bwd_kernel<kKIN, true, false, false>)
                   : (exact ? decoder_bwd_kernel<kKIN, false, true, false> : decoder_bwd_kernel<kKIN, false, false, false>);
      else
        fn = leaky ? (exact ? decoder_bwd_kernel<kKIN, true, true, true> : decoder_bwd_kernel<kKIN, true, false, true>)
                   : (exact ? decoder_bwd_kernel<kKIN, false, true, true> : decoder_bwd_kernel<kKIN, false, false, true>);
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      fn<<<dim3((unsigned)nslabs), dim3(kDecThreads), smem, s>>>(enc, rgb, drgb, W0, b0, W1, b1, W2, denc, slabs, hidden, P, in_dim, out_dim,
                                                                 target, gloss, nullptr, zf, zv);
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (reduce_here)
    decoder_reduce_kernel<<<dim3((unsigned)((nslab + 63) / 64)), dim3(1024), 0, s>>>(slabs, nslabs, nslab, in_dim, out_dim, dW0,
                                                                                     db0, dW1, db1, dW2, db2, denc_absmax, nullptr, nullptr);
  GNGF_RETURN_LAUNCH();
}

// Forward AND backward of the decoder for a training step whose loss is MSELoss(rgb, target) with a known upstream gradient
// *gloss (device scalar): rgb (P,out_dim) is written, d enc and the parameter-gradient slabs as by gngf_decoder_bwd with
// target / gloss — in ONE launch, without the hidden-layer buffer.  in_dim == 32 only (returns hipErrorInvalidValue otherwise:
// the caller falls back to gngf_decoder_fwd + gngf_decoder_bwd).  Gradient pointers / slabs / denc_absmax as gngf_decoder_bwd.
// zero_fill (optional; zero_floats floats, a multiple of 4, 16-byte aligned): cleared by the kernel on the way (see above) —
// complete when the launch is, i.e. before anything later on the stream.
extern "C" int gngf_decoder_train(const float* enc, const float* target, const float* gloss, const float* W0, const float* b0,
                                  const float* W1, const float* b1, const float* W2, const float* b2, float* rgb, float* denc,
                                  float* dW0, float* db0, float* dW1, float* db1, float* dW2, float* db2, float* slabs,
                                  float* denc_absmax, float* zero_fill, int64_t zero_floats, int64_t P, int in_dim, int out_dim,
                                  int leaky, void* stream) {
  GNGF_CHECK_ARG(P > 0 && in_dim == 32 && out_dim > 0 && out_dim <= 4);
  GNGF_CHECK_ARG(!zero_fill || (zero_floats >= 0 && (zero_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_fill) & 15) == 0));
  const bool reduce_here = dW0 || db0 || dW1 || db1 || dW2 || db2;
  GNGF_CHECK_ARG(slabs && (!reduce_here || (dW0 && db0 && dW1 && db1 && dW2 && db2)));
  GNGF_CHECK_ARG(enc && target && gloss && W0 && b0 && W1 && b1 && W2 && b2 && rgb && denc);
  const int nslab = slab_size(in_dim, out_dim);
  hipStream_t s = as_stream(stream);
  const int nslabs = gngf_decoder_bwd_slabs(P);
  const size_t smem = bwd_smem_bytes_hybrid(out_dim);
  auto fn = leaky ? decoder_bwd_kernel<32, true, true, false, true, true> : decoder_bwd_kernel<32, false, true, false, true, true>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (e != hipSuccess) return (int)e;
  fn<<<dim3((unsigned)nslabs), dim3(kDecThreads), smem, s>>>(enc, rgb, nullptr, W0, b0, W1, b1, W2, denc, slabs, nullptr, P, in_dim, out_dim,
                                                             target, gloss, b2, reinterpret_cast<float4*>(zero_fill),
                                                             zero_fill ? zero_floats / 4 : 0);
  e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (reduce_here)
    decoder_reduce_kernel<<<dim3((unsigned)((nslab + 63) / 64)), dim3(1024), 0, s>>>(slabs, nslabs, nslab, in_dim, out_dim, dW0,
                                                                                     db0, dW1, db1, dW2, db2, denc_absmax, nullptr, nullptr);
  GNGF_RETURN_LAUNCH();
}

// Second half of gngf_decoder_bwd when it was called without gradient pointers: slabs -> the six gradients (+ max |denc|).
// Separate so that it can run on another stream beside the encoder backward, which only needs the per-slab maxima
// (the last word of every slab: slabs[s * gngf_decoder_slab_floats() + gngf_decoder_slab_floats() - 1]).
// gloss_promised / gloss_arrived (device scalars, both or neither): the loss gradient gngf_decoder_train was given and the one
// autograd delivered afterwards; if they differ (relative 1e-6) every gradient is written as NaN instead (see promise_broken).
extern "C" int gngf_decoder_reduce(const float* slabs, float* dW0, float* db0, float* dW1, float* db1, float* dW2, float* db2,
                                   float* denc_absmax, const float* gloss_promised, const float* gloss_arrived, int64_t P,
                                   int in_dim, int out_dim, void* stream) {
  GNGF_CHECK_ARG(P >= 0 && in_dim > 0 && in_dim <= 64 && out_dim > 0 && out_dim <= 4);
  GNGF_CHECK_ARG(slabs && dW0 && db0 && dW1 && db1 && dW2 && db2 && (!gloss_promised == !gloss_arrived));
  const int nslab = slab_size(in_dim, out_dim);
  decoder_reduce_kernel<<<dim3((unsigned)((nslab + 63) / 64)), dim3(1024), 0, as_stream(stream)>>>(
      slabs, gngf_decoder_bwd_slabs(P), nslab, in_dim, out_dim, dW0, db0, dW1, db1, dW2, db2, denc_absmax, gloss_promised,
      gloss_arrived);
  GNGF_RETURN_LAUNCH();
}

// Duration of the most recent gngf_decoder_bwd main kernel on the device's constant 100 MHz clock (first workgroup start
// to last workgroup end), in nanoseconds.  Synchronises the device.
extern "C" int gngf_decoder_bwd_last_span_ns(double* ns) {
  GNGF_CHECK_ARG(ns);
  unsigned long long span[2] = {0ull, 0ull};
  hipError_t e = hipMemcpyFromSymbol(span, HIP_SYMBOL(gngf::g_bwd_span), sizeof(span));
  if (e != hipSuccess) return (int)e;
  *ns = span[1] > span[0] ? (double)(span[1] - span[0]) * 10.0 : 0.0;
  return 0;
}

#if defined(GNGF_STAMPS)
extern "C" int gngf_debug_read_fwd_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(gngf::g_fstamps), 16 * sizeof(unsigned long long));
}
extern "C" int gngf_debug_read_blocktime(unsigned long long* host1024) {
  return (int)hipMemcpyFromSymbol(host1024, HIP_SYMBOL(gngf::g_blocktime), 2 * 256 * 2 * sizeof(unsigned long long));
}
extern "C" int gngf_debug_read_stamps(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(gngf::g_stamps), 16 * sizeof(unsigned long long));
}
#endif
