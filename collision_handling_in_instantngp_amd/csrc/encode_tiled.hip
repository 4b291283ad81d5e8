// Tiled / LDS-privatised form of the encoder hot path — the fast path whenever a level's vertex grid is
// small enough to stage (all 16 levels at N_max = 512, T = 2^19).
//
// The indices of BOTH index sources depend only on (level, vertex): the spatial hash by definition
// (reference models.py:504-528) and the GNGF top-K because the HPD input is the integer vertex alone
// (models.py:416-418).  So the work splits into
//   vertex stage  G_l[gy][gx][f] = E_l[hash(gx,gy)][f]            or  sum_k w_k(v) E_l[idx_k(v)][f]
//                 (once per DISTINCT vertex: ~0.7 M rows instead of 67 M per-instance gathers at P = 2^20)
//   pixel stage   enc[p][l][f] = sum_v c_v(p,l) G_l[cell(p,l)+v][f]   — dense-grid bilinear interpolation,
// and backward mirrors it: the pixel stage accumulates dG, the vertex stage scatters dG into dE (and dL/dw).
//
// Pixel stage layout for gfx950: pixels are binned by spatial tile (2^tile_shift tiles per axis); one
// 256-thread workgroup owns one (tile, chunk of pixels) work item, stages the tile's sub-grid of EVERY level in
// LDS (~10 KB), and then runs one lane per (pixel, level): 4 x ds_read_b64 forward, 8 x ds_add_f32 backward.
// The 16 level-lanes of a pixel read / write one contiguous 128-byte row of enc / d enc.  Gradients leave the
// workgroup once per touched vertex instead of once per (pixel, corner): the 134 M global float atomics of the
// direct form become ~2 M row-contiguous ones.
#include "gngf_common.h"

namespace gngf {

#ifndef GNGF_TBF
#define GNGF_TBF 512
#endif
#ifndef GNGF_TBB
#define GNGF_TBB 1024
#endif
constexpr int kTBF = GNGF_TBF;  // pixel-stage workgroup, forward
constexpr int kTB = GNGF_TBB;   // pixel-stage workgroup, backward
#ifndef GNGF_BIN_THREADS
#define GNGF_BIN_THREADS 1024
#endif
constexpr int kBinThreads = GNGF_BIN_THREADS;
constexpr int kBinU = 8;          // pixels per thread and trip in the binning kernels

__device__ __forceinline__ int g_max0(int v) { return v < 0 ? 0 : v; }

__device__ __forceinline__ int tile_of(float x, float y, int tile_shift) {
  const int TS = 1 << tile_shift;
  int tx = (int)(x * (float)TS), ty = (int)(y * (float)TS);      // exact: TS is a power of two
  tx = tx < 0 ? 0 : (tx >= TS ? TS - 1 : tx);
  ty = ty < 0 ? 0 : (ty >= TS ? TS - 1 : ty);
  return (ty << tile_shift) | tx;
}

// ---------------------------------------------------------------------------------------------- binning
// K1: per-block histogram over a contiguous pixel range -> blockhist[tile][block]
// (NT = threads of the calling workgroup: the binning kernels' own 1024, or the 512 of the pixel-stage forward when the count of
// the NEXT batch rides on its launch — the pixel -> block partition depends on per_block only, not on the thread count)
template <int NT = kBinThreads>
__device__ __forceinline__ void bin_count_body(int blk, const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift,
                                               int NB, int32_t* __restrict__ blockhist, int* hist, int32_t* __restrict__ tot_atomic = nullptr) {
  const int ntiles = 1 << (2 * tile_shift);
  for (int i = threadIdx.x; i < ntiles; i += NT) hist[i] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blk * per_block;
  const int64_t hi = lo + per_block < P ? lo + per_block : P;
  // kBinU pixels per thread and trip, loads issued together: one pixel per trip made the block's time the SUM of eight memory
  // round trips (one workgroup per CU: there is nobody else to hide them)
  for (int64_t p0 = lo + threadIdx.x; p0 < hi; p0 += (int64_t)kBinU * NT) {
    float2 c[kBinU];
#pragma unroll
    for (int u = 0; u < kBinU; ++u) { const int64_t p = p0 + (int64_t)u * NT; c[u] = xy[p < hi ? p : hi - 1]; }
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      if (p0 + (int64_t)u * NT < hi) atomicAdd(&hist[tile_of(c[u].x, c[u].y, tile_shift)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ntiles; i += NT) {
    const int c = hist[i];
    blockhist[(int64_t)i * NB + blk] = c;
    if (tot_atomic && c) atomicAdd(tot_atomic + i, c);      // two-launch binning: the tile totals meet in global atomics (bin_scatter2_kernel)
  }
}

__global__ void __launch_bounds__(kBinThreads)
bin_count_kernel(const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift, int NB,
                 int32_t* __restrict__ blockhist) {
  extern __shared__ int hist[];
  bin_count_body((int)blockIdx.x, xy, P, per_block, tile_shift, NB, blockhist, hist);
}

// K2a: one WAVE per tile row of blockhist [tile][NB] (NB <= 512: 8 consecutive entries per lane): exclusive scan inside the
// row, row total to tot[tile].  grid = ceil(ntiles / 4) blocks of 4 waves.
constexpr int kBinMaxBlocks = 512;
__global__ void __launch_bounds__(256)
bin_rowscan_kernel(int32_t* __restrict__ blockhist, int NB, int ntiles, int32_t* __restrict__ tot) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  int32_t* row = blockhist + (int64_t)t * NB;
  int v[8], mine = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const int b = 8 * lane + q; v[q] = b < NB ? row[b] : 0; mine += v[q]; }
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  int ex = incl - mine;
#pragma unroll
  for (int q = 0; q < 8; ++q) { const int b = 8 * lane + q; if (b < NB) row[b] = ex; ex += v[q]; }
  if (lane == 63) tot[t] = incl;
}

// K2b (one block): exclusive scan over tiles of (pixels, items); tile_off, tile_item_base, work-item table.
__global__ void __launch_bounds__(kBinThreads)
bin_scan_kernel(const int32_t* __restrict__ tot, int tile_shift, int chunk, int32_t* __restrict__ tile_off,
                int32_t* __restrict__ tile_item_base, int4* __restrict__ items, int32_t* __restrict__ n_items) {
  __shared__ int wsum[kBinThreads / 64], wsum2[kBinThreads / 64];
  const int ntiles = 1 << (2 * tile_shift);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (ntiles + kBinThreads - 1) / kBinThreads;      // consecutive tiles per thread
  int mytot = 0, myit = 0;
  for (int q = 0; q < per; ++q) {
    const int t = tid * per + q;
    if (t < ntiles) { const int c = tot[t]; mytot += c; myit += (c + chunk - 1) / chunk; }
  }
  int a = mytot, n2 = myit;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(a, o, 64), un = __shfl_up(n2, o, 64);
    if (lane >= o) { a += ua; n2 += un; }
  }
  if (lane == 63) { wsum[wave] = a; wsum2[wave] = n2; }
  __syncthreads();
  int off = a - mytot, ioff = n2 - myit;
  for (int w = 0; w < wave; ++w) { off += wsum[w]; ioff += wsum2[w]; }
  for (int q = 0; q < per; ++q) {
    const int t = tid * per + q;
    if (t < ntiles) {
      const int total = tot[t];
      const int nit = (total + chunk - 1) / chunk;
      tile_off[t] = off;
      tile_item_base[t] = ioff;
      for (int jj = 0; jj < nit; ++jj) {
        const int cnt = (total - jj * chunk) < chunk ? (total - jj * chunk) : chunk;
        items[ioff + jj] = make_int4(off + jj * chunk, cnt, t, nit);
      }
      off += total;
      ioff += nit;
    }
  }
  if (tid == kBinThreads - 1) {
    tile_off[ntiles] = off; tile_item_base[ntiles] = ioff;
    n_items[0] = ioff;
    n_items[1] = n_items[2] = n_items[3] = 0;     // work / exit counters of the persistent interleaved kernels (il_claim, il_done)
  }
}

// K3: scatter (x, y, original index) into tile order.  Same pixel->block partition as K1.
__device__ __forceinline__ void bin_scatter_body(int blk, const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift,
                                                 int NB, const int32_t* __restrict__ blockhist, const int32_t* __restrict__ tile_off,
                                                 float4* __restrict__ sorted, int* cursor) {
  const int ntiles = 1 << (2 * tile_shift);
  for (int i = threadIdx.x; i < ntiles; i += kBinThreads) cursor[i] = tile_off[i] + blockhist[(int64_t)i * NB + blk];
  __syncthreads();
  const int64_t lo = (int64_t)blk * per_block;
  const int64_t hi = lo + per_block < P ? lo + per_block : P;
  for (int64_t p0 = lo + threadIdx.x; p0 < hi; p0 += (int64_t)kBinU * kBinThreads) {
    float2 c[kBinU];
    int pos[kBinU];
#pragma unroll
    for (int u = 0; u < kBinU; ++u) { const int64_t p = p0 + (int64_t)u * kBinThreads; c[u] = xy[p < hi ? p : hi - 1]; }
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      pos[u] = (p0 + (int64_t)u * kBinThreads < hi) ? atomicAdd(&cursor[tile_of(c[u].x, c[u].y, tile_shift)], 1) : -1;
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      if (pos[u] >= 0) sorted[pos[u]] = make_float4(c[u].x, c[u].y, __int_as_float((int)(p0 + (int64_t)u * kBinThreads)), 0.f);
  }
}

__global__ void __launch_bounds__(kBinThreads)
bin_scatter_kernel(const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift, int NB,
                   const int32_t* __restrict__ blockhist, const int32_t* __restrict__ tile_off, float4* __restrict__ sorted) {
  extern __shared__ int cursor[];
  bin_scatter_body((int)blockIdx.x, xy, P, per_block, tile_shift, NB, blockhist, tile_off, sorted, cursor);
}

// TWO-LAUNCH BINNING (count -> scatter): the row scan and the tile scan — two launches of ~5 us each inside a replayed step, all
// of it launch latency — fold into the scatter launch.  The count blocks add their histograms to per-tile totals with global
// atomics (`pws`: a PERSISTENT zero-initialised workspace [totals ntiles | cursors ntiles | ticket]); every scatter block
// scans the totals itself (1024 tiles: one per thread), reserves its pixels' places in each tile with one atomic on the tile's
// cursor (the order of the blocks inside a tile is whatever the atomics make it — as the order of pixels inside a block
// already is), block 0 also writes the tables the pixel stage reads, and the LAST block out (ticket) puts the workspace back to
// zero for the next call.  No block ever waits for another block.
// `cursor` [ntiles], `wsum` / `wsum2` [kBinThreads / 64] and `s_last` live in the caller's LDS (the kernel below, or the tail of the
// pixel-stage backward when the scatter of the NEXT batch rides there as claimed tasks: tiled_bwd_il_kernel).  pws = [totals
// ntiles | cursors ntiles | ticket | task-claim counter of the riding form].
__device__ __forceinline__ void bin_scatter2_body(const int blk, const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift,
                                                  int NB, int chunk, const int32_t* __restrict__ blockhist, int32_t* __restrict__ pws,
                                                  int32_t* __restrict__ tile_off, int32_t* __restrict__ tile_item_base,
                                                  int4* __restrict__ items, int32_t* __restrict__ n_items, float4* __restrict__ sorted,
                                                  int* cursor, int* wsum, int* wsum2, int* s_last) {
  const int ntiles = 1 << (2 * tile_shift);
  int32_t* tot = pws;
  int32_t* gcur = pws + ntiles;
  int32_t* ticket = pws + 2 * ntiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (ntiles + kBinThreads - 1) / kBinThreads;      // consecutive tiles per thread
  // exclusive scans over tiles of (pixels, items), as bin_scan_kernel
  int mytot = 0, myit = 0;
  for (int q = 0; q < per; ++q) {
    const int t = tid * per + q;
    if (t < ntiles) { const int c = tot[t]; mytot += c; myit += (c + chunk - 1) / chunk; }
  }
  int a = mytot, n2 = myit;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(a, o, 64), un = __shfl_up(n2, o, 64);
    if (lane >= o) { a += ua; n2 += un; }
  }
  if (lane == 63) { wsum[wave] = a; wsum2[wave] = n2; }
  __syncthreads();
  int off = a - mytot, ioff = n2 - myit;
  for (int w = 0; w < wave; ++w) { off += wsum[w]; ioff += wsum2[w]; }
  for (int q = 0; q < per; ++q) {
    const int t = tid * per + q;
    if (t < ntiles) {
      const int total = tot[t];
      const int nit = (total + chunk - 1) / chunk;
      // this block's first place in tile t: behind the blocks that reserved before it
      const int mine = blockhist[(int64_t)t * NB + blk];
      cursor[t] = off + (mine ? atomicAdd(gcur + t, mine) : 0);
      if (blk == 0) {
        tile_off[t] = off;
        tile_item_base[t] = ioff;
        for (int jj = 0; jj < nit; ++jj) {
          const int cnt = (total - jj * chunk) < chunk ? (total - jj * chunk) : chunk;
          items[ioff + jj] = make_int4(off + jj * chunk, cnt, t, nit);
        }
      }
      off += total;
      ioff += nit;
    }
  }
  if (blk == 0 && tid == kBinThreads - 1) {
    tile_off[ntiles] = off; tile_item_base[ntiles] = ioff;
    n_items[0] = ioff;
    n_items[1] = n_items[2] = n_items[3] = 0;
  }
  __syncthreads();
  const int64_t lo = (int64_t)blk * per_block;
  const int64_t hi = lo + per_block < P ? lo + per_block : P;
  for (int64_t p0 = lo + tid; p0 < hi; p0 += (int64_t)kBinU * kBinThreads) {     // (loads together: see bin_count_body)
    float2 c[kBinU];
    int pos[kBinU];
#pragma unroll
    for (int u = 0; u < kBinU; ++u) { const int64_t p = p0 + (int64_t)u * kBinThreads; c[u] = xy[p < hi ? p : hi - 1]; }
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      pos[u] = (p0 + (int64_t)u * kBinThreads < hi) ? atomicAdd(&cursor[tile_of(c[u].x, c[u].y, tile_shift)], 1) : -1;
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      if (pos[u] >= 0) sorted[pos[u]] = make_float4(c[u].x, c[u].y, __int_as_float((int)(p0 + (int64_t)u * kBinThreads)), 0.f);
  }
  // last block out clears the workspace (every block has finished reading the totals and reserving on the cursors by then)
  __syncthreads();
  if (tid == 0) *s_last = atomicAdd(ticket, 1) == NB - 1;
  __syncthreads();
  if (*s_last) {
    for (int t = tid; t < ntiles; t += kBinThreads) { tot[t] = 0; gcur[t] = 0; }
    if (tid == 0) *ticket = 0;
  }
}

__global__ void __launch_bounds__(kBinThreads)
bin_scatter2_kernel(const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift, int NB, int chunk,
                    const int32_t* __restrict__ blockhist, int32_t* __restrict__ pws, int32_t* __restrict__ tile_off,
                    int32_t* __restrict__ tile_item_base, int4* __restrict__ items, int32_t* __restrict__ n_items,
                    float4* __restrict__ sorted) {
  extern __shared__ int cursor[];                 // [ntiles]
  __shared__ int wsum[kBinThreads / 64], wsum2[kBinThreads / 64];
  __shared__ int s_last;
  bin_scatter2_body((int)blockIdx.x, xy, P, per_block, tile_shift, NB, chunk, blockhist, pws, tile_off, tile_item_base, items, n_items,
                    sorted, cursor, wsum, wsum2, &s_last);
}

// RESERVING COUNT -> SCATTER (round 4, late): the form gngf_bin_pixels2 and the riders use.  A scatter task of the form above
// spends a third of its ~14 us on work every task repeats — the scan over the tile totals, one global atomic per non-empty tile
// to reserve its places — and as a task in the tail of the pixel-stage backward it has to fit a hole of one work item (22 us): one
// fits, two do not.  Here the COUNT side does that work once: a count block reserves its places in every tile as soon as it
// knows its histogram (one RETURNING atomic per non-empty tile on the tile's cursor — the cursors end up holding the tile
// totals) and keeps the offsets, block-major, for the scatter block of the same index; the LAST count block out (ticket: its
// own atomics have returned before it takes one, so every reservation is performed) scans the totals — read with device-scope
// atomic loads: they were only ever touched by atomics — and writes the tile tables and the work items.  A scatter block then
// reads two coalesced rows (tile offsets + its own offsets) and moves its pixels: no scan, no global atomics at all.
// Nobody puts the cursors back to zero either (that took a "last scatter block out" ticket: a returning atomic and two barriers per
// task): they RUN ON from job to job, and the last count block out notes where this job's successor starts (`start`); offsets and
// totals are differences in unsigned arithmetic, correct through any wrap-around.
// pws = [cursors ntiles | start ntiles | (unused) | task counter | count ticket]: zero before its first use, never reset.
struct BinJobDev {
  const float2* xy;
  int64_t P, per_block;
  int NB, tile_shift, chunk;
  int32_t* blockbase;            // [NB][ntiles]: first place of block b in tile t, relative to the tile's start
  int32_t* pws;
  int32_t *tile_off, *tile_item_base;
  int4* items;
  int32_t* n_items;
  float4* sorted;
};

template <int NT>
__device__ __forceinline__ void bin_count_reserve_body(const int blk, const BinJobDev& j, int* hist /* LDS: ntiles + 2 NT/64 + 1 ints */) {
  const int ntiles = 1 << (2 * j.tile_shift);
  int* wsum = hist + ntiles;
  int* wsum2 = wsum + NT / 64;
  int* s_last = wsum2 + NT / 64;
  unsigned* gcur = reinterpret_cast<unsigned*>(j.pws);
  unsigned* start = gcur + ntiles;                            // the cursors' values when this job began (written by its predecessor)
  int32_t* tcount = j.pws + 2 * ntiles + 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < ntiles; i += NT) hist[i] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blk * j.per_block;
  const int64_t hi = lo + j.per_block < j.P ? lo + j.per_block : j.P;
  for (int64_t p0 = lo + tid; p0 < hi; p0 += (int64_t)kBinU * NT) {              // (loads together: see bin_count_body)
    float2 c[kBinU];
#pragma unroll
    for (int u = 0; u < kBinU; ++u) { const int64_t p = p0 + (int64_t)u * NT; c[u] = j.xy[p < hi ? p : hi - 1]; }
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      if (p0 + (int64_t)u * NT < hi) atomicAdd(&hist[tile_of(c[u].x, c[u].y, j.tile_shift)], 1);
  }
  __syncthreads();
  int32_t* mybase = j.blockbase + (int64_t)blk * ntiles;
  for (int i = tid; i < ntiles; i += NT) {
    const int c = hist[i];
    mybase[i] = c ? (int)(atomicAdd(gcur + i, (unsigned)c) - start[i]) : 0;      // the store needs the atomic's return: it has been performed by then
  }
  __syncthreads();
  if (tid == 0) *s_last = atomicAdd(tcount, 1) == j.NB - 1;
  __syncthreads();
  if (!*s_last) return;
  // last count block out: cursors = tile totals.  Exclusive scans over tiles of (pixels, items), as bin_scan_kernel.
  const int per = (ntiles + NT - 1) / NT;                   // consecutive tiles per thread
  int mytot = 0, myit = 0;
  for (int q = 0; q < per; ++q) {
    const int t = tid * per + q;
    if (t < ntiles) {
      const unsigned now = __hip_atomic_load(gcur + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int c = (int)(now - start[t]);
      start[t] = now;                                        // where the NEXT job's reservations begin (every block of this one is done)
      hist[t] = c;                                           // (the histogram is no longer needed: keep the totals there)
      mytot += c; myit += (c + j.chunk - 1) / j.chunk;
    }
  }
  int a = mytot, n2 = myit;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(a, o, 64), un = __shfl_up(n2, o, 64);
    if (lane >= o) { a += ua; n2 += un; }
  }
  if (lane == 63) { wsum[wave] = a; wsum2[wave] = n2; }
  __syncthreads();
  int off = a - mytot, ioff = n2 - myit;
  for (int w = 0; w < wave; ++w) { off += wsum[w]; ioff += wsum2[w]; }
  for (int q = 0; q < per; ++q) {
    const int t = tid * per + q;
    if (t < ntiles) {
      const int total = hist[t];
      const int nit = (total + j.chunk - 1) / j.chunk;
      j.tile_off[t] = off;
      j.tile_item_base[t] = ioff;
      for (int jj = 0; jj < nit; ++jj) {
        const int cnt = (total - jj * j.chunk) < j.chunk ? (total - jj * j.chunk) : j.chunk;
        j.items[ioff + jj] = make_int4(off + jj * j.chunk, cnt, t, nit);
      }
      off += total;
      ioff += nit;
    }
  }
  if (tid == NT - 1) {
    j.tile_off[ntiles] = off; j.tile_item_base[ntiles] = ioff;
    j.n_items[0] = ioff;
    j.n_items[1] = j.n_items[2] = j.n_items[3] = 0;
    *tcount = 0;                                             // (read again by atomics of a LATER launch only)
  }
}

// scatter block `blk` (1024 threads): cursor [ntiles] in the caller's LDS
__device__ __forceinline__ void bin_scatter3_body(const int blk, const BinJobDev& j, int* cursor) {
  const int ntiles = 1 << (2 * j.tile_shift);
  const int tid = threadIdx.x;
  const int32_t* mybase = j.blockbase + (int64_t)blk * ntiles;
  for (int t = tid; t < ntiles; t += kBinThreads) cursor[t] = j.tile_off[t] + mybase[t];
  __syncthreads();
  const int64_t lo = (int64_t)blk * j.per_block;
  const int64_t hi = lo + j.per_block < j.P ? lo + j.per_block : j.P;
  for (int64_t p0 = lo + tid; p0 < hi; p0 += (int64_t)kBinU * kBinThreads) {
    float2 c[kBinU];
    int pos[kBinU];
#pragma unroll
    for (int u = 0; u < kBinU; ++u) { const int64_t p = p0 + (int64_t)u * kBinThreads; c[u] = j.xy[p < hi ? p : hi - 1]; }
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      pos[u] = (p0 + (int64_t)u * kBinThreads < hi) ? atomicAdd(&cursor[tile_of(c[u].x, c[u].y, j.tile_shift)], 1) : -1;
#pragma unroll
    for (int u = 0; u < kBinU; ++u)
      if (pos[u] >= 0) j.sorted[pos[u]] = make_float4(c[u].x, c[u].y, __int_as_float((int)(p0 + (int64_t)u * kBinThreads)), 0.f);
  }
}

// the two as launches of their own (gngf_bin_pixels2: the first step of a replay, eager steps); workgroups [NB, NB + zblocks)
// of the count launch clear `zero` (as bin_count_ride_kernel)
__global__ void __launch_bounds__(kBinThreads)
bin_count_reserve_kernel(const BinJobDev j, float4* __restrict__ zero, int64_t nvec, int zblocks) {
  extern __shared__ int hist[];
  if ((int)blockIdx.x < j.NB) {
    if (blockIdx.x == 0 && threadIdx.x == 0) j.pws[2 * (1 << (2 * j.tile_shift)) + 1] = 0;      // the riding form's task counter
    bin_count_reserve_body<kBinThreads>((int)blockIdx.x, j, hist);
    return;
  }
  const int64_t zb = (int)blockIdx.x - j.NB;
  const int64_t per = (nvec + zblocks - 1) / zblocks;
  const int64_t lo = zb * per, hi = lo + per < nvec ? lo + per : nvec;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t e = lo + threadIdx.x; e < hi; e += kBinThreads) zero[e] = z;
}

__global__ void __launch_bounds__(kBinThreads)
bin_scatter3_kernel(const BinJobDev j) {
  extern __shared__ int cursor[];                 // [ntiles]
  bin_scatter3_body((int)blockIdx.x, j, cursor);
}

// K1 with a ZERO-FILL riding on the launch: workgroups [NB, NB + zblocks) clear `zero` (nvec float4) instead — the gradient
// buffer the backward pass will accumulate into (64 MiB at T = 2^19).  The count keeps 128 of the 256 CUs busy for ~8 us;
// the fill runs on the others, instead of being a launch (or a stream) of its own.
__global__ void __launch_bounds__(kBinThreads)
bin_count_ride_kernel(const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift, int NB,
                      int32_t* __restrict__ blockhist, float4* __restrict__ zero, int64_t nvec, int zblocks,
                      int32_t* __restrict__ tot_atomic = nullptr) {
  extern __shared__ int hist[];
  if ((int)blockIdx.x < NB) {
    bin_count_body((int)blockIdx.x, xy, P, per_block, tile_shift, NB, blockhist, hist, tot_atomic);
    return;
  }
  const int64_t zb = (int)blockIdx.x - NB;
  const int64_t per = (nvec + zblocks - 1) / zblocks;
  const int64_t lo = zb * per, hi = lo + per < nvec ? lo + per : nvec;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t e = lo + threadIdx.x; e < hi; e += kBinThreads) zero[e] = z;
}

// ---------------------------------------------------------------------------------------------- vertex stage
// grid = (ceil(maxverts/256), Ls): one lane per (level, vertex).  Level grids are stored back to back:
// G[(goff_l + gy*(N_l+2) + gx) * F + f], goff_l = sum_{j<l} (N_j+2)^2.
__device__ __forceinline__ int64_t level_offset(const int32_t* n_ls, int l) {
  int64_t o = 0;
  for (int j = 0; j < l; ++j) { const int64_t g = n_ls[j] + 2; o += g * g; }
  return o;
}

// one (level, vertex): i = gy * (N_l + 2) + gx inside level l, goff = the level's offset in G (vertices)
template <int F, bool VT, typename TT>
__device__ __forceinline__ void vertex_fwd_lane(const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx,
                                                const float* __restrict__ vert_w, float* __restrict__ G, float* __restrict__ dG_zero,
                                                int64_t T, int K, int vstride, int64_t NV, bool pow2, int l, int gw, int i, int64_t goff,
                                                int zero_words = 1, float* __restrict__ clear_rows = nullptr) {
  const int gy = i / gw, gx = i - gy * gw;
  const TT* tab = tables + (int64_t)l * T * F;
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  if constexpr (!VT) {
    const int64_t row = spatial_hash(gx, gy, T, pow2);
    if (G) {                                      // (G == nullptr: the pixel stage gathers from the tables itself — this lane only clears)
      const TT* r = tab + row * F;
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] = tload(r + f);
    }
    if (clear_rows) {                             // a step-to-step table gradient: the row this vertex can add to starts from zero
      float* z = clear_rows + ((int64_t)l * T + row) * F;       // (as gngf_clear_hashed_rows, without a launch of its own)
#pragma unroll
      for (int f = 0; f < F; ++f) z[f] = 0.f;
    }
  } else {
    const int64_t vid = (int64_t)gy * vstride + gx;
    if (gx < vstride && vid < NV) {
      for (int k = 0; k < K; ++k) {
        const float w = vert_w[vid * K + k];
        const TT* r = tab + (int64_t)vert_idx[vid * K + k] * F;
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] += tload(r + f) * w;
      }
    }
  }
  if (G) {
    float* o = G + (goff + i) * F;
#pragma unroll
    for (int f = 0; f < F; ++f) o[f] = acc[f];
  }
  if (dG_zero) {                                  // the vertex-grid gradient of the coming backward pass starts from zero
    float* z = dG_zero + (goff + i) * F * zero_words;          // (zero_words = 2: the 64-bit fixed-point form, see tiled_bwd_il_kernel)
    for (int f = 0; f < F * zero_words; ++f) z[f] = 0.f;
  }
}

template <int F, bool VT, typename TT>
__global__ void __launch_bounds__(256)
vertex_fwd_kernel(const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx, const float* __restrict__ vert_w,
                  const int32_t* __restrict__ n_ls, float* __restrict__ G, int64_t T, int K, int vstride, int64_t NV,
                  bool pow2) {
  const int l = blockIdx.y;
  const int gw = n_ls[l] + 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= gw * gw) return;
  vertex_fwd_lane<F, VT, TT>(tables, vert_idx, vert_w, G, nullptr, T, K, vstride, NV, pow2, l, gw, i, level_offset(n_ls, l));
}

// rider block `vb` of the vertex stage forward: one (level, vertex) per thread, flat over the level grids
template <int F, bool VT, typename TT>
__device__ __forceinline__ void vertex_ride_block(int vb, const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx,
                                                  const float* __restrict__ vert_w, const int32_t* __restrict__ n_ls, float* __restrict__ G,
                                                  float* __restrict__ dG_zero, int Ls, int64_t T, int K, int vstride, int64_t NV, bool pow2,
                                                  int64_t vtot, int zero_words, float* __restrict__ clear_rows = nullptr) {
  const int64_t e = (int64_t)vb * kBinThreads + threadIdx.x;
  if (e >= vtot) return;
  if (e == 0 && dG_zero && zero_words == 2) {     // the two 64-bit words behind the fixed-point grid: scale and poison flag
    float* t = dG_zero + vtot * F * 2;
    t[0] = t[1] = t[2] = t[3] = 0.f;
  }
  int l = 0, gw = n_ls[0] + 2;
  int64_t goff = 0;
  while (l + 1 < Ls && e >= goff + (int64_t)gw * gw) { goff += (int64_t)gw * gw; ++l; gw = n_ls[l] + 2; }
  vertex_fwd_lane<F, VT, TT>(tables, vert_idx, vert_w, G, dG_zero, T, K, vstride, NV, pow2, l, gw, (int)(e - goff), goff, zero_words,
                             clear_rows);
}

// K3 with the VERTEX STAGE FORWARD riding on the launch: workgroups [NB, NB + ceil(vtot / 1024)) evaluate one (level, vertex)
// per thread (flat over the level grids) — work that does not depend on the binned pixels and used to be a launch on a
// helper stream.  Measured: parallel branches of a replayed hipGraph run on different hardware queues, and every
// cross-queue dependency costs ~10 us (+ ~12 us between replays); one linear chain with riders has no such gaps.
// (The riders sit on the SCATTER launch, 13 us alone, and the gradient clear on the count launch, 8 us alone: the other way
// round the scatter launch took 21 us.)
template <int F, bool VT, typename TT>
__global__ void __launch_bounds__(kBinThreads)
bin_scatter_ride_kernel(const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift, int NB,
                        const int32_t* __restrict__ blockhist, const int32_t* __restrict__ tile_off, float4* __restrict__ sorted,
                        const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx,
                        const float* __restrict__ vert_w, const int32_t* __restrict__ n_ls, float* __restrict__ G,
                        float* __restrict__ dG_zero, int Ls, int64_t T, int K, int vstride, int64_t NV, bool pow2, int64_t vtot,
                        int zero_words, float* __restrict__ clear_rows) {
  extern __shared__ int cursor[];
  if ((int)blockIdx.x < NB) {
    bin_scatter_body((int)blockIdx.x, xy, P, per_block, tile_shift, NB, blockhist, tile_off, sorted, cursor);
    return;
  }
  vertex_ride_block<F, VT, TT>((int)blockIdx.x - NB, tables, vert_idx, vert_w, n_ls, G, dG_zero, Ls, T, K, vstride, NV, pow2, vtot, zero_words,
                               clear_rows);
}

// K1 with the vertex stage forward riding on it (the count keeps half of the CUs busy for ~8 us): used when the launch does
// not carry the 64 MiB gradient clear (the fused training decoder clears that buffer between its MFMAs) — the scatter launch
// then runs alone.
template <int F, bool VT, typename TT>
__global__ void __launch_bounds__(kBinThreads)
bin_count_vride_kernel(const float2* __restrict__ xy, int64_t P, int64_t per_block, int tile_shift, int NB,
                       int32_t* __restrict__ blockhist, const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx,
                       const float* __restrict__ vert_w, const int32_t* __restrict__ n_ls, float* __restrict__ G,
                       float* __restrict__ dG_zero, int Ls, int64_t T, int K, int vstride, int64_t NV, bool pow2, int64_t vtot,
                       int zero_words, int32_t* __restrict__ tot_atomic, float* __restrict__ clear_rows) {
  extern __shared__ int hist[];
  if ((int)blockIdx.x < NB) {
    bin_count_body((int)blockIdx.x, xy, P, per_block, tile_shift, NB, blockhist, hist, tot_atomic);
    return;
  }
  vertex_ride_block<F, VT, TT>((int)blockIdx.x - NB, tables, vert_idx, vert_w, n_ls, G, dG_zero, Ls, T, K, vstride, NV, pow2, vtot, zero_words,
                               clear_rows);
}

template <int F, bool VT, typename TT>
__global__ void __launch_bounds__(256)
vertex_bwd_kernel(const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx, const float* __restrict__ vert_w,
                  const int32_t* __restrict__ n_ls, const float* __restrict__ dG, float* __restrict__ dtables,
                  float* __restrict__ dvert_w, int Ls, int64_t T, int K, int vstride, int64_t NV, bool pow2) {
  // flat over the level grids (see gather_partials_kernel): blockIdx.y is not used
  int l = 0, gw = n_ls[0] + 2;
  int64_t goff = 0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  while (l + 1 < Ls && e >= goff + (int64_t)gw * gw) { goff += (int64_t)gw * gw; ++l; gw = n_ls[l] + 2; }
  if (e - goff >= (int64_t)gw * gw) return;
  const int i = (int)(e - goff);
  const float* gp = dG + (goff + i) * F;
  float g[F];
  bool any = false;
#pragma unroll
  for (int f = 0; f < F; ++f) { g[f] = gp[f]; any |= (g[f] != 0.f); }
  if (!any) return;                               // vertices no pixel touched
  const int gy = i / gw, gx = i - gy * gw;
  const TT* tab = tables + (int64_t)l * T * F;
  float* dtab = dtables + (int64_t)l * T * F;
  if constexpr (!VT) {
    float* r = dtab + spatial_hash(gx, gy, T, pow2) * F;
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(r + f, g[f]);
  } else {
    const int64_t vid = (int64_t)gy * vstride + gx;
    if (gx >= vstride || vid >= NV) return;
    for (int k = 0; k < K; ++k) {
      const float w = vert_w[vid * K + k];
      const int64_t row = vert_idx[vid * K + k];
      float dot = 0.f;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        dot += g[f] * tload(tab + row * F + f);
        atomicAdd(dtab + row * F + f, g[f] * w);
      }
      if (dvert_w) atomicAdd(dvert_w + vid * K + k, dot);
    }
  }
}

// ---------------------------------------------------------------------------------------------- pixel stage
// A decoder slab reduction (gngf_common.h: decoder_reduce_block) that rides on the tiled backward's launch: workgroups
// [first_block, first_block + ceil(nslab / 64)) run it instead of a work item.  slabs == nullptr: none.
struct RideAlong {
  const float* slabs;
  float *dW0, *db0, *dW1, *db1, *dW2, *db2;
  int nslabs, nslab, in_dim, out_dim, first_block;
  const float *promised, *arrived;      // loss gradient the fused training decoder ran with / the one that arrived (gngf_common.h: promise_broken)
};

// The value of the fused pixel loss (gngf_common.h: mse_sum_block) riding on the same launch: workgroups
// [first_block, first_block + nblocks).  pred == nullptr: none.
struct MseRide {
  const float* pred;
  const float* label;
  float* loss;
  double* acc;
  unsigned* counter;
  int64_t n;
  int first_block, nblocks;
};

struct TileMeta {          // per-level placement of the tile's sub-grid (in LDS)
  int n[GNGF_MAX_LEVELS], gw[GNGF_MAX_LEVELS], cx[GNGF_MAX_LEVELS], cy[GNGF_MAX_LEVELS], wx[GNGF_MAX_LEVELS],
      wy[GNGF_MAX_LEVELS], loff[GNGF_MAX_LEVELS];
  int64_t goff[GNGF_MAX_LEVELS];
};

// Fills meta for tile (tx,ty).  A level whose sub-grid does not fit the provided LDS gets wx = 0 (global fallback).
__device__ __forceinline__ void setup_tile(TileMeta& m, const int32_t* n_ls, int Ls, int tx, int ty, int tile_shift, int F,
                                           int lds_floats) {
  const int tid = threadIdx.x;
  if (tid < Ls) {
    const int n = n_ls[tid];
    m.n[tid] = n;
    m.gw[tid] = n + 2;
    const int cx = (tx * n) >> tile_shift, cy = (ty * n) >> tile_shift;
    int hx = (((tx + 1) * n) >> tile_shift) + 1, hy = (((ty + 1) * n) >> tile_shift) + 1;
    hx = hx > n + 1 ? n + 1 : hx;
    hy = hy > n + 1 ? n + 1 : hy;
    m.cx[tid] = cx; m.cy[tid] = cy; m.wx[tid] = hx - cx + 1; m.wy[tid] = hy - cy + 1;
  }
  __syncthreads();
  if (tid == 0) {
    int64_t go = 0;
    int lo = 0;
    for (int l = 0; l < Ls; ++l) {
      m.goff[l] = go;
      go += (int64_t)m.gw[l] * m.gw[l];
      const int sz = m.wx[l] * m.wy[l] * F;
      if (lo + sz <= lds_floats) { m.loff[l] = lo; lo += sz; } else { m.wx[l] = 0; m.loff[l] = 0; }
    }
  }
  __syncthreads();
}

// HSRC (round 5, spatial-hash source): the sub-grids are staged from the level tables themselves — G_l[v] = E_l[hash(v)], the value
// vertex_fwd_lane would have put into the vertex grid G, so enc is bit-identical — and the vertex grid (69 MB written and read
// back at the 4096^2 shape, 90 MB at the 8192^2 one) and the vertex riders' gathers disappear, as on the interleaved kernel.
template <int F, bool HSRC = false, typename TT = float>
__global__ void __launch_bounds__(kTBF)
tiled_fwd_kernel(const float4* __restrict__ sorted, const int4* __restrict__ items, const int32_t* __restrict__ n_items,
                 const int32_t* __restrict__ n_ls, const float* __restrict__ G, float* __restrict__ enc, int L, int Ls,
                 int tile_shift, int lds_floats, const TT* __restrict__ tables = nullptr, int64_t T = 0, bool pow2 = false) {
  extern __shared__ float lds[];
  __shared__ TileMeta m;
  if ((int)blockIdx.x >= *n_items) return;
  const int4 it = items[blockIdx.x];
  const int tid = threadIdx.x;
  const int TSm = (1 << tile_shift) - 1;
  setup_tile(m, n_ls, Ls, it.z & TSm, it.z >> tile_shift, tile_shift, F, lds_floats);
  // Staging, flattened over (level, vertex): a loop per level would expose one L2 round trip per level (16 x ~1.5 us per
  // work item — the fixed cost that made many small items slow); here every thread's few loads are in flight together.
  // Staged sub-grids are back to back in LDS (loff ascending), so element e of the image belongs to the last staged
  // level whose offset is <= e.
  {
    int used = 0;
    for (int l = 0; l < Ls; ++l) used += m.wx[l] * m.wy[l];            // vertices (levels that did not fit have wx = 0)
    for (int e = tid; e < used; e += kTBF) {
      int l = 0, base = 0;
      for (int q = 0, acc = 0; q < Ls; ++q) {
        const int sz = m.wx[q] * m.wy[q];
        if (sz > 0 && e >= acc) { l = q; base = acc; }
        acc += sz;
      }
      const int i = e - base, wx = m.wx[l];
      const int iy = i / wx, ix = i - iy * wx;
      float* dst = lds + m.loff[l] + i * F;
      if constexpr (HSRC) {
        const TT* sp = tables + ((int64_t)l * T + spatial_hash(m.cx[l] + ix, m.cy[l] + iy, T, pow2)) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) dst[f] = tload(sp + f);
      } else {
        const float* sp = G + (m.goff[l] + (int64_t)(m.cy[l] + iy) * m.gw[l] + m.cx[l] + ix) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) dst[f] = sp[f];
      }
    }
  }
  __syncthreads();
  const int ppp = kTBF / Ls;                     // pixels per pass: one lane per (pixel, level)
  const int lp = tid / Ls, l = tid - lp * Ls;
  if (lp >= ppp) return;
  const int n = m.n[l], cx = m.cx[l], cy = m.cy[l], wx = m.wx[l], wy = m.wy[l], gw = m.gw[l];
  const float* sub = lds + m.loff[l];
  const float* Gl = G + m.goff[l] * F;
  const int LF = L * F;
  // 4 pixels per lane per trip, their loads issued together.  The kernel is VALU-bound (~26 M wave-instructions), so the
  // common case — every pixel of the trip inside its staged sub-grid, true for all in-domain coordinates — runs without
  // per-lane branches; a wave-uniform vote sends trips with an outlier to the general path.
  constexpr int U = 4;
  float* enc_l = enc + l * F;
  for (int j0 = lp; j0 < it.y; j0 += U * ppp) {
    float4 sv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int j = j0 + u * ppp; sv[u] = sorted[it.x + (j < it.y ? j : it.y - 1)]; }
    Cell cs[U];
    int off[U];
    bool inside = true;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cs[u] = make_cell(sv[u].x, sv[u].y, n);
      const int lx = cs[u].gx - cx, ly = cs[u].gy - cy;
      inside = inside && ((unsigned)lx < (unsigned)(wx - 1)) && ((unsigned)ly < (unsigned)(wy - 1));
      off[u] = (ly * wx + lx) * F;
    }
    if (__ballot(!inside) == 0ull) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* a = sub + off[u];
        const float* b = a + wx * F;
        float r[F];
#pragma unroll
        for (int f = 0; f < F; ++f)
          r[f] = ((a[f] * cs[u].c[0] + a[F + f] * cs[u].c[1]) + b[f] * cs[u].c[2]) + b[F + f] * cs[u].c[3];
        if (j0 + u * ppp < it.y) {
          float* o = enc_l + (int64_t)__float_as_int(sv[u].z) * LF;
#pragma unroll
          for (int f = 0; f < F; ++f) o[f] = r[f];
        }
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j0 + u * ppp >= it.y) break;
      const float4 s = sv[u];
      const int64_t p = (int64_t)__float_as_int(s.z);
      const Cell c = cs[u];
      const int lx = c.gx - cx, ly = c.gy - cy;
      float v[4][F];
      if (lx >= 0 && ly >= 0 && lx + 1 < wx && ly + 1 < wy) {
        const float* a = sub + (ly * wx + lx) * F;
        const float* b = a + wx * F;
#pragma unroll
        for (int f = 0; f < F; ++f) { v[0][f] = a[f]; v[1][f] = a[F + f]; v[2][f] = b[f]; v[3][f] = b[F + f]; }
      } else {                                     // outside the staged sub-grid (never for in-domain coords): global
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int gx = c.gx + (q & 1), gy = c.gy + (q >> 1);
          gx = gx < 0 ? 0 : (gx > n + 1 ? n + 1 : gx);
          gy = gy < 0 ? 0 : (gy > n + 1 ? n + 1 : gy);
          if constexpr (HSRC) {
            const TT* sp = tables + ((int64_t)l * T + spatial_hash(gx, gy, T, pow2)) * F;
#pragma unroll
            for (int f = 0; f < F; ++f) v[q][f] = tload(sp + f);
          } else {
#pragma unroll
            for (int f = 0; f < F; ++f) v[q][f] = Gl[((int64_t)gy * gw + gx) * F + f];
          }
        }
      }
      float* o = enc + p * LF + l * F;
#pragma unroll
      for (int f = 0; f < F; ++f) o[f] = ((v[0][f] * c.c[0] + v[1][f] * c.c[1]) + v[2][f] * c.c[2]) + v[3][f] * c.c[3];
    }
  }
}

// fp32 term -> 64-bit fixed point, exactly, in four VALU operations.  The term arrives PRE-SCALED by 2^(S-32) (the
// gradient is scaled once per item; scaling by a power of two commutes with the rounding of the product g*c, so the
// term is bit-identical to ldexp(g*c, S-32)): floor-convert gives the high word, the fraction (exact in fp32) times 2^32
// the low word.  |u| < 2^31 by the choice of S (the whole SUM must stay below 2^62); floor semantics.
__device__ __forceinline__ long long to_fixed_scaled(float u) {
  int hi;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(hi) : "v"(u));
  const unsigned lo = (unsigned)ldexpf(__builtin_amdgcn_fractf(u), 32);
  return ((long long)hi << 32) + (long long)(unsigned long long)lo;
}

// The same in TWO instructions (level-interleaved kernel): the product g * c of two fp32 values is exact in double precision, and
// one double-precision fma with the constant 1.5 * 2^52 leaves RNE(g * c) — g already scaled by 2^S — as a two's-complement integer
// in the low 52 bits of the result; subtracting the constant's bit pattern (its low word is zero: one 32-bit add) yields the 64-bit
// fixed-point term.  Needs |g * c| < 2^51 (S leaves that room).  The term is the exactly rounded product instead of the fp32-rounded
// one the reference forms: closer to the real sum, and — like it — independent of the order of the additions.
#ifndef GNGF_FIXED_FMA
#define GNGF_FIXED_FMA 1
#endif
__device__ __forceinline__ unsigned long long to_fixed_fma(double g, double c) {
  const double r = __builtin_fma(g, c, 6755399441055744.0);
  return (unsigned long long)__double_as_longlong(r) - 0x4338000000000000ull;
}

// Backward pixel stage.  gfx950's LDS float atomic (ds_add_f32) retires ~3 cycles PER LANE (193 cycles per
// wave-instruction, measured: tools/micro/lds_atomic.cpp) while ds_add_u64 takes 7-12 cycles per wave-instruction,
// so the privatised sub-grids accumulate in 64-bit FIXED POINT: every term g*c (one fp32 multiply, as in the
// reference) is scaled by 2^S, S = 61 - ceil(log2(chunk)) - exponent(max |g| over the item's pixels), which makes
// overflow impossible and the quantisation (2^-S per term, ~2^-49 of the item's largest gradient) far below one fp32
// ulp of any partial sum.  Integer adds commute: the per-item image is bitwise reproducible.
// HDT (round 5; spatial-hash source on a single rank): as in tiled_bwd_il_kernel, the store pass adds the item's sums — rounded to
// fp32 once — straight to row hash(gx, gy) of the level's table gradient: no partial images, no gather pass (245 us at the 8192^2
// shape, 121 us at the 4096^2 one), no vertex grid.
template <int F, bool HDT = false>
__global__ void __launch_bounds__(kTB)
tiled_bwd_kernel(const float4* __restrict__ sorted, const int4* __restrict__ items, const int32_t* __restrict__ n_items,
                 const int32_t* __restrict__ n_ls, const float* __restrict__ genc, float* __restrict__ dG,
                 float* __restrict__ partials, const float* __restrict__ gmax_hint, int hint_count, int hint_stride, int L,
                 int Ls, int tile_shift, int lds_floats, int log2_chunk, RideAlong ride, MseRide mride,
                 float* __restrict__ hash_dt = nullptr, int64_t hash_T = 0, bool hash_pow2 = false) {
  extern __shared__ unsigned long long acc64[];
  __shared__ TileMeta m;
  __shared__ float wmax[kTB / 64];
  if (mride.pred && (int)blockIdx.x >= mride.first_block) {
    mse_sum_block((int)blockIdx.x - mride.first_block, mride.nblocks, mride.pred, mride.label, mride.loss, mride.acc, mride.counter,
                  mride.n);
    return;
  }
  if (ride.slabs && (int)blockIdx.x >= ride.first_block) {
    static_assert(kTB == 1024, "decoder_reduce_block is written for 1024-thread workgroups");
    decoder_reduce_block((int)blockIdx.x - ride.first_block, ride.slabs, ride.nslabs, ride.nslab, ride.in_dim, ride.out_dim, ride.dW0,
                         ride.db0, ride.dW1, ride.db1, ride.dW2, ride.db2, nullptr, ride.promised, ride.arrived);
    return;
  }
  if ((int)blockIdx.x >= *n_items) return;
  const int4 it = items[blockIdx.x];
  const int tid = threadIdx.x;
  const int TSm = (1 << tile_shift) - 1;
  setup_tile(m, n_ls, Ls, it.z & TSm, it.z >> tile_shift, tile_shift, F, lds_floats);
  int used = 0;
  for (int l = 0; l < Ls; ++l) used += m.wx[l] * m.wy[l] * F;
  for (int i = tid; i < used; i += kTB) acc64[i] = 0ull;
  const int ppp = kTB / Ls;
  const int lp = tid / Ls, l = tid - lp * Ls;
  const int LF = L * F;
  // pass 1: largest |gradient| over the item's pixels — skipped when the producer of genc handed over a bound on
  // max |genc| (the fused decoder backward does): any bound >= the item's own maximum keeps the sums overflow-free.
  float gmax = 0.f;
  constexpr int U = 4;                                     // 4 pixels per lane per trip: loads issued together
  if (gmax_hint)                                            // the bound is the largest of hint_count values (one per decoder slab)
    for (int q = tid; q < hint_count; q += kTB) { const float a = gmax_hint[(int64_t)q * hint_stride]; gmax = (a > gmax || a != a) ? a : gmax; }
  if (lp < ppp && !gmax_hint)
    for (int j0 = lp; j0 < it.y; j0 += U * ppp) {
      int64_t pp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int j = j0 + u * ppp; pp[u] = (int64_t)__float_as_int(sorted[it.x + (j < it.y ? j : it.y - 1)].z); }
      float gv[U][F];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int f = 0; f < F; ++f) gv[u][f] = genc[pp[u] * LF + l * F + f];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int f = 0; f < F; ++f) { const float a = fabsf(gv[u][f]); gmax = (a > gmax || a != a) ? a : gmax; }   // NaN sticks
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(gmax, o, 64); gmax = (ov > gmax || ov != ov) ? ov : gmax; }
  if ((tid & 63) == 0) wmax[tid >> 6] = gmax;
  __syncthreads();
  gmax = wmax[0];
#pragma unroll
  for (int w = 1; w < kTB / 64; ++w) gmax = (wmax[w] > gmax || wmax[w] != wmax[w]) ? wmax[w] : gmax;
  // d enc of a fused training decoder whose promised loss gradient did not arrive: the image goes out as NaN (below)
  const bool finite = gmax < INFINITY && !promise_broken(ride.promised, ride.arrived);     // false for inf and NaN
  int eg = 0;
  if (finite && gmax > 0.f) (void)frexpf(gmax, &eg);        // gmax < 2^eg
  // fma form: every term must stay below 2^51 — |g| 2^S < 2^(60 - log2_chunk), so a bound of fewer than 2^10 terms counts as 2^10
  int S = GNGF_FIXED_FMA ? 60 - (log2_chunk < 10 ? 10 : log2_chunk) - eg : 61 - log2_chunk - eg;
  S = S > 126 ? 126 : S;                                         // 2^S stays an fp32 number for any gradient magnitude
  if (lp < ppp && finite) {
    const int n = m.n[l], cx = m.cx[l], cy = m.cy[l], wx = m.wx[l], wy = m.wy[l], gw = m.gw[l];
    unsigned long long* sub = acc64 + m.loff[l];
    float* dGl = dG + m.goff[l] * F;
    for (int j0 = lp; j0 < it.y; j0 += U * ppp) {
      float4 sv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int j = j0 + u * ppp; sv[u] = sorted[it.x + (j < it.y ? j : it.y - 1)]; }
      float gv[U][F];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int f = 0; f < F; ++f) gv[u][f] = genc[(int64_t)__float_as_int(sv[u].z) * LF + l * F + f];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j0 + u * ppp >= it.y) break;
        const float4 s = sv[u];
        const Cell c = make_cell(s.x, s.y, n);
        const int lx = c.gx - cx, ly = c.gy - cy;
        float g[F];
#pragma unroll
        for (int f = 0; f < F; ++f) g[f] = gv[u][f];
        if (lx >= 0 && ly >= 0 && lx + 1 < wx && ly + 1 < wy) {
          unsigned long long* a = sub + (ly * wx + lx) * F;
          unsigned long long* b = a + wx * F;
#pragma unroll
          for (int f = 0; f < F; ++f) {
#if GNGF_FIXED_FMA
            const double gd = (double)ldexpf(g[f], S);
            atomicAdd(a + f, to_fixed_fma(gd, (double)c.c[0]));
            atomicAdd(a + F + f, to_fixed_fma(gd, (double)c.c[1]));
            atomicAdd(b + f, to_fixed_fma(gd, (double)c.c[2]));
            atomicAdd(b + F + f, to_fixed_fma(gd, (double)c.c[3]));
#else
            const float gs = ldexpf(g[f], S - 32);
            atomicAdd(a + f, (unsigned long long)to_fixed_scaled(gs * c.c[0]));
            atomicAdd(a + F + f, (unsigned long long)to_fixed_scaled(gs * c.c[1]));
            atomicAdd(b + f, (unsigned long long)to_fixed_scaled(gs * c.c[2]));
            atomicAdd(b + F + f, (unsigned long long)to_fixed_scaled(gs * c.c[3]));
#endif
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int gx = c.gx + (q & 1), gy = c.gy + (q >> 1);
            gx = gx < 0 ? 0 : (gx > n + 1 ? n + 1 : gx);
            gy = gy < 0 ? 0 : (gy > n + 1 ? n + 1 : gy);
            if constexpr (HDT) {
              float* r = hash_dt + ((int64_t)l * hash_T + spatial_hash(gx, gy, hash_T, hash_pow2)) * F;
#pragma unroll
              for (int f = 0; f < F; ++f) atomicAdd(r + f, g[f] * c.c[q]);
            } else {
#pragma unroll
              for (int f = 0; f < F; ++f) atomicAdd(dGl + ((int64_t)gy * gw + gx) * F + f, g[f] * c.c[q]);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const double inv = finite ? ldexp(1.0, -S) : 0.0;
  if constexpr (HDT) {
    // element i of the image = feature f of vertex v of the q-th level that fits (the levels are placed back to back in order)
    for (int i = tid; i < used; i += kTB) {
      int q = 0, lo = 0, sz = 0;
      for (; q < Ls; ++q) { sz = m.wx[q] * m.wy[q] * F; if (i < lo + sz) break; lo += sz; }
      const long long sum = (long long)acc64[i];
      if (sum != 0 || !finite) {
        const int j = i - lo, v = j / F, f = j - v * F;
        const int wxq = m.wx[q], iy = v / wxq, ix = v - iy * wxq;
        atomicAdd(hash_dt + ((int64_t)q * hash_T + spatial_hash(m.cx[q] + ix, m.cy[q] + iy, hash_T, hash_pow2)) * F + f,
                  finite ? (float)((double)sum * inv) : __int_as_float(0x7fc00000));
      }
    }
    return;
  }
  // store pass: the item's privatised sub-grid image goes out as fp32 with plain coalesced stores; the gather pass
  // sums, per destination vertex, the images of the (few) items that cover it — no global float atomics at all.
  float* part = partials + (int64_t)blockIdx.x * lds_floats;
  for (int i = tid; i < used; i += kTB)
    part[i] = finite ? (float)((double)(long long)acc64[i] * inv) : __int_as_float(0x7fc00000);
}

// ---------------------------------------------------------------------------------------------- pixel stage, LEVEL-INTERLEAVED images
// Round 3.  The kernels above place the sub-grids of the staged levels back to back in LDS; with one lane per (pixel, level)
// the 16 level-lanes of a pixel then hit 16 unrelated LDS columns per instruction: 70 % (forward) / 72 % (backward) of the
// kernels' LDS cycles were bank conflicts (PMC, profiles/r02_pmc_sq_counters.json).  For F = 2 and <= 16 staged levels the
// images below are INTERLEAVED BY LEVEL instead: vertex i of level l lives in row i, column l of a [rows][16] array of 8-byte
// slots (forward: the vertex's two fp32 features; backward: one 64-bit fixed-point accumulator per feature, rows 2 i and
// 2 i + 1).  A 16-lane group — the 16 levels of one pixel — then touches 16 DIFFERENT 8-byte columns whatever the cells are:
// the write / atomic path (banked in 16-lane groups over 128 bytes, MI355X_MICROARCH.md section LDS) is conflict-free by
// construction, and the forward's ds_read_b64 (32-lane groups over 256 bytes) sees at most a 2-way conflict.
// Cost: the image is as tall as the finest level for every column (46 KB forward, 92 KB backward at N = 512 and 32 x 32
// tiles instead of 10 / 20 KB), so one 1024-thread workgroup per CU.  Same items, same partial-image format, same results.
constexpr int kIL = 16;             // columns = levels per interleaved row

// worst-case vertex count of one level's sub-grid in a tile (setup_tile: wx, wy <= (n >> tile_shift) + 3)
static int interleaved_rows(const int32_t* n_ls_host, int Ls, int tile_shift) {
  int m = 0;
  for (int l = 0; l < Ls; ++l) { const int w = (n_ls_host[l] >> tile_shift) + 3; m = w * w > m ? w * w : m; }
  return m;
}

// private accumulator copies of a level with `rows_l` rows in a column of `rows2` rows: 4, 2 or 1
__device__ __forceinline__ int il_copies(int rows_l, int rows2) {
  return rows_l * 4 <= rows2 ? 4 : (rows_l * 2 <= rows2 ? 2 : 1);
}

typedef float v2f __attribute__((ext_vector_type(2)));

// diagnostic (-DGNGF_STAMPS builds only, as csrc/decoder.hip): per-phase cycle totals of workgroup 0 of the interleaved backward
// kernel (tools/perf_tiled_il.py).  The production build carries no stamps and no read-modify-writes of this array.
#ifdef GNGF_STAMPS
__device__ unsigned long long g_il_stamps[8];
#define IL_STAMP_INIT() unsigned long long il_t = __builtin_readcyclecounter()
#define IL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    g_il_stamps[k] += t_ - il_t; il_t = t_; } } while (0)
#define IL_STAMP_ITEM(px) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_il_stamps[6] += 1; g_il_stamps[7] += (unsigned long long)(px); } } while (0)
#else
#define IL_STAMP_INIT() do {} while (0)
#define IL_STAMP(k) do {} while (0)
#define IL_STAMP_ITEM(px) do {} while (0)
#endif

// Per-item geometry of the interleaved kernels, filled by the first 16 lanes of the workgroup (one level each; the prefix
// sums meet in shuffles — setup_tile's serial walk over the levels cost ~0.5 us per item) and read by everybody after one
// barrier.  `next`: the item the workgroup takes after this one (persistent workgroups, see il_next_item).
struct ILMeta {
  int n[kIL], cx[kIL], cy[kIL], wx[kIL], wy[kIL], loff[kIL], copies[kIL];
  int64_t goff[kIL];
  int nls[kIL];                 // the level resolutions (the same for every item: read from global memory once per workgroup)
  int used, rows_used;
};

__device__ __forceinline__ void il_setup(ILMeta& m, int Ls, int tx, int ty, int tile_shift, int rows2) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    int n = 0, cx = 0, cy = 0, wx = 0, wy = 0;
    if (tid < Ls) {
      n = m.nls[tid];
      cx = (tx * n) >> tile_shift; cy = (ty * n) >> tile_shift;
      int hx = (((tx + 1) * n) >> tile_shift) + 1, hy = (((ty + 1) * n) >> tile_shift) + 1;
      hx = hx > n + 1 ? n + 1 : hx;
      hy = hy > n + 1 ? n + 1 : hy;
      wx = hx - cx + 1; wy = hy - cy + 1;
    }
    const int sz = wx * wy * 2;                           // floats of the level in the compact image (F = 2)
    const int64_t g2 = tid < Ls ? (int64_t)(n + 2) * (n + 2) : 0;
    int lo = sz;                                          // inclusive prefix sums over the 16 level lanes
    int64_t go = g2;
#pragma unroll
    for (int o = 1; o < kIL; o <<= 1) {
      const int ul = __shfl_up(lo, o, 64);
      const int64_t ug = __shfl_up(go, o, 64);
      if ((tid & 15) >= o) { lo += ul; go += ug; }
    }
    const int copies = rows2 > 0 ? il_copies(sz, rows2) : 1;
    int tall = sz * copies, tot = __shfl(lo, kIL - 1, 64);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { const int t2 = __shfl_xor(tall, o, 64); tall = t2 > tall ? t2 : tall; }
    if (tid < kIL) {
      m.n[tid] = n; m.cx[tid] = cx; m.cy[tid] = cy; m.wx[tid] = wx; m.wy[tid] = wy;
      m.loff[tid] = lo - sz; m.goff[tid] = go - g2; m.copies[tid] = copies;
    }
    if (tid == 0) { m.used = tot; m.rows_used = tall; }
  }
}

// Persistent workgroups: workgroup b of `nwork` takes items b, b + nwork, ... (the items are about equally heavy: at most
// `chunk` pixels of one tile each); the next item's record is requested before the current item is processed.  (Claiming
// items through a global counter was tried: the atomic's round trip sat in front of every item's first barrier.)

// Where the vertex values of the staged sub-grids come from (template parameter SRC of the forward kernel):
//   0  the vertex grid G written by the vertex stage forward (vertex_fwd_kernel / the riders of the binning launches);
//   1  the level tables themselves, spatial-hash source:   G_l[v] = E_l[hash(v)]                 (models.py:504-528, 181-191)
//   2  the level tables, vertex-table source:              G_l[v] = sum_k w_k(v) E_l[idx_k(v)]   (models.py:193-222)
// With 1 and 2 the vertex stage forward is FUSED into the staging loop — the same separately rounded operations in the same
// order as vertex_fwd_lane, so enc is bit-identical — and its launch (14-18 us of L2 latency inside a step) and the G round
// trip disappear: the gathers of a work item are in flight together and hide among the two to three workgroups of a CU.
struct VertexSrc {
  const float* G;
  const float* tables;            // (L, T, 2) fp32
  const int32_t* vert_idx;        // (NV, K)
  const float* vert_w;            // (NV, K)
  int64_t T, NV;
  int K, vstride;
  bool pow2;
};

// The COUNT of the NEXT batch's binning riding on the forward launch: workgroups [0, NB) run bin_count_body on that batch's
// coordinates (histogram per block + the tile totals in global atomics of its persistent workspace) instead of a work item.
// The forward is bound by its 128 MiB of enc stores; the count reads 8 MiB and works in LDS atomics.  NB = 0: none.
typedef BinJobDev BinCountRide;   // (the reserving count: bin_count_reserve_body)

template <int SRC>
__device__ __forceinline__ v2f vertex_value(const VertexSrc& vs, int l, int gx, int gy, int64_t goff, int gw) {
  if constexpr (SRC == 0) {
    return reinterpret_cast<const v2f*>(vs.G)[goff + (int64_t)gy * gw + gx];
  } else if constexpr (SRC == 1) {
    return reinterpret_cast<const v2f*>(vs.tables)[(int64_t)l * vs.T + spatial_hash(gx, gy, vs.T, vs.pow2)];
  } else {
    v2f acc = {0.f, 0.f};
    const int64_t vid = (int64_t)gy * vs.vstride + gx;
    if (gx < vs.vstride && vid < vs.NV) {
      const v2f* tab = reinterpret_cast<const v2f*>(vs.tables) + (int64_t)l * vs.T;
      for (int k = 0; k < vs.K; ++k) {
        const float w = vs.vert_w[vid * vs.K + k];
        acc = acc + tab[vs.vert_idx[vid * vs.K + k]] * w;            // (vertex_fwd_lane: acc[f] += row[f] * w, k ascending)
      }
    }
    return acc;
  }
}

template <bool L16, int U, int SRC = 0>
__global__ void __launch_bounds__(kTBF)
tiled_fwd_il_kernel(const float4* __restrict__ sorted, const int4* __restrict__ items, const int32_t* __restrict__ n_items,
                    int32_t* __restrict__ counter, const int32_t* __restrict__ n_ls, const VertexSrc vs,
                    float* __restrict__ enc, int L, int Ls, int tile_shift, int nwork, const BinCountRide cride) {
  constexpr int F = 2;
  extern __shared__ float2 img_raw[];             // [rows][kIL]: (feature 0, feature 1) of vertex `row` of level `column`
  v2f* img = reinterpret_cast<v2f*>(img_raw);
  __shared__ ILMeta m;
  // The riders come LAST in the grid.  At the headline shape the 670 work items fit the chip at once (three workgroups per CU:
  // 768 places), so the launch lasts as long as its slowest item; riders placed first took 128 of those places and sent 30 items
  // into a second round (41.3 us), riders placed last take the ~100 free places at once and the rest as items retire (40.1 us).
  if ((int)blockIdx.x >= nwork) {
    const int rb = (int)blockIdx.x - nwork;
    if (rb == 0 && threadIdx.x == 0) cride.pws[2 * (1 << (2 * cride.tile_shift)) + 1] = 0;      // the scatter riders' task counter
    bin_count_reserve_body<kTBF>(rb, cride, reinterpret_cast<int*>(img_raw));
    return;
  }
  const int wg = (int)blockIdx.x;
  const int tid = threadIdx.x;
  const int nit = *n_items;
  const int TSm = (1 << tile_shift) - 1;
  constexpr int ppp = kTBF / kIL;                 // pixels per pass: lane = 16 * pixel + level
  const int lp = tid >> 4, l = tid & 15;
  const int LF = L16 ? 32 : L * F, LF2 = LF / 2;
  if (tid < kIL) m.nls[tid] = tid < Ls ? n_ls[tid] : 0;
  int4 it_next = items[wg < nit ? wg : 0];
  for (int item = wg; item < nit; item += nwork) {
    const int4 it = it_next;
    it_next = items[item + nwork < nit ? item + nwork : item];
    __syncthreads();                              // (the previous item's image is no longer read; m.nls is there)
    il_setup(m, Ls, it.z & TSm, it.z >> tile_shift, tile_shift, 0);
    __syncthreads();
    {
      // Staging: one vertex per thread and level, ALL the loads issued before the first LDS store (a load-store loop per level
      // exposes one memory round trip per level: 16 per item).  The row of a vertex comes from a float reciprocal —
      // (i + 0.5) / wx is at least 0.5 / wx away from an integer, far more than fp32 rounding moves it.
      // Flattened over (level, vertex) — element e of the compact image — three elements per thread and pass, their loads in
      // flight together; the level of e by a 4-step binary search over the compact starts (m.loff / 2, non-decreasing).
      const int used_v = m.used / F;
      auto locate = [&](int e, int& lv, int& i, int& gx, int& gy) {
        int q = (e >= (m.loff[8] >> 1) && 8 < Ls) ? 8 : 0;
        if (q + 4 < Ls && e >= (m.loff[q + 4] >> 1)) q += 4;
        if (q + 2 < Ls && e >= (m.loff[q + 2] >> 1)) q += 2;
        if (q + 1 < Ls && e >= (m.loff[q + 1] >> 1)) q += 1;
        lv = q;
        i = e - (m.loff[q] >> 1);
        const int wx = m.wx[q];
        const int iy = (int)(((float)i + 0.5f) * (1.0f / (float)wx)), ix = i - iy * wx;
        gx = m.cx[q] + ix; gy = m.cy[q] + iy;
      };
      for (int base = 0; base < used_v; base += 3 * kTBF) {
        int lv[3], iv[3], gx[3], gy[3];
        v2f val[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int e = base + k * kTBF + tid;
          locate(e < used_v ? e : used_v - 1, lv[k], iv[k], gx[k], gy[k]);
        }
        if constexpr (SRC == 2) {
          if (vs.K == 4) {
            // the usual K: the four (slot, weight) pairs of a vertex are one 16-byte load each, and all twelve table rows of
            // the thread's three vertices are requested before the first is used
            int4 id[3];
            float4 ww[3];
            bool ok[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const int64_t vid = (int64_t)gy[k] * vs.vstride + gx[k];
              ok[k] = gx[k] < vs.vstride && vid < vs.NV;
              const int64_t v = ok[k] ? vid : 0;
              id[k] = reinterpret_cast<const int4*>(vs.vert_idx)[v];
              ww[k] = reinterpret_cast<const float4*>(vs.vert_w)[v];
            }
            v2f r[3][4];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const v2f* tab = reinterpret_cast<const v2f*>(vs.tables) + (int64_t)lv[k] * vs.T;
              r[k][0] = tab[id[k].x]; r[k][1] = tab[id[k].y]; r[k][2] = tab[id[k].z]; r[k][3] = tab[id[k].w];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              v2f acc = {0.f, 0.f};
              acc = acc + r[k][0] * ww[k].x;
              acc = acc + r[k][1] * ww[k].y;
              acc = acc + r[k][2] * ww[k].z;
              acc = acc + r[k][3] * ww[k].w;
              val[k] = ok[k] ? acc : (v2f){0.f, 0.f};
            }
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) val[k] = vertex_value<2>(vs, lv[k], gx[k], gy[k], 0, 0);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) val[k] = vertex_value<SRC>(vs, lv[k], gx[k], gy[k], m.goff[lv[k]], m.n[lv[k]] + 2);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (base + k * kTBF + tid < used_v) img[iv[k] * kIL + lv[k]] = val[k];
      }
    }
    __syncthreads();
    if (l < Ls) {
      const int n = m.n[l], cx = m.cx[l], cy = m.cy[l], wx = m.wx[l], wy = m.wy[l], gw = n + 2;
      const int64_t goff_l = m.goff[l];
      const v2f* col = img + l;                   // this level's column
      v2f* enc_l = reinterpret_cast<v2f*>(enc) + l;
      const float fn = (float)n;
      const float4* rec = sorted + it.x;
      const int last = it.y - 1;
      auto fetch = [&](int j) { return rec[j < last ? j : last]; };
      // one pixel outside the fast path (a pixel outside its tile's staged sub-grid: never for coordinates in [0,1]^2)
      auto slow = [&](const float4 sv, int j) {
        if (j > last) return;
        const Cell cc = make_cell(sv.x, sv.y, n);
        const int lx = cc.gx - cx, ly = cc.gy - cy;
        float vv[4][F];
        if (lx >= 0 && ly >= 0 && lx + 1 < wx && ly + 1 < wy) {
          const v2f* a = col + (ly * wx + lx) * kIL;
          const v2f* b = a + wx * kIL;
          vv[0][0] = a[0].x; vv[0][1] = a[0].y; vv[1][0] = a[kIL].x; vv[1][1] = a[kIL].y;
          vv[2][0] = b[0].x; vv[2][1] = b[0].y; vv[3][0] = b[kIL].x; vv[3][1] = b[kIL].y;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int vx = cc.gx + (q & 1), vy = cc.gy + (q >> 1);
            vx = vx < 0 ? 0 : (vx > n + 1 ? n + 1 : vx);
            vy = vy < 0 ? 0 : (vy > n + 1 ? n + 1 : vy);
            const v2f g = vertex_value<SRC>(vs, l, vx, vy, goff_l, gw);
            vv[q][0] = g.x; vv[q][1] = g.y;
          }
        }
        float* o = enc + (int64_t)__float_as_int(sv.z) * LF + l * F;
#pragma unroll
        for (int f = 0; f < F; ++f) o[f] = ((vv[0][f] * cc.c[0] + vv[1][f] * cc.c[1]) + vv[2][f] * cc.c[2]) + vv[3][f] * cc.c[3];
      };
      // U pixels per trip: their 4 U LDS reads are in flight together, and a wave-uniform vote keeps the common case — every
      // pixel inside its staged sub-grid — free of per-lane branches.  Pairs of fp32 (coordinates, the two features) go through
      // packed instructions: same separately rounded operations as make_cell / the reference, half the instructions.
      // (U = 2: 72 VGPRs, three workgroups per CU: 33 us at 2^20 pixels; U = 4: 92 VGPRs, two workgroups: 35 us.)
      // (TAIL = false: every pixel of the trip exists — no clamped record index, no predicated store; the one trip that runs
      // past the item's end is peeled off below)
      auto trip = [&](const int j0, auto TAIL) {
        constexpr bool tail = decltype(TAIL)::value;
        float4 sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sv[u] = tail ? fetch(j0 + u * ppp) : rec[j0 + u * ppp];
        float c[U][4];
        int v[U];
        bool inside = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const v2f sxy = (v2f){sv[u].x, sv[u].y} * fn;
          const v2f a = __builtin_elementwise_floor(sxy);
          const v2f d = a + 1.0f;
          const v2f w0 = d - sxy, w1 = sxy - a;
          c[u][0] = w0.x * w0.y; c[u][1] = w1.x * w0.y; c[u][2] = w0.x * w1.y; c[u][3] = w1.x * w1.y;
          const int lx = (int)a.x - cx, ly = (int)a.y - cy;
          inside = inside && ((unsigned)lx < (unsigned)(wx - 1)) && ((unsigned)ly < (unsigned)(wy - 1));
          v[u] = __mul24(ly, wx) + lx;                      // (24-bit multiply: the plain product became a quarter-rate v_mad_u64_u32)
        }
        if (__ballot(!inside) == 0ull) {
          v2f a0[U], a1[U], b0[U], b1[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const v2f* a = col + v[u] * kIL;
            const v2f* b = a + wx * kIL;
            a0[u] = a[0]; a1[u] = a[kIL]; b0[u] = b[0]; b1[u] = b[kIL];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const v2f r = ((a0[u] * c[u][0] + a1[u] * c[u][1]) + b0[u] * c[u][2]) + b1[u] * c[u][3];
            if (!tail || j0 + u * ppp <= last) enc_l[(int64_t)__float_as_int(sv[u].z) * LF2] = r;
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) slow(sv[u], j0 + u * ppp);
        }
      };
      int j0 = lp;
      for (; j0 + (U - 1) * ppp <= last; j0 += U * ppp) trip(j0, std::false_type{});
      if (j0 <= last) trip(j0, std::true_type{});
    }
  }
}

// Backward with the level-interleaved accumulator image acc[rows][16] (row 2 i + f = feature f of vertex i, column = level):
// the eight ds_add_u64 of a (pixel, level) lane go to base + {0, 128, 256, 384} bytes of two row pairs — immediate offsets —
// and every 16-lane group (one pixel, 16 levels) hits 16 distinct columns.  The four pixels of a wave-instruction still meet on
// the SAME address whenever they share a cell — the rule at the coarse levels, where a tile covers a handful of cells, and an
// LDS atomic serialises on equal addresses (tools/micro/lds_atomic64_patterns.cpp) — so a level whose sub-grid is small
// enough keeps up to four private copies of its accumulators stacked in its column (copy = pixel slot of the lane; the column
// is as tall as the finest level anyway, so the copies are free) and the store pass adds them up as integers.
// Fixed point, scale, store pass and riders exactly as tiled_bwd_kernel: the partial image leaves bit-identical, in the same
// compact (level-after-level) format, so gather_partials is shared.  Workgroups [0, nwork) are persistent.
// dG64 (needs the bound on |genc| handed over by its producer — ONE scale for the whole launch; log2_chunk then bounds the
// pixels of the whole batch, not of an item): the store pass adds the item's 64-bit sums straight into a fixed-point vertex
// grid with global integer atomics (fire and forget: they drain while other workgroups compute) — no partial images, no gather
// pass; the sums are exact and order-free, so the vertex-grid gradient is bitwise reproducible.  dG64[vtot * F] = the scale S,
// dG64[vtot * F + 1] != 0: poisoned (non-finite gradient or broken promise) — read by the kernels that turn dG64 into fp32.
// The SCATTER of the NEXT batch's binning (bin_scatter2_body; its count rode on the forward launch: BinCountRide) as tasks the
// persistent workgroups claim from a counter once their own work items are done.  The items are about equally heavy and there
// are 2.6 of them per workgroup at the headline shape, so two workgroups in five finish a third of the launch early: the
// tasks run in that hole instead of in two launches of their own at the head of the next step (binning depends on the
// coordinates only, and the batches of an epoch are fixed slices of one permutation, known in advance: functions.py:186-194).
// No task waits for another one (the totals were completed by the previous launch); NB = 0: none.
typedef BinJobDev BinScatterRide;   // (bin_scatter3_body)

// HDT (spatial-hash index source on a single rank, bound on |genc| given: round 5): the vertex stage backward is GONE.  The store
// pass converts the item's exact 64-bit sums to fp32 (one rounding) and adds them straight to row hash(gx, gy) of the level's
// table gradient with fire-and-forget float atomics — the same number of memory-side atomic requests as the adds into the
// fixed-point vertex grid they replace (one per non-zero vertex of the item's sub-grids), no fixed-point grid to clear, no
// vertex_bwd_hash64 launch behind the kernel (10.7 us of a 396 us step).  What is given up: a vertex shared by several items (tile
// borders; every vertex of the coarse levels, whose cells span several tiles) now receives one fp32 add per item instead of one
// exact sum, so a table row is an fp32 sum of up to ~16 exactly rounded partial sums (was: one per vertex that hashes to it).
template <bool L16, bool HDT = false>
__global__ void __launch_bounds__(kTB)
tiled_bwd_il_kernel(const float4* __restrict__ sorted, const int4* __restrict__ items, const int32_t* __restrict__ n_items,
                    int32_t* __restrict__ counter, const int32_t* __restrict__ n_ls, const float* __restrict__ genc,
                    float* __restrict__ dG, float* __restrict__ partials, const float* __restrict__ gmax_hint, int hint_count,
                    int hint_stride, int L, int Ls, int tile_shift, int lds_floats, int rows2, int log2_chunk, int nwork,
                    RideAlong ride, MseRide mride, unsigned long long* __restrict__ dG64, const BinScatterRide bride,
                    float* __restrict__ hash_dt = nullptr, int64_t hash_T = 0, bool hash_pow2 = false) {
  constexpr int F = 2;
  extern __shared__ unsigned long long accil[];   // [rows2][kIL], then the compact fp32 image of the store pass
  __shared__ ILMeta m;
  __shared__ float wmax[kTB / 64];
  __shared__ int s_task;
  if (mride.pred && (int)blockIdx.x >= mride.first_block) {
    mse_sum_block((int)blockIdx.x - mride.first_block, mride.nblocks, mride.pred, mride.label, mride.loss, mride.acc, mride.counter,
                  mride.n);
    return;
  }
  if (ride.slabs && (int)blockIdx.x >= ride.first_block) {
    decoder_reduce_block((int)blockIdx.x - ride.first_block, ride.slabs, ride.nslabs, ride.nslab, ride.in_dim, ride.out_dim, ride.dW0,
                         ride.db0, ride.dW1, ride.db1, ride.dW2, ride.db2, nullptr, ride.promised, ride.arrived);
    return;
  }
  const int tid = threadIdx.x;
  const int nit = *n_items;
  const int TSm = (1 << tile_shift) - 1;
  constexpr int ppp = kTB / kIL;
  const int lp = tid >> 4, l = tid & 15;
  const bool lane_on = l < Ls;
  const int LF = L16 ? 32 : L * F, LF2 = LF / 2;
  const v2f* genc_l = reinterpret_cast<const v2f*>(genc) + l;
  float* cimg = reinterpret_cast<float*>(accil + (size_t)rows2 * kIL);
  const bool broken = promise_broken(ride.promised, ride.arrived);
  // the bound on |genc| handed over by its producer (one value per decoder slab): the same for every item
  float hint = 0.f;
  if (gmax_hint) {
    for (int q = tid; q < hint_count; q += kTB) { const float a = gmax_hint[(int64_t)q * hint_stride]; hint = (a > hint || a != a) ? a : hint; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(hint, o, 64); hint = (ov > hint || ov != ov) ? ov : hint; }
    if ((tid & 63) == 0) wmax[tid >> 6] = hint;
    __syncthreads();
    hint = wmax[0];
#pragma unroll
    for (int w = 1; w < kTB / 64; ++w) hint = (wmax[w] > hint || wmax[w] != wmax[w]) ? wmax[w] : hint;
  }
  if (tid < kIL) m.nls[tid] = tid < Ls ? n_ls[tid] : 0;
  IL_STAMP_INIT();
  int4 it_next = items[(int)blockIdx.x < nit ? (int)blockIdx.x : 0];
  for (int item = blockIdx.x; item < nit; item += nwork) {
    const int4 it = it_next;
    it_next = items[item + nwork < nit ? item + nwork : item];
    __syncthreads();                              // the previous item's compact image has left; m.nls is there
    IL_STAMP(0);
    il_setup(m, Ls, it.z & TSm, it.z >> tile_shift, tile_shift, rows2);
    __syncthreads();
    IL_STAMP(1);
    const int used = m.used, rows_used = m.rows_used;
    {
      ulonglong2* z = reinterpret_cast<ulonglong2*>(accil);
      const ulonglong2 zero = {0ull, 0ull};
      for (int i = tid; i < rows_used * (kIL / 2); i += kTB) z[i] = zero;
    }
    const float4* rec = sorted + it.x;
    const int last = it.y - 1;
    auto fetch = [&](int j) { return rec[j < last ? j : last]; };
    auto grad_of = [&](const float4 sv) { return genc_l[(int64_t)__float_as_int(sv.z) * LF2]; };
    constexpr int U = 4;
    // pass 1: largest |gradient| over the item's pixels — skipped when the producer of genc handed over a bound
    float gmax = hint;
    if (!gmax_hint) {
      gmax = 0.f;
      if (lane_on)
        for (int j0 = lp; j0 <= last; j0 += U * ppp) {
          v2f gv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) gv[u] = grad_of(fetch(j0 + u * ppp));
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float a = fabsf(gv[u].x), b = fabsf(gv[u].y);
            gmax = (a > gmax || a != a) ? a : gmax;
            gmax = (b > gmax || b != b) ? b : gmax;                    // NaN sticks
          }
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(gmax, o, 64); gmax = (ov > gmax || ov != ov) ? ov : gmax; }
      if ((tid & 63) == 0) wmax[tid >> 6] = gmax;
      __syncthreads();
      gmax = wmax[0];
#pragma unroll
      for (int w = 1; w < kTB / 64; ++w) gmax = (wmax[w] > gmax || wmax[w] != wmax[w]) ? wmax[w] : gmax;
    }
    __syncthreads();                              // the image is zero (and wmax is free again)
    IL_STAMP(2);
    const bool finite = gmax < INFINITY && !broken;        // false for inf and NaN
    int eg = 0;
    if (finite && gmax > 0.f) (void)frexpf(gmax, &eg);       // gmax < 2^eg
    // fma form: every term must stay below 2^51 — |g| 2^S < 2^(60 - log2_chunk), so a bound of fewer than 2^10 terms counts as 2^10
    int S = GNGF_FIXED_FMA ? 60 - (log2_chunk < 10 ? 10 : log2_chunk) - eg : 61 - log2_chunk - eg;
    S = S > 126 ? 126 : S;                                               // 2^S stays an fp32 number for any gradient magnitude
    if (dG64 && tid == 0) {
      const int64_t vt = m.goff[Ls - 1] + (int64_t)(m.n[Ls - 1] + 2) * (m.n[Ls - 1] + 2);
      if (blockIdx.x == 0 && item == (int)blockIdx.x) dG64[vt * F] = (unsigned long long)(long long)S;
      if (!finite) dG64[vt * F + 1] = 1ull;
    }
    if (lane_on && finite) {
      const int n = m.n[l], cx = m.cx[l], cy = m.cy[l], wx = m.wx[l], wy = m.wy[l], gw = n + 2;
      float* dGl = dG + m.goff[l] * F;
      const int rows_l = wx * wy * F;
      unsigned long long* col = accil + l + ((lp & (m.copies[l] - 1)) * rows_l) * kIL;
      const float fn = (float)n;
      const float scale = ldexpf(1.0f, GNGF_FIXED_FMA ? S : S - 32);   // (a power of two: scaling by it is exact, as ldexp)
      // one fixed-point term g * c (g scaled by 2^S), the same value on every path a pixel can take
      auto term = [&](float g, float c) -> unsigned long long {
#if GNGF_FIXED_FMA
        return to_fixed_fma((double)g, (double)c);
#else
        return (unsigned long long)to_fixed_scaled(g * c);
#endif
      };
      // the eight terms of one (pixel, level): corners (a, a + 1) of row pa and of row pb, two features each
      auto add8 = [&](unsigned long long* pa, unsigned long long* pb, const v2f g, float c0, float c1, float c2, float c3) {
#if GNGF_FIXED_FMA
        const double gx = (double)g.x, gy = (double)g.y, d0 = (double)c0, d1 = (double)c1, d2 = (double)c2, d3 = (double)c3;
        atomicAdd(pa, to_fixed_fma(gx, d0));
        atomicAdd(pa + kIL, to_fixed_fma(gy, d0));
        atomicAdd(pa + 2 * kIL, to_fixed_fma(gx, d1));
        atomicAdd(pa + 3 * kIL, to_fixed_fma(gy, d1));
        atomicAdd(pb, to_fixed_fma(gx, d2));
        atomicAdd(pb + kIL, to_fixed_fma(gy, d2));
        atomicAdd(pb + 2 * kIL, to_fixed_fma(gx, d3));
        atomicAdd(pb + 3 * kIL, to_fixed_fma(gy, d3));
#else
        const v2f t0 = g * c0, t1 = g * c1, t2 = g * c2, t3 = g * c3;      // the terms g*c: one fp32 multiply each, as the reference's
        atomicAdd(pa, (unsigned long long)to_fixed_scaled(t0.x));
        atomicAdd(pa + kIL, (unsigned long long)to_fixed_scaled(t0.y));
        atomicAdd(pa + 2 * kIL, (unsigned long long)to_fixed_scaled(t1.x));
        atomicAdd(pa + 3 * kIL, (unsigned long long)to_fixed_scaled(t1.y));
        atomicAdd(pb, (unsigned long long)to_fixed_scaled(t2.x));
        atomicAdd(pb + kIL, (unsigned long long)to_fixed_scaled(t2.y));
        atomicAdd(pb + 2 * kIL, (unsigned long long)to_fixed_scaled(t3.x));
        atomicAdd(pb + 3 * kIL, (unsigned long long)to_fixed_scaled(t3.y));
#endif
      };
      auto stage = [&](const float4 sv, const v2f gv, int j) {
        if (j > last) return;
        const v2f sxy = (v2f){sv.x, sv.y} * fn;   // make_cell, on pairs: every operation separately rounded as in the reference
        const v2f a = __builtin_elementwise_floor(sxy);
        const v2f d = a + 1.0f;
        const v2f w0 = d - sxy, w1 = sxy - a;
        const float c0 = w0.x * w0.y, c1 = w1.x * w0.y, c2 = w0.x * w1.y, c3 = w1.x * w1.y;
        const int gx = (int)a.x, gy = (int)a.y;
        const int lx = gx - cx, ly = gy - cy;
        if (((unsigned)lx < (unsigned)(wx - 1)) && ((unsigned)ly < (unsigned)(wy - 1))) {
          unsigned long long* pa = col + (ly * wx + lx) * (F * kIL);
          unsigned long long* pb = pa + wx * (F * kIL);
          const v2f g = gv * scale;
          add8(pa, pb, g, c0, c1, c2, c3);
        } else {                                   // outside the staged sub-grid (never for in-domain coordinates): global
          const float cq[4] = {c0, c1, c2, c3};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            int vx = gx + (q & 1), vy = gy + (q >> 1);
            vx = vx < 0 ? 0 : (vx > n + 1 ? n + 1 : vx);
            vy = vy < 0 ? 0 : (vy > n + 1 ? n + 1 : vy);
            if constexpr (HDT) {
              float* d = hash_dt + ((int64_t)l * hash_T + spatial_hash(vx, vy, hash_T, hash_pow2)) * F;
              atomicAdd(d, gv.x * cq[q]);
              atomicAdd(d + 1, gv.y * cq[q]);
            } else if (dG64) {
              unsigned long long* d = dG64 + (m.goff[l] + (int64_t)vy * gw + vx) * F;
              atomicAdd(d, term(gv.x * scale, cq[q]));
              atomicAdd(d + 1, term(gv.y * scale, cq[q]));
            } else {
              atomicAdd(dGl + ((int64_t)vy * gw + vx) * F, gv.x * cq[q]);
              atomicAdd(dGl + ((int64_t)vy * gw + vx) * F + 1, gv.y * cq[q]);
            }
          }
        }
      };
      // fast path of one pixel, no branches: every lane is inside its staged sub-grid (voted per trip) and a pixel past the end
      // of the item arrives with a zero gradient (it adds zeros to the cell of the item's last pixel)
      auto fast = [&](const v2f a, const v2f w0, const v2f w1, const int v, const v2f gv) {
        const float c0 = w0.x * w0.y, c1 = w1.x * w0.y, c2 = w0.x * w1.y, c3 = w1.x * w1.y;
        unsigned long long* pa = col + v * (F * kIL);
        unsigned long long* pb = pa + wx * (F * kIL);
        add8(pa, pb, gv * scale, c0, c1, c2, c3);
      };
      // four pixels per trip, loads issued together.  (Requesting the records two trips and the gradient rows one trip ahead
      // made the first wave finish earlier and the item no sooner: the phase is bound by VALU issue plus the LDS atomic unit,
      // 7.6 cycles per conflict-free ds_add_u64, not by memory latency.)
      // TAIL = false: all four pixels of the trip exist (no clamped record index, no zeroed gradient: the phase is bound by
      // VALU issue, four instructions per pixel matter); the one trip that runs past the item's end is peeled off below
      auto trip = [&](const int j0, auto TAIL) {
        constexpr bool tail = decltype(TAIL)::value;
        float4 sv[U];
        v2f gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sv[u] = tail ? fetch(j0 + u * ppp) : rec[j0 + u * ppp];
#pragma unroll
        for (int u = 0; u < U; ++u) gv[u] = grad_of(sv[u]);
        v2f fa[U], fw0[U], fw1[U];
        int fv[U];
        bool inside = true;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const v2f sxy = (v2f){sv[u].x, sv[u].y} * fn;     // make_cell, on pairs: every operation separately rounded as in the reference
          fa[u] = __builtin_elementwise_floor(sxy);
          const v2f d = fa[u] + 1.0f;
          fw0[u] = d - sxy; fw1[u] = sxy - fa[u];
          const int lx = (int)fa[u].x - cx, ly = (int)fa[u].y - cy;
          inside = inside && ((unsigned)lx < (unsigned)(wx - 1)) && ((unsigned)ly < (unsigned)(wy - 1));
          fv[u] = __mul24(ly, wx) + lx;                     // (24-bit multiply: the plain product became a quarter-rate v_mad_u64_u32)
          if (tail && j0 + u * ppp > last) gv[u] = (v2f){0.f, 0.f};
        }
        if (__ballot(!inside) == 0ull) {
#pragma unroll
          for (int u = 0; u < U; ++u) fast(fa[u], fw0[u], fw1[u], fv[u], gv[u]);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) stage(sv[u], gv[u], j0 + u * ppp);
        }
      };
      int j0 = lp;
      for (; j0 + (U - 1) * ppp <= last; j0 += U * ppp) trip(j0, std::false_type{});
      if (j0 <= last) trip(j0, std::true_type{});
    }
    IL_STAMP(3);
    __syncthreads();
    IL_STAMP(4);
    // Store pass.  The lanes keep their (pixel slot, level) roles: lane (r, l) converts rows r, r + 64, ... of column l — 16
    // consecutive lanes read 16 different columns, no bank conflicts — and drops the fp32 value into a COMPACT image (levels
    // back to back, the format tiled_bwd_kernel writes and gather_partials reads) behind the accumulators; that image then
    // leaves with coalesced 16-byte stores.
    if constexpr (HDT) {
      // Straight into the table gradient (see the kernel's header): flattened over the compact image as below; a non-finite
      // gradient (or a broken promise) poisons every row the item's sub-grids reach
      const double inv = finite ? ldexp(1.0, -S) : 0.0;
      for (int e = tid; e < used; e += kTB) {
        int q = (8 < Ls && e >= m.loff[8]) ? 8 : 0;
        if (q + 4 < Ls && e >= m.loff[q + 4]) q += 4;
        if (q + 2 < Ls && e >= m.loff[q + 2]) q += 2;
        if (q + 1 < Ls && e >= m.loff[q + 1]) q += 1;
        const int i = e - m.loff[q];
        const int wx = m.wx[q], rows_l = wx * m.wy[q] * F, copies = m.copies[q];
        unsigned long long sum = accil[i * kIL + q];
        for (int c = 1; c < copies; ++c) sum += accil[(c * rows_l + i) * kIL + q];
        if (sum != 0ull || !finite) {
          const int v = i >> 1;
          const int iy = (int)(((float)v + 0.5f) * (1.0f / (float)wx)), ix = v - iy * wx;
          float* r = hash_dt + ((int64_t)q * hash_T + spatial_hash(m.cx[q] + ix, m.cy[q] + iy, hash_T, hash_pow2)) * F + (i & 1);
          atomicAdd(r, finite ? (float)((double)(long long)sum * inv) : __int_as_float(0x7fc00000));
        }
      }
      IL_STAMP(5);
      IL_STAMP_ITEM(it.y);
      continue;
    }
    if (dG64) {
      // Straight into the fixed-point vertex grid, FLATTENED over the compact image (element e = row i of level lv: feature
      // i & 1 of vertex i >> 1 of the level's sub-grid): two to three elements per thread instead of up to eleven trips in
      // which only the finest levels' lanes still had rows left (the lanes' (pixel slot, level) roles leave 78 % of the trips
      // empty: 6.0 k -> ~3 k cycles per item; the column reads of neighbouring rows conflict, but there are few of them).
      if (finite) {
        for (int e = tid; e < used; e += kTB) {
          int q = (8 < Ls && e >= m.loff[8]) ? 8 : 0;                       // level of e: binary search over the compact starts
          if (q + 4 < Ls && e >= m.loff[q + 4]) q += 4;
          if (q + 2 < Ls && e >= m.loff[q + 2]) q += 2;
          if (q + 1 < Ls && e >= m.loff[q + 1]) q += 1;
          const int i = e - m.loff[q];
          const int wx = m.wx[q], rows_l = wx * m.wy[q] * F, copies = m.copies[q], gw = m.n[q] + 2;
          unsigned long long sum = accil[i * kIL + q];
          for (int c = 1; c < copies; ++c) sum += accil[(c * rows_l + i) * kIL + q];
          if (sum != 0ull) {
            const int v = i >> 1;
            const int iy = (int)(((float)v + 0.5f) * (1.0f / (float)wx)), ix = v - iy * wx;
            atomicAdd(dG64 + (m.goff[q] + (int64_t)(__mul24(m.cy[q] + iy, gw) + m.cx[q] + ix)) * F + (i & 1), sum);
          }
        }
      }
      IL_STAMP(5);
      IL_STAMP_ITEM(it.y);
      continue;
    }
    const double inv = finite ? ldexp(1.0, -S) : 0.0;
    if (lane_on) {
      const int rows_l = m.wx[l] * m.wy[l] * F, copies = m.copies[l], lo = m.loff[l];
      for (int i = lp; i < rows_l; i += ppp) {
        unsigned long long sum = accil[i * kIL + l];
        for (int c = 1; c < copies; ++c) sum += accil[(c * rows_l + i) * kIL + l];      // integer adds: any order, same bits
        cimg[lo + i] = finite ? (float)((double)(long long)sum * inv) : __int_as_float(0x7fc00000);
      }
    }
    __syncthreads();
    float* part = partials + (int64_t)item * lds_floats;
    if (((lds_floats | used) & 3) == 0) {
      const float4* c4 = reinterpret_cast<const float4*>(cimg);
      float4* p4 = reinterpret_cast<float4*>(part);
      for (int e = tid; e < used / 4; e += kTB) p4[e] = c4[e];
    } else {
      for (int e = tid; e < used; e += kTB) part[e] = cimg[e];
    }
    IL_STAMP(5);
    IL_STAMP_ITEM(it.y);
  }
  if (bride.NB > 0) {
    // the next batch's scatter tasks, claimed one at a time (the accumulator image is free: its LDS holds the task's cursors)
    static_assert(kTB == kBinThreads, "bin_scatter3_body is written for the binning kernels' workgroup size");
    int* sh = reinterpret_cast<int*>(accil);
    const int ntiles_b = 1 << (2 * bride.tile_shift);
    int32_t* claim = bride.pws + 2 * ntiles_b + 1;
    for (;;) {
      __syncthreads();                            // the previous task (or the last item's store pass) is done with the LDS
      if (tid == 0) s_task = atomicAdd(claim, 1);
      __syncthreads();
      const int task = s_task;
      if (task >= bride.NB) break;
      bin_scatter3_body(task, bride, sh);
    }
  }
}

// Gather pass: dG[(l, gx, gy)] += sum over the items whose tile sub-grid contains the vertex.
// grid = ceil(vtot / 256): one thread per (level, vertex), FLAT over the level grids (a (vertices of the finest level, Ls)
// grid launched 16.5 k workgroups at N = 512 of which 2.8 k had work: the empty ones cost more dispatch time than the kernel's
// memory round trips).  Re-derives each covering tile's LDS layout (same rule as setup_tile).
// dG64 (fixed point, scale at [nvals], poison flag at [nvals + 1]) -> dG (fp32): what the data-parallel exchange and the
// slot-ordered vertex backward read
__global__ void __launch_bounds__(256)
dg64_to_float_kernel(const unsigned long long* __restrict__ dG64, float* __restrict__ dG, int64_t nvals) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= nvals) return;
  const int S = (int)(long long)dG64[nvals];
  const bool poisoned = dG64[nvals + 1] != 0ull;
  dG[e] = poisoned ? __int_as_float(0x7fc00000) : (float)((double)(long long)dG64[e] * ldexp(1.0, -S));
}

// dG64 -> table gradient, spatial-hash index source (= vertex_bwd_kernel<F, false> reading the fixed-point grid)
template <int F>
__global__ void __launch_bounds__(256)
vertex_bwd_hash64_kernel(const unsigned long long* __restrict__ dG64, const int32_t* __restrict__ n_ls, float* __restrict__ dtables,
                         int Ls, int64_t T, bool pow2, int64_t vtot) {
  int l = 0, gw = n_ls[0] + 2;
  int64_t goff = 0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  while (l + 1 < Ls && e >= goff + (int64_t)gw * gw) { goff += (int64_t)gw * gw; ++l; gw = n_ls[l] + 2; }
  if (e - goff >= (int64_t)gw * gw) return;
  const int i = (int)(e - goff);
  const int gy = i / gw, gx = i - gy * gw;
  const int S = (int)(long long)dG64[vtot * F];
  const bool poisoned = dG64[vtot * F + 1] != 0ull;
  float* r = dtables + ((int64_t)l * T + spatial_hash(gx, gy, T, pow2)) * F;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const long long v = (long long)dG64[e * F + f];
    if (poisoned) atomicAdd(r + f, __int_as_float(0x7fc00000));
    else if (v != 0) atomicAdd(r + f, (float)((double)v * ldexp(1.0, -S)));
  }
}

// HASHFUSE: the vertex stage backward of the spatial-hash index source rides along — the summed gradient of vertex (l, gx, gy)
// goes straight to row hash(gx, gy) of the level's table gradient (float atomics, as vertex_bwd_kernel) and dG is not written:
// one launch and one round trip of the vertex-grid gradient less (single rank only: a data-parallel exchange needs dG).
template <int F, bool HASHFUSE = false>
__global__ void __launch_bounds__(256)
gather_partials_kernel(const float* __restrict__ partials, const int32_t* __restrict__ tile_item_base,
                       const int32_t* __restrict__ tile_level_off, const int32_t* __restrict__ n_ls, float* __restrict__ dG,
                       int Ls, int tile_shift, int lds_floats, float* __restrict__ dtables = nullptr, int64_t T = 0,
                       bool pow2 = false) {
  __shared__ int s_n[GNGF_MAX_LEVELS];
  if (threadIdx.x < Ls) s_n[threadIdx.x] = n_ls[threadIdx.x];
  __syncthreads();
  int l = 0;
  int64_t goff = 0;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (;;) {
    const int64_t g2 = (int64_t)(s_n[l] + 2) * (s_n[l] + 2);
    if (e < goff + g2 || l + 1 >= Ls) break;
    goff += g2;
    ++l;
  }
  const int n = s_n[l], gw = n + 2;
  if (e - goff >= (int64_t)gw * gw) return;
  const int i = (int)(e - goff);
  const int gy = i / gw, gx = i - gy * gw;
  const int TS = 1 << tile_shift;
  // tiles whose [cx, hx] range can contain gx:  cx(t) = (t*n)>>s <= gx   and   hx(t) = min(((t+1)*n>>s)+1, n+1) >= gx
  // (32-bit arithmetic: g << tile_shift < 2^20; a 64-bit division costs ~100 instructions, and there were four per vertex)
  auto lo_of = [&](int g) { int t = (int)((((unsigned)g_max0(g - 1) << tile_shift) + (unsigned)n - 1u) / (unsigned)n) - 1; return t < 0 ? 0 : t; };
  auto hi_of = [&](int g) { int t = (int)((((unsigned)(g + 1) << tile_shift) + (unsigned)n - 1u) / (unsigned)n) - 1; return t > TS - 1 ? TS - 1 : t; };
  const int tx0 = g_max0(lo_of(gx)), tx1 = hi_of(gx), ty0 = g_max0(lo_of(gy)), ty1 = hi_of(gy);
  float acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = 0.f;
  // (first item, one past the last item, float offset of this vertex in the items' images) of tile (tx, ty); empty when
  // the tile has no items, its image does not hold level l, or the vertex lies outside its sub-grid
  auto tile_span = [&](int tx, int ty, int& it0, int& it1, int& o) {
    const int t = (ty << tile_shift) | tx;
    it0 = tile_item_base[t];
    it1 = tile_item_base[t + 1];
    const int cx = (tx * n) >> tile_shift, cy = (ty * n) >> tile_shift;
    int hx = (((tx + 1) * n) >> tile_shift) + 1, hy = (((ty + 1) * n) >> tile_shift) + 1;
    hx = hx > n + 1 ? n + 1 : hx;
    hy = hy > n + 1 ? n + 1 : hy;
    const int wx = hx - cx + 1, wy = hy - cy + 1;
    int lo = 0;
    bool fits = false;
    if (tile_level_off) {
      lo = tile_level_off[t * Ls + l];
      fits = lo >= 0;
    } else {                                      // setup_tile's rule, re-derived
      for (int j = 0; j <= l; ++j) {
        const int nj = s_n[j];
        const int cxj = (tx * nj) >> tile_shift, cyj = (ty * nj) >> tile_shift;
        int hxj = (((tx + 1) * nj) >> tile_shift) + 1, hyj = (((ty + 1) * nj) >> tile_shift) + 1;
        hxj = hxj > nj + 1 ? nj + 1 : hxj;
        hyj = hyj > nj + 1 ? nj + 1 : hyj;
        const int sz = (hxj - cxj + 1) * (hyj - cyj + 1) * F;
        const bool fj = lo + sz <= lds_floats;
        if (j == l) fits = fj;
        else if (fj) lo += sz;
      }
    }
    const int lx = gx - cx, ly = gy - cy;
    if (!fits || lx < 0 || ly < 0 || lx >= wx || ly >= wy) it1 = it0;
    o = lo + (ly * wx + lx) * F;
  };
  auto add_item = [&](int it, int o) {
    const float* p = partials + (int64_t)it * lds_floats + o;
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] += p[f];
  };
  if (tx1 - tx0 <= 1 && ty1 - ty0 <= 1) {
    // The usual case, at most 2 x 2 covering tiles: every table read of the four tiles is issued before the first image
    // read, and the first two items of each tile are read without a loop (the chain tile -> item list -> image was a
    // serial walk of ~8 dependent memory round trips per vertex).
    int it0[4], it1[4], o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tx = tx0 + (q & 1), ty = ty0 + (q >> 1);
      if (tx <= tx1 && ty <= ty1) tile_span(tx, ty, it0[q], it1[q], o[q]);
      else { it0[q] = it1[q] = 0; o[q] = 0; }
    }
    float v[4][2][F];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const bool on = it0[q] + e < it1[q];
        const float* p = partials + (int64_t)(on ? it0[q] + e : 0) * lds_floats + (on ? o[q] : 0);
#pragma unroll
        for (int f = 0; f < F; ++f) v[q][e][f] = on ? p[f] : 0.f;
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] += v[q][e][f];
      for (int it = it0[q] + 2; it < it1[q]; ++it) add_item(it, o[q]);
    }
  } else {
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) {
        int it0, it1, o;
        tile_span(tx, ty, it0, it1, o);
        for (int it = it0; it < it1; ++it) add_item(it, o);
      }
  }
  float* d = dG + (goff + i) * F;
  if constexpr (HASHFUSE) {
    float* r = dtables + ((int64_t)l * T + spatial_hash(gx, gy, T, pow2)) * F;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const float g = acc[f] + d[f];                 // (d: what the out-of-sub-grid fallback added, normally 0)
      if (g != 0.f) atomicAdd(r + f, g);
    }
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) d[f] += acc[f];      // += : the out-of-sub-grid fallback may already have added (atomically, earlier kernel)
  }
}

// DPP lane movement (row_shr:n = 0x110+n inside 16-lane rows, row_bcast:15 = 0x142, row_bcast:31 = 0x143); lanes without a
// source keep `old` (0 for values, -1 for slot numbers)
template <int CTRL, int ROWMASK> __device__ __forceinline__ float dppf(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ int dppi(int x) {
  return __builtin_amdgcn_update_dpp(-1, x, CTRL, ROWMASK, 0xF, false);
}

// Vertex stage backward for the vertex-table index source, contention-free for ANY slot distribution (a freshly
// initialised HPD maps most vertices to a handful of slots): the (vertex,k) entries are visited in SLOT order
// (`order` = argsort of vert_idx), each lane takes one entry and walks the levels it belongs to, and equal slots —
// adjacent lanes — are combined with a wave-level segmented scan before the one atomic per (wave, run).
// dvert_w needs no atomics at all: every entry is owned by exactly one lane.
constexpr int kVB = 512;           // vertex_bwd_sorted workgroup: 8 waves share one atomic per (run, level); measured 1024: 35 us, 512: 26, 256: 27
// FROM64: the vertex-grid gradient is read as the 64-bit fixed-point grid the pixel stage accumulated (dG64: vtot * F words,
// then the scale S and the poison flag) and converted on the fly — the launch of dg64_to_float_kernel and one round trip of the
// grid less.
template <int F, typename TT, bool FROM64>
__global__ void __launch_bounds__(kVB)
vertex_bwd_sorted_kernel(const TT* __restrict__ tables, const int32_t* __restrict__ vert_idx, const float* __restrict__ vert_w,
                         const int32_t* __restrict__ order, const int32_t* __restrict__ n_ls, const float* __restrict__ dG,
                         float* __restrict__ dtables, float* __restrict__ dvert_w, int Ls, int64_t T, int K, int vstride,
                         int64_t NE, const unsigned long long* __restrict__ dG64, int64_t vtot) {
  constexpr int NW = kVB / 64;
  __shared__ int s_n[GNGF_MAX_LEVELS];
  __shared__ int64_t s_goff[GNGF_MAX_LEVELS];
  __shared__ int64_t s_first[NW], s_last[NW];               // (first level << 32 | slot) of each wave's first / last lane
  __shared__ int s_nruns[NW], s_lstart[NW];
  __shared__ float s_tail[2][NW][F];
  if (threadIdx.x < Ls) s_n[threadIdx.x] = n_ls[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) { int64_t o = 0; for (int l = 0; l < Ls; ++l) { s_goff[l] = o; o += (int64_t)(s_n[l] + 2) * (s_n[l] + 2); } }
  const int64_t j = (int64_t)blockIdx.x * kVB + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = j < NE;
  const int e = live ? order[j] : 0;
  const int slot = live ? vert_idx[e] : -1 - lane;          // dead lanes never merge with anything
  const float w = live ? vert_w[e] : 0.f;
  const int vid = e / K;
  const int gy = vid / vstride, gx = vid - gy * vstride;
  // first level this lane's vertex belongs to (levels ascend, so it belongs to every later one); the workgroup starts at
  // the smallest such level among its lanes — with the (first level, slot) visiting order that is mostly the lanes' own.
  int lmin = Ls;
  if (live) { const int mg = gx > gy ? gx : gy; lmin = 0; while (lmin < Ls && mg > s_n[lmin] + 1) ++lmin; }
  // Runs = maximal stretches of ADJACENT lanes with the same (first level, slot) key — the sort key of `order`.  The slot
  // value alone is not enough: slots restart at every level-group boundary, so the same slot can sit on both sides of a
  // boundary inside one wave (5, 9 | 5) and a value comparison would fold the first run into the third.  Run numbers
  // (prefix count of key changes) make the segment predicates exact for any visiting order.
  const int64_t key = ((int64_t)lmin << 32) | (uint32_t)slot;
  const int slot_up = __shfl_up(slot, 1, 64), lmin_up = __shfl_up(lmin, 1, 64);
  int rid = (lane == 0 || slot_up != slot || lmin_up != lmin) ? 1 : 0;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(rid, o, 64); if (lane >= o) rid += u; }
  const int rid_dn = __shfl_down(rid, 1, 64);
  // segment predicates of the DPP scan: does the lane 1/2/4/8 below in the 16-lane row, the last lane of the previous
  // row (rows 1, 3), lane 31 (rows 2, 3) belong to the same run?  (out-of-row sources read as -1: never equal, rid >= 1)
  const bool p1 = dppi<0x111, 0xF>(rid) == rid, p2 = dppi<0x112, 0xF>(rid) == rid, p4 = dppi<0x114, 0xF>(rid) == rid,
             p8 = dppi<0x118, 0xF>(rid) == rid, pA = dppi<0x142, 0xA>(rid) == rid, pB = dppi<0x143, 0xC>(rid) == rid;
  float dw_acc = 0.f;
  double inv64 = 0.0;
  bool poisoned = false;
  if constexpr (FROM64) {
    inv64 = ldexp(1.0, -(int)(long long)dG64[vtot * F]);
    poisoned = dG64[vtot * F + 1] != 0ull;
  }
  int lstart = lmin;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int ov = __shfl_xor(lstart, o, 64); lstart = ov < lstart ? ov : lstart; }
  // A freshly initialised HPD sends a million (vertex,k) entries to a few dozen slots: one atomic per (wave, run, level)
  // would still pile ~300 same-address float atomics on each table row.  The waves of the workgroup chain their runs
  // through LDS instead: a run that continues into the next wave hands its partial sum over, and only the wave in which the
  // run ENDS (or the last wave of the workgroup) issues the atomic.  The chaining depends on the keys only, not the level.
  const int64_t first_key = __shfl(key, 0, 64), last_key = __shfl(key, 63, 64);
  const int nruns = __shfl(rid, 63, 64);
  if (lane == 0) { s_first[wave] = first_key; s_last[wave] = last_key; s_nruns[wave] = nruns; s_lstart[wave] = lstart; }
  __syncthreads();
  int lblock = Ls;
#pragma unroll
  for (int q = 0; q < NW; ++q) lblock = s_lstart[q] < lblock ? s_lstart[q] : lblock;
  const bool continues = wave + 1 < NW && s_first[wave + 1] == last_key;     // my last run goes on in the next wave
  // lanes that emit the atomic of their run: its last lane, unless the run is handed to the next wave
  const bool emit = live && ((lane == 63) ? !continues : (rid_dn != rid));
  const bool in_first_run = rid == 1;
  for (int l = lblock; l < Ls; ++l) {
    const int n = s_n[l];
    const bool in = live && l >= lmin;
    float v[F];
    float dot = 0.f;
    if (in) {
      const int64_t gi = (s_goff[l] + (int64_t)gy * (n + 2) + gx) * F;
      float g[F];
      if constexpr (FROM64) {
#pragma unroll
        for (int f = 0; f < F; ++f)
          g[f] = poisoned ? __int_as_float(0x7fc00000) : (float)((double)(long long)dG64[gi + f] * inv64);      // as dg64_to_float_kernel
      } else {
#pragma unroll
        for (int f = 0; f < F; ++f) g[f] = dG[gi + f];
      }
      if (dvert_w) {                       // the table row is only needed for d w (trainable HPD)
        const TT* r = tables + ((int64_t)l * T + slot) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) { const float gv = g[f]; v[f] = gv * w; dot += gv * tload(r + f); }
      } else {
#pragma unroll
        for (int f = 0; f < F; ++f) v[f] = g[f] * w;
      }
    } else {
#pragma unroll
      for (int f = 0; f < F; ++f) v[f] = 0.f;
    }
    dw_acc += dot;
    // segmented inclusive scan over equal-slot runs, entirely on DPP (no ds_bpermute): 16-lane rows first, then lane
    // 15 / 47 into the next row and lane 31 into the upper half.
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float x = v[f], y;
      // the lane movement must run with the FULL exec mask (a DPP source lane that is masked off reads as invalid):
      // the empty asm keeps hipcc from sinking it under the predicate
      y = dppf<0x111, 0xF>(x); asm volatile("" : "+v"(y)); x += p1 ? y : 0.f;
      y = dppf<0x112, 0xF>(x); asm volatile("" : "+v"(y)); x += p2 ? y : 0.f;
      y = dppf<0x114, 0xF>(x); asm volatile("" : "+v"(y)); x += p4 ? y : 0.f;
      y = dppf<0x118, 0xF>(x); asm volatile("" : "+v"(y)); x += p8 ? y : 0.f;
      y = dppf<0x142, 0xA>(x); asm volatile("" : "+v"(y)); x += pA ? y : 0.f;
      y = dppf<0x143, 0xC>(x); asm volatile("" : "+v"(y)); x += pB ? y : 0.f;
      v[f] = x;
    }
    // hand-over between waves (double-buffered by level parity: one barrier per level)
    float (*tail)[F] = s_tail[l & 1];
    if (lane == 63) {
#pragma unroll
      for (int f = 0; f < F; ++f) tail[wave][f] = v[f];
    }
    __syncthreads();
    float carry[F];
#pragma unroll
    for (int f = 0; f < F; ++f) carry[f] = 0.f;
    for (int q = wave - 1; q >= 0; --q) {                   // wave-uniform walk down the chain that ends in my first run
      if (s_last[q] != first_key) break;
#pragma unroll
      for (int f = 0; f < F; ++f) carry[f] += tail[q][f];
      if (s_nruns[q] != 1) break;                           // that wave's last run began inside it: the chain starts there
    }
    if (emit) {
      float* d = dtables + ((int64_t)l * T + slot) * F;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const float tot = v[f] + (in_first_run ? carry[f] : 0.f);
        if (tot != 0.f) atomicAdd(d + f, tot);
      }
    }
  }
  if (live && dvert_w) dvert_w[e] = dw_acc;
}

}  // namespace gngf

using namespace gngf;

#define DISPATCH_TT(dt, ...)                                                          \
  if ((dt) == GNGF_FEAT_F32) { using TT = float; __VA_ARGS__; }                       \
  else if ((dt) == GNGF_FEAT_F16) { using TT = __half; __VA_ARGS__; }                 \
  else return (int)hipErrorInvalidValue;

#define DISPATCH_F(F, ...)                          \
  switch (F) {                                      \
    case 1: { constexpr int kF = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int kF = 2; __VA_ARGS__; } break; \
    case 4: { constexpr int kF = 4; __VA_ARGS__; } break; \
    case 8: { constexpr int kF = 8; __VA_ARGS__; } break; \
    default: return (int)hipErrorInvalidValue;      \
  }

// Bins P pixels into 4^tile_shift spatial tiles.  NB = number of binning blocks (<= 512), chunk = max pixels per
// work item.  Outputs: sorted (P float4 = x, y, bits(original index), 0), items (max_items int4 = start, count,
// tile, items of that tile; max_items >= ceil(P/chunk) + 4^tile_shift), n_items (4: the count + three work counters), tile_off and tile_item_base
// (4^tile_shift + 1 each: exclusive prefixes of pixels / items per tile), blockhist (4^tile_shift * (NB + 1) scratch).
extern "C" int gngf_bin_pixels(const float* xy, int64_t P, int tile_shift, int NB, int chunk, int32_t* blockhist,
                               int32_t* tile_off, int32_t* tile_item_base, int32_t* items, int32_t* n_items, float* sorted,
                               void* stream) {
  GNGF_CHECK_ARG(P >= 0 && P < (1ll << 31) && tile_shift >= 0 && tile_shift <= 6 && NB > 0 && NB <= kBinMaxBlocks && chunk > 0);
  GNGF_CHECK_ARG(xy && blockhist && tile_off && tile_item_base && items && n_items && sorted);
  const int ntiles = 1 << (2 * tile_shift);
  const int64_t per_block = ceil_div(ceil_div(P, NB), kBinThreads) * kBinThreads;
  hipStream_t s = as_stream(stream);
  const size_t smem = (size_t)ntiles * sizeof(int);
  bin_count_kernel<<<dim3(NB), dim3(kBinThreads), smem, s>>>(reinterpret_cast<const float2*>(xy), P, per_block, tile_shift, NB,
                                                             blockhist);
  // row totals live behind the NB columns of blockhist: the caller's scratch is 4^tile_shift * (NB + 1) int32
  int32_t* tot = blockhist + (int64_t)ntiles * NB;
  bin_rowscan_kernel<<<dim3((unsigned)ceil_div(ntiles, 4)), dim3(256), 0, s>>>(blockhist, NB, ntiles, tot);
  bin_scan_kernel<<<dim3(1), dim3(kBinThreads), 0, s>>>(tot, tile_shift, chunk, tile_off, tile_item_base,
                                                         reinterpret_cast<int4*>(items), n_items);
  bin_scatter_kernel<<<dim3(NB), dim3(kBinThreads), smem, s>>>(reinterpret_cast<const float2*>(xy), P, per_block, tile_shift, NB,
                                                               blockhist, tile_off, reinterpret_cast<float4*>(sorted));
  GNGF_RETURN_LAUNCH();
}

// Everything in front of the pixel stage in FOUR launches of one chain: binning (as gngf_bin_pixels) with the vertex stage
// forward (as gngf_vertex_grid_fwd, levels [0, Ls)) riding on the count launch and two buffer clears riding along:
// dG_zero (the vertex-grid gradient, same shape as G; NULL: none) is cleared by the vertex riders, zero_fill (zero_floats
// floats, a multiple of 4, 16-byte aligned; NULL: none) by riders of the scatter launch.
extern "C" int gngf_encode_tiled_prepare(const float* xy, int64_t P, int tile_shift, int NB, int chunk, int32_t* blockhist,
                                         int32_t* tile_off, int32_t* tile_item_base, int32_t* items, int32_t* n_items,
                                         float* sorted, const void* tables, int feat_dtype, const int32_t* vert_idx,
                                         const float* vert_w, const int32_t* n_ls, const int32_t* n_ls_host, float* G,
                                         float* dG_zero, int dG_zero_words, int Ls, int F, int64_t T, int K, int mode, int vstride,
                                         int64_t NV, float* zero_fill, int64_t zero_floats, int32_t* persistent_ws, float* clear_rows,
                                         void* stream) {
  GNGF_CHECK_ARG(dG_zero_words == 1 || dG_zero_words == 2);
  // clear_rows (optional, spatial-hash source only): an (L,T,F) fp32 table gradient that lives from step to step — the vertex riders
  // zero row hash(gx, gy) of every staged vertex's level on the way (gngf_clear_hashed_rows without a launch of its own)
  GNGF_CHECK_ARG(!clear_rows || (mode == GNGF_MODE_HASH && (reinterpret_cast<uintptr_t>(clear_rows) & 3) == 0));
  GNGF_CHECK_ARG(P >= 0 && P < (1ll << 31) && tile_shift >= 0 && tile_shift <= 6 && NB > 0 && NB <= kBinMaxBlocks && chunk > 0);
  GNGF_CHECK_ARG(xy && blockhist && tile_off && tile_item_base && items && n_items && sorted);
  // G == NULL (spatial-hash source only): no vertex grid is wanted — the pixel stage gathers from the level tables itself
  // (gngf_encode_tiled_fwd_fused) — and the riders only clear (dG_zero / clear_rows), or do not run at all
  GNGF_CHECK_ARG(Ls > 0 && Ls <= GNGF_MAX_LEVELS && T > 0 && tables && n_ls && n_ls_host && (G || mode == GNGF_MODE_HASH));
  GNGF_CHECK_ARG(mode == GNGF_MODE_HASH || (vert_idx && vert_w && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0));
  GNGF_CHECK_ARG(!zero_fill || (zero_floats >= 0 && (zero_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_fill) & 15) == 0));
  const int ntiles = 1 << (2 * tile_shift);
  const int64_t per_block = ceil_div(ceil_div(P, NB), kBinThreads) * kBinThreads;
  hipStream_t s = as_stream(stream);
  const size_t smem = (size_t)ntiles * sizeof(int);
  int64_t vtot = 0;
  for (int l = 0; l < Ls; ++l) vtot += (int64_t)(n_ls_host[l] + 2) * (n_ls_host[l] + 2);
  const int vblocks = (G || dG_zero || clear_rows) ? (int)ceil_div(vtot, kBinThreads) : 0;
  const bool pow2 = (T & (T - 1)) == 0;
  const float2* xy2 = reinterpret_cast<const float2*>(xy);
  const int64_t nvec = zero_fill ? zero_floats / 4 : 0;
  const int zblocks = nvec > 0 ? (int)(ceil_div(nvec, 4096) < 1024 ? ceil_div(nvec, 4096) : 1024) : 0;
  int32_t* tot = blockhist + (int64_t)ntiles * NB;
  if (zblocks == 0) {
    // no gradient clear on this launch: the vertex riders move to the COUNT launch and the scatter launch runs alone
    if (mode == GNGF_MODE_HASH) {
      DISPATCH_TT(feat_dtype, DISPATCH_F(F, (bin_count_vride_kernel<kF, false, TT><<<dim3((unsigned)(NB + vblocks)), dim3(kBinThreads), smem, s>>>(
                                  xy2, P, per_block, tile_shift, NB, blockhist, static_cast<const TT*>(tables), nullptr, nullptr, n_ls, G,
                                  dG_zero, Ls, T, 0, 0, 0, pow2, vtot, dG_zero_words, persistent_ws, clear_rows))));
    } else {
      DISPATCH_TT(feat_dtype, DISPATCH_F(F, (bin_count_vride_kernel<kF, true, TT><<<dim3((unsigned)(NB + vblocks)), dim3(kBinThreads), smem, s>>>(
                                  xy2, P, per_block, tile_shift, NB, blockhist, static_cast<const TT*>(tables), vert_idx, vert_w, n_ls, G,
                                  dG_zero, Ls, T, K, vstride, NV, pow2, vtot, dG_zero_words, persistent_ws, nullptr))));
    }
    if (persistent_ws) {       // count -> scatter: the scans ride inside the scatter launch (bin_scatter2_kernel)
      bin_scatter2_kernel<<<dim3(NB), dim3(kBinThreads), smem, s>>>(xy2, P, per_block, tile_shift, NB, chunk, blockhist, persistent_ws,
                                                                   tile_off, tile_item_base, reinterpret_cast<int4*>(items), n_items,
                                                                   reinterpret_cast<float4*>(sorted));
      GNGF_RETURN_LAUNCH();
    }
    bin_rowscan_kernel<<<dim3((unsigned)ceil_div(ntiles, 4)), dim3(256), 0, s>>>(blockhist, NB, ntiles, tot);
    bin_scan_kernel<<<dim3(1), dim3(kBinThreads), 0, s>>>(tot, tile_shift, chunk, tile_off, tile_item_base,
                                                           reinterpret_cast<int4*>(items), n_items);
    bin_scatter_kernel<<<dim3(NB), dim3(kBinThreads), smem, s>>>(xy2, P, per_block, tile_shift, NB, blockhist, tile_off,
                                                                 reinterpret_cast<float4*>(sorted));
    GNGF_RETURN_LAUNCH();
  }
  bin_count_ride_kernel<<<dim3((unsigned)(NB + zblocks)), dim3(kBinThreads), smem, s>>>(xy2, P, per_block, tile_shift, NB, blockhist,
                                                                                       reinterpret_cast<float4*>(zero_fill), nvec, zblocks);
  bin_rowscan_kernel<<<dim3((unsigned)ceil_div(ntiles, 4)), dim3(256), 0, s>>>(blockhist, NB, ntiles, tot);
  bin_scan_kernel<<<dim3(1), dim3(kBinThreads), 0, s>>>(tot, tile_shift, chunk, tile_off, tile_item_base,
                                                         reinterpret_cast<int4*>(items), n_items);
  float4* sorted4 = reinterpret_cast<float4*>(sorted);
  if (mode == GNGF_MODE_HASH) {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (bin_scatter_ride_kernel<kF, false, TT><<<dim3((unsigned)(NB + vblocks)), dim3(kBinThreads), smem, s>>>(
                                xy2, P, per_block, tile_shift, NB, blockhist, tile_off, sorted4, static_cast<const TT*>(tables), nullptr,
                                nullptr, n_ls, G, dG_zero, Ls, T, 0, 0, 0, pow2, vtot, dG_zero_words, clear_rows))));
  } else {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (bin_scatter_ride_kernel<kF, true, TT><<<dim3((unsigned)(NB + vblocks)), dim3(kBinThreads), smem, s>>>(
                                xy2, P, per_block, tile_shift, NB, blockhist, tile_off, sorted4, static_cast<const TT*>(tables), vert_idx,
                                vert_w, n_ls, G, dG_zero, Ls, T, K, vstride, NV, pow2, vtot, dG_zero_words, nullptr))));
  }
  GNGF_RETURN_LAUNCH();
}

static int max_grid_side(const int32_t* n_ls_host, int Ls) {
  int m = 0;
  for (int l = 0; l < Ls; ++l) m = n_ls_host[l] + 2 > m ? n_ls_host[l] + 2 : m;
  return m;
}

// Vertex stage forward: G (sum_l (N_l+2)^2, F) for levels [0, Ls).  n_ls_host mirrors n_ls on the host (grid sizing).
extern "C" int gngf_vertex_grid_fwd(const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls,
                                    const int32_t* n_ls_host, float* G, int Ls, int F, int64_t T, int K, int mode,
                                    int vstride, int64_t NV, void* stream) {
  GNGF_CHECK_ARG(Ls > 0 && Ls <= GNGF_MAX_LEVELS && T > 0 && tables && n_ls && n_ls_host && G);
  GNGF_CHECK_ARG(mode == GNGF_MODE_HASH || (vert_idx && vert_w && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0));
  const int side = max_grid_side(n_ls_host, Ls);
  dim3 grid((unsigned)ceil_div((int64_t)side * side, 256), (unsigned)Ls), block(256);
  const bool pow2 = (T & (T - 1)) == 0;
  if (mode == GNGF_MODE_HASH) {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (vertex_fwd_kernel<kF, false, TT><<<grid, block, 0, as_stream(stream)>>>(
                                static_cast<const TT*>(tables), nullptr, nullptr, n_ls, G, T, 0, 0, 0, pow2))));
  } else {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (vertex_fwd_kernel<kF, true, TT><<<grid, block, 0, as_stream(stream)>>>(
                                static_cast<const TT*>(tables), vert_idx, vert_w, n_ls, G, T, K, vstride, NV, pow2))));
  }
  GNGF_RETURN_LAUNCH();
}

// Vertex stage backward: dG -> dtables (accumulated, caller zero-fills) and dvert_w (accumulated, may be NULL).
extern "C" int gngf_vertex_grid_bwd(const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls,
                                    const int32_t* n_ls_host, const float* dG, float* dtables, float* dvert_w, int Ls, int F,
                                    int64_t T, int K, int mode, int vstride, int64_t NV, void* stream) {
  GNGF_CHECK_ARG(Ls > 0 && Ls <= GNGF_MAX_LEVELS && T > 0 && tables && n_ls && n_ls_host && dG && dtables);
  GNGF_CHECK_ARG(mode == GNGF_MODE_HASH || (vert_idx && vert_w && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0));
  int64_t vtot = 0;
  for (int l = 0; l < Ls; ++l) vtot += (int64_t)(n_ls_host[l] + 2) * (n_ls_host[l] + 2);
  dim3 grid((unsigned)ceil_div(vtot, 256)), block(256);
  const bool pow2 = (T & (T - 1)) == 0;
  if (mode == GNGF_MODE_HASH) {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (vertex_bwd_kernel<kF, false, TT><<<grid, block, 0, as_stream(stream)>>>(
                                static_cast<const TT*>(tables), nullptr, nullptr, n_ls, dG, dtables, nullptr, Ls, T, 0, 0, 0, pow2))));
  } else {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (vertex_bwd_kernel<kF, true, TT><<<grid, block, 0, as_stream(stream)>>>(
                                static_cast<const TT*>(tables), vert_idx, vert_w, n_ls, dG, dtables, dvert_w, Ls, T, K, vstride, NV,
                                pow2))));
  }
  GNGF_RETURN_LAUNCH();
}

// Pixel stage.  enc / genc are (P, L*F) rows; levels [0, Ls) are produced / consumed here (the remaining levels, if any,
// by the direct form).  lds_bytes = dynamic LDS for the per-tile sub-grids (levels that do not fit fall back to global).
// diagnostic: reads and clears the phase stamps of tiled_bwd_il_kernel's workgroup 0 (8 x uint64: cycles before the item's
// first barrier, setup, zero + hint, main loop (thread 0), wait for the others, store pass; items; pixels)
extern "C" int gngf_debug_il_stamps(unsigned long long* host8) {
#ifdef GNGF_STAMPS
  hipError_t e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(gngf::g_il_stamps), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return (int)e;
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(gngf::g_il_stamps), z, sizeof(z));
#else
  (void)host8;
  return (int)hipErrorNotSupported;          // the production build carries no stamps: build with HIPFLAGS += -DGNGF_STAMPS
#endif
}

static int compute_units() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
  }
  return cus;
}

// 1 (default): F = 2 with <= 16 staged levels runs the level-interleaved kernels (tiled_fwd_il_kernel / tiled_bwd_il_kernel);
// 0: the back-to-back layout for every shape.  Returns the previous setting (measurement / tests).
static int g_tiled_interleaved = 1;
extern "C" int gngf_set_tiled_interleaved(int on) {
  const int prev = g_tiled_interleaved;
  g_tiled_interleaved = on < 0 ? 0 : (on > 3 ? 3 : on);      // (3: the forward kernel with four pixels per trip instead of two)
  return prev;
}

// the interleaved kernels apply: two features, at most 16 staged levels, every level's sub-grid fits the compact image
// (lds_floats, the host plan's worst case) and the interleaved image fits the LDS
static bool interleaved_applies(const int32_t* n_ls_host, int Ls, int F, int tile_shift, int lds_floats, bool backward) {
  if (!g_tiled_interleaved || F != 2 || Ls > kIL || !n_ls_host) return false;
  int64_t compact = 0;
  for (int l = 0; l < Ls; ++l) { const int64_t w = (n_ls_host[l] >> tile_shift) + 3; compact += w * w * F; }
  if (compact > lds_floats) return false;
  // (the column of every level is as tall as the finest staged level: beyond ~100 KB the image costs more occupancy than the
  // conflicts it removes — measured at the 4096^2 shape, finest staged level N = 1955: 139 KB forward image, 106 us vs 88 us)
  const int64_t bytes = (int64_t)interleaved_rows(n_ls_host, Ls, tile_shift) * kIL * 8 * (backward ? 2 : 1) + (backward ? lds_floats * 4 : 0);
  return bytes <= (backward ? 112 : 72) * 1024;
}

static bool bin_job_ok(const gngf_bin_job* j) {
  return j && j->xy && j->P > 0 && j->P < (1ll << 31) && j->tile_shift >= 0 && j->tile_shift <= 6 && j->NB > 0 && j->NB <= kBinMaxBlocks &&
         j->chunk > 0 && j->blockhist && j->persistent_ws && j->tile_off && j->tile_item_base && j->items && j->n_items && j->sorted;
}
static int64_t bin_per_block(int64_t P, int NB) { return ceil_div(ceil_div(P, NB), kBinThreads) * kBinThreads; }
static BinJobDev bin_job_dev(const gngf_bin_job* j) {
  BinJobDev d;
  d.xy = reinterpret_cast<const float2*>(j->xy); d.P = j->P; d.per_block = bin_per_block(j->P, j->NB);
  d.NB = j->NB; d.tile_shift = j->tile_shift; d.chunk = j->chunk;
  d.blockbase = j->blockhist; d.pws = j->persistent_ws; d.tile_off = j->tile_off; d.tile_item_base = j->tile_item_base;
  d.items = reinterpret_cast<int4*>(j->items); d.n_items = j->n_items; d.sorted = reinterpret_cast<float4*>(j->sorted);
  return d;
}
static BinJobDev bin_job_none() {
  BinJobDev d = {nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  return d;
}

// The launcher's own decision, for callers that size / initialise buffers differently for the two kernel families (the
// fixed-point vertex grid dG64 is only filled by the interleaved backward): 1 = the level-interleaved kernel will run.
extern "C" int gngf_tiled_interleaved_applies(const int32_t* n_ls_host, int Ls, int F, int tile_shift, int lds_bytes, int backward) {
  if (!n_ls_host || Ls <= 0 || Ls > GNGF_MAX_LEVELS || tile_shift < 0 || tile_shift > 6 || lds_bytes < 0) return 0;
  return interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, backward != 0) ? 1 : 0;
}

extern "C" int gngf_encode_tiled_fwd(const float* sorted, const int32_t* items, const int32_t* n_items, int max_items,
                                     const int32_t* n_ls, const int32_t* n_ls_host, const float* G, float* enc, int L, int Ls,
                                     int F, int tile_shift, int lds_bytes, void* stream) {
  GNGF_CHECK_ARG(max_items >= 0 && L > 0 && Ls > 0 && Ls <= L && L <= GNGF_MAX_LEVELS && lds_bytes >= 0 && lds_bytes <= 128 * 1024);
  if (max_items == 0) return 0;
  GNGF_CHECK_ARG(sorted && items && n_items && n_ls && G && enc);
  if (interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, false)) {
    const size_t smem = (size_t)interleaved_rows(n_ls_host, Ls, tile_shift) * kIL * 8;
    const int pv = g_tiled_interleaved;
    auto fn = (L == 16) ? (pv == 3 ? tiled_fwd_il_kernel<true, 4, 0> : tiled_fwd_il_kernel<true, 2, 0>) : tiled_fwd_il_kernel<false, 2, 0>;
    if (smem > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    // persistent workgroups: as many as fit the chip at once (LDS-bound), sharing the items through a counter
    const int per_cu = (int)((150 * 1024) / (smem + 2048)) < 1 ? 1 : (int)((150 * 1024) / (smem + 2048));
    (void)per_cu;
    const int nwork = max_items;                   // forward: one item per workgroup (two to three workgroups share a CU)
    const VertexSrc vs = {G, nullptr, nullptr, nullptr, 0, 0, 0, 0, false};
    const BinCountRide none = bin_job_none();
    fn<<<dim3((unsigned)nwork), dim3(kTBF), smem, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(sorted), reinterpret_cast<const int4*>(items), n_items, const_cast<int32_t*>(n_items) + 1, n_ls, vs,
        enc, L, Ls, tile_shift, nwork, none);
    GNGF_RETURN_LAUNCH();
  }
  DISPATCH_F(F, {
    if (lds_bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tiled_fwd_kernel<kF>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      if (e != hipSuccess) return (int)e;
    }
    tiled_fwd_kernel<kF><<<dim3((unsigned)max_items), dim3(kTBF), (size_t)lds_bytes, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(sorted), reinterpret_cast<const int4*>(items), n_items, n_ls, G, enc, L, Ls, tile_shift,
        lds_bytes / 4);
  });
  GNGF_RETURN_LAUNCH();
}

// Binning alone in TWO launches (count -> scatter, the scans ride inside the scatter launch: bin_scatter2_kernel) — the form the
// step uses when the vertex stage forward is fused into the pixel stage (gngf_encode_tiled_fwd_fused) and nothing else has to
// ride on the binning.  zero_fill (optional; zero_floats floats, a multiple of 4, 16-byte aligned): cleared by rider workgroups
// of the count launch.
extern "C" int gngf_bin_pixels2(const gngf_bin_job* job, float* zero_fill, int64_t zero_floats, void* stream) {
  GNGF_CHECK_ARG(bin_job_ok(job));
  GNGF_CHECK_ARG(!zero_fill || (zero_floats >= 0 && (zero_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_fill) & 15) == 0));
  const int ntiles = 1 << (2 * job->tile_shift);
  hipStream_t s = as_stream(stream);
  const BinJobDev j = bin_job_dev(job);
  const int64_t nvec = zero_fill ? zero_floats / 4 : 0;
  const int zblocks = nvec > 0 ? (int)(ceil_div(nvec, 4096) < 1024 ? ceil_div(nvec, 4096) : 1024) : 0;
  const size_t smem_c = ((size_t)ntiles + 2 * (kBinThreads / 64) + 1) * sizeof(int);
  bin_count_reserve_kernel<<<dim3((unsigned)(job->NB + zblocks)), dim3(kBinThreads), smem_c, s>>>(
      j, reinterpret_cast<float4*>(zero_fill), nvec, zblocks > 0 ? zblocks : 1);
  bin_scatter3_kernel<<<dim3(job->NB), dim3(kBinThreads), (size_t)ntiles * sizeof(int), s>>>(j);
  GNGF_RETURN_LAUNCH();
}

// Pixel stage forward with the vertex stage forward FUSED into its staging loop (level-interleaved kernel only: F = 2, fp32
// tables, <= 16 staged levels whose image fits the LDS — gngf_tiled_interleaved_applies(…, 0) — anything else is rejected):
// the staged sub-grids are gathered from the level tables themselves (mode / vert_idx / vert_w / vstride / NV as
// gngf_vertex_grid_fwd), same arithmetic in the same order, so enc equals gngf_vertex_grid_fwd + gngf_encode_tiled_fwd bit for
// bit.  next_count (optional): the COUNT half of another batch's binning (its per-block histograms and tile totals) runs in
// extra workgroups at the head of this launch; its scatter half rides on gngf_encode_tiled_bwd(…, next_bin).
extern "C" int gngf_encode_tiled_fwd_fused(const float* sorted, const int32_t* items, const int32_t* n_items, int max_items,
                                           const int32_t* n_ls, const int32_t* n_ls_host, const void* tables, int feat_dtype,
                                           const int32_t* vert_idx, const float* vert_w, float* enc, int L, int Ls, int F, int64_t T,
                                           int K, int mode, int vstride, int64_t NV, int tile_shift, int lds_bytes,
                                           const gngf_bin_job* next_count, void* stream) {
  GNGF_CHECK_ARG(max_items >= 0 && L > 0 && Ls > 0 && Ls <= L && L <= GNGF_MAX_LEVELS && lds_bytes >= 0 && lds_bytes <= 128 * 1024);
  GNGF_CHECK_ARG(!next_count || bin_job_ok(next_count));
  if (max_items == 0 && !next_count) return 0;
  GNGF_CHECK_ARG(sorted && items && n_items && n_ls && n_ls_host && tables && enc && T > 0);
  GNGF_CHECK_ARG(mode == GNGF_MODE_HASH || (vert_idx && vert_w && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0));
  if (!interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, false)) {
    // generic pixel-stage kernel (shapes whose level-interleaved image does not fit, F != 2): spatial-hash source, fp32 or fp16 tables
    GNGF_CHECK_ARG(mode == GNGF_MODE_HASH && !next_count);
    if (max_items == 0) return 0;
    const bool hp2 = (T & (T - 1)) == 0;
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, {
      auto fn = tiled_fwd_kernel<kF, true, TT>;
      if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return (int)e;
      }
      fn<<<dim3((unsigned)max_items), dim3(kTBF), (size_t)lds_bytes, as_stream(stream)>>>(
          reinterpret_cast<const float4*>(sorted), reinterpret_cast<const int4*>(items), n_items, n_ls, nullptr, enc, L, Ls, tile_shift,
          lds_bytes / 4, static_cast<const TT*>(tables), T, hp2);
    }));
    GNGF_RETURN_LAUNCH();
  }
  GNGF_CHECK_ARG(feat_dtype == GNGF_FEAT_F32);
  size_t smem = (size_t)interleaved_rows(n_ls_host, Ls, tile_shift) * kIL * 8;
  BinCountRide cride = bin_job_none();
  if (next_count) {
    cride = bin_job_dev(next_count);
    const size_t hist = (sizeof(int) << (2 * next_count->tile_shift)) + sizeof(int) * (2 * (kTBF / 64) + 1);
    smem = smem < hist ? hist : smem;
  }
  const bool hash = mode == GNGF_MODE_HASH;
  const VertexSrc vs = {nullptr, static_cast<const float*>(tables), vert_idx, vert_w, T, NV, K, vstride, (T & (T - 1)) == 0};
  auto fn = hash ? ((L == 16) ? tiled_fwd_il_kernel<true, 2, 1> : tiled_fwd_il_kernel<false, 2, 1>)
                 : ((L == 16) ? tiled_fwd_il_kernel<true, 2, 2> : tiled_fwd_il_kernel<false, 2, 2>);
  if (smem > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  const int nwork = max_items;
  fn<<<dim3((unsigned)(nwork + cride.NB)), dim3(kTBF), smem, as_stream(stream)>>>(
      reinterpret_cast<const float4*>(sorted), reinterpret_cast<const int4*>(items), n_items, const_cast<int32_t*>(n_items) + 1, n_ls, vs,
      enc, L, Ls, tile_shift, nwork, cride);
  GNGF_RETURN_LAUNCH();
}

extern "C" int gngf_encode_tiled_bwd(const float* sorted, const int32_t* items, const int32_t* n_items, int max_items,
                                     const int32_t* tile_item_base, const int32_t* tile_level_off, const int32_t* n_ls,
                                     const int32_t* n_ls_host, const float* genc, const float* genc_absmax, int absmax_count,
                                     int absmax_stride, float* dG, float* partials, int L, int Ls, int F, int tile_shift,
                                     int lds_bytes, int chunk, const float* ride_slabs, float* ride_dW0, float* ride_db0,
                                     float* ride_dW1, float* ride_db1, float* ride_dW2, float* ride_db2, int64_t ride_P,
                                     int ride_in_dim, int ride_out_dim, const float* gloss_promised, const float* gloss_arrived,
                                     const float* mse_pred, const float* mse_label,
                                     float* mse_loss, float* mse_workspace, int64_t mse_n, float* hash_dtables, int64_t hash_T,
                                     void* dG64, int log2_pixels, const gngf_bin_job* next_bin, void* stream) {
  GNGF_CHECK_ARG(!hash_dtables || hash_T > 0);
  // the scatter half of another batch's binning rides in the tail of the persistent interleaved kernel only
  GNGF_CHECK_ARG(!next_bin || (bin_job_ok(next_bin) && max_items > 0 && n_ls_host &&
                               interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, true)));
  const bool hpow2 = hash_dtables && (hash_T & (hash_T - 1)) == 0;
  GNGF_CHECK_ARG(max_items >= 0 && L > 0 && Ls > 0 && Ls <= L && L <= GNGF_MAX_LEVELS && lds_bytes >= 0 && lds_bytes <= 64 * 1024);
  GNGF_CHECK_ARG(!gloss_promised == !gloss_arrived);
  RideAlong ride = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, gloss_promised, gloss_arrived};
  int ride_blocks = 0;
  if (ride_slabs) {
    GNGF_CHECK_ARG(ride_dW0 && ride_db0 && ride_dW1 && ride_db1 && ride_dW2 && ride_db2 && ride_P >= 0 && ride_in_dim > 0 &&
                   ride_in_dim <= 64 && ride_out_dim > 0 && ride_out_dim <= 4);
    ride.slabs = ride_slabs; ride.dW0 = ride_dW0; ride.db0 = ride_db0; ride.dW1 = ride_dW1; ride.db1 = ride_db1;
    ride.dW2 = ride_dW2; ride.db2 = ride_db2;
    ride.nslabs = gngf_decoder_bwd_slabs(ride_P);
    ride.nslab = gngf_decoder_slab_floats(ride_in_dim, ride_out_dim);
    ride.in_dim = ride_in_dim; ride.out_dim = ride_out_dim; ride.first_block = max_items;
    ride_blocks = (ride.nslab + 63) / 64;
  }
  MseRide mride = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  if (mse_pred) {
    GNGF_CHECK_ARG(mse_label && mse_loss && mse_workspace && mse_n > 0 && (reinterpret_cast<uintptr_t>(mse_pred) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(mse_label) & 15) == 0 && (reinterpret_cast<uintptr_t>(mse_workspace) & 7) == 0);
    mride.pred = mse_pred; mride.label = mse_label; mride.loss = mse_loss;
    mride.acc = reinterpret_cast<double*>(mse_workspace); mride.counter = reinterpret_cast<unsigned*>(mse_workspace + 2);
    mride.n = mse_n; mride.first_block = max_items + ride_blocks; mride.nblocks = gngf_mse_blocks(mse_n);
    ride_blocks += mride.nblocks;
  }
  GNGF_CHECK_ARG(chunk > 0 && chunk <= (1 << 20) && (!genc_absmax || (absmax_count > 0 && absmax_stride >= 0)));
  int log2_chunk = 0;
  while ((1 << log2_chunk) < chunk) ++log2_chunk;
  if (max_items == 0 && ride_blocks == 0) return 0;
  GNGF_CHECK_ARG(max_items == 0 || (sorted && items && n_items && tile_item_base && n_ls && n_ls_host && genc && partials));
  // dG may be NULL only when the launch is going to fill dG64 and nothing else (the caller reads the fixed-point grid itself)
  // ... or, spatial-hash source, when the interleaved kernel adds to the table gradient itself (bound given, no fixed-point grid)
  GNGF_CHECK_ARG(max_items == 0 || dG ||
                 (genc_absmax && dG64 && log2_pixels > 0 && log2_pixels <= 40 && !hash_dtables &&
                  interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, true)) ||
                 (!dG64 && hash_dtables));
  // a fixed-point grid (with the bound that makes it usable) is only ever filled by the interleaved kernel: a caller that
  // passes one for a shape the generic kernels take has sized / initialised dG for the wrong path (it would read an
  // uninitialised dG) — rejected instead of computed (callers ask gngf_tiled_interleaved_applies first)
  GNGF_CHECK_ARG(max_items == 0 || !(dG64 && genc_absmax) || interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, true));
  int64_t vtot_h = 0;
  if (max_items > 0)
    for (int l = 0; l < Ls; ++l) vtot_h += (int64_t)(n_ls_host[l] + 2) * (n_ls_host[l] + 2);
  if (max_items > 0 && interleaved_applies(n_ls_host, Ls, F, tile_shift, lds_bytes / 4, true)) {
    const int rows2 = 2 * interleaved_rows(n_ls_host, Ls, tile_shift);
    size_t smem = (size_t)rows2 * kIL * 8 + (size_t)lds_bytes;          // accumulators + the compact fp32 image of the store pass
    BinScatterRide bride = bin_job_none();
    if (next_bin) {
      bride = bin_job_dev(next_bin);
      const size_t need = (sizeof(int) << (2 * next_bin->tile_shift)) + sizeof(int);
      smem = smem < need ? need : smem;
    }
    // spatial-hash source and neither vertex grid handed in: the store pass adds to the table gradient itself
    const bool hdt = hash_dtables && !dG && !dG64;
    auto fn = hdt ? ((L == 16) ? tiled_bwd_il_kernel<true, true> : tiled_bwd_il_kernel<false, true>)
                  : ((L == 16) ? tiled_bwd_il_kernel<true> : tiled_bwd_il_kernel<false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    const int per_cu = (int)((150 * 1024) / (smem + 2048)) < 1 ? 1 : (int)((150 * 1024) / (smem + 2048));
    const int fit = compute_units() * (per_cu > 2 ? 2 : per_cu);
    const int nwork = max_items < fit ? max_items : fit;
    ride.first_block = nwork;                      // the riders follow the persistent workgroups
    if (mse_pred) mride.first_block = nwork + (ride_slabs ? (ride.nslab + 63) / 64 : 0);
    // one scale for the launch (a bound on |genc| from its producer) -> the items add their sums into a fixed-point vertex grid
    unsigned long long* g64 = (genc_absmax && dG64 && log2_pixels > 0 && log2_pixels <= 40) ? static_cast<unsigned long long*>(dG64) : nullptr;
    fn<<<dim3((unsigned)(nwork + ride_blocks)), dim3(kTB), smem, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(sorted), reinterpret_cast<const int4*>(items), n_items, const_cast<int32_t*>(n_items) + 2, n_ls,
        genc, dG, partials, genc_absmax, absmax_count, absmax_stride, L, Ls, tile_shift, lds_bytes / 4, rows2,
        g64 ? log2_pixels : log2_chunk, nwork, ride, mride, g64, bride, hdt ? hash_dtables : nullptr, hash_T, hpow2);
    if (hdt) GNGF_RETURN_LAUNCH();
    if (g64) {
      if (hash_dtables)
        vertex_bwd_hash64_kernel<2><<<dim3((unsigned)ceil_div(vtot_h, 256)), dim3(256), 0, as_stream(stream)>>>(
            g64, n_ls, hash_dtables, Ls, hash_T, hpow2, vtot_h);
      else if (dG)        // (dG NULL: the caller's vertex stage reads the fixed-point grid itself — gngf_vertex_grid_bwd_sorted(dG64))
        dg64_to_float_kernel<<<dim3((unsigned)ceil_div(vtot_h * 2, 256)), dim3(256), 0, as_stream(stream)>>>(g64, dG, vtot_h * 2);
      GNGF_RETURN_LAUNCH();
    }
    if (hash_dtables)
      gather_partials_kernel<2, true><<<dim3((unsigned)ceil_div(vtot_h, 256)), dim3(256), 0, as_stream(stream)>>>(
          partials, tile_item_base, tile_level_off, n_ls, dG, Ls, tile_shift, lds_bytes / 4, hash_dtables, hash_T, hpow2);
    else
      gather_partials_kernel<2><<<dim3((unsigned)ceil_div(vtot_h, 256)), dim3(256), 0, as_stream(stream)>>>(
          partials, tile_item_base, tile_level_off, n_ls, dG, Ls, tile_shift, lds_bytes / 4);
    GNGF_RETURN_LAUNCH();
  }
  const bool hdt_g = hash_dtables && !dG && !dG64;       // (generic kernels: see tiled_bwd_kernel<F, HDT>)
  DISPATCH_F(F, {
    auto fn = hdt_g ? tiled_bwd_kernel<kF, true> : tiled_bwd_kernel<kF, false>;
    if (2 * lds_bytes > 48 * 1024) {       // 64-bit accumulators: twice the forward image
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * lds_bytes);
      if (e != hipSuccess) return (int)e;
    }
    fn<<<dim3((unsigned)(max_items + ride_blocks)), dim3(kTB), (size_t)2 * lds_bytes, as_stream(stream)>>>(
        reinterpret_cast<const float4*>(sorted), reinterpret_cast<const int4*>(items), n_items, n_ls, genc, dG, partials,
        genc_absmax, absmax_count, absmax_stride, L, Ls, tile_shift, lds_bytes / 4, log2_chunk, ride, mride,
        hdt_g ? hash_dtables : nullptr, hash_T, hpow2);
    if (max_items > 0 && !hdt_g) {
      if (hash_dtables)
        gather_partials_kernel<kF, true><<<dim3((unsigned)ceil_div(vtot_h, 256)), dim3(256), 0, as_stream(stream)>>>(
            partials, tile_item_base, tile_level_off, n_ls, dG, Ls, tile_shift, lds_bytes / 4, hash_dtables, hash_T, hpow2);
      else
        gather_partials_kernel<kF><<<dim3((unsigned)ceil_div(vtot_h, 256)), dim3(256), 0,
                                     as_stream(stream)>>>(partials, tile_item_base, tile_level_off, n_ls, dG, Ls, tile_shift,
                                                          lds_bytes / 4);
    }
  });
  GNGF_RETURN_LAUNCH();
}

// Vertex stage backward, vertex-table source, slot-ordered and contention-free (see vertex_bwd_sorted_kernel).
// order (NV*K) int32 = argsort of vert_idx viewed flat.  dtables accumulated; dvert_w (NV,K) WRITTEN (may be NULL).
extern "C" int gngf_vertex_grid_bwd_sorted(const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w,
                                           const int32_t* order, const int32_t* n_ls, const float* dG, const void* dG64,
                                           int64_t vtot, float* dtables, float* dvert_w, int Ls, int F, int64_t T, int K,
                                           int vstride, int64_t NV, void* stream) {
  GNGF_CHECK_ARG(Ls > 0 && Ls <= GNGF_MAX_LEVELS && T > 0 && K > 0 && K <= GNGF_MAX_TOPK && vstride > 0 && NV > 0);
  GNGF_CHECK_ARG(tables && vert_idx && vert_w && order && n_ls && (dG || (dG64 && vtot > 0)) && dtables && NV * K < (1ll << 31));
  const int64_t NE = NV * K;
  if (dG64) {
    DISPATCH_TT(feat_dtype, DISPATCH_F(F, (vertex_bwd_sorted_kernel<kF, TT, true><<<dim3((unsigned)ceil_div(NE, kVB)), dim3(kVB), 0,
                                                                                  as_stream(stream)>>>(
                                static_cast<const TT*>(tables), vert_idx, vert_w, order, n_ls, nullptr, dtables, dvert_w, Ls, T, K,
                                vstride, NE, static_cast<const unsigned long long*>(dG64), vtot))));
    GNGF_RETURN_LAUNCH();
  }
  DISPATCH_TT(feat_dtype, DISPATCH_F(F, (vertex_bwd_sorted_kernel<kF, TT, false><<<dim3((unsigned)ceil_div(NE, kVB)), dim3(kVB), 0,
                                                                                 as_stream(stream)>>>(
                              static_cast<const TT*>(tables), vert_idx, vert_w, order, n_ls, dG, dtables, dvert_w, Ls, T, K,
                              vstride, NE, nullptr, 0))));
  GNGF_RETURN_LAUNCH();
}
