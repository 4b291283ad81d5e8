/*
 * gngf.h — C-ABI of libgngf_hip.so: the MI355X (gfx950) hot path of the multi-resolution hash-grid encoder
 * with learned (GNGF) collision handling.
 *
 * The reference (FedeMont/collision_handling_in_instantNGP) has NO FFI/plugin boundary: its boundary is the
 * Python nn.Module API of models.py.  These entry points are therefore what a maintainer binds with ctypes
 * underneath that module API; each one names the reference code it replaces
 * (file:line relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns every buffer;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are asynchronous;
 *   - the return value is a hipError_t as int (0 = hipSuccess); nothing throws;
 *     hipErrorInvalidValue (1) = rejected arguments (null pointer, unsupported F/K, negative size);
 *   - fp32 everywhere unless stated (`tables` of the fused encoder may be fp16 storage: feat_dtype; table GRADIENTS are
 *     always accumulated in fp32); indices handed to / from the reference-shaped API are int64,
 *     internal per-vertex tables use int32;
 *   - layouts:  xy (P,2) = (row, col) in [0,1];  tables (L,T,F) contiguous, level-major;
 *               enc (P, L*F) level-major / feature-minor (models.py:651);
 *               corner order v = dx + 2*dy, dx on input dim 0 (models.py:322-331);
 *               vertex table id  vid = gy * vstride + gx  (gx = row-axis vertex, fast).
 */
#ifndef GNGF_H_
#define GNGF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNGF_ABI_VERSION 13
#define GNGF_MAX_LEVELS 32
#define GNGF_MAX_TOPK 32

/* blend of the K looked-up rows — models.py:212-217 (global should_softmax_topk_features) */
enum { GNGF_BLEND_SOFTMAX = 0 /* True */, GNGF_BLEND_RAW = 1 /* None */, GNGF_BLEND_NORM = 2 /* False */ };

/* storage type of the level tables: fp32 (the reference) or fp16 (BASELINE.json config 5); arithmetic and gradients are fp32 */
enum { GNGF_FEAT_F32 = 0, GNGF_FEAT_F16 = 1 };

/* index source of the fused encoder */
enum { GNGF_MODE_HASH = 0 /* models.py:412-414 */, GNGF_MODE_VERTEX_TABLE = 1 /* models.py:416-418, de-duplicated */ };

int gngf_abi_version(void);

/* ---- a5+a6: _scale_to_grid + _fast_hash (models.py:486-528) -> idx (P,L,4) int64, as the module returns it */
int gngf_hash_indices(const float* xy, const int32_t* n_ls, int64_t* idx, int64_t P, int L, int64_t T, void* stream);

/* ---- a10/a11: MultiResHashEncoding.forward (models.py:173-229).
 * idx (P,L,4) when K == 0 (hash branch) or (P,L,4,K) (GNGF branch); probs (P,L,4,K) or NULL; out (P,F,L,4). */
int gngf_mrhe_fwd(const void* tables, int feat_dtype, const int64_t* idx, const float* probs, float* out,
                  int64_t P, int L, int F, int64_t T, int K, int blend, void* stream);
/* a15 at the same boundary: dtables (L,T,F) is ACCUMULATED into (caller zero-fills), dprobs (P,L,4,K) is written. */
int gngf_mrhe_bwd(const void* tables, int feat_dtype, const int64_t* idx, const float* probs, const float* gout,
                  float* dtables, float* dprobs, int64_t P, int L, int F, int64_t T, int K, int blend, void* stream);

/* ---- a12: _bilinear_interpolate (models.py:621-655): feats (P,F,L,4) -> enc (P,L*F); bwd writes dfeats. */
int gngf_bilinear_fwd(const float* xy, const int32_t* n_ls, const float* feats, float* enc,
                      int64_t P, int L, int F, void* stream);
int gngf_bilinear_bwd(const float* xy, const int32_t* n_ls, const float* genc, float* dfeats,
                      int64_t P, int L, int F, void* stream);

/* ---- a5..a12 fused, "direct" form: one lane per (pixel, level); table rows gathered straight from HBM/L2.
 * Only levels [l0, l1) are produced / consumed (enc rows stay (P, L*F)); the tiled form below covers the others.
 * mode HASH: vert_idx/vert_w NULL, K ignored.
 * mode VERTEX_TABLE: vert_idx (NV,K) int32 slots and vert_w (NV,K) blend weights per grid vertex
 *   (vid = gy*vstride + gx), i.e. HPD(top-K) evaluated once per DISTINCT vertex instead of per instance.
 * pixel_order (ABI 12, may be NULL): as in gngf_encode_bwd_bucketed — the pixels walked in the tiled form's binned order (the
 *   gathers of neighbouring lanes then fall into the same aligned blocks of table rows); enc is written by original index. */
int gngf_encode_fwd(const float* xy, const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w,
                    const int32_t* n_ls, float* enc, int64_t P, int L, int F, int64_t T, int K,
                    int mode, int vstride, int64_t NV, int l0, int l1, const float* pixel_order, void* stream);
/* backward: dtables (L,T,F) accumulated (caller zero-fills); dvert_w (NV,K) accumulated (caller zero-fills),
 * = sum over instances of c_v * <g, E_l[idx_k]>  (gradient w.r.t. the blend WEIGHT, before the blend's own backward). */
int gngf_encode_bwd(const float* xy, const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w,
                    const int32_t* n_ls, const float* genc, float* dtables, float* dvert_w,
                    int64_t P, int L, int F, int64_t T, int K, int mode, int vstride, int64_t NV, int l0, int l1,
                    void* stream);

/* backward of levels [l0, l1), spatial-hash source, WITHOUT one memory-side atomic per contribution (csrc/encode_bucket.hip: the
 * chip retires 20.6 G atomic row updates per second wherever the rows lie): the contributions are counting-sorted by table slice
 * ("bucket" = slot >> bucket_shift) and one workgroup per (level, bucket) sums its slice in a 64-bit fixed-point LDS image — the
 * result is bitwise reproducible.  Replaces the same reference lines as gngf_encode_bwd (models.py:382-392 backward).
 * gngf_encode_bwd_bucketed_plan: 1 and plan[0..5] = bucket_shift, buckets per level, pixel blocks, matrix ints, base ints, bytes of
 *   the item buffer — or 0 when the shape is not served (F not in {1,2,4}, > 8192 buckets per level, >= 2^31 contributions);
 *   image_bytes: LDS bytes of a bucket's image (1 KiB .. 128 KiB).
 * gngf_encode_bwd_bucketed: accumulate = 1 adds to dtables (L,T,F) fp32; 0 WRITES every row of levels [l0, l1) (no clear needed).
 *   matrix / base / items: scratch of the sizes the plan names.
 *   pixel_order (ABI 12, may be NULL): the batch's pixels as P records {x, y, bits(original index), 0} in TILE order (the `sorted`
 *   output of gngf_bin_pixels / gngf_bin_pixels2 for the same xy): walked in that order, a workgroup's contributions fall into a few
 *   hundred buckets instead of all of them (the hash keeps the low bits of gx) and leave as whole lines.  Same result bit for bit. */
int gngf_encode_bwd_bucketed_plan(int64_t P, int F, int64_t T, int nl, int image_bytes, int64_t* plan);
int gngf_encode_bwd_bucketed(const float* xy, const int32_t* n_ls, const float* genc, float* dtables, int64_t P, int L, int F,
                             int64_t T, int l0, int l1, int image_bytes, int accumulate, int32_t* matrix, int32_t* base,
                             void* items, const float* pixel_order, void* stream);

/* zeroes, in dtables (L,T,F) fp32, row hash(gx, gy) of level l for every vertex of levels [0, Ls): all the rows the staged levels
 * of a hash-indexed encoder can touch, whatever the batch (a gradient buffer kept from step to step needs no dense clear).
 * vtot = sum over those levels of (N_l + 2)^2.  Serves the zero_grad of functions.py:199 for such a buffer. */
int gngf_clear_hashed_rows(float* dtables, const int32_t* n_ls, int Ls, int F, int64_t T, int64_t vtot, void* stream);

/* ---- a5..a12 fused, "tiled" form: vertex stage + spatially binned, LDS-privatised pixel stage.
 * gngf_bin_pixels: bins P pixels into 4^tile_shift tiles of [0,1]^2.  NB binning blocks (<= 512); `chunk` = max pixels
 *   per work item.  Outputs: sorted (P,4) fp32 = x, y, bits(original index), 0;  items (max_items,4) int32 = start, count,
 *   tile, items-of-tile with max_items >= ceil(P/chunk) + 4^tile_shift;  n_items (4 int32: [0] = number of items, [1..3] = work counters of the persistent pixel-stage kernels, zeroed here and put
 *   back to zero by those kernels when they finish);  tile_off and tile_item_base
 *   (4^tile_shift + 1 each: exclusive prefixes of pixels / items per tile);  blockhist: scratch of 4^tile_shift * (NB + 1)
 *   int32. */
int gngf_bin_pixels(const float* xy, int64_t P, int tile_shift, int NB, int chunk, int32_t* blockhist, int32_t* tile_off,
                    int32_t* tile_item_base, int32_t* items, int32_t* n_items, float* sorted, void* stream);
/* gngf_encode_tiled_prepare: everything in front of the pixel stage as ONE chain of four launches — gngf_bin_pixels plus
 *   gngf_vertex_grid_fwd (levels [0, Ls), riding on the count launch as extra workgroups) plus two buffer clears:
 *   dG_zero (same shape as G; NULL: none) and zero_fill (zero_floats floats, a multiple of 4, 16-byte aligned — the table
 *   gradient buffer; NULL: none; riding on the scatter launch).  Parallel branches of a replayed hipGraph cost ~10 us per
 *   cross-queue dependency on this stack; riders on one chain cost nothing.
 *   G may be NULL for the spatial-hash source (round 5): no vertex grid is built — the pixel stage gathers from the level tables
 *   itself (gngf_encode_tiled_fwd_fused) — and the riders only clear (dG_zero / clear_rows), or do not run at all. */
int gngf_encode_tiled_prepare(const float* xy, int64_t P, int tile_shift, int NB, int chunk, int32_t* blockhist,
                              int32_t* tile_off, int32_t* tile_item_base, int32_t* items, int32_t* n_items, float* sorted,
                              const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w,
                              const int32_t* n_ls, const int32_t* n_ls_host, float* G, float* dG_zero, int dG_zero_words, int Ls,
                              int F, int64_t T, int K, int mode, int vstride, int64_t NV, float* zero_fill, int64_t zero_floats,
                              int32_t* persistent_ws, float* clear_rows, void* stream);
/* persistent_ws (optional): (2 * 4^tile_shift + 1) int32, ZERO before its first use and owned by one stream — with it (and
 * without zero_fill) the binning is TWO launches instead of four: the tile totals meet in global atomics, every scatter
 * workgroup scans them itself, and the last one out puts the workspace back to zero. */
/* clear_rows (optional, spatial-hash source only): an (L,T,F) fp32 table gradient kept from step to step — the vertex riders zero
 * row hash(gx, gy) of every staged vertex's level on the way (what gngf_clear_hashed_rows does, without a launch of its own). */
/* dG_zero_words: 1 = dG_zero is the fp32 vertex-grid gradient (vtot * F floats); 2 = it is the 64-bit fixed-point form of
 * gngf_encode_tiled_bwd's dG64 ((vtot * F + 2) 64-bit words, the two trailing words cleared as well). */
/* vertex stage: G[(goff_l + gy*(N_l+2) + gx)*F + f] for levels [0, Ls), goff_l = sum_{j<l} (N_j+2)^2.
 * n_ls_host mirrors n_ls on the host (grid sizing only). */
int gngf_vertex_grid_fwd(const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls,
                         const int32_t* n_ls_host, float* G, int Ls, int F, int64_t T, int K, int mode, int vstride,
                         int64_t NV, void* stream);
/* dG -> dtables (L,T,F) accumulated (caller zero-fills) and dvert_w (NV,K) accumulated (NULL when not needed). */
int gngf_vertex_grid_bwd(const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls,
                         const int32_t* n_ls_host, const float* dG, float* dtables, float* dvert_w, int Ls, int F, int64_t T,
                         int K, int mode, int vstride, int64_t NV, void* stream);
/* pixel stage over the work items of gngf_bin_pixels: enc / genc rows are (P, L*F); levels [0, Ls) handled here.
 * lds_bytes: dynamic LDS for the per-tile sub-grids (a level that does not fit falls back to global memory).
 * bwd ACCUMULATES into dG (caller zero-fills) in two passes with no global float atomics: every work item stores its
 * privatised sub-grid image to partials (max_items * lds_bytes/4 floats), then a gather pass sums, per vertex, the
 * images of the items that cover it.  Inside a work item the sub-grids accumulate in 64-bit fixed point (LDS float
 * atomics are ~20x slower than 64-bit integer ones on gfx950); `chunk` = the binning chunk (bounds the term count).
 * genc_absmax (optional): absmax_count floats, absmax_stride apart, whose maximum is a bound >= max |genc| that fixes the
 * fixed-point scale (one value, or the per-slab maxima gngf_decoder_bwd leaves in its workspace); NULL: each work item
 * scans its own rows first.
 * tile_level_off (4^tile_shift * Ls int32, optional): float offset of level l's sub-grid inside tile t's image at
 * [t * Ls + l], -1 when the level does not fit (levels are laid out back to back in ascending order, a level that
 * would exceed lds_bytes is skipped); NULL: the gather pass re-derives it per vertex. */
/* One binning job (the arguments of gngf_bin_pixels as a host struct): lets the binning of ANOTHER batch ride on the pixel-stage
 * launches of the current one — binning depends on the coordinates only, and the batches of an epoch are fixed slices of one
 * permutation, known in advance (functions.py:186-194).  persistent_ws: (2 * 4^tile_shift + 3) int32, ZERO before its first
 * use, owned by one sequence of steps and used by these entry points only (NOT shared with gngf_encode_tiled_prepare's): running
 * tile cursors, their values at the start of the current job, the count half's ticket, the task counter of the riding scatter —
 * never reset.  blockhist: 4^tile_shift * (NB + 1) int32 of scratch that carries each count block's reserved offsets to the
 * scatter block of the same index. */
typedef struct gngf_bin_job {
  const float* xy;            /* (P, 2) coordinates of the batch to bin */
  int64_t P;
  int tile_shift, NB, chunk;
  int32_t* blockhist;         /* 4^tile_shift * (NB + 1) scratch */
  int32_t* persistent_ws;
  int32_t* tile_off;          /* outputs, as gngf_bin_pixels */
  int32_t* tile_item_base;
  int32_t* items;
  int32_t* n_items;
  float* sorted;
} gngf_bin_job;
/* binning in two launches (count -> scatter with the scans inside); zero_fill (optional, zero_floats a multiple of 4, 16-byte
 * aligned) is cleared by rider workgroups of the count launch */
int gngf_bin_pixels2(const gngf_bin_job* job, float* zero_fill, int64_t zero_floats, void* stream);
/* pixel stage forward with the vertex stage forward fused into its staging loop (bit-identical to gngf_vertex_grid_fwd +
 * gngf_encode_tiled_fwd).  Level-interleaved kernel (F = 2, fp32 tables — see gngf_tiled_interleaved_applies): both index sources;
 * generic kernel (any other shape; round 5): spatial-hash source, fp32 or fp16 tables, next_count must be NULL.
 * next_count (optional): the count half of another batch's binning runs in extra workgroups at the head of the launch; its
 * scatter half rides on gngf_encode_tiled_bwd(..., next_bin) of the same step. */
int gngf_encode_tiled_fwd_fused(const float* sorted, const int32_t* items, const int32_t* n_items, int max_items,
                                const int32_t* n_ls, const int32_t* n_ls_host, const void* tables, int feat_dtype,
                                const int32_t* vert_idx, const float* vert_w, float* enc, int L, int Ls, int F, int64_t T, int K,
                                int mode, int vstride, int64_t NV, int tile_shift, int lds_bytes, const gngf_bin_job* next_count,
                                void* stream);
int gngf_encode_tiled_fwd(const float* sorted, const int32_t* items, const int32_t* n_items, int max_items,
                          const int32_t* n_ls, const int32_t* n_ls_host, const float* G, float* enc, int L, int Ls, int F,
                          int tile_shift, int lds_bytes, void* stream);
/* n_ls_host (host copy of n_ls; may be NULL): lets the launcher size the LEVEL-INTERLEAVED LDS images (F = 2, <= 16 staged
 * levels: vertex i of level l in row i, column l of 8-byte slots, so that the 16 level-lanes of a pixel never share an LDS
 * column); NULL, or gngf_set_tiled_interleaved(0): the back-to-back layout.  Results do not depend on the layout. */
int gngf_set_tiled_interleaved(int on);       /* returns the previous setting (default 1) */
/* 1 when gngf_encode_tiled_fwd (backward = 0) / gngf_encode_tiled_bwd (backward = 1) will run the level-interleaved kernel for
 * this shape under the current setting, else 0 (host logic only, no launch).  Only the interleaved backward fills dG64: callers
 * ask here before handing one over (gngf_encode_tiled_bwd rejects a dG64 + bound for a shape the generic kernels take). */
int gngf_tiled_interleaved_applies(const int32_t* n_ls_host, int Ls, int F, int tile_shift, int lds_bytes, int backward);
/* diagnostic: per-phase cycle totals of workgroup 0 of the interleaved backward kernel since the last call (read and cleared;
 * synchronises): [0] tail wait, [1] setup, [2] clear + bound, [3] main loop, [4] wait for the other waves, [5] store pass,
 * [6] items, [7] pixels */
int gngf_debug_il_stamps(unsigned long long* host8);
int gngf_encode_tiled_bwd(const float* sorted, const int32_t* items, const int32_t* n_items, int max_items,
                          const int32_t* tile_item_base, const int32_t* tile_level_off, const int32_t* n_ls,
                          const int32_t* n_ls_host, const float* genc, const float* genc_absmax, int absmax_count,
                          int absmax_stride, float* dG, float* partials, int L, int Ls, int F, int tile_shift, int lds_bytes,
                          int chunk, const float* ride_slabs, float* ride_dW0, float* ride_db0, float* ride_dW1, float* ride_db1,
                          float* ride_dW2, float* ride_db2, int64_t ride_P, int ride_in_dim, int ride_out_dim,
                          const float* gloss_promised, const float* gloss_arrived,
                          const float* mse_pred, const float* mse_label, float* mse_loss, float* mse_workspace, int64_t mse_n,
                          float* hash_dtables, int64_t hash_T, void* dG64, int log2_pixels, const gngf_bin_job* next_bin,
                          void* stream);
/* next_bin (optional; level-interleaved kernel only): the SCATTER half of another batch's binning — whose count half ran on
 * gngf_encode_tiled_fwd_fused(..., next_count) of the same step — as tasks the persistent workgroups claim once their own work
 * items are done (they run in the tail of the launch, where two workgroups in five are idle at the headline shape). */
/* ride_* (optional, ride_slabs NULL = none): the slab reduction of a preceding gngf_decoder_bwd that was called without
 * gradient pointers (= gngf_decoder_reduce(ride_slabs, ride_dW0 .. ride_db2, NULL, ride_P, ride_in_dim, ride_out_dim)) runs in
 * extra workgroups of this launch instead of a launch of its own (one dependent launch less on the step's critical path).
 * gloss_promised / gloss_arrived (optional device scalars, both or neither): genc came out of gngf_decoder_train, which ran its
 * backward with *gloss_promised before autograd delivered *gloss_arrived; if the two differ (relative 1e-6) every gradient
 * this launch writes (the vertex-grid gradient and the ridden decoder gradients) is NaN — checked on the device, no sync.
 * hash_dtables (optional; (L, hash_T, F) fp32 gradient buffer): spatial-hash index source on a single rank — the gather pass
 * adds every vertex's gradient straight to row _fast_hash(gx, gy) of the level's table gradient (= gngf_vertex_grid_bwd in hash
 * mode for levels [0, Ls)) and leaves dG unwritten: one launch and one round trip of dG less.
 * dG64 (optional; (vtot * F + 2) 64-bit words, ZERO on entry — gngf_encode_tiled_prepare clears it with dG_zero_words = 2) with
 * log2_pixels >= ceil(log2(P)) and a bound on |genc| (genc_absmax): the work items add their exact 64-bit fixed-point sums
 * straight into this vertex grid with global integer atomics (no partial images, no gather pass; bitwise reproducible), and
 * the launch ends with the conversion to dG (fp32) or — with hash_dtables — with the hash-source vertex stage reading dG64.
 * Used by the interleaved kernels (F = 2, <= 16 staged levels); ignored otherwise.
 * mse_* (optional, mse_pred NULL = none): likewise gngf_mse_fwd(mse_pred, mse_label, mse_loss, mse_workspace, mse_n) — the
 * VALUE of the pixel loss, which no kernel of the step reads. */
/* vertex stage backward for the vertex-table source in SLOT order (order (NV*K) int32 = argsort of vert_idx, flat):
 * contention-free for any slot distribution (wave-level segmented reduction, one atomic per (wave, slot run));
 * dtables accumulated, dvert_w (NV,K) written without atomics (NULL when not needed).
 * dG64 (optional, with vtot = sum_l (N_l+2)^2; dG may then be NULL): read the vertex-grid gradient from the 64-bit fixed-point grid
 * gngf_encode_tiled_bwd filled (called with dG = NULL, which skips its conversion to fp32) — one launch less. */
int gngf_vertex_grid_bwd_sorted(const void* tables, int feat_dtype, const int32_t* vert_idx, const float* vert_w, const int32_t* order,
                                const int32_t* n_ls, const float* dG, const void* dG64, int64_t vtot, float* dtables, float* dvert_w,
                                int Ls, int F, int64_t T, int K, int vstride, int64_t NV, void* stream);

/* Large aligned GEMMs (full 128 x 128 tiles, contraction a multiple of 32) of the dense-layer entry points run on the
 * split-bf16 kernels while this is non-zero: every fp32 operand is split exactly into three bf16 terms and six of the nine cross
 * products are accumulated in fp32 on the bf16 matrix pipe (error <= ~2e-7 sum |a_k b_k|, at or below the rounding of an
 * fp32 fma chain; 2-2.7x the fp32-MFMA rate).  Process-wide; returns the previous setting.  Default 0 (exact fp32 MFMA).
 *   1  (any other non-zero value too): operands split once on the way into LDS, bf16 planes (round 5)
 *   2  as 1, and GEMMs that ACCUMULATE into their output (gngf_linear_bwd_weight, gngf_gemm_acc) use two planes and three
 *      products: |error| <= 3 * 2^-18 |a b| per product (the HashProbDistribution's dW / dh, contractions of >= 4096 terms)
 *   17 the round-4 kernel: every wave splits the fragments it reads (kept for A/B; same numbers as 1) */
int gngf_set_gemm_split_bf16(int mode);
/* The same switch for the fused decoder (gngf_decoder_fwd / gngf_decoder_bwd) at in_dim 32 or 64: exact three-way bf16
 * split of every operand, six cross products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  While on, the hidden-layer
 * buffer is neither written by the forward nor read by the backward (the backward recomputes the two layers).  Process-wide;
 * returns the previous setting. */
int gngf_set_decoder_split_bf16(int on);
/* gngf_decoder_bwd at in_dim == 32 with the saved hidden layers: 1 (default) = hybrid kernel — the two products with the pixel
 * on the lane (dh1 = W1^T dz2, d enc = W0^T dz1) on the bf16 pipe with the exact three-way split, the weight-gradient products
 * on the fp32 pipe; 0 = everything on the fp32 pipe.  Process-wide; returns the previous setting. */
int gngf_set_decoder_bwd_hybrid(int on);
/* Forward AND backward of the decoder in ONE launch for a training step whose loss is torch.nn.MSELoss()(rgb, target)
 * (utils.py:99) with a KNOWN upstream gradient *gloss (device scalar, e.g. the loss weight l_mse of functions.py:243): rgb
 * (P,out_dim) is written, d enc and the parameter gradients as by gngf_decoder_bwd(target, gloss).  The tile's hidden layers
 * stay in registers between its forward and backward part (no hidden-layer buffer, no separate forward launch).  in_dim == 32
 * only (hipErrorInvalidValue otherwise: use gngf_decoder_fwd + gngf_decoder_bwd).  Gradient pointers all NULL: the caller
 * runs gngf_decoder_reduce on the slabs (as with gngf_decoder_bwd). */
int gngf_decoder_train(const float* enc, const float* target, const float* gloss, const float* W0, const float* b0,
                       const float* W1, const float* b1, const float* W2, const float* b2, float* rgb, float* denc,
                       float* dW0, float* db0, float* dW1, float* db1, float* dW2, float* db2, float* slabs,
                       float* denc_absmax, float* zero_fill, int64_t zero_floats, int64_t P, int in_dim, int out_dim, int leaky,
                       void* stream);
/* zero_fill (optional): zero_floats floats (a multiple of 4, 16-byte aligned) cleared by this launch on the way — the table
 * gradient the encoder backward will accumulate into; the stores issue between the kernel's MFMAs (no issue time, idle HBM)
 * instead of in rider workgroups of gngf_encode_tiled_prepare. */
/* ---- dense layers on the matrix cores (exact-fp32 MFMA).  act: 0 none, 1 ReLU, 2 LeakyReLU(0.01), 3 Sigmoid.
 * nn.Linear + activation of HashProbDistribution (models.py:80-88,105-106) and of the decoder (models.py:382-392). */
int gngf_linear_fwd(const float* X, const float* W, const float* b, float* Y, int64_t M, int N, int K, int act, void* stream);
/* dX[M,K] = (dY .* act'(Y)) W ;  Y = activated output of the layer (may be NULL when act == 0) */
int gngf_linear_bwd_input(const float* dY, const float* Y, const float* W, float* dX, int64_t M, int N, int K, int act,
                          void* stream);
/* dW[N,K] += (dY .* act'(Y))^T X ; db[N] += column sums.  Accumulates (float atomics): zero-fill for a fresh gradient. */
int gngf_linear_bwd_weight(const float* dY, const float* Y, const float* X, float* dW, float* db, int64_t M, int N, int K,
                           int act, void* stream);
/* C[M,N] += opA(A) opB(B), contraction Kc split over blocks.  ta: A stored (Kc,M); tb: B stored (N,Kc). */
int gngf_gemm_acc(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t Kc, int ta, int tb, void* stream);

/* ---- a13: the decoder MLP fused (models.py:382-392,469-470): Linear(in,64)+act, Linear(64,64)+act, Linear(64,out)+Sigmoid,
 * act = ReLU (leaky = 0) or LeakyReLU(0.01) (leaky = 1); in_dim <= 64, out_dim <= 4.  W* are (out,in) like nn.Linear. */
/* hidden (optional): gngf_decoder_hidden_floats(P) floats that receive the two activated hidden layers (512 B / pixel) for
 * gngf_decoder_bwd — the stores ride under the forward kernel's MFMAs, and the backward kernel then reads them back
 * instead of recomputing them (96 of its 292 MFMAs per 32 pixels). */
int gngf_decoder_fwd(const float* enc, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                     const float* b2, float* rgb, float* hidden, int64_t P, int in_dim, int out_dim, int leaky, void* stream);
int64_t gngf_decoder_hidden_floats(int64_t P);
/* backward: denc (P,in_dim) and the six parameter gradients, each WRITTEN (not accumulated); rgb = the forward output.
 * slabs: workspace of gngf_decoder_bwd_slabs(P) * gngf_decoder_slab_floats(in_dim, out_dim) floats.
 * denc_absmax (1 float, optional): receives max |denc| (NaN if any element is NaN) — a bound the tiled encoder backward
 * can take instead of scanning its input once more.
 * hidden (optional): the buffer gngf_decoder_fwd filled for the same enc and weights; NULL: recompute. */
int gngf_decoder_bwd(const float* enc, const float* rgb, const float* drgb, const float* target, const float* gloss,
                     const float* W0, const float* b0, const float* W1,
                     const float* b1, const float* W2, float* denc, float* dW0, float* db0, float* dW1, float* db1, float* dW2,
                     float* db2, float* slabs, float* denc_absmax, const float* hidden, float* zero_fill, int64_t zero_floats,
                     int64_t P, int in_dim, int out_dim, int leaky, void* stream);
/* zero_fill (optional; zero_floats floats, a multiple of 4, 16-byte aligned): cleared by the kernel on the way, a few 16-byte
 * stores per lane and tile (as gngf_decoder_train's) — the table gradient the encoder backward is about to accumulate into, at
 * shapes that have no fused training kernel (64 input features: BASELINE config 5). */
/* target + gloss (1 float), optional, instead of drgb: the gradient of the fused pixel loss,
 * d rgb = gloss[0] * 2 / (P * out_dim) * (rgb - target) (MSELoss backward), is formed inside the kernel (drgb may be NULL). */
 /* The six gradient pointers may ALL be NULL: the kernel then stops at the slabs and gngf_decoder_reduce finishes (e.g. on a
  * second stream, beside the encoder backward — which only needs max |denc|: the last float of every slab is that slab's
  * maximum, i.e. genc_absmax = slabs + slab_floats - 1, absmax_count = gngf_decoder_bwd_slabs(P), absmax_stride = slab_floats). */
int gngf_decoder_reduce(const float* slabs, float* dW0, float* db0, float* dW1, float* db1, float* dW2, float* db2,
                        float* denc_absmax, const float* gloss_promised, const float* gloss_arrived, int64_t P, int in_dim,
                        int out_dim, void* stream);
/* gloss_promised / gloss_arrived: as for gngf_encode_tiled_bwd (NULL, NULL: nothing was promised). */
int gngf_decoder_bwd_slabs(int64_t P);
/* measurement: duration (ns, device 100 MHz clock; first workgroup start -> last workgroup end) of the most recent
 * gngf_decoder_bwd main kernel — usable when the call sits inside a replayed hipGraph, where events cannot be recorded.
 * Synchronises the device. */
int gngf_decoder_bwd_last_span_ns(double* ns);
int gngf_decoder_slab_floats(int in_dim, int out_dim);

/* ---- a7/a8 tail, per DISTINCT vertex: Softmax(dim=-1) + nan_to_num + top-K (models.py:85,111,116; 5-19).
 * logits_probs (U,T): logits in, probabilities out (in place).  topk_val (U,K) sorted descending, topk_idx (U,K) int32;
 * ties resolve to the LOWER slot index (torch.topk leaves tie order unspecified).  rowstat (U,2) = (row max, row sum of
 * exp(z - max); NaN marks a NaN-poisoned row), optional (NULL): lets the backward rebuild p from recomputed logits. */
int gngf_softmax_topk(float* logits_probs, float* topk_val, int32_t* topk_idx, float* rowstat, int64_t U, int64_t T, int K,
                      void* stream);
/* streaming form for the chunked per-vertex path: logits are only READ (the (rows,T) distribution is never written):
 * one pass gives the online row statistics and the top-K (selected on the logits; probabilities = exp(z_k - max)/sum),
 * a second pass accumulates pbar (L,T) += mw (U,L)^T softmax(logits) when mw is given (L = 0: skipped). */
int gngf_logits_topk_pbar(const float* logits, float* topk_val, int32_t* topk_idx, float* rowstat, const float* mw, int L,
                          float* pbar, int64_t U, int64_t T, int K, void* stream);
/* DifferentiableTopk.forward alone (models.py:11) on arbitrary rows x (U,T); same ordering rules. */
int gngf_topk(const float* x, float* topk_val, int32_t* topk_idx, int64_t U, int64_t T, int K, void* stream);
/* backward of the above without the dense zero-filled scatter of models.py:27-35:
 *   g = gdense (U,T, optional) + mw (U,L) * G (L,T) (optional low-rank term of the batch-mean loss) + dq at topk_idx
 *   dlogits = p .* (g - <p, g>)        dlogits may alias probs. */
int gngf_softmax_bwd(const float* probs, const float* dq, const int32_t* topk_idx, const float* gdense, const float* mw,
                     const float* G, int L, float* dlogits, int64_t U, int64_t T, int K, void* stream);

/* The logits product with ROW STATISTICS in its epilogue, and their consumers (round 4): gngf_linear_fwd_rowstats = gngf_linear_fwd
 * without activation + rowparts (M, N / 64) float pairs (max, sum exp(y - max)) per row and 64-column block (whole 128 x 128 tiles
 * only: M % 128 == 0, N % 128 == 0, K % 32 == 0, 16-byte aligned — else hipErrorInvalidValue and the caller takes gngf_linear_fwd +
 * gngf_logits_topk_pbar); gngf_rowstats_topk = the statistics / top-K half of gngf_logits_topk_pbar from those partials (it reads
 * K x 64 logits per row instead of all T: the K largest logits lie inside the K blocks with the largest maxima);
 * gngf_pbar_accumulate = its batch-mean half.  Same results as gngf_logits_topk_pbar (models.py:85,105-123; utils.py:138,159). */
int gngf_linear_fwd_rowstats(const float* X, const float* W, const float* b, float* Y, float* rowparts, int64_t M, int N, int K,
                             void* stream);
int gngf_rowstats_topk(const float* logits, const float* rowparts, float* topk_val, int32_t* topk_idx, float* rowstat, int64_t U,
                       int64_t T, int K, void* stream);
int gngf_pbar_accumulate(const float* logits, const float* rowstat, const float* mw, int L, float* pbar, int64_t U, int64_t T,
                         void* stream);
/* the same backward from RECOMPUTED LOGITS, in place (logits in, d logits out), for the chunked per-vertex path:
 * p = exp(z - rowstat.max) / rowstat.sum;  g = mw (U,L) * G (L,T) + dq at topk_idx;  dz = p .* (g - <p,g>);
 * db (T) += column sums of dz (NULL: skipped).  scratch: U + U*K floats.  L = 0 / K = 0 drop the respective term. */
int gngf_softmax_bwd_lowrank(float* logits_dz, const float* rowstat, const float* dq, const int32_t* topk_idx, const float* mw,
                             const float* G, int L, float* db, float* scratch, const float* topk_p, int64_t U, int64_t T, int K,
                             void* stream);
/* The same backward WITHOUT the d-logits matrix (round 5): gngf_hpd_bwd_dot leaves the row dots
 *   dot[r] = sum_k topk_p[r,k] dq[r,k] + sum_t p[r,t] (mw G)[r,t]
 * (one read of the logits), and gngf_hpd_bwd_fused forms dz = p .* (mw G - dot) (+ p_k dq_k at the K top-K slots of a row) inside
 * the operand loaders of the last layer's two backward GEMMs, which read the LOGITS:
 *   dW (T,hidden) += dz^T h,  db (T) += column sums of dz (NULL: skipped),  dH (U,hidden) += dz W        (all three accumulate)
 * — what gngf_softmax_bwd_lowrank + gngf_linear_bwd_weight + gngf_gemm_acc compute (the autograd backward of models.py:84-85,
 * 105-116 under the batch-mean loss utils.py:138,159), with two passes over the (U,T) matrix instead of five.
 * The small operands are split ONCE per backward pass into bf16 planes by gngf_hpd_bwd_prepare:
 *   h (rows,hidden) = the last hidden layer of ALL vertices -> hp (planes, rows, hidden);  mw (rows,L) -> mwp (3, rows, 16), zero-padded
 *   W (T,hidden) = the last layer's weight -> Wp (planes, T, hidden);                      G (L,T) -> Gtp (3, T, 16), transposed
 * (either half may be skipped: h == NULL / W == NULL; L = 0: mw / G may be NULL).  planes: 3 = every fp32 value split exactly into
 * three bf16 terms (six products, fp32 accumulation), 2 = two terms, three products (|error| <= 3 * 2^-18 |a b| per product); the
 * operands of the small product mw G always keep the exact three.  gngf_hpd_bwd_fused takes hp / mwp AT THE CHUNK'S FIRST ROW
 * (pointer + u0 * hidden resp. + u0 * 16 elements) and rows_total = the rows they were prepared with (the plane stride); h (U,hidden)
 * and W in fp32 serve the K top-K terms only (K = 0: unused).  Diagnostic: planes + 16 / + 32 / + 48 run the dW + db part / the dH
 * part / the top-K terms alone.
 * Shapes: gngf_hpd_bwd_fused_applies(U, T, L, K, hidden) != 0 (U % 128 == 0, T % 128 == 0, T < 2^22, hidden == 128, L <= 16);
 * anything else is rejected (hipErrorInvalidValue) — the caller then takes the three separate entry points. */
int gngf_hpd_bwd_dot(const float* logits, const float* rowstat, const float* dq, const float* topk_p, const float* mw,
                     const float* G, int L, float* dot, int64_t U, int64_t T, int K, void* stream);
int gngf_hpd_bwd_fused_applies(int64_t U, int64_t T, int L, int K, int hidden);
int gngf_hpd_bwd_prepare(const float* h, const float* mw, int64_t rows, void* hp, void* mwp, const float* W, const float* G, int64_t T,
                         void* Wp, void* Gtp, int L, int hidden, int planes, void* stream);
int gngf_hpd_bwd_fused(const float* logits, const float* rowstat, const float* dot, const float* dq, const float* topk_p,
                       const int32_t* topk_idx, const void* hp, const void* mwp, int64_t rows_total, const void* Wp, const void* Gtp,
                       const float* h, const float* W, float* dW, float* db, float* dH, int64_t U, int64_t T, int K, int hidden,
                       int planes, void* stream);
/* topk_p (optional; (U,K)): the top-K probabilities gngf_logits_topk_pbar / gngf_softmax_topk returned for these logits — the
 * backward then takes p at the top-K slots from them instead of reading the logits again at random. */

/* verts[i] = (gx, gy) fp32 of vertex id u0+i (vid = gy*vstride + gx): the raw-integer HPD input of models.py:416-418 */
int gngf_vertex_coords(float* verts, int64_t u0, int64_t count, int vstride, void* stream);
/* blend weights of the K rows (models.py:212-217) on the per-vertex table, and their backward (dw -> dq) */
int gngf_blend_fwd(const float* q, float* w, int64_t U, int K, int blend, void* stream);
int gngf_blend_bwd(const float* q, const float* dw, float* dq, int64_t U, int K, int blend, void* stream);
/* counts (L,NV) int32 += number of (pixel, corner) instances of each vertex per level (caller zero-fills);
 * mw (NV,L) = counts / denom : weights of the batch-mean distribution p-bar_l (utils.py:138,159). */
int gngf_vertex_multiplicity(const float* xy, const int32_t* n_ls, int32_t* counts, int64_t P, int L, int vstride, int64_t NV,
                             void* stream);
int gngf_multiplicity_weights(const int32_t* counts, float* mw, int64_t NV, int L, float denom, void* stream);
/* reference-shaped outputs rebuilt from the per-vertex table: vid_out (P,L,4) int64, out_idx (P,L,4,K) int64,
 * out_val (P,L,4,K) fp32 (any of the three may be NULL) — models.py:478-484 return contract. */
int gngf_expand_vertex_table(const float* xy, const int32_t* n_ls, const int32_t* src_idx, const float* src_val,
                             int64_t* vid_out, int64_t* out_idx, float* out_val, int64_t P, int L, int K, int vstride,
                             int64_t NV, void* stream);

/* ---- pixel loss (row f1 of the scope table: the caller of the path) -------------------------------------------------
 * torch.nn.MSELoss() of utils.py:99 on the (P,out_dim) outputs: loss[0] = mean((pred - label)^2) over n elements.
 * workspace: gngf_mse_workspace_floats() floats, 8-byte aligned, zero-filled once before the first call (the kernel
 * resets it).  One launch; workgroup partials meet in a double-precision atomic. */
int gngf_mse_workspace_floats(void);
int gngf_mse_blocks(int64_t n);      /* workgroups gngf_mse_fwd launches for n elements */
int gngf_mse_fwd(const float* pred, const float* label, float* loss, float* workspace, int64_t n, void* stream);
/* its backward: dpred (n) = gout[0] * 2 (pred - label) / n   (gout: device scalar, the gradient of the loss value) */
int gngf_mse_bwd(const float* pred, const float* label, const float* gout, float* dpred, int64_t n, void* stream);

/* the distribution term of the same Loss (utils.py:122-174) on the batch-mean distribution pbar (L,T):
 *   out[l] = -(gamma + eps) * JS(pbar_l, uniform) + eps * KL(uniform || pbar_l),
 * both built from torch.nn.KLDivLoss(reduction='batchmean') on 1-D rows, i.e. divided by T.
 * workspace: gngf_js_kl_workspace_doubles(L) doubles (8-byte aligned, no initialisation needed). */
int gngf_js_kl_workspace_doubles(int L);
int gngf_js_kl_fwd(const float* pbar, float* out, double* workspace, int L, int64_t T, float gamma, float eps, void* stream);
/* its backward: dpbar (L,T) = gout[l] * d out[l] / d pbar[l,t] */
int gngf_js_kl_bwd(const float* pbar, const float* gout, float* dpbar, int L, int64_t T, float gamma, float eps, void* stream);

/* ---- collision statistics (row f3) ----------------------------------------------------------------------------------
 * counts (K,L) int32 = number of distinct slots in [0,T) among indices[:, l, :, k], indices (P,L,V,K) int64 contiguous
 * (K = 1: the hash source's (P,L,V)) — the torch.unique(...).numel() of calc_hash_collisions, models.py:568-619.
 * bitmap: gngf_slot_bitmap_words(L, K, T) uint32 words of workspace (cleared by the call). */
int64_t gngf_slot_bitmap_words(int L, int K, int64_t T);
int gngf_distinct_slot_counts(const int64_t* indices, int64_t P, int L, int V, int K, int64_t T, uint32_t* bitmap,
                              int32_t* counts, void* stream);
/* The same statistic without the index tensor (2 GiB per step at the headline shape): both index sources depend on (level,
 * vertex) only, so a batch's distinct slots are the slots of its touched vertices.  Marks the slots batch `xy` uses into
 * `bitmap` (gngf_slot_bitmap_words(L, K, T) 32-bit words, ACCUMULATED over calls: the caller clears it once per epoch);
 * vert_idx (NV,K) int32 with vid = gy * vstride + gx, or NULL for the spatial hash (K = 1, vstride >= N_max + 2, NV = vstride^2);
 * touched: workspace of L * ceil(NV / 32) words.  gngf_count_slot_bits then gives counts (K,L) as gngf_distinct_slot_counts. */
int gngf_mark_batch_slots(const float* xy, const int32_t* n_ls, int64_t P, int L, const int32_t* vert_idx, int K, int64_t T,
                          int vstride, int64_t NV, uint32_t* touched, uint32_t* bitmap, void* stream);
int gngf_count_slot_bits(const uint32_t* bitmap, int L, int K, int64_t T, int32_t* counts, void* stream);

/* ---- optimizer (row f2: the caller of the path) ---------------------------------------------------------------------
 * torch.optim.Adam as get_optimizer builds it (functions.py:96-127: betas (0.9, 0.99), eps 1e-15, L2-style weight
 * decay, dense moments, per-group lr), every tensor of every group in ONE launch.
 * segments: device array of nseg 64-byte records
 *     { void* param; const void* grad; float* exp_avg; float* exp_avg_sq; float* master; int64_t n; int64_t first_block;
 *       int32_t group; int32_t flags; }
 * with first_block = running sum of ceil(n / gngf_adam_block_elems()) and total_blocks that sum over all segments.
 * flags bit 0: param and grad are fp16 (fp16 level tables, BASELINE.json config 5) and `master` is the fp32 master copy
 * the update is applied to (param = round(master)); moments are fp32 either way; bit 1 (only with bit 0): `grad` is fp32
 * all the same — the buffer the table gradient was accumulated in, no fp16 copy.  inv_grad_scale multiplies every gradient
 * before use (1 / loss scale of fp16 training; 1 = off).
 * step: device float holding the number of steps taken so far; the call increments it and uses the new value (bias
 * corrections in double precision), so a step is capturable in a hipGraph.  lr, weight_decay: HOST arrays of ngroups
 * (<= GNGF_ADAM_MAX_GROUPS) values, indexed by a segment's `group`. */
#define GNGF_ADAM_MAX_GROUPS 4
int gngf_adam_block_elems(void);
int gngf_adam_step(const void* segments, int nseg, int64_t total_blocks, float* step, const float* lr,
                   const float* weight_decay, int ngroups, float beta1, float beta2, float eps, float inv_grad_scale,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNGF_H_ */
