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
 *   - fp32 everywhere unless stated; indices handed to / from the reference-shaped API are int64,
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

#define GNGF_ABI_VERSION 1
#define GNGF_MAX_LEVELS 32
#define GNGF_MAX_TOPK 32

/* blend of the K looked-up rows — models.py:212-217 (global should_softmax_topk_features) */
enum { GNGF_BLEND_SOFTMAX = 0 /* True */, GNGF_BLEND_RAW = 1 /* None */, GNGF_BLEND_NORM = 2 /* False */ };

/* index source of the fused encoder */
enum { GNGF_MODE_HASH = 0 /* models.py:412-414 */, GNGF_MODE_VERTEX_TABLE = 1 /* models.py:416-418, de-duplicated */ };

int gngf_abi_version(void);

/* ---- a5+a6: _scale_to_grid + _fast_hash (models.py:486-528) -> idx (P,L,4) int64, as the module returns it */
int gngf_hash_indices(const float* xy, const int32_t* n_ls, int64_t* idx, int64_t P, int L, int64_t T, void* stream);

/* ---- a10/a11: MultiResHashEncoding.forward (models.py:173-229).
 * idx (P,L,4) when K == 0 (hash branch) or (P,L,4,K) (GNGF branch); probs (P,L,4,K) or NULL; out (P,F,L,4). */
int gngf_mrhe_fwd(const float* tables, const int64_t* idx, const float* probs, float* out,
                  int64_t P, int L, int F, int64_t T, int K, int blend, void* stream);
/* a15 at the same boundary: dtables (L,T,F) is ACCUMULATED into (caller zero-fills), dprobs (P,L,4,K) is written. */
int gngf_mrhe_bwd(const float* tables, const int64_t* idx, const float* probs, const float* gout,
                  float* dtables, float* dprobs, int64_t P, int L, int F, int64_t T, int K, int blend, void* stream);

/* ---- a12: _bilinear_interpolate (models.py:621-655): feats (P,F,L,4) -> enc (P,L*F); bwd writes dfeats. */
int gngf_bilinear_fwd(const float* xy, const int32_t* n_ls, const float* feats, float* enc,
                      int64_t P, int L, int F, void* stream);
int gngf_bilinear_bwd(const float* xy, const int32_t* n_ls, const float* genc, float* dfeats,
                      int64_t P, int L, int F, void* stream);

/* ---- a5..a12 fused, "direct" form: one lane per (pixel, level); table rows gathered straight from HBM/L2.
 * mode HASH: vert_idx/vert_w NULL, K ignored.
 * mode VERTEX_TABLE: vert_idx (NV,K) int32 slots and vert_w (NV,K) blend weights per grid vertex
 *   (vid = gy*vstride + gx), i.e. HPD(top-K) evaluated once per DISTINCT vertex instead of per instance. */
int gngf_encode_fwd(const float* xy, const float* tables, const int32_t* vert_idx, const float* vert_w,
                    const int32_t* n_ls, float* enc, int64_t P, int L, int F, int64_t T, int K,
                    int mode, int vstride, int64_t NV, void* stream);
/* backward: dtables (L,T,F) accumulated (caller zero-fills); dvert_w (NV,K) accumulated (caller zero-fills),
 * = sum over instances of c_v * <g, E_l[idx_k]>  (gradient w.r.t. the blend WEIGHT, before the blend's own backward). */
int gngf_encode_bwd(const float* xy, const float* tables, const int32_t* vert_idx, const float* vert_w,
                    const int32_t* n_ls, const float* genc, float* dtables, float* dvert_w,
                    int64_t P, int L, int F, int64_t T, int K, int mode, int vstride, int64_t NV, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNGF_H_ */
