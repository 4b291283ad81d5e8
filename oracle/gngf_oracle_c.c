/*
 * CPU ORACLE in plain C (OpenMP) — TEST INFRASTRUCTURE ONLY (same status as oracle/gngf_oracle.py).
 *
 * Restates the fused per-pixel hot path of the reference for timing on the host cores (bench.py `cpu_baseline`,
 * kind "port") and for cross-checking the numpy oracle at sizes numpy is too slow for:
 *   _scale_to_grid            models.py:486-502
 *   _fast_hash                models.py:504-528 (int32 wrap-around of the prime product, non-negative remainder)
 *   MultiResHashEncoding      models.py:173-229 (hash branch / GNGF branch with per-vertex (idx, weight) tables)
 *   _bilinear_interpolate     models.py:621-655
 *   decoder MLP               models.py:382-392, 469-470
 *   and the backward of all of it: scatter-add into the tables, decoder weight gradients.
 * Parity: pinned through tests/test_oracle_c.py against oracle/gngf_oracle.py, which is pinned to the reference goldens.
 * The product never links or calls this file.
 *
 * Build: make -C oracle   ->  oracle/libgngf_oracle_c.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HID 64

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static inline int64_t hash2(int gx, int gy, int64_t T) {
  int32_t h = gx ^ (int32_t)((uint32_t)gy * 2654435761u);
  int64_t r = (int64_t)h % T;
  return r < 0 ? r + T : r;
}

typedef struct { int gx, gy; float c[4]; } cell_t;

static inline cell_t make_cell(float x, float y, int n) {
  cell_t r;
  float fn = (float)n, sx = x * fn, sy = y * fn;
  float ax = floorf(sx), ay = floorf(sy), dx = ax + 1.0f, dy = ay + 1.0f;
  float wx0 = dx - sx, wx1 = sx - ax, wy0 = dy - sy, wy1 = sy - ay;
  r.c[0] = wx0 * wy0; r.c[1] = wx1 * wy0; r.c[2] = wx0 * wy1; r.c[3] = wx1 * wy1;
  r.gx = (int)ax; r.gy = (int)ay;
  return r;
}

/* enc (P, L*F).  vert_idx/vert_w NULL -> hash mode; else (NV,K) tables with vid = gy*vstride + gx. */
void orc_encode_fwd(const float* xy, const float* tables, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls,
                    float* enc, int64_t P, int L, int F, int64_t T, int K, int vstride) {
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < P; ++p)
    for (int l = 0; l < L; ++l) {
      cell_t c = make_cell(xy[2 * p], xy[2 * p + 1], n_ls[l]);
      const float* tab = tables + (int64_t)l * T * F;
      float feat[4][8];
      for (int v = 0; v < 4; ++v) {
        int gx = c.gx + (v & 1), gy = c.gy + (v >> 1);
        if (!vert_idx) {
          const float* r = tab + hash2(gx, gy, T) * F;
          for (int f = 0; f < F; ++f) feat[v][f] = r[f];
        } else {
          int64_t vid = (int64_t)gy * vstride + gx;
          for (int f = 0; f < F; ++f) feat[v][f] = 0.f;
          for (int k = 0; k < K; ++k) {
            const float* r = tab + (int64_t)vert_idx[vid * K + k] * F;
            float w = vert_w[vid * K + k];
            for (int f = 0; f < F; ++f) feat[v][f] += r[f] * w;
          }
        }
      }
      for (int f = 0; f < F; ++f)
        enc[(p * L + l) * F + f] = ((feat[0][f] * c.c[0] + feat[1][f] * c.c[1]) + feat[2][f] * c.c[2]) + feat[3][f] * c.c[3];
    }
}

/* dtables (L,T,F) += scatter of genc; double accumulation is NOT used: float atomics as a CPU port would do. */
void orc_encode_bwd(const float* xy, const float* tables, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls,
                    const float* genc, float* dtables, int64_t P, int L, int F, int64_t T, int K, int vstride) {
  (void)tables;
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < P; ++p)
    for (int l = 0; l < L; ++l) {
      cell_t c = make_cell(xy[2 * p], xy[2 * p + 1], n_ls[l]);
      float* dtab = dtables + (int64_t)l * T * F;
      for (int v = 0; v < 4; ++v) {
        int gx = c.gx + (v & 1), gy = c.gy + (v >> 1);
        if (!vert_idx) {
          float* r = dtab + hash2(gx, gy, T) * F;
          for (int f = 0; f < F; ++f) {
            float add = genc[(p * L + l) * F + f] * c.c[v];
#pragma omp atomic
            r[f] += add;
          }
        } else {
          int64_t vid = (int64_t)gy * vstride + gx;
          for (int k = 0; k < K; ++k) {
            float* r = dtab + (int64_t)vert_idx[vid * K + k] * F;
            float w = vert_w[vid * K + k];
            for (int f = 0; f < F; ++f) {
              float add = (genc[(p * L + l) * F + f] * c.c[v]) * w;
#pragma omp atomic
              r[f] += add;
            }
          }
        }
      }
    }
}

/* The same scatter with the SUM kept in double precision (dtables64 (L,T,F) double, zero on entry).
 * exact_products = 0: every term is the reference's own fp32 product chain (genc * c, then * w), only the accumulation is wide
 * (an fp32 accumulation of ~10^4 terms per coarse-level row carries ~1e-5 of order-dependent rounding of its own);
 * exact_products = 1: the products are formed in double as well — the mathematically exact gradient of the fp32 inputs, what a
 * relative check of rows that are small through CANCELLATION has to be made against (each fp32-rounded product is off by 6e-8 of
 * its own size, which is not small next to a sum that cancels). */
void orc_encode_bwd_f64(const float* xy, const int32_t* vert_idx, const float* vert_w, const int32_t* n_ls, const float* genc,
                        double* dtables64, int64_t P, int L, int F, int64_t T, int K, int vstride, int exact_products) {
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < P; ++p)
    for (int l = 0; l < L; ++l) {
      cell_t c = make_cell(xy[2 * p], xy[2 * p + 1], n_ls[l]);
      double* dtab = dtables64 + (int64_t)l * T * F;
      for (int v = 0; v < 4; ++v) {
        int gx = c.gx + (v & 1), gy = c.gy + (v >> 1);
        if (!vert_idx) {
          double* r = dtab + hash2(gx, gy, T) * F;
          for (int f = 0; f < F; ++f) {
            const float g = genc[(p * L + l) * F + f];
            const double add = exact_products ? (double)g * (double)c.c[v] : (double)(g * c.c[v]);
#pragma omp atomic
            r[f] += add;
          }
        } else {
          int64_t vid = (int64_t)gy * vstride + gx;
          for (int k = 0; k < K; ++k) {
            double* r = dtab + (int64_t)vert_idx[vid * K + k] * F;
            float w = vert_w[vid * K + k];
            for (int f = 0; f < F; ++f) {
              const float g = genc[(p * L + l) * F + f];
              const double add = exact_products ? (double)g * (double)c.c[v] * (double)w : (double)((g * c.c[v]) * w);
#pragma omp atomic
              r[f] += add;
            }
          }
        }
      }
    }
}

/* decoder in -> 64 -> 64 -> out (ReLU, ReLU, Sigmoid).  W* are (out,in).  h1,h2 (P,64) kept for backward.
 * Every dot product keeps its k-ordered chain of separately rounded multiply-adds (what a scalar port does), but the loops
 * run over the OUTPUT index innermost on transposed weights, so that the compiler vectorises them (a dependent sum over k
 * cannot be vectorised without re-association): the port is not a straw man next to the reference's MKL GEMMs. */
static void transpose(const float* W, float* WT, int rows, int cols) {      /* W (rows, cols) -> WT (cols, rows) */
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) WT[c * rows + r] = W[r * cols + c];
}

void orc_decoder_fwd(const float* x, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
                     const float* b2, float* h1, float* h2, float* y, int64_t P, int in_dim, int out_dim) {
  float* W0T = (float*)malloc(sizeof(float) * (size_t)in_dim * HID);
  float* W1T = (float*)malloc(sizeof(float) * HID * HID);
  transpose(W0, W0T, HID, in_dim);
  transpose(W1, W1T, HID, HID);
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < P; ++p) {
    const float* xp = x + p * in_dim;
    float* a1 = h1 + p * HID;
    float* a2 = h2 + p * HID;
    float s[HID];
    for (int j = 0; j < HID; ++j) s[j] = b0[j];
    for (int k = 0; k < in_dim; ++k) {
      const float xk = xp[k];
      const float* w = W0T + k * HID;
#pragma omp simd
      for (int j = 0; j < HID; ++j) s[j] += w[j] * xk;
    }
#pragma omp simd
    for (int j = 0; j < HID; ++j) a1[j] = s[j] > 0.f ? s[j] : 0.f;
    for (int j = 0; j < HID; ++j) s[j] = b1[j];
    for (int k = 0; k < HID; ++k) {
      const float ak = a1[k];
      const float* w = W1T + k * HID;
#pragma omp simd
      for (int j = 0; j < HID; ++j) s[j] += w[j] * ak;
    }
#pragma omp simd
    for (int j = 0; j < HID; ++j) a2[j] = s[j] > 0.f ? s[j] : 0.f;
    for (int c = 0; c < out_dim; ++c) {
      float t = b2[c];
      for (int k = 0; k < HID; ++k) t += W2[c * HID + k] * a2[k];
      y[p * out_dim + c] = 1.0f / (1.0f + expf(-t));
    }
  }
  free(W0T);
  free(W1T);
}

/* dx (P,in) and dW0,db0,dW1,db1,dW2,db2 (written).  Per-thread partial weight gradients (double), summed at the end. */
void orc_decoder_bwd(const float* x, const float* h1, const float* h2, const float* y, const float* dy, const float* W0,
                     const float* W1, const float* W2, float* dx, float* dW0, float* db0, float* dW1, float* db1, float* dW2,
                     float* db2, int64_t P, int in_dim, int out_dim) {
  const int nt = orc_num_threads();
  const int n0 = HID * in_dim, n1 = HID * HID, n2 = out_dim * HID;
  const int tot = n0 + HID + n1 + HID + n2 + out_dim;
  double* part = (double*)calloc((size_t)nt * tot, sizeof(double));
#pragma omp parallel
  {
#ifdef _OPENMP
    double* g = part + (size_t)omp_get_thread_num() * tot;
#else
    double* g = part;
#endif
    double *g0 = g, *gb0 = g0 + n0, *g1 = gb0 + HID, *gb1 = g1 + n1, *g2 = gb1 + HID, *gb2 = g2 + n2;
#pragma omp for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
      float dz3[4], d2[HID], d1[HID], s[HID];
      const float* h1p = h1 + p * HID;
      const float* h2p = h2 + p * HID;
      const float* xp = x + p * in_dim;
      for (int c = 0; c < out_dim; ++c) {
        float yy = y[p * out_dim + c];
        dz3[c] = dy[p * out_dim + c] * (yy * (1.f - yy));
        gb2[c] += dz3[c];
        const float dc = dz3[c];
        double* gr = g2 + c * HID;
#pragma omp simd
        for (int k = 0; k < HID; ++k) gr[k] += dc * h2p[k];
      }
      for (int k = 0; k < HID; ++k) s[k] = 0.f;
      for (int c = 0; c < out_dim; ++c) {
        const float dc = dz3[c];
        const float* w = W2 + c * HID;
#pragma omp simd
        for (int k = 0; k < HID; ++k) s[k] += w[k] * dc;
      }
#pragma omp simd
      for (int k = 0; k < HID; ++k) { d2[k] = h2p[k] > 0.f ? s[k] : 0.f; gb1[k] += d2[k]; }
      for (int j = 0; j < HID; ++j) {
        const float dj = d2[j];
        double* gr = g1 + j * HID;
#pragma omp simd
        for (int k = 0; k < HID; ++k) gr[k] += dj * h1p[k];
      }
      for (int k = 0; k < HID; ++k) s[k] = 0.f;
      for (int j = 0; j < HID; ++j) {
        const float dj = d2[j];
        const float* w = W1 + j * HID;
#pragma omp simd
        for (int k = 0; k < HID; ++k) s[k] += w[k] * dj;
      }
#pragma omp simd
      for (int k = 0; k < HID; ++k) { d1[k] = h1p[k] > 0.f ? s[k] : 0.f; gb0[k] += d1[k]; }
      for (int j = 0; j < HID; ++j) {
        const float dj = d1[j];
        double* gr = g0 + j * in_dim;
#pragma omp simd
        for (int k = 0; k < in_dim; ++k) gr[k] += dj * xp[k];
      }
      {
        float t[64];
        for (int k = 0; k < in_dim; ++k) t[k] = 0.f;
        for (int j = 0; j < HID; ++j) {
          const float dj = d1[j];
          const float* w = W0 + j * in_dim;
#pragma omp simd
          for (int k = 0; k < in_dim; ++k) t[k] += w[k] * dj;
        }
        for (int k = 0; k < in_dim; ++k) dx[p * in_dim + k] = t[k];
      }
    }
  }
  float* outs[6] = {dW0, db0, dW1, db1, dW2, db2};
  int sizes[6] = {n0, HID, n1, HID, n2, out_dim};
  int off = 0;
  for (int a = 0; a < 6; ++a) {
    for (int e = 0; e < sizes[a]; ++e) {
      double t = 0.0;
      for (int th = 0; th < nt; ++th) t += part[(size_t)th * tot + off + e];
      outs[a][e] = (float)t;
    }
    off += sizes[a];
  }
  free(part);
}
