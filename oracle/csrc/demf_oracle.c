/*
 * demf_oracle.c — CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Plain-C restatement of the native operators the DeMF fusion hot path executes.
 * The reference tree (haoy945/DeMF) contains no native code: these operators live
 * in its pinned, un-vendored dependencies mmdet3d==0.18.1 (mmdet3d/ops/*) and
 * mmcv-full==1.3.18 (mmcv/ops ms_deform_attn), see requirements.txt:2-4.  Each
 * function below restates the published algorithm of the upstream operator and
 * cites the reference call site that consumes it.
 *
 * PARITY STATUS: the reference ships no tests, golden vectors or fixtures for
 * these operators and the dependencies are absent from this environment, so the
 * operator restatements are "parity unpinned" against upstream binaries; they
 * are pinned instead against independent in-container implementations
 * (brute-force numpy, torch grid_sample, transformers' pure-PyTorch deformable
 * attention) in tests/test_oracle_*.py.  See DESIGN.md section "Oracle".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.
 *
 * Canonical arithmetic (shared bit-for-bit with the HIP kernels; built with
 * -ffp-contract=off so only the explicit fmaf calls fuse):
 *   squared distance   d2 = fmaf(dz, dz, fmaf(dx, dx, dy*dy))
 *   3-tap interpolate  o  = fmaf(w2, f2, fmaf(w0, f0, w1*f1))
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float dist2f(float dx, float dy, float dz) {
  return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
}

/* ---- furthest_point_sample ------------------------------------------------
 * Consumed at demf/modeling/heads/class_agnostic_vote_head.py:429-430 and by
 * every PointSAModule (configs/demf/demf_votenet.py:48-62,155-162).
 * Upstream (mmdet3d/ops/furthest_point_sample): one block of
 * bs = min(1024, 2^floor(log2 N)) threads per scene; thread t owns points
 * t, t+bs, ...; per round temp[k] = min(temp[k], d2(k, last)); the thread keeps
 * its first strict maximum; the block tree-reduction keeps the lower thread id
 * on ties.  Hence the winner is the maximum of temp with ties broken by
 * (k mod bs) first, then k.                                                   */
void oracle_fps(int B, int N, int M, const float* xyz, int* idx) {
  int bs = 1;
  while (bs * 2 <= N && bs < 1024) bs *= 2;
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    const float* p = xyz + (size_t)b * N * 3;
    int* out = idx + (size_t)b * M;
    float* temp = (float*)malloc(sizeof(float) * (size_t)N);
    for (int k = 0; k < N; ++k) temp[k] = 1e10f;
    int old = 0;
    if (M > 0) out[0] = 0;
    for (int j = 1; j < M; ++j) {
      const float x1 = p[3 * old], y1 = p[3 * old + 1], z1 = p[3 * old + 2];
      float best = -1.f;
      int besti = 0, bestt = bs;
      for (int k = 0; k < N; ++k) {
        const float d = dist2f(p[3 * k] - x1, p[3 * k + 1] - y1, p[3 * k + 2] - z1);
        const float t = d < temp[k] ? d : temp[k]; /* min(d, temp) */
        temp[k] = t;
        const int tt = k & (bs - 1);
        if (t > best || (t == best && tt < bestt)) {
          best = t;
          besti = k;
          bestt = tt;
        }
      }
      old = besti;
      out[j] = old;
    }
    free(temp);
  }
}

/* ---- ball_query -----------------------------------------------------------
 * Consumed through QueryAndGroup in the SA modules built at
 * class_agnostic_vote_head.py:383 (radii: demf_votenet.py:51-53,158).
 * Upstream: per centre scan k = 0..N-1; accept if d2 == 0 or
 * min_r^2 <= d2 < max_r^2; the first hit pre-fills all nsample slots; stop after
 * nsample hits; idx is zero-initialised.                                       */
void oracle_ball_query(int B, int N, int M, float min_radius, float max_radius, int ns,
                       const float* center, const float* xyz, int* idx) {
  const float min_r2 = min_radius * min_radius, max_r2 = max_radius * max_radius;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b) {
    for (int m = 0; m < M; ++m) {
      const float* p = xyz + (size_t)b * N * 3;
      const float* c = center + ((size_t)b * M + m) * 3;
      int* out = idx + ((size_t)b * M + m) * ns;
      for (int l = 0; l < ns; ++l) out[l] = 0;
      int cnt = 0;
      for (int k = 0; k < N && cnt < ns; ++k) {
        const float d2 = dist2f(c[0] - p[3 * k], c[1] - p[3 * k + 1], c[2] - p[3 * k + 2]);
        if (d2 == 0.f || (d2 >= min_r2 && d2 < max_r2)) {
          if (cnt == 0)
            for (int l = 0; l < ns; ++l) out[l] = k;
          out[cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* ---- grouping_operation / gather_points (channel-major) -------------------- */
void oracle_group_points_fwd(int B, int C, int N, int J, const float* feat, const int* idx,
                             float* out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float* f = feat + ((size_t)b * C + c) * N;
      float* o = out + ((size_t)b * C + c) * J;
      const int* ii = idx + (size_t)b * J;
      for (int j = 0; j < J; ++j) o[j] = f[ii[j]];
    }
}

void oracle_group_points_bwd(int B, int C, int N, int J, const float* gout, const int* idx,
                             float* gfeat /* zeroed */) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      float* f = gfeat + ((size_t)b * C + c) * N;
      const float* g = gout + ((size_t)b * C + c) * J;
      const int* ii = idx + (size_t)b * J;
      for (int j = 0; j < J; ++j) f[ii[j]] += g[j];
    }
}

/* ---- three_nn ---------------------------------------------------------------
 * PointFPModule of the backbone (demf_votenet.py:56).  Upstream: strict '<'
 * insertion into (best1,best2,best3) initialised to 1e40f (= +inf in fp32);
 * returns SQUARED distances (the Python wrapper takes the sqrt).               */
void oracle_three_nn(int B, int n, int m, const float* target, const float* source,
                     float* dist2, int* idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < n; ++t) {
      const float* u = target + ((size_t)b * n + t) * 3;
      const float* s = source + (size_t)b * m * 3;
      float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
      int i1 = 0, i2 = 0, i3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = dist2f(u[0] - s[3 * k], u[1] - s[3 * k + 1], u[2] - s[3 * k + 2]);
        if (d < b1) {
          b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
        } else if (d < b2) {
          b3 = b2; i3 = i2; b2 = d; i2 = k;
        } else if (d < b3) {
          b3 = d; i3 = k;
        }
      }
      float* dd = dist2 + ((size_t)b * n + t) * 3;
      int* ii = idx + ((size_t)b * n + t) * 3;
      dd[0] = b1; dd[1] = b2; dd[2] = b3;
      ii[0] = i1; ii[1] = i2; ii[2] = i3;
    }
}

/* ---- three_interpolate (channel-major) -------------------------------------- */
void oracle_three_interpolate_fwd(int B, int C, int m, int n, const float* feat,
                                  const int* idx, const float* w, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float* f = feat + ((size_t)b * C + c) * m;
      float* o = out + ((size_t)b * C + c) * n;
      for (int t = 0; t < n; ++t) {
        const int* ii = idx + ((size_t)b * n + t) * 3;
        const float* ww = w + ((size_t)b * n + t) * 3;
        o[t] = fmaf(ww[2], f[ii[2]], fmaf(ww[0], f[ii[0]], ww[1] * f[ii[1]]));
      }
    }
}

void oracle_three_interpolate_bwd(int B, int C, int n, int m, const float* gout,
                                  const int* idx, const float* w, float* gfeat /* zeroed */) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      float* f = gfeat + ((size_t)b * C + c) * m;
      const float* g = gout + ((size_t)b * C + c) * n;
      for (int t = 0; t < n; ++t) {
        const int* ii = idx + ((size_t)b * n + t) * 3;
        const float* ww = w + ((size_t)b * n + t) * 3;
        f[ii[0]] += g[t] * ww[0];
        f[ii[1]] += g[t] * ww[1];
        f[ii[2]] += g[t] * ww[2];
      }
    }
}

/* ---- multi-scale deformable attention ---------------------------------------
 * Consumed at demf/modeling/layers/transformer.py:73 via mmcv
 * MultiScaleDeformableAttention (config demf_votenet.py:79-85).  Restates mmcv's
 * ms_deformable_im2col / col2im: h_im = loc_y*H_l - 0.5, w_im = loc_x*W_l - 0.5;
 * sample only if -1 < h_im < H_l and -1 < w_im < W_l; bilinear over the 4
 * neighbours with out-of-image neighbours contributing zero.
 * REAL is float for the product-parity oracle and double for the fp64 variant
 * used by gradient checks.                                                     */
#define MSDA_IMPL(NAME, REAL)                                                               \
  void NAME##_fwd(int B, int S, int H, int Dh, int L, int Q, int P, const REAL* value,      \
                  const int64_t* shapes, const int64_t* lsi, const REAL* loc,               \
                  const REAL* attw, REAL* out) {                                            \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                \
    for (int b = 0; b < B; ++b)                                                             \
      for (int q = 0; q < Q; ++q)                                                           \
        for (int h = 0; h < H; ++h) {                                                       \
          const size_t item = ((size_t)b * Q + q) * H + h;                                  \
          REAL* o = out + item * Dh;                                                        \
          for (int c = 0; c < Dh; ++c) o[c] = 0;                                            \
          for (int l = 0; l < L; ++l) {                                                     \
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];                 \
            const REAL* vl = value + ((size_t)b * S + lsi[l]) * H * Dh + (size_t)h * Dh;    \
            for (int p = 0; p < P; ++p) {                                                   \
              const size_t si = (item * L + l) * P + p;                                     \
              const REAL lx = loc[2 * si], ly = loc[2 * si + 1], aw = attw[si];             \
              const REAL h_im = ly * Hl - (REAL)0.5, w_im = lx * Wl - (REAL)0.5;            \
              if (!(h_im > -1 && w_im > -1 && h_im < Hl && w_im < Wl)) continue;            \
              const int h_low = (int)floor((double)h_im), w_low = (int)floor((double)w_im); \
              const int h_high = h_low + 1, w_high = w_low + 1;                             \
              const REAL lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;    \
              const size_t hs = (size_t)H * Dh;                                             \
              for (int c = 0; c < Dh; ++c) {                                                \
                REAL v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                        \
                if (h_low >= 0 && w_low >= 0) v1 = vl[((size_t)h_low * Wl + w_low) * hs + c];          \
                if (h_low >= 0 && w_high <= Wl - 1) v2 = vl[((size_t)h_low * Wl + w_high) * hs + c];   \
                if (h_high <= Hl - 1 && w_low >= 0) v3 = vl[((size_t)h_high * Wl + w_low) * hs + c];   \
                if (h_high <= Hl - 1 && w_high <= Wl - 1)                                   \
                  v4 = vl[((size_t)h_high * Wl + w_high) * hs + c];                         \
                const REAL val = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4; \
                o[c] += aw * val;                                                           \
              }                                                                             \
            }                                                                               \
          }                                                                                 \
        }                                                                                   \
  }                                                                                         \
  /* gvalue zeroed by the caller; batches are independent so parallel over b only */       \
  void NAME##_bwd(int B, int S, int H, int Dh, int L, int Q, int P, const REAL* value,      \
                  const int64_t* shapes, const int64_t* lsi, const REAL* loc,               \
                  const REAL* attw, const REAL* gout, REAL* gvalue, REAL* gloc,             \
                  REAL* gattw) {                                                            \
    _Pragma("omp parallel for schedule(static)")                                            \
    for (int b = 0; b < B; ++b)                                                             \
      for (int q = 0; q < Q; ++q)                                                           \
        for (int h = 0; h < H; ++h) {                                                       \
          const size_t item = ((size_t)b * Q + q) * H + h;                                  \
          const REAL* go = gout + item * Dh;                                                \
          for (int l = 0; l < L; ++l) {                                                     \
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];                 \
            const size_t base = ((size_t)b * S + lsi[l]) * H * Dh + (size_t)h * Dh;         \
            const REAL* vl = value + base;                                                  \
            REAL* gvl = gvalue + base;                                                      \
            for (int p = 0; p < P; ++p) {                                                   \
              const size_t si = (item * L + l) * P + p;                                     \
              const REAL lx = loc[2 * si], ly = loc[2 * si + 1], aw = attw[si];             \
              const REAL h_im = ly * Hl - (REAL)0.5, w_im = lx * Wl - (REAL)0.5;            \
              gloc[2 * si] = 0; gloc[2 * si + 1] = 0; gattw[si] = 0;                        \
              if (!(h_im > -1 && w_im > -1 && h_im < Hl && w_im < Wl)) continue;            \
              const int h_low = (int)floor((double)h_im), w_low = (int)floor((double)w_im); \
              const int h_high = h_low + 1, w_high = w_low + 1;                             \
              const REAL lh = h_im - h_low, lw = w_im - w_low, hh = 1 - lh, hw = 1 - lw;    \
              const size_t hs = (size_t)H * Dh;                                             \
              REAL g_w = 0, g_x = 0, g_y = 0;                                               \
              for (int c = 0; c < Dh; ++c) {                                                \
                const REAL top = go[c], tgv = top * aw;                                     \
                REAL ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                      \
                if (h_low >= 0 && w_low >= 0) {                                             \
                  const size_t o = ((size_t)h_low * Wl + w_low) * hs + c;                   \
                  v1 = vl[o]; ghw -= hw * v1; gww -= hh * v1; gvl[o] += hh * hw * tgv;      \
                }                                                                           \
                if (h_low >= 0 && w_high <= Wl - 1) {                                       \
                  const size_t o = ((size_t)h_low * Wl + w_high) * hs + c;                  \
                  v2 = vl[o]; ghw -= lw * v2; gww += hh * v2; gvl[o] += hh * lw * tgv;      \
                }                                                                           \
                if (h_high <= Hl - 1 && w_low >= 0) {                                       \
                  const size_t o = ((size_t)h_high * Wl + w_low) * hs + c;                  \
                  v3 = vl[o]; ghw += hw * v3; gww -= lh * v3; gvl[o] += lh * hw * tgv;      \
                }                                                                           \
                if (h_high <= Hl - 1 && w_high <= Wl - 1) {                                 \
                  const size_t o = ((size_t)h_high * Wl + w_high) * hs + c;                 \
                  v4 = vl[o]; ghw += lw * v4; gww += lh * v4; gvl[o] += lh * lw * tgv;      \
                }                                                                           \
                const REAL val = hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4; \
                g_w += top * val;                                                           \
                g_x += Wl * gww * tgv;                                                      \
                g_y += Hl * ghw * tgv;                                                      \
              }                                                                             \
              gattw[si] = g_w; gloc[2 * si] = g_x; gloc[2 * si + 1] = g_y;                  \
            }                                                                               \
          }                                                                                 \
        }                                                                                   \
  }

MSDA_IMPL(oracle_msda_f32, float)
MSDA_IMPL(oracle_msda_f64, double)

int oracle_version(void) { return 1; }
