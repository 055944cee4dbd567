/*
 * gh_oracle.c — CPU ORACLE ("Oracle B") for the Gaussian-splatting hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under guassianhand_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" by the reference. The algorithm restated here lives in the
 * un-vendored third-party dependency `diff-gaussian-rasterization` (pinned only as `==0.0.0` at
 * /root/reference/environment.yml:129; imported at tgs/models/renderer_one_shot.py:3). Its source is
 * not under /root/reference, the reference ships no tests or golden vectors, and the package is
 * CUDA-only, so nothing of the reference can be run here for this path. This file therefore restates
 * the published tile-based 3DGS rasterisation algorithm (Kerbl et al., SIGGRAPH 2023) as specified in
 * SURVEY.md Appendix A, and is pinned by
 *   (a) an independent dense PyTorch-autograd restatement (oracle/oracle_torch.py, "Oracle A"),
 *   (b) closed-form known-answer tests (tests/test_oracle_known_answers.py),
 *   (c) float64 finite differences of Oracle A,
 * while the host-side half of the path (camera matrices, attribute blend, call protocol) IS pinned
 * against the reference's own Python (tests/golden/make_host_fixtures.py).
 * Call sites this restates: tgs/models/renderer_one_shot.py:281-296, :338-346, :355-379 and the
 * attribute blend at :298-334.
 *
 * ARITHMETIC CONTRACT (shared in prose with the HIP kernels): fp32 throughout,
 * no implicit contraction (-ffp-contract=off), FMAs only where written as fmaf(), IEEE division and
 * sqrt, and a software exp() (gho_exp) built from IEEE primitives so that every discrete decision
 * (cull, alpha < 1/255, T < 1e-4, power > 0) is taken on bit-identical values on CPU and GPU.
 *
 * Build: see oracle/Makefile  (gcc -O2 -mfma -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gh_raster.h"

#define TILE GH_TILE

/* Optional host arrays the oracle fills for stage-by-stage comparison (any may be NULL). */
typedef struct GhoDebug {
  float* xy;           /* (n_views,P,2) */
  float* depth;        /* (n_views,P) */
  float* conic_opacity;/* (n_views,P,4)  A,B,C,opacity */
  float* rgb;          /* (n_views,P,3) */
  uint32_t* rect;      /* (n_views,P)  packed like GhLayout.rect */
  uint32_t* offsets;   /* (n_views,P)  inclusive scan of tiles touched */
  uint64_t* sorted_keys;/* (D) */
  uint32_t* sorted_gid; /* (D) view*P+i */
  uint32_t* ranges;    /* (n_views*tiles,2) */
  float* final_T;      /* (n_views,H,W) */
  uint32_t* n_contrib; /* (n_views,H,W) */
  int64_t capacity;    /* capacity of sorted_* arrays */
  int64_t num_rendered;/* out */
} GhoDebug;

/* ------------------------------------------------------------------------------------------- */
/* exp(x) for x <= 0 from IEEE primitives only: 2^(x*log2e), n = rne(t), Taylor-6 on [-.5,.5].  */
static inline float gho_exp(float x) {
  float t = x * 1.44269504088896341f;
  if (t < -126.0f) return 0.0f;
  float n = nearbyintf(t);
  float f = t - n;
  float p = 1.5403530393381608e-04f;
  p = fmaf(p, f, 1.3333558146428443e-03f);
  p = fmaf(p, f, 9.6181291076284772e-03f);
  p = fmaf(p, f, 5.5504108664821580e-02f);
  p = fmaf(p, f, 2.4022650695910072e-01f);
  p = fmaf(p, f, 6.9314718055994531e-01f);
  p = fmaf(p, f, 1.0f);
  return ldexpf(p, (int)n);
}

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* Per (view, Gaussian) intermediate state kept between forward and backward. */
typedef struct GView {
  float px, py, depth;
  float A, B, C, opac;  /* conic + blended opacity */
  float rgb[3];
  int radius;
  int minx, miny, maxx, maxy;
  uint32_t tiles;
  uint8_t clamped;      /* SH clamp flags */
} GView;

typedef struct Inst { uint64_t key; uint32_t slot; uint32_t gid; } Inst;

static int inst_cmp(const void* a, const void* b) {
  const Inst* x = (const Inst*)a; const Inst* y = (const Inst*)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  if (x->slot < y->slot) return -1;   /* emit order == stable tie-break */
  if (x->slot > y->slot) return 1;
  return 0;
}

/* ---- baseline mode (bench.py's cpu_baseline leg only) -------------------------------------------------------------
 * gho_set_parallel(1): the stages that the checker runs on one thread (instance emit + sort, the A.5 chain rule, the final
 * write-out) run under OpenMP too — per-tile buckets + per-tile stable sort instead of one global qsort (same (key, slot)
 * total order, so the lists are identical), the chain rule over Gaussians in parallel with the views of a Gaussian summed in
 * the checker's order (per-Gaussian results bit-identical; the global (48,) color_w sum is reduced from per-thread double
 * partials). Default 0: the serial checker path every parity test uses.
 * gho_timing(): wall seconds inside gho_forward / gho_backward since the last reset, and the part of it spent in sections
 * that ran on one thread whatever the mode (what bounds the all-core number: bench.py reports serial / total). */
static int g_parallel = 0;
static double g_t_total = 0.0, g_t_serial = 0.0;
#ifdef _OPENMP
extern double omp_get_wtime(void);
static inline double gho_now(void) { return omp_get_wtime(); }
#else
#include <time.h>
static inline double gho_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#endif
void gho_set_parallel(int on) { g_parallel = on ? 1 : 0; }
int gho_get_parallel(void) { return g_parallel; }
void gho_timing_reset(void) { g_t_total = 0.0; g_t_serial = 0.0; }
void gho_timing(double* total, double* serial) { if (total) *total = g_t_total; if (serial) *serial = g_t_serial; }

/* Persistent oracle context (host memory) so backward can reuse forward state. */
typedef struct GhoCtx {
  GhDims dims;
  GView* g;          /* n_views*P */
  uint32_t* offsets; /* n_views*P inclusive */
  Inst* inst;        /* D sorted */
  int64_t D;
  uint32_t* ranges;  /* n_views*tiles*2 */
  float* final_T;
  uint32_t* n_contrib;
} GhoCtx;

void gho_free(GhoCtx* c) {
  if (!c) return;
  free(c->g); free(c->offsets); free(c->inst); free(c->ranges); free(c->final_T); free(c->n_contrib);
  free(c);
}

/* blended SH coefficient k, channel ch of Gaussian i (renderer_one_shot.py:330-334) */
static inline float blended_sh(const GhDims* d, const GhInputs* in, int i, int k, int ch) {
  float s = in->shs[((size_t)i * d->M + k) * 3 + ch];
  if (in->blend_color_w) {
    const float* w = in->blend_color_w + ((d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? (size_t)i * 48 : 0);
    s = s * w[k * 3 + ch];
    if (in->blend_color_b) {
      s = s * w[k * 3 + ch];
      s = s + in->blend_color_b[(size_t)i * 48 + k * 3 + ch];
    }
  }
  return s;
}

/* SH basis values for direction (x,y,z), up to degree deg; returns number of coefficients. */
static int sh_basis(int deg, float x, float y, float z, float* Bv) {
  Bv[0] = SH_C0;
  if (deg < 1) return 1;
  Bv[1] = -SH_C1 * y; Bv[2] = SH_C1 * z; Bv[3] = -SH_C1 * x;
  if (deg < 2) return 4;
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  Bv[4] = SH_C2[0] * xy;
  Bv[5] = SH_C2[1] * yz;
  Bv[6] = SH_C2[2] * (2.0f * zz - xx - yy);
  Bv[7] = SH_C2[3] * xz;
  Bv[8] = SH_C2[4] * (xx - yy);
  if (deg < 3) return 9;
  Bv[9]  = SH_C3[0] * y * (3.0f * xx - yy);
  Bv[10] = SH_C3[1] * xy * z;
  Bv[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
  Bv[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
  Bv[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
  Bv[14] = SH_C3[5] * z * (xx - yy);
  Bv[15] = SH_C3[6] * x * (xx - 3.0f * yy);
  return 16;
}

/* d(basis_k)/d(x,y,z) */
static void sh_basis_grad(int deg, float x, float y, float z, float (*dB)[3]) {
  for (int k = 0; k < 16; ++k) dB[k][0] = dB[k][1] = dB[k][2] = 0.0f;
  if (deg < 1) return;
  dB[1][1] = -SH_C1; dB[2][2] = SH_C1; dB[3][0] = -SH_C1;
  if (deg < 2) return;
  float xx = x * x, yy = y * y, zz = z * z;
  dB[4][0] = SH_C2[0] * y;  dB[4][1] = SH_C2[0] * x;
  dB[5][1] = SH_C2[1] * z;  dB[5][2] = SH_C2[1] * y;
  dB[6][0] = SH_C2[2] * -2.0f * x; dB[6][1] = SH_C2[2] * -2.0f * y; dB[6][2] = SH_C2[2] * 4.0f * z;
  dB[7][0] = SH_C2[3] * z;  dB[7][2] = SH_C2[3] * x;
  dB[8][0] = SH_C2[4] * 2.0f * x; dB[8][1] = SH_C2[4] * -2.0f * y;
  if (deg < 3) return;
  dB[9][0]  = SH_C3[0] * 6.0f * x * y;           dB[9][1]  = SH_C3[0] * (3.0f * xx - 3.0f * yy);
  dB[10][0] = SH_C3[1] * y * z; dB[10][1] = SH_C3[1] * x * z; dB[10][2] = SH_C3[1] * x * y;
  dB[11][0] = SH_C3[2] * -2.0f * x * y; dB[11][1] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy); dB[11][2] = SH_C3[2] * 8.0f * y * z;
  dB[12][0] = SH_C3[3] * -6.0f * x * z; dB[12][1] = SH_C3[3] * -6.0f * y * z; dB[12][2] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
  dB[13][0] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); dB[13][1] = SH_C3[4] * -2.0f * x * y; dB[13][2] = SH_C3[4] * 8.0f * x * z;
  dB[14][0] = SH_C3[5] * 2.0f * x * z; dB[14][1] = SH_C3[5] * -2.0f * y * z; dB[14][2] = SH_C3[5] * (xx - yy);
  dB[15][0] = SH_C3[6] * (3.0f * xx - 3.0f * yy); dB[15][1] = SH_C3[6] * -6.0f * x * y;
}

/* Shared per-Gaussian geometry (forward values needed again by the backward chain rule). */
typedef struct Geo {
  float mx, my, mz;            /* blended mean */
  float tx, ty, tz;            /* view space */
  float hx, hy, hw, winv;
  float S[6];                  /* Sigma3D: 00 01 02 11 12 22 */
  float R[9], s[3];            /* rotation, modulated scales */
  float T[6];                  /* J*W, 2x3 */
  float cx, cy;                /* clamped tx, ty */
  int xclamped, yclamped;
  float fx, fy;
  float a, b, c, det;          /* dilated cov2D */
} Geo;

static void geo_forward(const GhDims* d, const GhInputs* in, const float* cam, int i, Geo* o) {
  const float* V = cam; const float* PM = cam + 16;
  float mx = in->means3D[3 * i], my = in->means3D[3 * i + 1], mz = in->means3D[3 * i + 2];
  if (in->blend_xyz_b) { mx = mx + in->blend_xyz_b[0]; my = my + in->blend_xyz_b[1]; mz = mz + in->blend_xyz_b[2]; }
  o->mx = mx; o->my = my; o->mz = mz;
  o->tx = fmaf(V[0], mx, fmaf(V[4], my, fmaf(V[8], mz, V[12])));
  o->ty = fmaf(V[1], mx, fmaf(V[5], my, fmaf(V[9], mz, V[13])));
  o->tz = fmaf(V[2], mx, fmaf(V[6], my, fmaf(V[10], mz, V[14])));
  o->hx = fmaf(PM[0], mx, fmaf(PM[4], my, fmaf(PM[8], mz, PM[12])));
  o->hy = fmaf(PM[1], mx, fmaf(PM[5], my, fmaf(PM[9], mz, PM[13])));
  o->hw = fmaf(PM[3], mx, fmaf(PM[7], my, fmaf(PM[11], mz, PM[15])));
  o->winv = 1.0f / (o->hw + 1e-7f);
  /* Sigma3D = R S S^T R^T — or given (the published module's cov3D_precomp: used as it stands, scale_modifier not applied) */
  if (in->cov3D_precomp) {
    for (int k = 0; k < 6; ++k) o->S[k] = in->cov3D_precomp[6 * i + k];
    for (int k = 0; k < 9; ++k) o->R[k] = 0.0f;
    o->s[0] = o->s[1] = o->s[2] = 0.0f;
  } else {
  float mod = d->scale_modifier;
  o->s[0] = mod * in->scales[3 * i]; o->s[1] = mod * in->scales[3 * i + 1]; o->s[2] = mod * in->scales[3 * i + 2];
  float r = in->rotations[4 * i], x = in->rotations[4 * i + 1], y = in->rotations[4 * i + 2], z = in->rotations[4 * i + 3];
  float* R = o->R;
  R[0] = 1.0f - 2.0f * fmaf(y, y, z * z); R[1] = 2.0f * fmaf(x, y, -(r * z)); R[2] = 2.0f * fmaf(x, z, r * y);
  R[3] = 2.0f * fmaf(x, y, r * z); R[4] = 1.0f - 2.0f * fmaf(x, x, z * z); R[5] = 2.0f * fmaf(y, z, -(r * x));
  R[6] = 2.0f * fmaf(x, z, -(r * y)); R[7] = 2.0f * fmaf(y, z, r * x); R[8] = 1.0f - 2.0f * fmaf(x, x, y * y);
  float M[9];
  for (int a = 0; a < 3; ++a) for (int j = 0; j < 3; ++j) M[3 * a + j] = R[3 * a + j] * o->s[j];
  o->S[0] = fmaf(M[0], M[0], fmaf(M[1], M[1], M[2] * M[2]));
  o->S[1] = fmaf(M[0], M[3], fmaf(M[1], M[4], M[2] * M[5]));
  o->S[2] = fmaf(M[0], M[6], fmaf(M[1], M[7], M[2] * M[8]));
  o->S[3] = fmaf(M[3], M[3], fmaf(M[4], M[4], M[5] * M[5]));
  o->S[4] = fmaf(M[3], M[6], fmaf(M[4], M[7], M[5] * M[8]));
  o->S[5] = fmaf(M[6], M[6], fmaf(M[7], M[7], M[8] * M[8]));
  }
  /* EWA projection */
  float tanx = cam[35], tany = cam[36];
  float limx = 1.3f * tanx, limy = 1.3f * tany;
  float txtz = o->tx / o->tz, tytz = o->ty / o->tz;
  float cxr = fminf(limx, fmaxf(-limx, txtz)), cyr = fminf(limy, fmaxf(-limy, tytz));
  o->xclamped = (cxr != txtz); o->yclamped = (cyr != tytz);
  o->cx = cxr * o->tz; o->cy = cyr * o->tz;
  o->fx = (float)d->W / (2.0f * tanx); o->fy = (float)d->H / (2.0f * tany);
  float tz2 = o->tz * o->tz;
  float J00 = o->fx / o->tz, J02 = -(o->fx * o->cx) / tz2, J11 = o->fy / o->tz, J12 = -(o->fy * o->cy) / tz2;
  float* T = o->T;
  T[0] = fmaf(J00, V[0], J02 * V[2]); T[1] = fmaf(J00, V[4], J02 * V[6]); T[2] = fmaf(J00, V[8], J02 * V[10]);
  T[3] = fmaf(J11, V[1], J12 * V[2]); T[4] = fmaf(J11, V[5], J12 * V[6]); T[5] = fmaf(J11, V[9], J12 * V[10]);
  const float* S = o->S;
  float U0 = fmaf(T[0], S[0], fmaf(T[1], S[1], T[2] * S[2]));
  float U1 = fmaf(T[0], S[1], fmaf(T[1], S[3], T[2] * S[4]));
  float U2 = fmaf(T[0], S[2], fmaf(T[1], S[4], T[2] * S[5]));
  float U3 = fmaf(T[3], S[0], fmaf(T[4], S[1], T[5] * S[2]));
  float U4 = fmaf(T[3], S[1], fmaf(T[4], S[3], T[5] * S[4]));
  float U5 = fmaf(T[3], S[2], fmaf(T[4], S[4], T[5] * S[5]));
  float c00 = fmaf(U0, T[0], fmaf(U1, T[1], U2 * T[2]));
  float c01 = fmaf(U0, T[3], fmaf(U1, T[4], U2 * T[5]));
  float c11 = fmaf(U3, T[3], fmaf(U4, T[4], U5 * T[5]));
  o->a = c00 + 0.3f; o->b = c01; o->c = c11 + 0.3f;
  o->det = fmaf(o->a, o->c, -(o->b * o->b));
}

/* float -> int as the GPU converts (v_cvt_i32_f32): truncation, NaN -> 0, saturation at the int range */
static inline int f2i(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int)x;
}

/* ------------------------------------------------------------------------------------------- */
int gho_forward(const GhDims* d, const GhInputs* in, const GhOutputs* out, GhoCtx** ctx_out, GhoDebug* dbg) {
  if (!d || !in || !out || !ctx_out) return GH_ERR_INVALID_ARG;
  if (d->abi != GH_ABI_TAG) return GH_ERR_ABI;           /* (built from another version of the header than its caller) */
  if (d->P != 0 && (in->shs != NULL) == (in->colors_precomp != NULL)) return GH_ERR_INVALID_ARG;   /* (P = 0: nothing is read, the background only) */
  /* exactly one of {scales AND rotations, cov3D_precomp} (the published wrapper's second validation, App. A.0) */
  if (d->P != 0 && (((in->scales != NULL) != (in->rotations != NULL)) || ((in->scales != NULL) == (in->cov3D_precomp != NULL)))) return GH_ERR_INVALID_ARG;
  const int P = d->P, NV = d->n_views, H = d->H, W = d->W;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, tiles = gx * gy;
  if (gx > 255 || gy > 255 || d->sh_degree > 3) return GH_ERR_UNSUPPORTED;
  /* layout variants of the product (compact colour biases, pose batch) are checked against this oracle through their
   * reference-layout equivalents (tests/test_gpu_dropin.py); the oracle itself only speaks the reference layouts */
  if (d->flags & ~GH_FLAG_BLEND_W_PER_GAUSSIAN) return GH_ERR_UNSUPPORTED;
  const double t_begin = gho_now();
  double t_ser = 0.0, t_mark;
  GhoCtx* c = (GhoCtx*)calloc(1, sizeof(GhoCtx));
  c->dims = *d;
  c->g = (GView*)calloc((size_t)NV * P + 1, sizeof(GView));
  c->offsets = (uint32_t*)calloc((size_t)NV * P + 1, sizeof(uint32_t));
  c->ranges = (uint32_t*)calloc((size_t)NV * tiles * 2 + 2, sizeof(uint32_t));
  c->final_T = (float*)calloc((size_t)NV * H * W + 1, sizeof(float));
  c->n_contrib = (uint32_t*)calloc((size_t)NV * H * W + 1, sizeof(uint32_t));

  /* A.1 preprocess */
#pragma omp parallel for schedule(static)
  for (long n = 0; n < (long)NV * P; ++n) {
    int v = (int)(n / P), i = (int)(n % P);
    const float* cam = in->cams + (size_t)v * GH_CAM_FLOATS;
    GView* g = &c->g[n];
    memset(g, 0, sizeof(*g));
    Geo e; geo_forward(d, in, cam, i, &e);
    /* Non-finite inputs: the published code is undefined there (a NaN radius cast to int; a radius of INT_MIN makes BOTH rect
     * extents negative and their product positive). The checker stays defined by doing what the GPU's conversions do — NaN -> 0,
     * saturation (f2i below) — and by culling a NaN depth like the kernels' `tz > 0.2` test; for finite inputs nothing changes. */
    if (!(e.tz > 0.2f)) continue;
    if (e.det == 0.0f) continue;
    float dinv = 1.0f / e.det;
    float mid = 0.5f * (e.a + e.c);
    float sq = sqrtf(fmaxf(0.1f, fmaf(mid, mid, -e.det)));
    float lam1 = mid + sq, lam2 = mid - sq;
    int radius = f2i(ceilf(3.0f * sqrtf(fmaxf(lam1, lam2))));
    float ndcx = e.hx * e.winv, ndcy = e.hy * e.winv;
    float px = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
    float py = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
    int minx = f2i((px - (float)radius) / (float)TILE); minx = minx < 0 ? 0 : (minx > gx ? gx : minx);
    int miny = f2i((py - (float)radius) / (float)TILE); miny = miny < 0 ? 0 : (miny > gy ? gy : miny);
    int maxx = f2i((px + (float)radius + (float)(TILE - 1)) / (float)TILE); maxx = maxx < 0 ? 0 : (maxx > gx ? gx : maxx);
    int maxy = f2i((py + (float)radius + (float)(TILE - 1)) / (float)TILE); maxy = maxy < 0 ? 0 : (maxy > gy ? gy : maxy);
    if ((maxx - minx) * (maxy - miny) <= 0) continue;
    g->radius = radius; g->minx = minx; g->miny = miny; g->maxx = maxx; g->maxy = maxy;
    g->tiles = (uint32_t)((maxx - minx) * (maxy - miny));
    g->px = px; g->py = py; g->depth = e.tz;
    g->A = e.c * dinv; g->B = -e.b * dinv; g->C = e.a * dinv;
    float op = in->opacities[i];
    if (in->blend_opacity_b) op = op + in->blend_opacity_b[i];
    g->opac = op;
    if (in->colors_precomp) {
      for (int ch = 0; ch < 3; ++ch) {
        float col = in->colors_precomp[3 * i + ch];
        if (in->blend_color_w) {
          const float* w = in->blend_color_w + ((d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? (size_t)i * 48 : 0);
          col = col * w[ch]; col = col + w[3 + ch]; col = col - 1.0f;
        }
        if (in->blend_color_b) col = col + in->blend_color_b[(size_t)i * 48 + ch];
        g->rgb[ch] = col;
      }
    } else {
      float dx = e.mx - cam[32], dy = e.my - cam[33], dz = e.mz - cam[34];
      float len = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
      dx = dx / len; dy = dy / len; dz = dz / len;
      float Bv[16]; int nb = sh_basis(d->sh_degree, dx, dy, dz, Bv);
      if (nb > d->M) nb = d->M;
      for (int ch = 0; ch < 3; ++ch) {
        float acc = 0.0f;
        for (int k = 0; k < nb; ++k) acc = fmaf(Bv[k], blended_sh(d, in, i, k, ch), acc);
        acc = acc + 0.5f;
        if (acc < 0.0f) { g->clamped |= (uint8_t)(1u << ch); acc = 0.0f; }
        g->rgb[ch] = acc;
      }
    }
  }
  /* radii */
  t_mark = gho_now();
  if (out->radii) for (long n = 0; n < (long)NV * P; ++n) out->radii[n] = c->g[n].radius;

  /* A.2 binning: inclusive scan, emit, stable sort, ranges */
  uint64_t run = 0;
  for (long n = 0; n < (long)NV * P; ++n) { run += c->g[n].tiles; c->offsets[n] = (uint32_t)run; }
  c->D = (int64_t)run;
  c->inst = (Inst*)malloc(sizeof(Inst) * (size_t)(run + 1));
  t_ser += gho_now() - t_mark;
  if (g_parallel) {
    /* baseline mode: bucket the instances by tile (count -> prefix -> scatter), then sort every tile's bucket by the
     * checker's own comparator (key, emit slot): a total order, so the result equals the global stable sort below */
    const long NT = (long)NV * tiles;
    const long NG = (long)NV * P;
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    const int nth = omp_get_max_threads();
#else
    const int nth = 1;
#endif
    /* thread-private tile counters (no atomics: 128 threads hammering 672 counters is slower than one thread): thread k owns
     * Gaussians [NG k / nth, NG (k+1) / nth), counts their instances per tile, then writes them at its own offsets */
    uint32_t* cnt = (uint32_t*)calloc((size_t)NT + 1, sizeof(uint32_t));
    uint32_t* cur = (uint32_t*)malloc(((size_t)NT + 1) * sizeof(uint32_t));
    uint32_t* lc = (uint32_t*)calloc((size_t)nth * (size_t)NT + 1, sizeof(uint32_t));
    /* nth LOGICAL slices, one per loop iteration: every slice is processed whatever team the runtime delivers (a smaller
     * team under OMP_THREAD_LIMIT / OMP_DYNAMIC / a cgroup quota just takes several slices per thread) — ADVICE r4 */
#pragma omp parallel for schedule(static, 1)
    for (int k = 0; k < nth; ++k) {
      uint32_t* mine = lc + (size_t)k * NT;
      for (long n = NG * k / nth; n < NG * (k + 1) / nth; ++n) {
        const GView* g = &c->g[n];
        if (!g->tiles) continue;
        int v = (int)(n / P);
        for (int ty = g->miny; ty < g->maxy; ++ty)
          for (int tx = g->minx; tx < g->maxx; ++tx) mine[(size_t)v * tiles + (size_t)ty * gx + tx]++;
      }
    }
    t_mark = gho_now();
    uint32_t acc = 0;
    for (long t = 0; t < NT; ++t) {
      cur[t] = acc;
      for (int k = 0; k < nth; ++k) { uint32_t x = lc[(size_t)k * NT + t]; lc[(size_t)k * NT + t] = acc; acc += x; cnt[t] += x; }
    }
    t_ser += gho_now() - t_mark;
#pragma omp parallel for schedule(static, 1)
    for (int k = 0; k < nth; ++k) {
      uint32_t* mine = lc + (size_t)k * NT;
      for (long n = NG * k / nth; n < NG * (k + 1) / nth; ++n) {
        const GView* g = &c->g[n];
        if (!g->tiles) continue;
        int v = (int)(n / P);
        uint32_t off = c->offsets[n] - g->tiles;
        uint32_t dbits; memcpy(&dbits, &g->depth, 4);
        for (int ty = g->miny; ty < g->maxy; ++ty)
          for (int tx = g->minx; tx < g->maxx; ++tx) {
            uint64_t tile = (uint64_t)v * tiles + (uint64_t)ty * gx + tx;
            uint32_t pos = mine[tile]++;
            c->inst[pos].key = (tile << 32) | dbits; c->inst[pos].slot = off; c->inst[pos].gid = (uint32_t)n;
            ++off;
          }
      }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (long t = 0; t < NT; ++t) {
      if (!cnt[t]) continue;
      uint32_t s0 = cur[t];
      qsort(c->inst + s0, (size_t)cnt[t], sizeof(Inst), inst_cmp);
      c->ranges[2 * t] = s0; c->ranges[2 * t + 1] = s0 + cnt[t];
    }
    free(lc);
    free(cnt); free(cur);
  } else {
  t_mark = gho_now();
  for (long n = 0; n < (long)NV * P; ++n) {
    const GView* g = &c->g[n];
    if (!g->tiles) continue;
    int v = (int)(n / P);
    uint32_t off = c->offsets[n] - g->tiles;
    uint32_t dbits; memcpy(&dbits, &g->depth, 4);
    for (int ty = g->miny; ty < g->maxy; ++ty)
      for (int tx = g->minx; tx < g->maxx; ++tx) {
        uint64_t tile = (uint64_t)v * tiles + (uint64_t)ty * gx + tx;
        c->inst[off].key = (tile << 32) | dbits; c->inst[off].slot = off; c->inst[off].gid = (uint32_t)n;
        ++off;
      }
  }
  qsort(c->inst, (size_t)c->D, sizeof(Inst), inst_cmp);
  for (int64_t k = 0; k < c->D; ++k) {
    uint32_t t = (uint32_t)(c->inst[k].key >> 32);
    if (k == 0 || t != (uint32_t)(c->inst[k - 1].key >> 32)) c->ranges[2 * t] = (uint32_t)k;
    if (k == c->D - 1 || t != (uint32_t)(c->inst[k + 1].key >> 32)) c->ranges[2 * t + 1] = (uint32_t)(k + 1);
  }
  t_ser += gho_now() - t_mark;
  }

  /* A.3 render */
#pragma omp parallel for schedule(dynamic, 4)
  for (long t = 0; t < (long)NV * tiles; ++t) {
    int v = (int)(t / tiles), tt = (int)(t % tiles), ty = tt / gx, tx = tt % gx;
    const float* bg = in->cams + (size_t)v * GH_CAM_FLOATS + 37;
    uint32_t s0 = c->ranges[2 * t], s1 = c->ranges[2 * t + 1];
    for (int ly = 0; ly < TILE; ++ly) for (int lx = 0; lx < TILE; ++lx) {
      int x = tx * TILE + lx, y = ty * TILE + ly;
      if (x >= W || y >= H) continue;
      float pxf = (float)x, pyf = (float)y;
      float Tr = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f; uint32_t last = 0, cnt = 0;
      for (uint32_t k = s0; k < s1; ++k) {
        ++cnt;
        const GView* g = &c->g[c->inst[k].gid];
        float dx = g->px - pxf, dy = g->py - pyf;
        float power = -0.5f * (g->A * dx * dx + g->C * dy * dy) - g->B * dx * dy;
        if (power > 0.0f) continue;
        float alpha = fminf(0.99f, g->opac * gho_exp(power));
        if (alpha < 1.0f / 255.0f) continue;
        float test_T = Tr * (1.0f - alpha);
        if (test_T < 0.0001f) break;
        float w = alpha * Tr;
        C0 = C0 + g->rgb[0] * w; C1 = C1 + g->rgb[1] * w; C2 = C2 + g->rgb[2] * w;   /* mul then add (contract §4) */
        Tr = test_T; last = cnt;
      }
      size_t pix = ((size_t)v * H + y) * W + x;
      c->final_T[pix] = Tr; c->n_contrib[pix] = last;
      float* img = out->image + (size_t)v * 3 * H * W;
      img[(size_t)0 * H * W + (size_t)y * W + x] = fmaf(Tr, bg[0], C0);
      img[(size_t)1 * H * W + (size_t)y * W + x] = fmaf(Tr, bg[1], C1);
      img[(size_t)2 * H * W + (size_t)y * W + x] = fmaf(Tr, bg[2], C2);
    }
  }

  if (dbg) {
    dbg->num_rendered = c->D;
    for (long n = 0; n < (long)NV * P; ++n) {
      const GView* g = &c->g[n];
      if (dbg->xy) { dbg->xy[2 * n] = g->px; dbg->xy[2 * n + 1] = g->py; }
      if (dbg->depth) dbg->depth[n] = g->depth;
      if (dbg->conic_opacity) { dbg->conic_opacity[4 * n] = g->A; dbg->conic_opacity[4 * n + 1] = g->B; dbg->conic_opacity[4 * n + 2] = g->C; dbg->conic_opacity[4 * n + 3] = g->opac; }
      if (dbg->rgb) { dbg->rgb[3 * n] = g->rgb[0]; dbg->rgb[3 * n + 1] = g->rgb[1]; dbg->rgb[3 * n + 2] = g->rgb[2]; }
      if (dbg->rect) dbg->rect[n] = (uint32_t)g->minx | ((uint32_t)g->miny << 8) | ((uint32_t)g->maxx << 16) | ((uint32_t)g->maxy << 24);
      if (dbg->offsets) dbg->offsets[n] = c->offsets[n];
    }
    for (int64_t k = 0; k < c->D && k < dbg->capacity; ++k) {
      if (dbg->sorted_keys) dbg->sorted_keys[k] = c->inst[k].key;
      if (dbg->sorted_gid) dbg->sorted_gid[k] = c->inst[k].gid;
    }
    if (dbg->ranges) memcpy(dbg->ranges, c->ranges, sizeof(uint32_t) * 2 * (size_t)NV * tiles);
    if (dbg->final_T) memcpy(dbg->final_T, c->final_T, sizeof(float) * (size_t)NV * H * W);
    if (dbg->n_contrib) memcpy(dbg->n_contrib, c->n_contrib, sizeof(uint32_t) * (size_t)NV * H * W);
  }
  *ctx_out = c;
  g_t_total += gho_now() - t_begin; g_t_serial += t_ser;
  return GH_OK;
}

/* A.5 for one (view, Gaussian) n = v*P + i: sums its instance records, chain rule, adds into the per-Gaussian accumulators.
 * acc_cw: where the color_w gradient goes (the global array, or a thread's private 48 doubles in baseline mode). */
typedef struct A5Acc { double *m, *o, *s, *q, *c, *sh, *cb, *cov; } A5Acc;
static void a5_one(const GhoCtx* c, const GhInputs* in, const GhGrads* gr, long n, const double* rec, const A5Acc* A, double* acc_cw) {
  const GhDims* d = &c->dims;
  const int P = d->P, H = d->H, W = d->W, M = d->M;
  const int wpg = (d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? 1 : 0;
  double *acc_m = A->m, *acc_o = A->o, *acc_s = A->s, *acc_q = A->q, *acc_c = A->c, *acc_sh = A->sh, *acc_cb = A->cb, *acc_cov = A->cov;
  {
    int v = (int)(n / P), i = (int)(n % P);
    const GView* g = &c->g[n];
    if (gr->dL_dmeans2D) { gr->dL_dmeans2D[3 * n] = 0; gr->dL_dmeans2D[3 * n + 1] = 0; gr->dL_dmeans2D[3 * n + 2] = 0; }
    if (!g->tiles) return;
    const float* cam = in->cams + (size_t)v * GH_CAM_FLOATS;
    const float* V = cam; const float* PM = cam + 16;
    double s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t o1 = c->offsets[n], o0 = o1 - g->tiles;
    for (uint32_t sidx = o0; sidx < o1; ++sidx) for (int q = 0; q < 9; ++q) s9[q] += rec[(size_t)sidx * 9 + q];
    float g_px = (float)s9[0], g_py = (float)s9[1], gA = (float)s9[2], gB = (float)s9[3], gC = (float)s9[4], g_o = (float)s9[5];
    float g_rgb[3] = {(float)s9[6], (float)s9[7], (float)s9[8]};
    if (gr->dL_dmeans2D) { gr->dL_dmeans2D[3 * n] = g_px * 0.5f * (float)W; gr->dL_dmeans2D[3 * n + 1] = g_py * 0.5f * (float)H; }
    Geo e; geo_forward(d, in, cam, i, &e);
    float dm[3] = {0, 0, 0};

    /* colour */
    if (in->colors_precomp) {
      const float* w = in->blend_color_w ? in->blend_color_w + (wpg ? (size_t)i * 48 : 0) : NULL;
      for (int ch = 0; ch < 3; ++ch) {
        float gc = g_rgb[ch];
        if (w) {
          acc_cw[(wpg ? (size_t)i * 48 : 0) + ch] += (double)(gc * in->colors_precomp[3 * i + ch]);
          acc_cw[(wpg ? (size_t)i * 48 : 0) + 3 + ch] += (double)gc;
          acc_c[3 * i + ch] += (double)(gc * w[ch]);
        } else acc_c[3 * i + ch] += (double)gc;
        if (in->blend_color_b) acc_cb[(size_t)i * 48 + ch] += (double)gc;
      }
    } else {
      float dx = e.mx - cam[32], dy = e.my - cam[33], dz = e.mz - cam[34];
      float len = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
      float ux = dx / len, uy = dy / len, uz = dz / len;
      float Bv[16]; float dB[16][3];
      int nb = sh_basis(d->sh_degree, ux, uy, uz, Bv); if (nb > M) nb = M;
      sh_basis_grad(d->sh_degree, ux, uy, uz, dB);
      float ddir[3] = {0, 0, 0};
      for (int ch = 0; ch < 3; ++ch) {
        float gc = (g->clamped & (1u << ch)) ? 0.0f : g_rgb[ch];
        for (int k = 0; k < nb; ++k) {
          float shv = blended_sh(d, in, i, k, ch);
          float gk = Bv[k] * gc;        /* dL/d(blended sh) */
          for (int a = 0; a < 3; ++a) ddir[a] += dB[k][a] * shv * gc;
          float raw = in->shs[((size_t)i * M + k) * 3 + ch];
          if (in->blend_color_w) {
            const float* w = in->blend_color_w + (wpg ? (size_t)i * 48 : 0);
            float wv = w[k * 3 + ch];
            if (in->blend_color_b) {
              acc_sh[((size_t)i * M + k) * 3 + ch] += (double)(gk * wv * wv);
              acc_cw[(wpg ? (size_t)i * 48 : 0) + k * 3 + ch] += (double)(gk * 2.0f * raw * wv);
              acc_cb[(size_t)i * 48 + k * 3 + ch] += (double)gk;
            } else {
              acc_sh[((size_t)i * M + k) * 3 + ch] += (double)(gk * wv);
              acc_cw[(wpg ? (size_t)i * 48 : 0) + k * 3 + ch] += (double)(gk * raw);
            }
          } else acc_sh[((size_t)i * M + k) * 3 + ch] += (double)gk;
        }
      }
      /* normalize backward: u = d/len */
      float dot = ux * ddir[0] + uy * ddir[1] + uz * ddir[2];
      dm[0] += (ddir[0] - ux * dot) / len; dm[1] += (ddir[1] - uy * dot) / len; dm[2] += (ddir[2] - uz * dot) / len;
    }

    /* conic -> dilated cov2D (a,b,c) */
    float a = e.a, b = e.b, cc = e.c, det = e.det;
    float det2inv = 1.0f / (det * det);
    float dL_da = (-cc * cc * gA + b * cc * gB - b * b * gC) * det2inv;
    float dL_dc = (-b * b * gA + a * b * gB - a * a * gC) * det2inv;
    float dL_db = (2.0f * b * cc * gA - (a * cc + b * b) * gB + 2.0f * a * b * gC) * det2inv;
    /* G2 = [[da, db/2],[db/2, dc]];  dSigma(full) = T^T G2 T ; dT = 2 G2 T Sigma */
    float g00 = dL_da, g01 = 0.5f * dL_db, g11 = dL_dc;
    const float* T = e.T;
    float GT[6]; /* G2*T, 2x3 */
    for (int k = 0; k < 3; ++k) { GT[k] = g00 * T[k] + g01 * T[3 + k]; GT[3 + k] = g01 * T[k] + g11 * T[3 + k]; }
    float dS[9]; /* T^T (G2 T), 3x3 */
    for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) dS[3 * p + q] = T[p] * GT[q] + T[3 + p] * GT[3 + q];
    float Sf[9] = {e.S[0], e.S[1], e.S[2], e.S[1], e.S[3], e.S[4], e.S[2], e.S[4], e.S[5]};
    float dT[6];
    for (int r2 = 0; r2 < 2; ++r2) for (int q = 0; q < 3; ++q)
      dT[3 * r2 + q] = 2.0f * (GT[3 * r2] * Sf[q] + GT[3 * r2 + 1] * Sf[3 + q] + GT[3 * r2 + 2] * Sf[6 + q]);
    if (in->cov3D_precomp) {
      /* Sigma itself is the input: symmetric storage (xx xy xz yy yz zz), an off-diagonal entry carries both matrix positions */
      acc_cov[6 * i + 0] += (double)dS[0]; acc_cov[6 * i + 1] += (double)(dS[1] + dS[3]); acc_cov[6 * i + 2] += (double)(dS[2] + dS[6]);
      acc_cov[6 * i + 3] += (double)dS[4]; acc_cov[6 * i + 4] += (double)(dS[5] + dS[7]); acc_cov[6 * i + 5] += (double)dS[8];
    } else {
    /* Sigma = M M^T, M = R diag(s): dM = 2 dS M */
    float Mm[9]; for (int p = 0; p < 3; ++p) for (int j = 0; j < 3; ++j) Mm[3 * p + j] = e.R[3 * p + j] * e.s[j];
    float dM[9];
    for (int p = 0; p < 3; ++p) for (int j = 0; j < 3; ++j)
      dM[3 * p + j] = 2.0f * (dS[3 * p] * Mm[j] + dS[3 * p + 1] * Mm[3 + j] + dS[3 * p + 2] * Mm[6 + j]);
    float dR[9];
    for (int j = 0; j < 3; ++j) {
      float ds = dM[j] * e.R[j] + dM[3 + j] * e.R[3 + j] + dM[6 + j] * e.R[6 + j];
      acc_s[3 * i + j] += (double)(ds * d->scale_modifier);
      for (int p = 0; p < 3; ++p) dR[3 * p + j] = dM[3 * p + j] * e.s[j];
    }
    {
      float r = in->rotations[4 * i], x = in->rotations[4 * i + 1], y = in->rotations[4 * i + 2], z = in->rotations[4 * i + 3];
      float dr = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      float dxq = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
      float dyq = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
      float dzq = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
      acc_q[4 * i] += dr; acc_q[4 * i + 1] += dxq; acc_q[4 * i + 2] += dyq; acc_q[4 * i + 3] += dzq;
    }
    }
    /* T = J W : dJ_ab = sum_c dT_ac W_bc, W_bc = V[4c+b] */
    float dJ00 = dT[0] * V[0] + dT[1] * V[4] + dT[2] * V[8];
    float dJ02 = dT[0] * V[2] + dT[1] * V[6] + dT[2] * V[10];
    float dJ11 = dT[3] * V[1] + dT[4] * V[5] + dT[5] * V[9];
    float dJ12 = dT[3] * V[2] + dT[4] * V[6] + dT[5] * V[10];
    float tz = e.tz, tzi = 1.0f / tz, tz2i = tzi * tzi, tz3i = tz2i * tzi;
    float dtx = e.xclamped ? 0.0f : -e.fx * tz2i * dJ02;
    float dty = e.yclamped ? 0.0f : -e.fy * tz2i * dJ12;
    float dtz = -e.fx * tz2i * dJ00 - e.fy * tz2i * dJ11 + 2.0f * e.fx * e.cx * tz3i * dJ02 + 2.0f * e.fy * e.cy * tz3i * dJ12;
    for (int a2 = 0; a2 < 3; ++a2) dm[a2] += dtx * V[4 * a2] + dty * V[4 * a2 + 1] + dtz * V[4 * a2 + 2];
    /* projection path */
    float dndcx = g_px * 0.5f * (float)W, dndcy = g_py * 0.5f * (float)H;
    float dhx = dndcx * e.winv, dhy = dndcy * e.winv;
    float dhw = -(dndcx * e.hx + dndcy * e.hy) * e.winv * e.winv;
    for (int a2 = 0; a2 < 3; ++a2) dm[a2] += dhx * PM[4 * a2] + dhy * PM[4 * a2 + 1] + dhw * PM[4 * a2 + 3];
    for (int a2 = 0; a2 < 3; ++a2) acc_m[3 * i + a2] += (double)dm[a2];
    acc_o[i] += (double)g_o;
  }
}

/* ------------------------------------------------------------------------------------------- */
int gho_backward(const GhoCtx* c, const GhInputs* in, const GhGrads* gr) {
  if (!c || !in || !gr || !gr->dL_dimage) return GH_ERR_INVALID_ARG;
  const GhDims* d = &c->dims;
  const int P = d->P, NV = d->n_views, H = d->H, W = d->W, M = d->M;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, tiles = gx * gy;
  const double t_begin = gho_now();
  double t_ser = 0.0, t_mark;
  /* per-instance records: dpx dpy dA dB dC do dr dg db */
  double* rec = (double*)calloc((size_t)c->D * 9 + 9, sizeof(double));

  /* A.4 render backward: per pixel, back-to-front */
#pragma omp parallel for schedule(dynamic, 4)
  for (long t = 0; t < (long)NV * tiles; ++t) {
    int v = (int)(t / tiles), tt = (int)(t % tiles), ty = tt / gx, tx = tt % gx;
    const float* bg = in->cams + (size_t)v * GH_CAM_FLOATS + 37;
    uint32_t s0 = c->ranges[2 * t];
    const float* dimg = gr->dL_dimage + (size_t)v * 3 * H * W;
    for (int ly = 0; ly < TILE; ++ly) for (int lx = 0; lx < TILE; ++lx) {
      int x = tx * TILE + lx, y = ty * TILE + ly;
      if (x >= W || y >= H) continue;
      size_t pix = ((size_t)v * H + y) * W + x;
      float pxf = (float)x, pyf = (float)y;
      float T_final = c->final_T[pix];
      uint32_t last = c->n_contrib[pix];
      float dpix[3] = {dimg[(size_t)0 * H * W + (size_t)y * W + x], dimg[(size_t)1 * H * W + (size_t)y * W + x], dimg[(size_t)2 * H * W + (size_t)y * W + x]};
      float bg_dot = bg[0] * dpix[0] + bg[1] * dpix[1] + bg[2] * dpix[2];
      float Tr = T_final, last_alpha = 0.0f, last_col[3] = {0, 0, 0}, accum[3] = {0, 0, 0};
      for (int64_t j = (int64_t)last - 1; j >= 0; --j) {
        uint32_t k = s0 + (uint32_t)j;
        const GView* g = &c->g[c->inst[k].gid];
        float dx = g->px - pxf, dy = g->py - pyf;
        float power = -0.5f * (g->A * dx * dx + g->C * dy * dy) - g->B * dx * dy;
        if (power > 0.0f) continue;
        float G = gho_exp(power);
        float alpha = fminf(0.99f, g->opac * G);
        if (alpha < 1.0f / 255.0f) continue;
        Tr = Tr / (1.0f - alpha);
        float dchannel_dcolor = alpha * Tr;
        float dL_dalpha = 0.0f;
        double* r = rec + (size_t)c->inst[k].slot * 9;
        for (int ch = 0; ch < 3; ++ch) {
          accum[ch] = last_alpha * last_col[ch] + (1.0f - last_alpha) * accum[ch];
          last_col[ch] = g->rgb[ch];
          dL_dalpha += (g->rgb[ch] - accum[ch]) * dpix[ch];
          r[6 + ch] += (double)(dchannel_dcolor * dpix[ch]);
        }
        dL_dalpha *= Tr;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
        float dL_dG = g->opac * dL_dalpha;      /* straight-through the 0.99 clamp (App. A.4-2) */
        float gdx = G * dx, gdy = G * dy;
        float dG_ddelx = -gdx * g->A - gdy * g->B;
        float dG_ddely = -gdy * g->C - gdx * g->B;
        r[0] += (double)(dL_dG * dG_ddelx);      /* d(delta)/d(px) = +1 */
        r[1] += (double)(dL_dG * dG_ddely);
        r[2] += (double)(-0.5f * gdx * dx * dL_dG);
        r[3] += (double)(-gdx * dy * dL_dG);
        r[4] += (double)(-0.5f * gdy * dy * dL_dG);
        r[5] += (double)(G * dL_dalpha);
      }
    }
  }

  /* A.5 preprocess backward: reduce records per (view,Gaussian), chain rule, sum over views */
  double* acc_m = (double*)calloc((size_t)P * 3 + 3, sizeof(double));
  double* acc_o = (double*)calloc((size_t)P + 1, sizeof(double));
  double* acc_s = (double*)calloc((size_t)P * 3 + 3, sizeof(double));
  double* acc_q = (double*)calloc((size_t)P * 4 + 4, sizeof(double));
  double* acc_cov = (double*)calloc((size_t)P * 6 + 6, sizeof(double));
  double* acc_c = (double*)calloc((size_t)P * 3 + 3, sizeof(double));
  double* acc_sh = (double*)calloc((size_t)P * (M > 0 ? M : 1) * 3 + 3, sizeof(double));
  double* acc_cb = (double*)calloc((size_t)P * 48 + 48, sizeof(double));
  int wpg = (d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? 1 : 0;
  double* acc_cw = (double*)calloc((wpg ? (size_t)P * 48 : 48) + 48, sizeof(double));

  const A5Acc A5 = {acc_m, acc_o, acc_s, acc_q, acc_c, acc_sh, acc_cb, acc_cov};
  if (g_parallel) {
    /* baseline mode: Gaussians in parallel, the views of a Gaussian in the checker's (ascending) order */
#pragma omp parallel
    {
      double cw_local[48];
      for (int a = 0; a < 48; ++a) cw_local[a] = 0.0;
#pragma omp for schedule(static)
      for (long i = 0; i < (long)P; ++i)
        for (int v = 0; v < NV; ++v) a5_one(c, in, gr, (long)v * P + i, rec, &A5, wpg ? acc_cw : cw_local);
      if (!wpg) {
#pragma omp critical
        for (int a = 0; a < 48; ++a) acc_cw[a] += cw_local[a];
      }
    }
  } else {
    t_mark = gho_now();
    for (long n = 0; n < (long)NV * P; ++n) a5_one(c, in, gr, n, rec, &A5, acc_cw);
    t_ser += gho_now() - t_mark;
  }


  t_mark = gho_now();
#pragma omp parallel for schedule(static) if (g_parallel)
  for (int i = 0; i < P; ++i) {
    if (gr->dL_dmeans3D) for (int a = 0; a < 3; ++a) gr->dL_dmeans3D[3 * i + a] = (float)acc_m[3 * i + a];
    if (gr->dL_dopacities) gr->dL_dopacities[i] = (float)acc_o[i];
    if (gr->dL_dblend_opacity_b) gr->dL_dblend_opacity_b[i] = (float)acc_o[i];
    if (gr->dL_dscales) for (int a = 0; a < 3; ++a) gr->dL_dscales[3 * i + a] = (float)acc_s[3 * i + a];
    if (gr->dL_drotations) for (int a = 0; a < 4; ++a) gr->dL_drotations[4 * i + a] = (float)acc_q[4 * i + a];
    if (gr->dL_dcov3D) for (int a = 0; a < 6; ++a) gr->dL_dcov3D[6 * i + a] = (float)acc_cov[6 * i + a];
    if (gr->dL_dcolors) for (int a = 0; a < 3; ++a) gr->dL_dcolors[3 * i + a] = (float)acc_c[3 * i + a];
    if (gr->dL_dshs) for (int a = 0; a < M * 3; ++a) gr->dL_dshs[(size_t)i * M * 3 + a] = (float)acc_sh[(size_t)i * M * 3 + a];
    if (gr->dL_dblend_color_b) for (int a = 0; a < 48; ++a) gr->dL_dblend_color_b[(size_t)i * 48 + a] = (float)acc_cb[(size_t)i * 48 + a];
    if (gr->dL_dblend_color_w && wpg) for (int a = 0; a < 48; ++a) gr->dL_dblend_color_w[(size_t)i * 48 + a] = (float)acc_cw[(size_t)i * 48 + a];
  }
  if (gr->dL_dblend_color_w && !wpg) for (int a = 0; a < 48; ++a) gr->dL_dblend_color_w[a] = (float)acc_cw[a];
  if (gr->dL_dblend_xyz_b) {
    double s[3] = {0, 0, 0};
    for (int i = 0; i < P; ++i) for (int a = 0; a < 3; ++a) s[a] += acc_m[3 * i + a];
    for (int a = 0; a < 3; ++a) gr->dL_dblend_xyz_b[a] = (float)s[a];
  }
  if (!g_parallel) t_ser += gho_now() - t_mark;
  free(rec); free(acc_m); free(acc_o); free(acc_s); free(acc_q); free(acc_cov); free(acc_c); free(acc_sh); free(acc_cb); free(acc_cw);
  g_t_total += gho_now() - t_begin; g_t_serial += t_ser;
  return GH_OK;
}

/* exposed so tests can check the reproducible exp against libm */
float gho_exp_public(float x) { return gho_exp(x); }
/* number of OpenMP threads of the following calls (bench.py's cpu_baseline picks the count that is fastest on the host it runs on:
 * a container's CPU share is usually far smaller than the host's core count) */
void gho_set_num_threads(int n) {
#ifdef _OPENMP
  extern void omp_set_num_threads(int);
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int gho_num_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * kNN oracle for the interaction mask (infer_one_shot.py:247-250). knn_points is pytorch3d.ops (third-party, not in
 * /root/reference; "parity unpinned"): brute force over all points, squared distance accumulated in x, y, z order as
 * `dist += diff * diff` — which its compiler contracts to fma(dz,dz, fma(dy,dy, dx*dx)) —, the K smallest returned
 * sorted by distance. Ties are ordered by ascending index (the library leaves them unspecified).
 * queries: indices of the nq query points (NULL = all N points, nq ignored). idx_out (nq,K), dist_out (nq,K) or NULL.
 */
static void gho_sift_down(uint64_t* heap, int n, int i) {          /* max-heap on the u64 keys */
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && heap[l] > heap[m]) m = l;
    if (r < n && heap[r] > heap[m]) m = r;
    if (m == i) return;
    uint64_t t = heap[i]; heap[i] = heap[m]; heap[m] = t;
    i = m;
  }
}

static int gho_cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

int gho_knn(const float* points, int N, const int32_t* queries, int nq, int K, int32_t* idx_out, float* dist_out) {
  if (N < 1 || K < 1 || K > N || !points || !idx_out) return -1;
  if (!queries) nq = N;
  int status = 0;
#pragma omp parallel
  {
    uint64_t* heap = (uint64_t*)malloc((size_t)K * sizeof(uint64_t));
    if (!heap) {
#pragma omp atomic write
      status = -2;
    }
#pragma omp for schedule(dynamic, 16)
    for (int qi = 0; qi < nq; ++qi) {
      if (!heap) continue;
      const int q = queries ? queries[qi] : qi;
      const float qx = points[3 * q], qy = points[3 * q + 1], qz = points[3 * q + 2];
      int n = 0;
      for (int j = 0; j < N; ++j) {
        const float dx = qx - points[3 * j], dy = qy - points[3 * j + 1], dz = qz - points[3 * j + 2];
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        uint32_t bits;
        memcpy(&bits, &d2, 4);
        const uint64_t key = ((uint64_t)bits << 32) | (uint32_t)j;      /* (distance, index) lexicographic */
        if (n < K) {
          heap[n++] = key;
          if (n == K) for (int i = K / 2 - 1; i >= 0; --i) gho_sift_down(heap, K, i);
        } else if (key < heap[0]) {
          heap[0] = key;
          gho_sift_down(heap, K, 0);
        }
      }
      qsort(heap, (size_t)K, sizeof(uint64_t), gho_cmp_u64);
      for (int k = 0; k < K; ++k) {
        idx_out[(size_t)qi * K + k] = (int32_t)(uint32_t)heap[k];
        if (dist_out) { uint32_t b = (uint32_t)(heap[k] >> 32); memcpy(&dist_out[(size_t)qi * K + k], &b, 4); }
      }
    }
    free(heap);
  }
  return status;
}
