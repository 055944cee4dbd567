// gh_internal.h — shared device helpers and launcher prototypes for the gfx950 rasteriser kernels.
// Written for CDNA4 only: 64-lane wavefronts, DPP cross-lane ops, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gh_raster.h"

#define GH_WAVE 64
#define GH_BLOCK 256                 // 4 waves: one 8x8 pixel quadrant of a 16x16 tile per wave
#define GH_REC 9                     // LDS stride (floats) of a partial gradient record: 18 KB per block keeps 8 blocks per CU
#define GH_REC_G 9                   // floats per (instance, quadrant) sub-record in HBM: packed, three 12-byte accesses
#define GH_FINE_TILES 2048           // launches of at most this many tiles run the backward with one wave per 4x4 block
// Small launches of the forward (gh_render_fwd_kernel<.., FINE>: at most GH_FWD_FINE_TILES tiles, up to four views of 512x334 — there the
// kernel is as long as its heaviest waves): (1) the launch order goes by what the PREVIOUS forward over the workspace measured
// per tile (the most list entries one 4x4-pixel block let through = its longest wave's work) instead of the list length; (2) the
// GH_FWD_FINE_K tiles at the head of that order whose list holds at least GH_FWD_FINE_MIN entries are walked by 64 waves of 2x2 pixels x
// 16 depth slots instead of 16 waves of 4x4 pixels x 4 slots — the same arithmetic in the same order per pixel, a quarter of the trips
// per wave at 2.4x the instructions per tile; (3) the backward's work list keeps the first quarter of the launch order in a region of
// its own, taken first. Measured optimum K = 24..48 at one, two and four views alike; from six views up all three cost more than they buy.
#ifndef GH_FWD_FINE_TILES
#define GH_FWD_FINE_TILES 3072
#endif
#define GH_FWD_FINE_K 32
// The backward's work list: regions by the cost (cycles >> 8 of the slowest quadrant) the previous backward measured per (tile, depth
// segment); GH_BWD_COST_SLOTS history words per tile (segments past the last share it), classes of 2^GH_BWD_CLASS_SHIFT x 256 cycles.
#define GH_BWD_CLASSES 16
#define GH_BWD_COST_SLOTS 8
#ifndef GH_BWD_CLASS_SHIFT
#define GH_BWD_CLASS_SHIFT 5
#endif
#define GH_FWD_FINE_MIN 256

struct GhF3 { float x, y, z; };      // 12-byte access (global_load/store_dwordx3)

// Streaming stores (global_store ... nt): data written once here and read by a LATER kernel does not need a place in the L2 —
// kept out of it, the lines this kernel gathers or re-reads stay resident (gh_ranges_kernel: 54 -> 45 us at 512x334, 158 -> 128 us
// at 1024x1024 with the per-instance records streamed).
typedef float gh_v4f __attribute__((ext_vector_type(4)));
typedef float gh_v2f __attribute__((ext_vector_type(2)));
typedef uint32_t gh_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gh_stream(float4* p, const float4& v) { __builtin_nontemporal_store((gh_v4f){v.x, v.y, v.z, v.w}, (gh_v4f*)p); }
__device__ __forceinline__ void gh_stream(float2* p, const float2& v) { __builtin_nontemporal_store((gh_v2f){v.x, v.y}, (gh_v2f*)p); }
__device__ __forceinline__ void gh_stream(uint2* p, const uint2& v) { __builtin_nontemporal_store((gh_v2u){v.x, v.y}, (gh_v2u*)p); }
__device__ __forceinline__ void gh_stream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void gh_stream(float* p, float v) { __builtin_nontemporal_store(v, p); }

struct GhGrid {
  int P, NV, H, W, gx, gy, tiles, N;  // N = NV*P
  int tile_bits, n_pass;
  int64_t cap;                        // max_instances
  int n_items;                        // capacity of the backward work list: NV*tiles + cap/GH_SEGMENT + 2
  uint32_t flags;                     // GhDims.flags
  int64_t total_tiles;                // tiles of the whole call (a half of a GH_FLAG_SPLIT_STREAMS call keeps the full call's
                                      // kernel variants, so that its results are the unsplit call's bit for bit)
};

static inline GhGrid gh_make_grid(const GhDims* d) {
  GhGrid g;
  g.P = d->P; g.NV = d->n_views; g.H = d->H; g.W = d->W;
  g.gx = (d->W + GH_TILE - 1) / GH_TILE; g.gy = (d->H + GH_TILE - 1) / GH_TILE;
  g.tiles = g.gx * g.gy; g.N = d->n_views * d->P;
  int tb = 1; while ((1ll << tb) < (long long)g.tiles * d->n_views) ++tb;
  g.tile_bits = tb; g.n_pass = 4 + (tb + 7) / 8;   // level-1 depth passes + level-3 tile passes
  g.cap = d->max_instances;
  g.n_items = (int)((size_t)g.NV * g.tiles + (size_t)g.cap / GH_SEGMENT + 2);
  g.total_tiles = (int64_t)g.NV * g.tiles;
  g.flags = d->flags;
  return g;
}

// Floats of the fused image loss's partial sums (GhLayout.loss_partials): one per 8x8-pixel quadrant; a launch that may run tiles in
// the fine-grained form keeps four per quadrant (one per workgroup of a fine tile; a coarse tile's workgroup zeroes the other three).
static inline bool gh_fwd_fine_launch(const GhGrid& g) { return g.total_tiles <= GH_FWD_FINE_TILES; }
// The forward's launch order is ranked by spare workgroups of the projection kernel, from what the previous forward over the workspace
// left per tile (tile_walk[3]: small launches the most entries one 4x4 block took, large ones half the entries the tile walked),
// instead of gh_tile_order_kernel behind the binning — whenever there is a projection kernel (N > 0) and no A/B switch says otherwise.
bool gh_heavy_order_enabled();                         // (GH_FWD_HEAVY_ORDER=0 in the environment: launch order by list length, gh_binning.hip)
// Launches of at most GH_ORDER_TILES tiles: from ten rounds of workgroups up the tail a bad order leaves is short against the kernel and
// what concurrent workgroups share in the L2 weighs more (1024^2 x 8 views: backward +25 us with the classed list, 16 views of 512x334: +-0).
#ifndef GH_ORDER_TILES
#define GH_ORDER_TILES 8192
#endif
#ifndef GH_CLASS_TILES
#define GH_CLASS_TILES GH_ORDER_TILES
#endif
static inline bool gh_order_in_projection(const GhGrid& g) {
  return g.total_tiles <= GH_ORDER_TILES && g.N > 0 && !(g.flags & GH_FLAG_FRESH_ORDER) && gh_heavy_order_enabled();
}
bool gh_bwd_classes_enabled();                         // (GH_BWD_CLASSES=0 in the environment: the work list in one piece, gh_binning.hip)
// 0: one region (completion order, taken from the end: rounds 2-5); 1: regions by the cost the previous backward measured per (tile,
// segment); 2: two regions by the forward's launch order (launches whose backward runs in its four-wave form)
static inline uint32_t gh_bwd_class_mode(const GhGrid& g) {
  if (!gh_bwd_classes_enabled() || g.total_tiles > GH_CLASS_TILES) return 0u;
  return g.total_tiles <= GH_FINE_TILES ? 2u : 1u;
}
static inline size_t gh_loss_partial_count(const GhGrid& g) {
  return (size_t)g.NV * g.tiles * (gh_fwd_fine_launch(g) ? 16 : 4);
}

// ---- launchers (each enqueues on `s`, never synchronises) --------------------------------------
void gh_launch_preprocess_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, int32_t* radii,
                              char* ws, const GhLayout& L, hipStream_t s);
// tile_depth_bound: GhInputs.tile_depth_bound (the kernels that decide list membership for rects without a hit mask repeat the test)
void gh_launch_binning(const GhDims* d, const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s, const float* tile_depth_bound = nullptr);
// wg: workspace that owns the geometry / binning arrays (ranges, tile order, inst_r0, sorted_slot, slot_begin, tiles_touched,
// vals); ws: workspace of this call's own state (colour records, image state, backward scratch). wg == ws for a plain call;
// they differ for a second call over the same geometry (gh_forward_shared).
// seen / seen_scale: GhOutputs.tile_depth_seen (optional; full forwards only)
void gh_launch_render_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, float* image, float* alpha,
                          const char* wg, char* ws, const GhLayout& L, hipStream_t s, float* seen = nullptr, float seen_scale = 1.0f,
                          uint32_t seen_slack = 0u, const GhOutputs* fused = nullptr,     // fused: GhOutputs.l1_* (full forwards only)
                          bool own_order = false);    // the launch order is ws's own (gh_launch_refresh re-ranked it), not wg's
// out[0] = scale * (fixed-order sum of n floats, n a multiple of 4, 16-byte aligned): one workgroup (gh_loss.hip)
// scale_behind: the factor is the float the forward left behind the last partial (times `scale`)
void gh_launch_partials_sum(const float* partials, size_t n, float scale, float* out, hipStream_t s, bool scale_behind = false);
// geom: gh_records_need_geometry(in, gr) — false: the sub-records carry the colour / opacity moments only
void gh_launch_render_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const float* dL_dimage,
                          const float* dL_dalpha, const float* upstream_scale, const char* wg, char* ws, const GhLayout& L, hipStream_t s,
                          bool geom, float* deferred_loss = nullptr);     // deferred_loss: GhGrads.deferred_loss
// Is a gradient that flows through the projection wanted (means, scales, rotations, means2D, xyz_b)?
static inline bool gh_wants_geometry(const GhInputs* in, const GhGrads* gr) {
  return gr->dL_dmeans3D || gr->dL_dmeans2D || gr->dL_dscales || gr->dL_drotations || gr->dL_dcov3D ||
         (in->blend_xyz_b && gr->dL_dblend_xyz_b);
}
// Do the render backward's sub-records need their five position / conic moments? Not when no geometry gradient is wanted and
// the colours are precomputed (with SH colours gh_record_sum_kernel reads whole records).
static inline bool gh_records_need_geometry(const GhInputs* in, const GhGrads* gr) {
  return gh_wants_geometry(in, gr) || !in->colors_precomp;
}
void gh_launch_recolour(const GhDims* d, const GhGrid& g, const GhInputs* in, const char* wg, char* ws, const GhLayout& L, hipStream_t s);
// gh_forward_refresh: per-instance records (opacity, colour, block mask) of the CURRENT opacities / colours over the lists of wg
// returns true when it re-ranked the forward's launch order into ws's own tile_order (static lists: the order of the BUILD call is by
// list length — no measurement existed then; every refresh has the previous step's)
bool gh_launch_refresh(const GhDims* d, const GhGrid& g, const GhInputs* in, const char* wg, char* ws, const GhLayout& L, hipStream_t s);
// LSD radix sort of (keys, vals) on bits [0, nbits): ceil(nbits/8) stable passes, element count read from device memory
// (*n_ptr <= cap); pointers are swapped so that on return k_in / v_in hold the result. table: gh_radix_table_words(cap).
void gh_radix_sort(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr, uint32_t cap,
                   int nbits, uint32_t* table, hipStream_t s);
size_t gh_radix_table_words(size_t cap);
int gh_radix_passes(size_t cap, int nbits);
// Variable-length segments [vstart[v], vstart[v+1]) (device memory) of one array, each sorted among itself (gh_binning.hip)
void gh_radix_sort_var(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* vstart, int nseg,
                       uint32_t cap, int nbits, uint32_t* table, hipStream_t s);
// Is the tile partition of this launch done per view (tile ids inside the view as keys)? From two views on, when that takes no
// more launches than the partition by global tile id (small launches sort in two launches per pass, the per-view form in three).
static inline bool gh_partition_per_view(const GhGrid& g) {
  if (g.NV < 2 || g.cap == 0) return false;
  int vbits = 1; while ((1 << vbits) < g.tiles) ++vbits;
  const int items = (size_t)g.cap <= ((size_t)1 << 21) ? 4 : ((size_t)g.cap <= ((size_t)1 << 25) ? 8 : 16);
  const size_t nblk = ((size_t)g.cap + (size_t)GH_BLOCK * items - 1) / ((size_t)GH_BLOCK * items);
  const int per_pass_global = nblk <= 128 ? 2 : 3;
  const int launches_global = gh_radix_passes((size_t)g.cap, g.tile_bits) * per_pass_global, launches_var = ((vbits + 7) / 8) * 3;
  return launches_var <= launches_global;
}
// General form: `segs` independent segments of exactly seg_len elements each (seg_len = 0: one segment, count from *n_ptr).
// table: gh_radix_table_words(per-segment capacity, segs).
// key_bits / n_bits: optional per-producer-block (OR, AND) of the key bits: a pass whose digit no two keys differ in is a copy.
void gh_radix_sort_ex(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr, uint32_t cap,
                      int nbits, uint32_t seg_len, int segs, uint32_t* table, hipStream_t s, const uint2* key_bits = nullptr,
                      int n_bits = 0, uint32_t* wide_flag = nullptr);    // wide_flag (needs key_bits): |= GH_COUNTER_DEPTH24_OK when no key bit
                                                                         // above 23 varies, |= 8 when one does and nbits <= 24
size_t gh_radix_table_words(size_t per_segment, int segs);
void gh_launch_sh_colour_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, char* ws, const GhLayout& L, hipStream_t s);
// wg: workspace whose tiles_touched (visibility of a (view, Gaussian)) counts — the geometry owner's
int gh_launch_sh_colour_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const GhGrads* gr, const char* wg, char* ws,
                            const GhLayout& L, hipStream_t s);
// parts: GH_PBWD_RECORD_SUM (fixed-order sums of the render backward's sub-records, per (view, Gaussian)) and / or
// GH_PBWD_CHAIN (SH colour backward, chain rule, blend-parameter reductions). With precomputed colours the chain-rule kernel
// sums the sub-records itself and GH_PBWD_RECORD_SUM launches nothing. v_split >= 0: the chain call of a split backward — the
// sub-records of views >= v_split sit cap_a instances further on (the second half's share of the per-instance arrays).
#define GH_PBWD_RECORD_SUM 1
#define GH_PBWD_CHAIN 2
void gh_launch_preprocess_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const GhGrads* gr,
                              const char* wg, char* ws, const GhLayout& L, hipStream_t s,
                              int parts = GH_PBWD_RECORD_SUM | GH_PBWD_CHAIN, int v_split = -1, size_t cap_a = 0);

#if defined(__HIPCC__)
// ---- arithmetic contract: fp32, no implicit contraction, FMAs only where written ----

// exp(x) from IEEE primitives only so CPU oracle and GPU agree bit for bit wherever the value is USED (x <= 0, result not
// negligible): 2^(x*log2e): n = rne(t), Taylor-6 of 2^f on [-.5,.5], v_ldexp_f32.
// No clamps (round 3: v_min / v_max cost 4.4 cycles each, as much as an fma — profiles/r3_valu_cycles_pmc.txt). Below t = -126
// the oracle returns 0; here v_cvt_i32_f32 saturates and v_ldexp_f32 underflows to a denormal or 0 — the only consumer is
// alpha = min(0.99, o * exp), and o * (<= 2^-126) < 1/255 is skipped exactly like o * 0 (any finite opacity below 3e35). For
// x > 0 (a conic that is not positive definite) the value may overflow to +inf: every caller rejects the entry on `power > 0`
// by a select, never by arithmetic on this value.
// PRECONDITION: x is finite or NaN-free. x = -inf gives t - rint(t) = NaN and a NaN result, which the callers' fminf(0.99f, ·)
// would turn into alpha = 0.99 where the oracle's clamped exp gives 0: reachable only through a non-finite conic or pixel
// offset, which the +0.3 dilation (det >= 0.09 for finite inputs) and the finite-input contract of the projection rule out.
__device__ __forceinline__ float gh_exp(float x) {
  float t = x * 1.44269504088896341f;
  float n = __builtin_rintf(t);
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

struct GhGeo {
  float mx, my, mz, tx, ty, tz, hx, hy, hw, winv;
  float S[6], R[9], s[3], T[6];
  float cx, cy, fx, fy, a, b, c, det;
  bool xclamped, yclamped;
};

// View transform, Sigma3D, EWA projection — one Gaussian, one camera. cam = GH_CAM_FLOATS record.
// COV: Sigma3D is given (GhInputs.cov3D_precomp) — a template parameter, so that the common kernels carry neither the branch nor
// its registers (as a run-time branch it cost the projection kernel 2 us and the chain rule 2 us of 39 / 70).
template <bool COV = false>
__device__ __forceinline__ void gh_geo_forward(const GhInputs& in, const float* __restrict__ cam, int i,
                                               float mod, int H, int W, GhGeo& o) {
  const float* V = cam; const float* PM = cam + 16;
  float mx = in.means3D[3 * i], my = in.means3D[3 * i + 1], mz = in.means3D[3 * i + 2];
  if (in.blend_xyz_b) { mx = mx + in.blend_xyz_b[0]; my = my + in.blend_xyz_b[1]; mz = mz + in.blend_xyz_b[2]; }
  o.mx = mx; o.my = my; o.mz = mz;
  o.tx = fmaf(V[0], mx, fmaf(V[4], my, fmaf(V[8], mz, V[12])));
  o.ty = fmaf(V[1], mx, fmaf(V[5], my, fmaf(V[9], mz, V[13])));
  o.tz = fmaf(V[2], mx, fmaf(V[6], my, fmaf(V[10], mz, V[14])));
  o.hx = fmaf(PM[0], mx, fmaf(PM[4], my, fmaf(PM[8], mz, PM[12])));
  o.hy = fmaf(PM[1], mx, fmaf(PM[5], my, fmaf(PM[9], mz, PM[13])));
  o.hw = fmaf(PM[3], mx, fmaf(PM[7], my, fmaf(PM[11], mz, PM[15])));
  o.winv = 1.0f / (o.hw + 1e-7f);
  if (COV) {                                         // the published module's cov3D_precomp: Sigma as given (no scale_modifier)
#pragma unroll
    for (int k = 0; k < 6; ++k) o.S[k] = in.cov3D_precomp[6 * i + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) o.R[k] = 0.0f;
    o.s[0] = o.s[1] = o.s[2] = 0.0f;
  } else {
    o.s[0] = mod * in.scales[3 * i]; o.s[1] = mod * in.scales[3 * i + 1]; o.s[2] = mod * in.scales[3 * i + 2];
    float r = in.rotations[4 * i], x = in.rotations[4 * i + 1], y = in.rotations[4 * i + 2], z = in.rotations[4 * i + 3];
    float* R = o.R;
    R[0] = 1.0f - 2.0f * fmaf(y, y, z * z); R[1] = 2.0f * fmaf(x, y, -(r * z)); R[2] = 2.0f * fmaf(x, z, r * y);
    R[3] = 2.0f * fmaf(x, y, r * z); R[4] = 1.0f - 2.0f * fmaf(x, x, z * z); R[5] = 2.0f * fmaf(y, z, -(r * x));
    R[6] = 2.0f * fmaf(x, z, -(r * y)); R[7] = 2.0f * fmaf(y, z, r * x); R[8] = 1.0f - 2.0f * fmaf(x, x, y * y);
    float M[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int j = 0; j < 3; ++j) M[3 * a + j] = R[3 * a + j] * o.s[j];
    o.S[0] = fmaf(M[0], M[0], fmaf(M[1], M[1], M[2] * M[2]));
    o.S[1] = fmaf(M[0], M[3], fmaf(M[1], M[4], M[2] * M[5]));
    o.S[2] = fmaf(M[0], M[6], fmaf(M[1], M[7], M[2] * M[8]));
    o.S[3] = fmaf(M[3], M[3], fmaf(M[4], M[4], M[5] * M[5]));
    o.S[4] = fmaf(M[3], M[6], fmaf(M[4], M[7], M[5] * M[8]));
    o.S[5] = fmaf(M[6], M[6], fmaf(M[7], M[7], M[8] * M[8]));
  }
  float tanx = cam[35], tany = cam[36];
  float limx = 1.3f * tanx, limy = 1.3f * tany;
  float txtz = o.tx / o.tz, tytz = o.ty / o.tz;
  float cxr = fminf(limx, fmaxf(-limx, txtz)), cyr = fminf(limy, fmaxf(-limy, tytz));
  o.xclamped = (cxr != txtz); o.yclamped = (cyr != tytz);
  o.cx = cxr * o.tz; o.cy = cyr * o.tz;
  o.fx = (float)W / (2.0f * tanx); o.fy = (float)H / (2.0f * tany);
  float tz2 = o.tz * o.tz;
  float J00 = o.fx / o.tz, J02 = -(o.fx * o.cx) / tz2, J11 = o.fy / o.tz, J12 = -(o.fy * o.cy) / tz2;
  float* T = o.T;
  T[0] = fmaf(J00, V[0], J02 * V[2]); T[1] = fmaf(J00, V[4], J02 * V[6]); T[2] = fmaf(J00, V[8], J02 * V[10]);
  T[3] = fmaf(J11, V[1], J12 * V[2]); T[4] = fmaf(J11, V[5], J12 * V[6]); T[5] = fmaf(J11, V[9], J12 * V[10]);
  const float* S = o.S;
  float U0 = fmaf(T[0], S[0], fmaf(T[1], S[1], T[2] * S[2]));
  float U1 = fmaf(T[0], S[1], fmaf(T[1], S[3], T[2] * S[4]));
  float U2 = fmaf(T[0], S[2], fmaf(T[1], S[4], T[2] * S[5]));
  float U3 = fmaf(T[3], S[0], fmaf(T[4], S[1], T[5] * S[2]));
  float U4 = fmaf(T[3], S[1], fmaf(T[4], S[3], T[5] * S[4]));
  float U5 = fmaf(T[3], S[2], fmaf(T[4], S[4], T[5] * S[5]));
  float c00 = fmaf(U0, T[0], fmaf(U1, T[1], U2 * T[2]));
  float c01 = fmaf(U0, T[3], fmaf(U1, T[4], U2 * T[5]));
  float c11 = fmaf(U3, T[3], fmaf(U4, T[4], U5 * T[5]));
  o.a = c00 + 0.3f; o.b = c01; o.c = c11 + 0.3f;
  o.det = fmaf(o.a, o.c, -(o.b * o.b));
}

// ---- SH basis (real, degree <= 3) ----------------------------------------------------------------
#define GH_SH_C0 0.28209479177387814f
#define GH_SH_C1 0.4886025119029199f
#define GH_SH_C2_0 1.0925484305920792f
#define GH_SH_C2_1 -1.0925484305920792f
#define GH_SH_C2_2 0.31539156525252005f
#define GH_SH_C2_3 -1.0925484305920792f
#define GH_SH_C2_4 0.5462742152960396f
#define GH_SH_C3_0 -0.5900435899266435f
#define GH_SH_C3_1 2.890611442640554f
#define GH_SH_C3_2 -0.4570457994644658f
#define GH_SH_C3_3 0.3731763325901154f
#define GH_SH_C3_4 -0.4570457994644658f
#define GH_SH_C3_5 1.445305721320277f
#define GH_SH_C3_6 -0.5900435899266435f

__device__ __forceinline__ int gh_sh_basis(int deg, float x, float y, float z, float* Bv) {
  Bv[0] = GH_SH_C0;
  if (deg < 1) return 1;
  Bv[1] = -GH_SH_C1 * y; Bv[2] = GH_SH_C1 * z; Bv[3] = -GH_SH_C1 * x;
  if (deg < 2) return 4;
  float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  Bv[4] = GH_SH_C2_0 * xy;
  Bv[5] = GH_SH_C2_1 * yz;
  Bv[6] = GH_SH_C2_2 * (2.0f * zz - xx - yy);
  Bv[7] = GH_SH_C2_3 * xz;
  Bv[8] = GH_SH_C2_4 * (xx - yy);
  if (deg < 3) return 9;
  Bv[9]  = GH_SH_C3_0 * y * (3.0f * xx - yy);
  Bv[10] = GH_SH_C3_1 * xy * z;
  Bv[11] = GH_SH_C3_2 * y * (4.0f * zz - xx - yy);
  Bv[12] = GH_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
  Bv[13] = GH_SH_C3_4 * x * (4.0f * zz - xx - yy);
  Bv[14] = GH_SH_C3_5 * z * (xx - yy);
  Bv[15] = GH_SH_C3_6 * x * (xx - 3.0f * yy);
  return 16;
}

// blended SH coefficient (renderer_one_shot.py:330-334, incl. the double multiply when color_b is given)
__device__ __forceinline__ float gh_blended_sh(const GhInputs& in, uint32_t flags, int M, int i, int k, int ch) {
  float s = in.shs[((size_t)i * M + k) * 3 + ch];
  if (in.blend_color_w) {
    const float* w = in.blend_color_w + ((flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? (size_t)i * 48 : 0);
    s = s * w[k * 3 + ch];
    if (in.blend_color_b) {
      s = s * w[k * 3 + ch];
      s = s + in.blend_color_b[(size_t)i * 48 + k * 3 + ch];
    }
  }
  return s;
}

// Blended precomputed colour of row i (renderer_one_shot.py:321-328): c*w[0:3] + w[3:6] - 1 (+ b[0:3]).
__device__ __forceinline__ void gh_blended_rgb(const GhInputs& in, uint32_t flags, int i, float* rgb) {
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float col = in.colors_precomp[3 * i + ch];
    if (in.blend_color_w) {
      const float* w = in.blend_color_w + ((flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? (size_t)i * 48 : 0);
      col = col * w[ch]; col = col + w[3 + ch]; col = col - 1.0f;
    }
    if (in.blend_color_b) col = col + in.blend_color_b[(size_t)i * ((flags & GH_FLAG_BLEND_COLOR_B_RGB) ? 3 : 48) + ch];
    rgb[ch] = col;
  }
}

// Culling test: can Gaussian (g0 = px,py,A,B; g1 = C,opacity,..) reach alpha >= 1/255 at any pixel centre of the
// block [qx0, qx0+ext] x [qy0, qy0+ext]?  alpha >= 1/255  <=>  q(d) = A dx^2 + 2B dx dy + C dy^2 <= 2 ln(255 o).
// For a positive definite conic q is convex with its minimum 0 at the centre and grows along every ray from it, so its
// minimum over the rectangle lies on an edge that FACES the centre: on the vertical line dx = clamp(0, lx, ux) or the
// horizontal line dy = clamp(0, ly, uy) (the near edges; a line through the centre when the centre's coordinate is inside
// the range — then the point found is inside the rectangle and at most ties the minimum; centre inside: q = 0). Each line
// is a clamped 1-D parabola: an exact ellipse/rectangle intersection up to rounding, from two edges instead of four.
// A conic that is not positive definite (a determinant that rounded to <= 0: needle-shaped covariances) or anything
// non-finite answers "hit". The threshold carries a margin for the approximate log / rcp. A false "hit" only costs
// time and a false "miss" is impossible within the margin: the exact per-pixel tests of App. A.3 still decide.
__device__ __forceinline__ bool gh_block_hit(const float4& g0, const float4& g1, float qx0, float qy0, float ext) {
  const float o = g1.y;
  if (!(o >= 1.0f / 255.0f)) return false;          // alpha = min(.99, o*exp(p<=0)) <= o < 1/255 everywhere
  const float thr = 2.0f * (__logf(255.0f * o) * 1.0001f + 1e-3f);
  const float A = g0.z, B = g0.w, C = g1.x;
  if (!(C > 0.0f && A * C - B * B > 0.0f)) return true;
  const float lx = qx0 - g0.x, ux = lx + ext, ly = qy0 - g0.y, uy = ly + ext;   // offset ranges of the block
  const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
  const float xn = fminf(fmaxf(0.0f, lx), ux), yn = fminf(fmaxf(0.0f, ly), uy);  // the lines facing the centre
  // on dx = xn the best dy = clamp(-B xn / C); on dy = yn the best dx = clamp(-B yn / A)
  const float dyb = fminf(fmaxf(-B * xn * rC, ly), uy), dxb = fminf(fmaxf(-B * yn * rA, lx), ux);
  const float q0 = A * xn * xn + 2.0f * B * xn * dyb + C * dyb * dyb;
  const float q1 = A * dxb * dxb + 2.0f * B * dxb * yn + C * yn * yn;
  const float qmin = fminf(q0, q1);
  // rounding of the (cancelling) terms of q: bounded by a few ulps of the largest term magnitude anywhere on the block,
  // A mx^2 + 2|B| mx my + C my^2 with (mx, my) the largest offsets — matters for far off-screen centres of elongated conics
  const float mx = fmaxf(fabsf(lx), fabsf(ux)), my = fmaxf(fabsf(ly), fabsf(uy));
  const float mag = A * mx * mx + 2.0f * fabsf(B) * mx * my + C * my * my;
  const bool miss = qmin * 0.9999f - 1e-6f * mag > thr;
  return !miss;                                       // NaN compares false -> hit
}
// Exact tile culling (binning) and the per-instance 4x4-block mask (render kernels) are both this test:
//   tile (tx, ty):            gh_block_hit(g0, g1, 16 tx, 16 ty, 15)
//   block (bx, by) of a tile: gh_block_hit(g0, g1, 16 tx + 4 bx, 16 ty + 4 by, 3)   -> bit by*4 + bx
// Same decision as 16 gh_block_hit(.., 3) calls, with the per-line terms shared: on the vertical line dx = u the
// parabola in dy is C (dy - dy*)^2 + (A - B^2/C) u^2 with dy* = -B u / C, so a line costs a clamp, a subtract and an
// fma (and symmetrically for horizontal lines); the facing line of a block depends on its column (row) only.
// The margin of the threshold absorbs the different rounding.
__device__ __forceinline__ uint32_t gh_block_mask16(const float4& g0, const float4& g1, float tx0, float ty0) {
  const float o = g1.y;
  if (!(o >= 1.0f / 255.0f)) return 0u;
  const float thr = 2.0f * (__logf(255.0f * o) * 1.0001f + 1e-3f);
  const float A = g0.z, B = g0.w, C = g1.x;
  const float rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
  const float sx = -B * rC, sy = -B * rA;            // dy* = sx * u on a vertical line, dx* = sy * v on a horizontal one
  const float Kx = A - B * B * rC, Ky = C - B * B * rA;
  const float ox = tx0 - g0.x, oy = ty0 - g0.y;      // tile origin relative to the centre
  // rounding margin for the whole tile (see gh_block_hit): a few ulps of the largest term magnitude on it
  const float mx = fmaxf(fabsf(ox), fabsf(ox + 15.0f)), my = fmaxf(fabsf(oy), fabsf(oy + 15.0f));
  const float thr_m = (thr + 1e-6f * (A * mx * mx + 2.0f * fabsf(B) * mx * my + C * my * my)) * 1.0002f;   // (q * 0.9999 > t)
  if (!(C > 0.0f && A * C - B * B > 0.0f && Kx > 0.0f && Ky > 0.0f && thr_m == thr_m)) return 0xFFFFu;    // not convex / non-finite
  float lo_x[4], hi_x[4], lo_y[4], hi_y[4], us[4], vs[4], ux2[4], vy2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lo_x[k] = ox + (float)(4 * k); hi_x[k] = ox + (float)(4 * k + 3);         // pixel-centre offsets of block column k
    lo_y[k] = oy + (float)(4 * k); hi_y[k] = oy + (float)(4 * k + 3);
    const float xn = fminf(fmaxf(0.0f, lo_x[k]), hi_x[k]), yn = fminf(fmaxf(0.0f, lo_y[k]), hi_y[k]);   // facing lines
    us[k] = sx * xn; vs[k] = sy * yn;
    ux2[k] = Kx * xn * xn; vy2[k] = Ky * yn * yn;
  }
  uint32_t m = 0;
#pragma unroll
  for (int by = 0; by < 4; ++by) {
#pragma unroll
    for (int bx = 0; bx < 4; ++bx) {
      const float d = fminf(fmaxf(us[bx], lo_y[by]), hi_y[by]) - us[bx];       // vertical line of column bx within row by
      const float e = fminf(fmaxf(vs[by], lo_x[bx]), hi_x[bx]) - vs[by];       // horizontal line of row by within column bx
      const float qmin = fminf(fmaf(C * d, d, ux2[bx]), fmaf(A * e, e, vy2[by]));
      m |= (qmin > thr_m) ? 0u : (1u << (by * 4 + bx));                         // NaN compares false -> hit
    }
  }
  return m;
}

// t / d for t < 2^24 and a quotient below 2^21 (tile ids, Gaussian indices inside a launch): an fp32 estimate with rd = 1 / d and
// one correction step — the integer division the compiler emits for a run-time divisor costs ~40 instructions.
__device__ __forceinline__ uint32_t gh_div_small(uint32_t t, uint32_t d, float rd) {
  uint32_t q = (uint32_t)((float)t * rd);
  const uint32_t r = t - q * d;                        // wraps below zero when the estimate is one too large
  if ((int32_t)r < 0) q -= 1u; else if (r >= d) q += 1u;
  return q;
}

// GH_FLAG_STATIC_LISTS: the opacity a Gaussian's tiles are culled with when the lists must outlive the call's opacities
// (gh_forward_refresh) — at least 2, and twice the current value. The alpha >= 1/255 ellipse grows with sqrt(ln(255 o)):
// doubling an opacity of 1 widens it by 6 %, so the bound is cheap (a few per cent more instances) and a fit's opacity
// bias has to double an opacity — or push it past 2 — before the guard asks for new lists.
__device__ __forceinline__ float gh_static_cull_opacity(float op) { return fmaxf(2.0f, 2.0f * op); }

// wave64 ballot of a predicate, straight from the compare (HIP's __ballot(int) goes through a 0/1 integer first)
__device__ __forceinline__ uint64_t gh_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// ---- wave64 cross-lane helpers (DPP; no LDS traffic) ------------------------------------------------
// DPP controls (gfx9 encoding): quad_perm 0x00-0xFF, row_shr:n 0x110+n, row_ror:n 0x120+n,
// row_mirror 0x140, row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float gh_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true));
}

// Sum over all 64 lanes; every lane of the wave must be active. Result is valid in lane 63.
__device__ __forceinline__ float gh_wave_sum_to63(float v) {
  v += gh_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
  v += gh_dpp<0x4E>(v);          // quad_perm [2,3,0,1]
  v += gh_dpp<0x141>(v);         // row_half_mirror: 8-lane sums
  v += gh_dpp<0x140>(v);         // row_mirror: 16-lane row sums in every lane
  v += gh_dpp<0x142, 0xA>(v);    // row_bcast15 -> rows 1,3
  v += gh_dpp<0x143, 0xC>(v);    // row_bcast31 -> rows 2,3
  return v;
}

// Exclusive scan of one value per thread over the block (thread order); returns the prefix, *total = block sum.
// s_w: GH_BLOCK / GH_WAVE words of LDS. Contains two barriers.
__device__ __forceinline__ uint32_t gh_block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
  __syncthreads();                                     // s_w may still be read by a previous scan
  if (lane == 63) s_w[wid] = x;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) { const uint32_t t = s_w[w]; if (w < wid) woff += t; tot += t; }
  *total = tot;
  return woff + x - v;
}


// The forward's launch order of one view's tiles: order[rank * NV + v] = global tile id, heaviest first (a counting sort over 256
// buckets by one workgroup of GH_BLOCK threads; ties in no particular order — placement only, never results).
// The key: the list length (ranges given), or — launches small enough for the fine-grained forward — what the previous forward over
// this workspace measured per tile (`heavy`: the most entries one 4x4-pixel block let through = its longest wave's work; a list of 544
// entries can keep a wave busier than one of 1,265). With `heavy` and NO ranges (the ranking runs as spare workgroups of the projection
// kernel, before this call's lists exist) the measurement alone decides: a tile never measured goes last. Stale values (another scene,
// the first call) only cost time. `heavy` is cleared behind the read: the forward of THIS call fills it again — also when the
// caller has said the hint is stale (use_heavy = false, GH_FLAG_FRESH_ORDER): the list lengths decide then, as before the hints.
__device__ __forceinline__ void gh_rank_tiles(const uint2* __restrict__ ranges, int tiles, int NV, int v, uint32_t* __restrict__ order,
                                              uint32_t* __restrict__ heavy, bool use_heavy = true) {
  __shared__ uint32_t s_cnt[256];
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  const int tid = threadIdx.x;
  if (ranges) ranges += (size_t)v * tiles;
  if (heavy) heavy += (size_t)v * tiles;
  auto bucket_of = [&](int t) {                         // bucket 255 = heaviest
    uint32_t b = 0u;
    bool listed = true;
    if (ranges) { const uint2 r = ranges[t]; b = (r.y - r.x + 15u) >> 4; listed = r.y != r.x; }
    if (heavy && use_heavy) { const uint32_t h = heavy[t]; if (h != 0u && listed) b = 128u + ((h + 7u) >> 3); else b = b > 127u ? 127u : b; }
    return b > 255u ? 255u : b;
  };
  // (a thread's first four tiles keep their bucket in registers between the two passes: every launch the heaviness key applies to)
  uint32_t bk[4];
  s_cnt[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tid + j * GH_BLOCK;
    if (t < tiles) { bk[j] = bucket_of(t); atomicAdd(&s_cnt[255u - bk[j]], 1u); }
  }
  for (int t = tid + 4 * GH_BLOCK; t < tiles; t += GH_BLOCK) atomicAdd(&s_cnt[255u - bucket_of(t)], 1u);
  __syncthreads();
  uint32_t total;
  const uint32_t c = s_cnt[tid];
  const uint32_t pre = gh_block_excl_scan(c, s_w, &total);
  __syncthreads();
  s_cnt[tid] = pre;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = tid + j * GH_BLOCK;
    if (t < tiles) {
      const uint32_t rank = atomicAdd(&s_cnt[255u - bk[j]], 1u);
      order[(size_t)rank * NV + v] = (uint32_t)(v * tiles + t);
      if (heavy) heavy[t] = 0u;
    }
  }
  for (int t = tid + 4 * GH_BLOCK; t < tiles; t += GH_BLOCK) {
    const uint32_t rank = atomicAdd(&s_cnt[255u - bucket_of(t)], 1u);
    order[(size_t)rank * NV + v] = (uint32_t)(v * tiles + t);
    if (heavy) heavy[t] = 0u;
  }
}

__device__ __forceinline__ unsigned gh_wave_sum_u32(unsigned v) {  // all lanes get the total
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#ifdef GH_EXP_WG_TIME
// (experiment: per-workgroup cycle counters of any kernel of this translation unit, read back by gh_exp_wg_read_<TU>; kernels declare
//  `GhWgTimer wg_timer(<id>);` as their first statement. tools/experiments/wg_lpt_sim.py)
#define GH_WG_TIMER_TU(NAME) \
  __device__ uint4 gh_exp_wg_buf_##NAME[1 << 19]; __device__ uint32_t gh_exp_wg_n_##NAME; \
  struct GhWgTimer { uint32_t id; uint64_t t0; \
    __device__ GhWgTimer(uint32_t k) : id(k), t0(__builtin_readcyclecounter()) {} \
    __device__ ~GhWgTimer() { const uint64_t t1 = __builtin_readcyclecounter(); \
      if (threadIdx.x == 0) { const uint32_t n = atomicAdd(&gh_exp_wg_n_##NAME, 1u); \
        if (n < (1u << 19)) gh_exp_wg_buf_##NAME[n] = make_uint4((uint32_t)(t1 - t0), blockIdx.x, id, (uint32_t)(t0 >> 4)); } } }; \
  extern "C" int gh_exp_wg_read_##NAME(void* dst, size_t bytes, uint32_t* n) { \
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(gh_exp_wg_n_##NAME), 4) != hipSuccess) return -1; \
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gh_exp_wg_buf_##NAME), bytes); } \
  extern "C" int gh_exp_wg_clear_##NAME() { uint32_t z = 0; return (int)hipMemcpyToSymbol(HIP_SYMBOL(gh_exp_wg_n_##NAME), &z, 4); }
#define GH_WG_TIMER(ID) GhWgTimer wg_timer(ID)
#else
#define GH_WG_TIMER_TU(NAME)
#define GH_WG_TIMER(ID)
#endif
#endif  // __HIPCC__
