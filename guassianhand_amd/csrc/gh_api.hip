// gh_api.hip — the C-ABI entry points of include/gh_raster.h: workspace layout + kernel sequencing.
// No allocation, no synchronisation, no host read-back: everything is enqueued on the caller's stream.
#include "gh_internal.h"

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

static int check_dims(const GhDims* d) {
  if (!d) return GH_ERR_INVALID_ARG;
  if (d->P < 0 || d->n_views < 1 || d->H < 1 || d->W < 1 || d->max_instances < 0) return GH_ERR_INVALID_ARG;
  if (d->sh_degree < 0 || d->sh_degree > 3) return GH_ERR_UNSUPPORTED;
  if (d->M != 0 && d->M != 1 && d->M != 4 && d->M != 9 && d->M != 16) return GH_ERR_UNSUPPORTED;
  int gx = (d->W + GH_TILE - 1) / GH_TILE, gy = (d->H + GH_TILE - 1) / GH_TILE;
  if (gx > 255 || gy > 255) return GH_ERR_UNSUPPORTED;                 // rect packs tile coords in 8 bits
  if ((long long)d->n_views * d->P >= (1ll << 31)) return GH_ERR_UNSUPPORTED;
  if (d->max_instances >= (1ll << 32)) return GH_ERR_UNSUPPORTED;      // uint32 slots
  if ((long long)gx * gy * d->n_views >= (1ll << 31)) return GH_ERR_UNSUPPORTED;
  return GH_OK;
}

extern "C" int gh_version(void) { return (GH_VERSION_MAJOR << 16) | GH_VERSION_MINOR; }

extern "C" int gh_workspace_layout(const GhDims* d, GhLayout* L) {
  int rc = check_dims(d);
  if (rc != GH_OK || !L) return rc != GH_OK ? rc : GH_ERR_INVALID_ARG;
  GhGrid g = gh_make_grid(d);
  size_t N = (size_t)g.N, cap = (size_t)g.cap, pix = (size_t)g.NV * g.H * g.W;
  size_t nblk_pre = (N + GH_BLOCK - 1) / GH_BLOCK;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  L->counters = take(sizeof(GhCounters));
  L->geom = take(N * 64);
  L->depth = take(N * 4);
  L->rect = take(N * 4);
  L->clamped = take(N);
  L->tiles_touched = take(N * 4);
  L->slot_begin = take(N * 4);
  L->depth_keys_a = take(N * 4);
  L->depth_keys_b = take(N * 4);
  L->depth_vals_a = take(N * 4);
  L->depth_vals_b = take(N * 4);
  L->block_sums = take((nblk_pre + 1) * 4);
  L->keys_a = take(cap * 4);
  L->keys_b = take(cap * 4);
  L->vals_a = take(cap * 4);
  L->vals_b = take(cap * 4);
  L->sorted_slot = take(cap * 4);
  L->inst_r0 = take(cap * 16);
  L->inst_r1 = take(cap * 16);
  L->inst_r2 = take(cap * 8);
  const size_t tab_n = gh_radix_table_words((size_t)g.P, g.NV);          // per-view depth sort: NV segments of P keys
  const size_t tab_d = gh_radix_table_words((size_t)g.cap);
  L->sort_tables = take((tab_n > tab_d ? tab_n : tab_d) * 4);
  L->ranges = take((size_t)g.NV * g.tiles * 8);
  L->tile_walk = take((size_t)g.NV * g.tiles * 8);       // walked entries [T] + completion counters [T]; directly after
                                                         // ranges: all cleared by one memset when there is nothing to project
  L->tile_order = take((size_t)g.NV * g.tiles * 4);
  const size_t n_items = (size_t)g.NV * g.tiles + cap / GH_SEGMENT + 2;       // backward work items / checkpoint slots
  L->bwd_items = take(n_items * 8);
  L->ckpt_rgb = take(n_items * 256 * 16);
  L->final_C = take(pix * 16);
  L->final_T = take(pix * 4);
  L->n_contrib = take(pix * 4);
  L->inst_grad = take(cap * 4 * GH_REC_G * 4);
  L->inst_flag = take(cap * 4);
  const bool sh_mode = d->M != 0;
  L->sh_rgb = take(sh_mode ? N * 16 : 0);
  L->dmean_sh = take(sh_mode ? N * 16 : 0);
  L->sh_scratch = take(sh_mode ? ((N * 16 + GH_BLOCK - 1) / GH_BLOCK + 1) * 64 * 4 : 0);   // sized for the pose-batch row count
  L->grad_sums = take(N * 48);
  L->bwd_scratch = take((2 * nblk_pre + 2) * 64 * 4);     // per-block partials of the chain-rule kernel (<= 2N lanes)
  const size_t nblk_proj = (((N > (size_t)g.NV * g.tiles ? N : (size_t)g.NV * g.tiles)) + GH_BLOCK - 1) / GH_BLOCK;
  L->key_bits = take((nblk_proj + 1) * 8);                // (OR, AND) of the visible depth keys per projection block
  L->total_bytes = off;
  return GH_OK;
}

extern "C" size_t gh_workspace_bytes(const GhDims* d) {
  GhLayout L;
  if (gh_workspace_layout(d, &L) != GH_OK) return 0;
  return L.total_bytes;
}

static int check_inputs(const GhDims* d, const GhInputs* in) {
  if (!in || !in->cams) return GH_ERR_INVALID_ARG;
  if (d->P == 0) return GH_OK;                      // nothing to read: only the background is composited
  if (d->P > 0 && (!in->means3D || !in->opacities || !in->scales || !in->rotations)) return GH_ERR_INVALID_ARG;
  if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) return GH_ERR_INVALID_ARG;  // exactly one
  if (in->shs && d->M == 0) return GH_ERR_INVALID_ARG;
  if (in->colors_precomp && d->M != 0) return GH_ERR_INVALID_ARG;
  if (in->blend_color_b && !in->blend_color_w && in->shs) return GH_ERR_INVALID_ARG;      // SH: b needs w (:334)
  if ((d->flags & GH_FLAG_BLEND_COLOR_B_RGB) && !in->colors_precomp) return GH_ERR_INVALID_ARG;   // (P,3) biases: RGB mode only
  if (in->shs && (in->blend_color_w || in->blend_color_b) && d->M != 16) return GH_ERR_INVALID_ARG;
  return GH_OK;
}

extern "C" int gh_forward(const GhDims* d, const GhInputs* in, const GhOutputs* out, void* workspace,
                          size_t ws_bytes, void* hip_stream) {
  return gh_forward_stages(d, in, out, workspace, ws_bytes, hip_stream, GH_FWD_ALL);
}

extern "C" int gh_forward_stages(const GhDims* d, const GhInputs* in, const GhOutputs* out, void* workspace,
                                 size_t ws_bytes, void* hip_stream, uint32_t stages) {
  int rc = check_dims(d);
  if (rc != GH_OK) return rc;
  rc = check_inputs(d, in);
  if (rc != GH_OK) return rc;
  if (!out || !out->image || !workspace) return GH_ERR_INVALID_ARG;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  char* ws = (char*)workspace;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  if (stages & GH_FWD_PREPROCESS) {
    // the projection kernel also resets the counters and the per-tile state (ranges, tile_walk) of the later stages
    gh_launch_sh_colour_fwd(d, g, in, ws, L, s);
    gh_launch_preprocess_fwd(d, g, in, out->radii, ws, L, s);
  }
  if (stages & GH_FWD_BINNING) {
    gh_launch_binning(d, g, ws, L, s);
  }
  if (stages & GH_FWD_RENDER) gh_launch_render_fwd(d, g, in, out->image, out->alpha, ws, ws, L, s);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_backward(const GhDims* d, const GhInputs* in, const GhGrads* gr, void* workspace,
                           size_t ws_bytes, void* hip_stream) {
  return gh_backward_stages(d, in, gr, workspace, ws_bytes, hip_stream, GH_BWD_ALL);
}

extern "C" int gh_backward_stages(const GhDims* d, const GhInputs* in, const GhGrads* gr, void* workspace,
                                  size_t ws_bytes, void* hip_stream, uint32_t stages) {
  int rc = check_dims(d);
  if (rc != GH_OK) return rc;
  rc = check_inputs(d, in);
  if (rc != GH_OK) return rc;
  if (!gr || !gr->dL_dimage || !workspace) return GH_ERR_INVALID_ARG;
  // 48-wide gradient rows are written as float4s
  if ((((uintptr_t)gr->dL_dblend_color_b | (uintptr_t)gr->dL_dblend_color_w) & 15) != 0) return GH_ERR_INVALID_ARG;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  char* ws = (char*)workspace;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  if (stages & GH_BWD_RENDER) gh_launch_render_bwd(d, g, in, gr->dL_dimage, gr->dL_dalpha, gr->upstream_scale, ws, ws, L, s);
  if (stages & GH_BWD_PREPROCESS) gh_launch_preprocess_bwd(d, g, in, gr, ws, ws, L, s);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

// ---- second call over the same geometry --------------------------------------------------------------------------
static int check_shared(const GhDims* d, const GhInputs* in) {
  int rc = check_dims(d);
  if (rc != GH_OK) return rc;
  rc = check_inputs(d, in);
  if (rc != GH_OK) return rc;
  if (d->P > 0 && !in->colors_precomp) return GH_ERR_UNSUPPORTED;       // colours must be precomputed (no SH stage here)
  return GH_OK;
}

extern "C" int gh_forward_shared(const GhDims* d, const GhInputs* in, const GhOutputs* out, const void* geometry_ws,
                                 void* workspace, size_t ws_bytes, void* hip_stream) {
  int rc = check_shared(d, in);
  if (rc != GH_OK) return rc;
  if (!out || !out->image || !workspace || !geometry_ws || geometry_ws == workspace) return GH_ERR_INVALID_ARG;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  gh_launch_recolour(d, g, in, (const char*)geometry_ws, (char*)workspace, L, s);
  gh_launch_render_fwd(d, g, in, out->image, out->alpha, (const char*)geometry_ws, (char*)workspace, L, s);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_backward_shared(const GhDims* d, const GhInputs* in, const GhGrads* gr, const void* geometry_ws,
                                  void* workspace, size_t ws_bytes, void* hip_stream) {
  int rc = check_shared(d, in);
  if (rc != GH_OK) return rc;
  if (!gr || !gr->dL_dimage || !workspace || !geometry_ws || geometry_ws == workspace) return GH_ERR_INVALID_ARG;
  if ((((uintptr_t)gr->dL_dblend_color_b | (uintptr_t)gr->dL_dblend_color_w) & 15) != 0) return GH_ERR_INVALID_ARG;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  gh_launch_render_bwd(d, g, in, gr->dL_dimage, gr->dL_dalpha, gr->upstream_scale, (const char*)geometry_ws, (char*)workspace, L, s);
  gh_launch_preprocess_bwd(d, g, in, gr, (const char*)geometry_ws, (char*)workspace, L, s);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
