// gh_api.hip — the C-ABI entry points of include/gh_raster.h: workspace layout + kernel sequencing.
// No allocation, no synchronisation, no host read-back: everything is enqueued on the caller's stream.
#include "gh_internal.h"

#include <mutex>

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

static int check_dims(const GhDims* d) {
  if (!d) return GH_ERR_INVALID_ARG;
  if (d->abi != GH_ABI_TAG) return GH_ERR_ABI;          // a host built against another header: its structs may be shorter than ours
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

// ---- GH_FLAG_SPLIT_STREAMS: the two halves of the views -----------------------------------------------------------
static bool gh_split_on(const GhDims* d) { return (d->flags & GH_FLAG_SPLIT_STREAMS) && d->n_views >= 2 && d->P > 0; }

// Dims of half h (0 / 1): views [0, NV/2) / [NV/2, NV), a share of the instance capacity proportional to the views
// (multiple of 64 slots). *v0 / *cap0: first view / first instance slot of the half.
static GhDims gh_half_dims(const GhDims* d, int h, int* v0, size_t* cap0) {
  const int nva = d->n_views / 2;
  const size_t capa = ((size_t)d->max_instances * (size_t)nva / (size_t)d->n_views) & ~(size_t)63;
  GhDims o = *d;
  o.flags &= ~GH_FLAG_SPLIT_STREAMS;
  o.n_views = h ? d->n_views - nva : nva;
  o.max_instances = h ? (int64_t)((size_t)d->max_instances - capa) : (int64_t)capa;
  *v0 = h ? nva : 0;
  *cap0 = h ? capa : 0;
  return o;
}

static size_t gh_sort_table_words(const GhGrid& g) {
  const size_t tab_n = gh_radix_table_words((size_t)g.P, g.NV);          // per-view depth sort: NV segments of P keys
  const size_t tab_d = gh_radix_table_words((size_t)g.cap) + 2048 * (size_t)g.NV;     // (+ the per-view partition's extra blocks and totals: up to
                                                                                       //  1024 digits x (one more block + one row of totals) per view)
  return tab_n > tab_d ? tab_n : tab_d;
}

static size_t gh_proj_blocks(const GhGrid& g) {                          // grid of gh_preprocess_fwd_kernel
  const size_t N = (size_t)g.N, T = (size_t)g.NV * g.tiles;
  return ((N > T ? N : T) + GH_BLOCK - 1) / GH_BLOCK;
}

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
  L->clamped = take(N);
  L->tiles_touched = take(N * 4);
  L->slot_begin = take(N * 4);
  L->depth_keys_a = take(N * 4);
  L->depth_keys_b = take(N * 4);
  L->depth_vals_a = take(N * 4);
  L->depth_vals_b = take(N * 4);
  L->block_sums = take((nblk_pre + 3) * 4);               // (+2: the halves of a split call keep one spare entry each)
  L->keys_a = take(cap * 4);
  L->keys_b = take(cap * 4);
  L->vals_a = take(cap * 4);
  L->vals_b = take(cap * 4);
  L->sorted_slot = take(cap * 4);
  L->inst_r0 = take(cap * 16);
  L->inst_r1 = take(cap * 16);
  L->inst_r2 = take(cap * 8);
  size_t tab = gh_sort_table_words(g);
  if (gh_split_on(d)) {                                   // each half sorts with tables of its own
    int v0; size_t c0;
    const GhDims da = gh_half_dims(d, 0, &v0, &c0), db = gh_half_dims(d, 1, &v0, &c0);
    const size_t both = gh_sort_table_words(gh_make_grid(&da)) + gh_sort_table_words(gh_make_grid(&db));
    if (both > tab) tab = both;
  }
  L->sort_tables = take(tab * 4);
  L->ranges = take((size_t)g.NV * g.tiles * 8);
  L->tile_walk = take((size_t)g.NV * g.tiles * (4 + GH_BWD_COST_SLOTS) * 4);      // walked entries [T] + completion counters [T] + stop positions [T] + heaviness of the previous FINE launch [T] (never cleared);
                                                         // directly after ranges: all cleared by one memset when there is
                                                         // nothing to project
  L->tile_order = take((size_t)g.NV * g.tiles * 4);
  const size_t n_items = (size_t)g.NV * g.tiles + cap / GH_SEGMENT + 4;       // backward work items / checkpoint slots (2 + 2 spare per half)
  L->bwd_items = take(n_items * 8 * GH_BWD_CLASSES);        // (one region of the list's full capacity per cost class)
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
  L->cull_bound = take(N * 4);                            // (a refresh call reads these two of the BUILD's workspace with its OWN
  L->inst_c = take(cap * 4);                              //  layout: both calls must agree on M == 0 / M != 0, see gh_forward_refresh)
  L->attr = take(N * 16);
  L->half_counters = take(512);
  L->key_bits = take((gh_proj_blocks(g) + 4) * 8);        // (OR, AND) of the visible depth keys per projection block (+1 word; two
                                                          // halves: + 1 block of rounding + 1 word each)
  L->tile_bound = take((size_t)g.NV * g.tiles * 4);
  L->block_tiles = take((gh_proj_blocks(g) + 4) * 4);     // (two halves: + 1 block of rounding + 1 spare word each)
  L->render_guard = take(512);                            // one word (+ a second, 256 bytes on, for the other half of a split call)
  L->loss_partials = take(gh_loss_partial_count(g) * 4 + 16);   // (+ the factor of the final sum, behind the last partial)
  L->view_start = take(((size_t)g.NV + 2) * 4);           // (a split call: each half's n_views + 1 bounds, the second half's behind the first's)
  L->total_bytes = off;
  return GH_OK;
}

extern "C" size_t gh_workspace_bytes(const GhDims* d) {
  GhLayout L;
  if (gh_workspace_layout(d, &L) != GH_OK) return 0;
  return L.total_bytes;
}

extern "C" int gh_partition_is_per_view(const GhDims* d) {
  const int rc = check_dims(d);
  if (rc != GH_OK) return rc;
  if (gh_split_on(d)) {                                  // the halves decide for themselves; a caller that inspects them asks per half
    int v0; size_t c0;
    const GhDims da = gh_half_dims(d, 0, &v0, &c0);
    return gh_partition_per_view(gh_make_grid(&da)) ? 1 : 0;
  }
  return gh_partition_per_view(gh_make_grid(d)) ? 1 : 0;
}

static int check_inputs(const GhDims* d, const GhInputs* in) {
  if (!in || !in->cams) return GH_ERR_INVALID_ARG;
  if (d->P == 0) return GH_OK;                      // nothing to read: only the background is composited
  if (d->P > 0 && (!in->means3D || !in->opacities)) return GH_ERR_INVALID_ARG;
  // exactly one of {scales AND rotations, cov3D_precomp} (the published wrapper's second validation, App. A.0)
  if ((in->scales != nullptr) != (in->rotations != nullptr)) return GH_ERR_INVALID_ARG;
  if ((in->scales != nullptr) == (in->cov3D_precomp != nullptr)) return GH_ERR_INVALID_ARG;
  if ((in->shs != nullptr) == (in->colors_precomp != nullptr)) return GH_ERR_INVALID_ARG;  // exactly one
  if (in->shs && d->M == 0) return GH_ERR_INVALID_ARG;
  if (in->colors_precomp && d->M != 0) return GH_ERR_INVALID_ARG;
  if (in->blend_color_b && !in->blend_color_w && in->shs) return GH_ERR_INVALID_ARG;      // SH: b needs w (:334)
  if ((d->flags & GH_FLAG_BLEND_COLOR_B_RGB) && !in->colors_precomp) return GH_ERR_INVALID_ARG;   // (P,3) biases: RGB mode only
  if (in->shs && (in->blend_color_w || in->blend_color_b) && d->M != 16) return GH_ERR_INVALID_ARG;
  return GH_OK;
}

// The half's view of the workspace: every array that is indexed by (view, Gaussian), tile or pixel keeps its place in the
// whole call's array (view-major), the per-instance arrays of the second half start at its first slot, and the few arrays
// with another shape (scan scratch, sort tables, work list, key bits, counters) are cut in two.
struct GhHalf {
  GhDims d; GhGrid g; GhLayout L; GhInputs in;
  int v0;
};

static void gh_make_halves(const GhDims* d, const GhLayout& L, const GhInputs* in, GhHalf hv[2]) {
  const GhGrid gf = gh_make_grid(d);
  const bool per_view = (d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;
  const bool sh_mode = d->M != 0;
  size_t blk_a = 0, items_a = 0, tab_a = 0, proj_a = 0;
  for (int h = 0; h < 2; ++h) {
    GhHalf& H = hv[h];
    size_t cap0;
    H.d = gh_half_dims(d, h, &H.v0, &cap0);
    H.g = gh_make_grid(&H.d);
    H.g.total_tiles = gf.total_tiles;
    const size_t n0 = (size_t)H.v0 * gf.P, t0 = (size_t)H.v0 * gf.tiles, p0 = (size_t)H.v0 * gf.H * gf.W;
    GhLayout& o = H.L;
    o = L;
    o.counters = L.half_counters + (size_t)h * 256;
    o.geom += n0 * 64; o.clamped += n0; o.tiles_touched += n0 * 4; o.slot_begin += n0 * 4;
    o.depth_keys_a += n0 * 4; o.depth_keys_b += n0 * 4; o.depth_vals_a += n0 * 4; o.depth_vals_b += n0 * 4;
    o.block_sums += h ? (blk_a + 1) * 4 : 0;
    o.keys_a += cap0 * 4; o.keys_b += cap0 * 4; o.vals_a += cap0 * 4; o.vals_b += cap0 * 4; o.sorted_slot += cap0 * 4;
    o.inst_r0 += cap0 * 16; o.inst_r1 += cap0 * 16; o.inst_r2 += cap0 * 8;
    o.sort_tables += h ? tab_a * 4 : 0;
    o.ranges += t0 * 8; o.tile_walk += t0 * (4 + GH_BWD_COST_SLOTS) * 4; o.tile_order += t0 * 4;
    o.bwd_items += h ? items_a * 8 * GH_BWD_CLASSES : 0; o.ckpt_rgb += h ? items_a * 256 * 16 : 0;
    o.final_C += p0 * 16; o.final_T += p0 * 4; o.n_contrib += p0 * 4;
    o.inst_grad += cap0 * 4 * GH_REC_G * 4; o.inst_flag += cap0 * 4;
    if (sh_mode) { o.sh_rgb += n0 * 16; o.dmean_sh += n0 * 16; }
    o.grad_sums += n0 * 48;
    o.cull_bound += n0 * 4; o.inst_c += cap0 * 4; o.attr += n0 * 16;
    o.key_bits += h ? (proj_a + 1) * 8 : 0;
    o.block_tiles += h ? (proj_a + 1) * 4 : 0;
    o.render_guard += (size_t)h * 256;
    o.view_start += h ? ((size_t)hv[0].d.n_views + 1) * 4 : 0;
    o.tile_bound += t0 * 4;
    if (h == 0) {
      blk_a = ((size_t)H.g.N + GH_BLOCK - 1) / GH_BLOCK; items_a = (size_t)H.g.n_items; tab_a = gh_sort_table_words(H.g);
      proj_a = gh_proj_blocks(H.g);
    }
    H.in = *in;
    H.in.cams = in->cams + (size_t)H.v0 * GH_CAM_FLOATS;
    if (in->tile_depth_bound) H.in.tile_depth_bound = in->tile_depth_bound + 2 * t0;     // (depth, block mask) pairs
    if (per_view) {                                       // pose batch: the half's own rows of every per-Gaussian array
      const size_t r0 = n0;
      H.in.means3D = in->means3D + r0 * 3; H.in.opacities = in->opacities + r0;
      if (in->scales) { H.in.scales = in->scales + r0 * 3; H.in.rotations = in->rotations + r0 * 4; }
      if (in->cov3D_precomp) H.in.cov3D_precomp = in->cov3D_precomp + r0 * 6;
      if (in->shs) H.in.shs = in->shs + r0 * (size_t)d->M * 3;
      if (in->colors_precomp) H.in.colors_precomp = in->colors_precomp + r0 * 3;
      if (in->blend_opacity_b) H.in.blend_opacity_b = in->blend_opacity_b + r0;
      if (in->blend_color_w && (d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN)) H.in.blend_color_w = in->blend_color_w + r0 * 48;
      if (in->blend_color_b) H.in.blend_color_b = in->blend_color_b + r0 * ((d->flags & GH_FLAG_BLEND_COLOR_B_RGB) ? 3 : 48);
    }
  }
}

// The library's own stream and fork / join events, one set per device, created on first use.
// `call` is held for the whole of a split call: the events are re-recorded by every call, so two host threads must not
// interleave their fork / join pairs (split calls of one device are enqueued one after the other; they still overlap on the GPU).
struct GhSide { hipStream_t s2 = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false; std::mutex call; };
static GhSide* gh_side() {
  static GhSide sides[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  GhSide& S = sides[dev];
  if (!S.ok) {
    if (hipStreamCreateWithFlags(&S.s2, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&S.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&S.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    S.ok = true;
  }
  return &S;
}
static bool gh_fork(GhSide* S, hipStream_t s) {
  return hipEventRecord(S->fork, s) == hipSuccess && hipStreamWaitEvent(S->s2, S->fork, 0) == hipSuccess;
}
static bool gh_join(GhSide* S, hipStream_t s) {
  return hipEventRecord(S->join, S->s2) == hipSuccess && hipStreamWaitEvent(s, S->join, 0) == hipSuccess;
}

// Counters of the whole call from the halves': D = sum, overflow = either, reserved[0] = the max_instances that would have
// given every half a share large enough.
__global__ void gh_merge_counters_kernel(const GhCounters* __restrict__ a, const GhCounters* __restrict__ b,
                                         GhCounters* __restrict__ out, uint32_t nv, uint32_t nva) {
  const unsigned long long da = a->num_rendered, db = b->num_rendered, nvb = nv - nva;
  unsigned long long need_a = (da * nv + nva - 1) / nva + 64ull * nv, need_b = (db * nv + nvb - 1) / nvb + 64ull * nv;
  unsigned long long need = need_a > need_b ? need_a : need_b;
  if (need > 0xFFFFFFFFull) need = 0xFFFFFFFFull;
  const unsigned long long tot = da + db;
  out->num_rendered = tot > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)tot;
  // errors: either half's (bit 0: a half exceeded its share; bit 2: depth-bound miss; ...); the depth-key information bit: both halves'
  out->overflow = ((a->overflow | b->overflow) & GH_COUNTER_ERROR_MASK) | (a->overflow & b->overflow & GH_COUNTER_DEPTH24_OK);
  out->reserved[0] = (uint32_t)need;
  out->reserved[1] = 0u;
}

static inline bool gh_ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
  const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
  return x < y + nb && y < x + na;
}

// GhOutputs.l1_* / fit_loss: the image loss from the render kernel's epilogue
static int check_fused_loss(const GhDims* d, const GhInputs* in, const GhOutputs* out) {
  if (!out->l1_target && !out->fit_loss) return GH_OK;
  if (out->l1_target && out->fit_loss) return GH_ERR_INVALID_ARG;
  // one walk, one loss: not with an occlusion bound / report (a miss found late could not take back the gradients of the waves
  // that finished early) or two halves on two streams
  if (out->tile_depth_seen || in->tile_depth_bound || gh_split_on(d)) return GH_ERR_UNSUPPORTED;
  if (out->l1_target) {
    if (!out->l1_dL_dimage || !out->l1_loss) return GH_ERR_INVALID_ARG;
    // the gradient is stored while other waves still read the target and store the image: three arrays whose byte ranges are disjoint
    const size_t nb = (size_t)d->n_views * 3 * (size_t)d->H * (size_t)d->W * 4;
    if (gh_ranges_overlap(out->l1_dL_dimage, nb, out->image, nb) || gh_ranges_overlap(out->l1_dL_dimage, nb, out->l1_target, nb) ||
        gh_ranges_overlap(out->image, nb, out->l1_target, nb))
      return GH_ERR_INVALID_ARG;
    if (out->alpha) return GH_ERR_UNSUPPORTED;             // (the L1 form has no mask channel)
    return GH_OK;
  }
  const GhFitLoss* f = out->fit_loss;
  if (!f->gt_rgb || !f->gt_mask || !f->dL_dimage || !f->dL_dalpha || !f->loss) return GH_ERR_INVALID_ARG;
  if (!out->alpha) return GH_ERR_INVALID_ARG;              // the mask term reads the fused mask channel
  {
    const size_t nb3 = (size_t)d->n_views * 3 * (size_t)d->H * (size_t)d->W * 4, nb1 = nb3 / 3;
    const void* arr[6] = {out->image, out->alpha, f->dL_dimage, f->dL_dalpha, f->gt_rgb, f->gt_mask};
    const size_t len[6] = {nb3, nb1, nb3, nb1, nb3, nb1};
    for (int a = 2; a < 4; ++a)                              // each gradient array against every other array of the call
      for (int b = 0; b < 6; ++b)
        if (a != b && gh_ranges_overlap(arr[a], len[a], arr[b], len[b])) return GH_ERR_INVALID_ARG;
    if (f->bbox && (gh_ranges_overlap(f->bbox, nb1, f->dL_dimage, nb3) || gh_ranges_overlap(f->bbox, nb1, f->dL_dalpha, nb1))) return GH_ERR_INVALID_ARG;
  }
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
  if (in->tile_depth_bound && (const float*)out->tile_depth_seen == in->tile_depth_bound) return GH_ERR_INVALID_ARG;
  // lists that must outlive this call's opacities cannot be truncated by a bound that holds for this call only
  if (in->tile_depth_bound && (d->flags & GH_FLAG_STATIC_LISTS)) return GH_ERR_UNSUPPORTED;
  rc = check_fused_loss(d, in, out);
  if (rc != GH_OK) return rc;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  char* ws = (char*)workspace;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  if (gh_split_on(d)) {
    GhSide* S = gh_side();
    if (!S) return GH_ERR_LAUNCH;
    std::lock_guard<std::mutex> one_call(S->call);
    GhHalf hv[2];
    gh_make_halves(d, L, in, hv);
    if (!gh_fork(S, s)) return GH_ERR_LAUNCH;
    for (int h = 0; h < 2; ++h) {
      const GhHalf& H = hv[h];
      hipStream_t sh = h ? S->s2 : s;
      const size_t pix0 = (size_t)H.v0 * g.H * g.W;
      if (stages & GH_FWD_PREPROCESS) {
        gh_launch_sh_colour_fwd(&H.d, H.g, &H.in, ws, H.L, sh);
        gh_launch_preprocess_fwd(&H.d, H.g, &H.in, out->radii ? out->radii + (size_t)H.v0 * g.P : nullptr, ws, H.L, sh);
      }
      if (stages & GH_FWD_BINNING) gh_launch_binning(&H.d, H.g, ws, H.L, sh, H.in.tile_depth_bound ? (const float*)(ws + H.L.tile_bound) : nullptr);
      if (stages & GH_FWD_RENDER)
        gh_launch_render_fwd(&H.d, H.g, &H.in, out->image + pix0 * 3, out->alpha ? out->alpha + pix0 : nullptr, ws, ws, H.L, sh,
                             out->tile_depth_seen ? out->tile_depth_seen + (size_t)2 * H.v0 * g.tiles : nullptr, out->tile_depth_seen_scale,
                             out->tile_depth_seen_slack);
    }
    if (!gh_join(S, s)) return GH_ERR_LAUNCH;
    hipLaunchKernelGGL(gh_merge_counters_kernel, dim3(1), dim3(1), 0, s, (const GhCounters*)(ws + hv[0].L.counters),
                       (const GhCounters*)(ws + hv[1].L.counters), (GhCounters*)(ws + L.counters), (uint32_t)d->n_views,
                       (uint32_t)hv[0].d.n_views);
    return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
  }
  if (stages & GH_FWD_PREPROCESS) {
    // the projection kernel also resets the counters and the per-tile state (ranges, tile_walk) of the later stages
    gh_launch_sh_colour_fwd(d, g, in, ws, L, s);
    gh_launch_preprocess_fwd(d, g, in, out->radii, ws, L, s);
  }
  if (stages & GH_FWD_BINNING) {
    gh_launch_binning(d, g, ws, L, s, in->tile_depth_bound ? (const float*)(ws + L.tile_bound) : nullptr);
  }
  if (stages & GH_FWD_RENDER)
    gh_launch_render_fwd(d, g, in, out->image, out->alpha, ws, ws, L, s, out->tile_depth_seen, out->tile_depth_seen_scale,
                         out->tile_depth_seen_slack, out);
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
  if (gr->deferred_loss && gh_split_on(d)) return GH_ERR_UNSUPPORTED;      // (a split forward fuses no loss)
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  char* ws = (char*)workspace;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  if (gh_split_on(d)) {
    // the halves walk their lists and sum their sub-records side by side; the chain rule then runs once over all views
    // (every per-(view, Gaussian) sum sits where the unsplit call puts it)
    GhSide* S = gh_side();
    if (!S) return GH_ERR_LAUNCH;
    std::lock_guard<std::mutex> one_call(S->call);
    GhHalf hv[2];
    gh_make_halves(d, L, in, hv);
    if (!gh_fork(S, s)) return GH_ERR_LAUNCH;
    for (int h = 0; h < 2; ++h) {
      const GhHalf& H = hv[h];
      hipStream_t sh = h ? S->s2 : s;
      const size_t pix0 = (size_t)H.v0 * g.H * g.W;
      if (stages & GH_BWD_RENDER)
        gh_launch_render_bwd(&H.d, H.g, &H.in, gr->dL_dimage + pix0 * 3, gr->dL_dalpha ? gr->dL_dalpha + pix0 : nullptr,
                             gr->upstream_scale, ws, ws, H.L, sh, gh_records_need_geometry(in, gr));
      if (stages & GH_BWD_PREPROCESS) gh_launch_preprocess_bwd(&H.d, H.g, &H.in, gr, ws, ws, H.L, sh, GH_PBWD_RECORD_SUM);
    }
    if (!gh_join(S, s)) return GH_ERR_LAUNCH;
    if (stages & GH_BWD_PREPROCESS)
      gh_launch_preprocess_bwd(d, g, in, gr, ws, ws, L, s, GH_PBWD_CHAIN, hv[1].v0, (size_t)hv[0].d.max_instances);
    return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
  }
  if (stages & GH_BWD_RENDER)
    gh_launch_render_bwd(d, g, in, gr->dL_dimage, gr->dL_dalpha, gr->upstream_scale, ws, ws, L, s, gh_records_need_geometry(in, gr),
                         gr->deferred_loss);
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
  if (d->flags & GH_FLAG_SPLIT_STREAMS) return GH_ERR_UNSUPPORTED;
  return GH_OK;
}

extern "C" int gh_forward_shared(const GhDims* d, const GhInputs* in, const GhOutputs* out, const void* geometry_ws,
                                 void* workspace, size_t ws_bytes, void* hip_stream) {
  int rc = check_shared(d, in);
  if (rc != GH_OK) return rc;
  if (!out || !out->image || !workspace || !geometry_ws || geometry_ws == workspace) return GH_ERR_INVALID_ARG;
  if (out->l1_target || out->fit_loss) return GH_ERR_UNSUPPORTED;            // the fused image losses are gh_forward's / gh_forward_refresh's
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
  if (gr->deferred_loss) return GH_ERR_UNSUPPORTED;                       // (gh_forward_shared fuses no loss)
  if ((((uintptr_t)gr->dL_dblend_color_b | (uintptr_t)gr->dL_dblend_color_w) & 15) != 0) return GH_ERR_INVALID_ARG;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  gh_launch_render_bwd(d, g, in, gr->dL_dimage, gr->dL_dalpha, gr->upstream_scale, (const char*)geometry_ws, (char*)workspace, L, s,
                       gh_records_need_geometry(in, gr));
  gh_launch_preprocess_bwd(d, g, in, gr, (const char*)geometry_ws, (char*)workspace, L, s);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

// ---- a later step over static geometry ------------------------------------------------------------------------------
static int check_refresh(const GhDims* d, const GhInputs* in) {
  int rc = check_dims(d);
  if (rc != GH_OK) return rc;
  rc = check_inputs(d, in);
  if (rc != GH_OK) return rc;
  if (d->flags & GH_FLAG_SPLIT_STREAMS) return GH_ERR_UNSUPPORTED;
  if (!(d->flags & GH_FLAG_STATIC_LISTS)) return GH_ERR_INVALID_ARG;    // the lists must have been built for re-use
  return GH_OK;
}

extern "C" int gh_forward_refresh(const GhDims* d, const GhInputs* in, const GhOutputs* out, const void* geometry_ws,
                                  void* workspace, size_t ws_bytes, void* hip_stream) {
  int rc = check_refresh(d, in);
  if (rc != GH_OK) return rc;
  if (!out || !out->image || !workspace || !geometry_ws || geometry_ws == workspace) return GH_ERR_INVALID_ARG;
  rc = check_fused_loss(d, in, out);
  if (rc != GH_OK) return rc;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  if (g.N == 0) {                                        // nothing to draw: the background, through the plain path's state
    gh_launch_preprocess_fwd(d, g, in, nullptr, (char*)workspace, L, s);
    gh_launch_render_fwd(d, g, in, out->image, out->alpha, (const char*)geometry_ws, (char*)workspace, L, s, nullptr, 1.0f, 0u, out);
    return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
  }
  gh_launch_sh_colour_fwd(d, g, in, (char*)workspace, L, s);           // SH mode only
  const bool own_order = gh_launch_refresh(d, g, in, (const char*)geometry_ws, (char*)workspace, L, s);
  gh_launch_render_fwd(d, g, in, out->image, out->alpha, (const char*)geometry_ws, (char*)workspace, L, s, nullptr, 1.0f, 0u, out, own_order);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_backward_refresh(const GhDims* d, const GhInputs* in, const GhGrads* gr, const void* geometry_ws,
                                   void* workspace, size_t ws_bytes, void* hip_stream) {
  int rc = check_refresh(d, in);
  if (rc != GH_OK) return rc;
  if (!gr || !gr->dL_dimage || !workspace || !geometry_ws || geometry_ws == workspace) return GH_ERR_INVALID_ARG;
  if ((((uintptr_t)gr->dL_dblend_color_b | (uintptr_t)gr->dL_dblend_color_w) & 15) != 0) return GH_ERR_INVALID_ARG;
  GhLayout L;
  gh_workspace_layout(d, &L);
  if (ws_bytes < L.total_bytes) return GH_ERR_WORKSPACE_SMALL;
  hipStream_t s = (hipStream_t)hip_stream;
  GhGrid g = gh_make_grid(d);
  (void)hipGetLastError();
  gh_launch_render_bwd(d, g, in, gr->dL_dimage, gr->dL_dalpha, gr->upstream_scale, (const char*)geometry_ws, (char*)workspace, L, s,
                       gh_records_need_geometry(in, gr), gr->deferred_loss);
  gh_launch_preprocess_bwd(d, g, in, gr, (const char*)geometry_ws, (char*)workspace, L, s);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
