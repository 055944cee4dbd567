/*
 * gh_raster.h — C-ABI of the MI355X-native differentiable Gaussian-splatting rasteriser.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json `north_star`.
 * What it replaces in the reference (XuanHuang0/GuassianHand):
 *   - the third-party CUDA extension imported at tgs/models/renderer_one_shot.py:3
 *     (`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`)
 *     and called at tgs/models/renderer_one_shot.py:338-346 (RGB pass) and :372-379 (mask pass);
 *   - the per-view attribute blend of tgs/models/renderer_one_shot.py:298-334, which the
 *     kernels fuse into the per-Gaussian projection stage (optional: all blend pointers may be NULL).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / STL types.
 *   - Every buffer (inputs, outputs, gradients, workspace) is allocated by the caller; the library
 *     owns no device memory. All functions are re-entrant. The only process-wide state exists with
 *     GH_FLAG_SPLIT_STREAMS: one side stream + fork / join events per device, created on first use and
 *     guarded by a per-device mutex (split calls of one device are enqueued one after the other).
 *     Without the flag, no call touches anything but its arguments.
 *   - All device work is enqueued on the caller-supplied stream (and, with GH_FLAG_SPLIT_STREAMS, on the
 *     library's side stream forked from and joined back into it). The library never synchronises,
 *     never allocates and never copies to the host: every entry point is HIP-graph capturable.
 *     Data-dependent sizes (the number of tile instances D) stay on the device; the caller bounds
 *     them with `max_instances` and reads `GhCounters` back whenever it chooses.
 *   - Return value: 0 on success, negative GhStatus on error. Nothing throws across the boundary.
 *   - Matrices follow the reference's row-vector convention (renderer_one_shot.py:96,106):
 *     viewmatrix = w2c^T, projmatrix = (P * w2c)^T, both read as 16 consecutive floats, i.e.
 *     element [4*c + r] is M[r][c] of the column-vector matrix.
 *
 * The CPU oracle (oracle/gh_oracle.c) implements the same structs on host pointers; it is test
 * infrastructure only and is never linked into or called by this library.
 */
#ifndef GH_RASTER_H
#define GH_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GH_VERSION_MAJOR 0
#define GH_VERSION_MINOR 8

#define GH_TILE 16           /* tile edge in pixels (binning granularity; fixes which Gaussians a pixel sees) */
#define GH_CAM_FLOATS 40     /* floats per camera record, see GhCamera */
#define GH_MAX_SH_COEFFS 16  /* degree 3 */

typedef enum GhStatus {
  GH_OK = 0,
  GH_ERR_INVALID_ARG = -1,     /* NULL / inconsistent pointers, bad dims */
  GH_ERR_WORKSPACE_SMALL = -2, /* ws_bytes < gh_workspace_bytes(dims) */
  GH_ERR_LAUNCH = -3,          /* hipGetLastError() != hipSuccess after a launch */
  GH_ERR_UNSUPPORTED = -4,     /* e.g. image wider than 255 tiles, sh_degree > 3 */
  GH_ERR_ABI = -5              /* GhDims.abi != GH_ABI_TAG: the caller was built against another version of this header */
} GhStatus;

/* ABI handshake (v0.8). The structs below grow between 0.x minor versions (0.5 -> 0.7 added fields to GhInputs, GhOutputs and
   GhGrads), and a host compiled against an older header would hand the library structs that are too short. Every entry point
   that takes a GhDims therefore checks its FIRST field against the tag of the header the library was built from and returns
   GH_ERR_ABI before it reads anything else; a host built against a header older than 0.8 has `P` in that place, which never
   equals the tag. 0.x minor bumps require a recompile of the host. */
#define GH_ABI_TAG (0x47480000u | ((uint32_t)GH_VERSION_MAJOR << 8) | (uint32_t)GH_VERSION_MINOR)   /* 'G' 'H' major minor */

/* Flags for GhDims.flags */
#define GH_FLAG_NONE 0u
#define GH_SEGMENT 256 /* depth segment (list entries) of the backward: segments of a tile are processed by different workgroups */
#define GH_FLAG_BLEND_W_PER_GAUSSIAN 1u /* color_w is (P,48) instead of (48,) — renderer_one_shot_edit.py:489-500 */
#define GH_FLAG_PER_VIEW_GAUSSIANS 4u /* pose batch (the batch loop of GS3DRenderer.forward, renderer_one_shot.py:615-633): every
                                         per-Gaussian input / gradient array holds n_views*P rows and view v renders rows
                                         [v*P, (v+1)*P) only (its own pose); color_w (48,), xyz_b stay shared */
#define GH_FLAG_BLEND_COLOR_B_RGB 2u    /* colors_precomp mode only: blend_color_b and dL_dblend_color_b are (P,3), the
                                           three columns of the (P,48) view that renderer_one_shot.py:328 reads */
#define GH_FLAG_STATIC_LISTS 16u       /* this forward's tile lists will be re-used by gh_forward_refresh with OTHER opacities and
                                           colours (same means / scales / rotations / cameras): tile culling treats every
                                           opacity o as max(2, 2 o), so the lists hold every tile the Gaussian can reach while
                                           its opacity stays <= that bound. Images and gradients are those of the plain call
                                           (a tile that holds no contributing pixel changes nothing): only D is larger. */
#define GH_FLAG_DEPTH24 32u             /* the caller expects the depth keys of the Gaussians that emit instances to differ in their
                                           low 24 bits only (all visible depths inside one factor-4 range such as [0.5, 2) m — the top
                                           byte of the float, sign + 7 exponent bits, is the same): the per-view depth sort then runs
                                           three 8-bit passes instead of four (the fourth would be a copy: two launches less per
                                           forward). Verified on the device before the sort; if the top byte does vary the call sets
                                           GhCounters.overflow bit 3 and returns a NaN image: re-run without the flag. */
#define GH_FLAG_DEFER_LOSS_SUM 64u      /* with a fused image loss (GhOutputs.l1_* / fit_loss): the forward leaves the per-quadrant
                                           partial sums in the workspace and launches NO sum kernel; *l1_loss / *fit_loss->loss is
                                           written by the matching backward instead (GhGrads.deferred_loss, a spare workgroup of its
                                           render kernel: the one-workgroup sum leaves the step's critical path, -4.7 us per step at
                                           512x334). Until that backward has run the loss value is undefined; the gradients
                                           l1_dL_dimage / fit_loss->dL_* are complete after the forward as always. Ignored without a
                                           fused loss. */
#define GH_FLAG_FRESH_ORDER 128u        /* the scheduling hints the previous forward left in this workspace (GhLayout.tile_walk[3]: the
                                           next launch order) do not describe THIS call — other cameras, another scene: rank the
                                           forward's tiles by this call's own list lengths (a kernel of its own behind the binning, as
                                           before v0.8's hints) instead of inside the projection kernel. Never changes a result; with
                                           8 views a step is ~2 % slower WITH stale hints than without, ~1 % faster with valid ones. A
                                           host that cannot tell sets it. */
#define GH_FLAG_SPLIT_STREAMS 8u        /* n_views >= 2: the views are rendered as two independent halves (views [0, n/2) and
                                           [n/2, n)), the second on a HIP stream of the library's own, forked from and joined
                                           back into the caller's stream inside every call (graph-capturable): the drain of one
                                           half's kernels is filled with the other half's work. Images, radii and gradients
                                           are bit-identical to the unsplit call. Each half gets its share of max_instances
                                           (proportional to its views): GhCounters.overflow is set when either half exceeds
                                           its share, and GhCounters.reserved[0] holds the max_instances that would have
                                           sufficed. Not available for gh_forward_shared / gh_backward_shared. */

/*
 * One camera, GH_CAM_FLOATS consecutive floats in DEVICE memory (built by the caller without a host sync):
 *   [ 0..15] viewmatrix   (renderer_one_shot.py:288  world_view_transform, flat)
 *   [16..31] projmatrix   (renderer_one_shot.py:289  full_proj_transform, flat)
 *   [32..34] campos       (renderer_one_shot.py:291)
 *   [35]     tanfovx      (renderer_one_shot.py:278)
 *   [36]     tanfovy      (renderer_one_shot.py:279)
 *   [37..39] bg           (renderer_one_shot.py:286)
 */
typedef struct GhDims {
  uint32_t abi;           /* GH_ABI_TAG of the header the caller was compiled against (see above); checked first by every call */
  int32_t P;              /* Gaussians */
  int32_t n_views;        /* cameras rendered by this call (grid.z); the reference loops views in Python (:494) */
  int32_t H, W;           /* image size */
  int32_t sh_degree;      /* active SH degree 0..3 (renderer_one_shot.py:290); ignored when colors are precomputed */
  int32_t M;              /* SH coefficients per Gaussian present in `shs` (1,4,9,16); 0 => colors_precomp path */
  float scale_modifier;   /* renderer_one_shot.py:287 */
  uint32_t flags;
  int64_t max_instances;  /* capacity for tile instances D summed over views; see GhCounters.overflow */
} GhDims;

/* Inputs: row-major fp32 device arrays exactly as the reference hands them to the rasteriser. */
typedef struct GhInputs {
  const float* cams;            /* (n_views, GH_CAM_FLOATS) */
  const float* means3D;         /* (P,3) */
  const float* opacities;       /* (P,)  — reference passes (P,1) */
  const float* scales;          /* (P,3)                                           (NULL with cov3D_precomp) */
  const float* rotations;       /* (P,4) quaternion (w,x,y,z), used as given       (NULL with cov3D_precomp) */
  const float* shs;             /* (P,M,3) or NULL */
  const float* colors_precomp;  /* (P,3)   or NULL — exactly one of shs / colors_precomp */
  /* Optional fused attribute blend (renderer_one_shot.py:298-334). NULL => that term is absent. */
  const float* blend_xyz_b;     /* (3,)    means3D += xyz_b                       (:300-301) */
  const float* blend_opacity_b; /* (P,)    opacity += opacity_b                   (:306-307) */
  const float* blend_color_w;   /* (48,) or (P,48): RGB: c*w[0:3] + w[3:6] - 1    (:323-324); SH: shs*w (:331-332) */
  const float* blend_color_b;   /* (P,48): RGB: + b[0:3] (:327-328); SH: (shs*w)*w + b (:333-334) */
  /* Optional speculative occlusion bound (gh_forward only; NULL = none): the array a previous gh_forward of the same views
     wrote as GhOutputs.tile_depth_seen — (n_views * tiles) pairs (depth as float, block mask as uint32), tile-major inside a
     view (tile = ty * ceil(W/16) + tx). A tile is bounded when all of its own pixels reached the early stop in that call AND so
     did every 4x4-pixel block of the eight neighbouring tiles that touches it (a silhouette that moves by up to four pixels
     cannot uncover a pixel of a bounded tile: an uncovered pixel never stops and needs its whole list).
     A (Gaussian, tile) instance whose view-space depth is ABOVE the tile's bound is not listed. The forward VERIFIES the
     speculation: every pixel of a bounded tile must reach the early stop of App. A.3 (T (1 - alpha) < 1e-4) inside the
     truncated list — then nothing behind the bound could have been looked at and the result is the unbounded call's bit for
     bit; a pixel that runs off the end of a truncated list gets NaN and GhCounters.overflow |= 4: re-run without the bound. */
  const float* tile_depth_bound;
  /* Precomputed 3-D covariance, the published module's `cov3D_precomp` (the reference never passes it, renderer_one_shot.py:313,
     :346): (P,6) = the upper triangle (xx, xy, xz, yy, yz, zz) of Sigma, used AS GIVEN — scale_modifier is not applied — in place
     of scales + rotations: exactly one of {scales AND rotations, cov3D_precomp} (App. A.1-3). NULL = scales + rotations. */
  const float* cov3D_precomp;
} GhInputs;

/* The image part of the one-shot fit loss (gh_fit_loss below: utils.py:180-252, :282-294; infer_one_shot.py:497, :507-510), evaluated
   by the render kernel's epilogue (GhOutputs.fit_loss):
     loss[0] = scale * sum_v [ lambda_l1 * mean|bbox * rgb - gt_rgb| + lambda_mask * mean((clip(alpha, -0.001, 1) - gt_mask)^2) ]
   with dL_dimage / dL_dalpha as GhGrads takes them — gh_fit_loss's values: the gradients bit for bit, the loss up to the order of
   its fixed-order float32 sums. An invalid call (GhCounters.overflow bits 0, 1, 3) yields loss NaN and zero gradients. */
typedef struct GhFitLoss {
  const float* gt_rgb;     /* (n_views,H,W,3) */
  const float* gt_mask;    /* (n_views,H,W) */
  const float* bbox;       /* (n_views,H,W) or NULL: colour is zeroed where bbox == 0 */
  float lambda_l1, lambda_mask, scale;
  float* dL_dimage;        /* (n_views,3,H,W) */
  float* dL_dalpha;        /* (n_views,H,W) */
  float* loss;             /* 1 float */
} GhFitLoss;

typedef struct GhOutputs {
  float* image;    /* (n_views,3,H,W) */
  int32_t* radii;  /* (n_views,P)   0 for culled Gaussians */
  float* alpha;    /* (n_views,H,W) or NULL. Accumulated alpha = the reference's mask pass (colour 1, bg 0,
                      renderer_one_shot.py:353-380) produced by the SAME walk as a 4th channel */
  /* (n_views * tiles) pairs (float, uint32) or NULL: per tile, [0] tile_depth_seen_scale x the depth of the LAST list entry any of
     its pixels looked at (the entry that stopped its last pixel, plus the slack below) when every pixel of the tile reached the
     early stop, else +inf (tiles at the silhouette, empty tiles); [1] bit b = every in-image pixel of the tile's 4x4-pixel block b
     (b = 4 * block row + block column) reached the stop. Fed back as GhInputs.tile_depth_bound of the NEXT step of a loop whose Gaussians move a little
     between steps (the one-shot fit's network-side trainables move them every step, infer_one_shot.py:340-343), it removes the
     instances behind a saturated surface — over half of them on the hand scenes — from emit, sort and record gather;
     tile_depth_seen_scale > 1 is the margin for the motion. Must not alias tile_depth_bound. */
  float* tile_depth_seen;
  float tile_depth_seen_scale;
  /* Slack, in list entries: the depth reported is that of the entry this many positions BEHIND the last one looked at (the
     list's own last entry if it ends first; the tile's previous bound, scaled, if that list was itself truncated). A pixel whose
     transmittance ended just below the 1e-4 stop needs a few more entries as soon as anything moves — possibly from the next
     surface centimetres behind; the slack entries are what it then finds. */
  uint32_t tile_depth_seen_slack;
  /* Fused image loss (v0.7; gh_forward / gh_forward_stages / gh_forward_refresh; NULL = none): the L1 term of utils.py:282-294 as bench.py and the
     fit loop consume the render — l1_loss[0] = mean|image - l1_target| over all n_views*3*H*W values and l1_dL_dimage =
     sign(image - l1_target) / (n_views*3*H*W) (sign(0) = 0, as torch.abs' backward), both (n_views,3,H,W) like `image` —
     produced by the render kernel's own epilogue while the pixel is still in registers: the same values gh_l1_loss computes from
     the stored image (the gradient bit for bit; the loss up to the order of its fixed-order sums), without the pass over the
     images. Bitwise reproducible. A call the device flags as invalid (GhCounters.overflow bits 0, 1, 3) yields l1_loss = NaN and
     l1_dL_dimage = 0, like gh_l1_loss under its guard. Not available together with `alpha`, `tile_depth_seen`,
     GhInputs.tile_depth_bound or GH_FLAG_SPLIT_STREAMS (GH_ERR_UNSUPPORTED). All three pointers or none; image, l1_target and
     l1_dL_dimage are three arrays whose byte ranges do not overlap (GH_ERR_INVALID_ARG). */
  const float* l1_target;
  float* l1_dL_dimage;
  float* l1_loss;
  /* The fit's image loss fused the same way (v0.7; gh_forward / gh_forward_stages / gh_forward_refresh; NULL = none; needs `alpha`):
     a HOST struct, read during the call. Same exclusions as l1_target, and not both. */
  const struct GhFitLoss* fit_loss;
} GhOutputs;

/* Device-side counters written by gh_forward (first bytes of the workspace, see GhLayout.counters). */
typedef struct GhCounters {
  uint32_t num_rendered; /* D: tile instances emitted (before clamping to max_instances) */
  uint32_t overflow;     /* bit 0: D > max_instances: image / gradients are invalid, re-run with a larger workspace;
                            bit 1: gh_forward_refresh met an opacity above the bound its lists were built for;
                            bit 2: a pixel of a tile with a GhInputs.tile_depth_bound ran off the end of its truncated list
                                   (that pixel is NaN): re-run without the bound;
                            bit 3: GH_FLAG_DEPTH24 did not hold (NaN image): re-run without the flag
                            bits 0-3 (GH_COUNTER_ERROR_MASK) are errors — the device-side guards of the loss / optimiser entry points
                            test exactly these;
                            bit 4: INFORMATION, set by every full forward: the depth keys of the Gaussians that emit instances did
                                   not differ in their top byte, i.e. GH_FLAG_DEPTH24 would have held (or did hold) for this call —
                                   how a caller learns, from a call made WITHOUT the flag, that the flag is safe for a call shape */
  uint32_t reserved[2];  /* [0] after a GH_FLAG_SPLIT_STREAMS forward: the max_instances that would have sufficed */
} GhCounters;

#define GH_COUNTER_ERROR_MASK 15u
#define GH_COUNTER_DEPTH24_OK 16u

/* Upstream gradient + outputs of gh_backward. Any output pointer may be NULL (that gradient is skipped). */
typedef struct GhGrads {
  const float* dL_dimage;   /* (n_views,3,H,W) */
  const float* dL_dalpha;   /* (n_views,H,W) or NULL: upstream gradient of GhOutputs.alpha (fused mask pass) */
  float* dL_dmeans3D;       /* (P,3)   summed over views */
  float* dL_dmeans2D;       /* (n_views,P,3)  [dL/dpx*W/2, dL/dpy*H/2, 0] — API parity with means2D.grad */
  float* dL_dopacities;     /* (P,) */
  float* dL_dscales;        /* (P,3) */
  float* dL_drotations;     /* (P,4) */
  float* dL_dshs;           /* (P,M,3) */
  float* dL_dcolors;        /* (P,3) */
  /* gradients of the fused blend parameters (only when the matching input pointer was given) */
  float* dL_dblend_xyz_b;     /* (3,) */
  float* dL_dblend_opacity_b; /* (P,) */
  float* dL_dblend_color_w;   /* (48,) or (P,48) */
  float* dL_dblend_color_b;   /* (P,48), 16-byte aligned (rows are written as float4s) */
  /* Optional device scalar that multiplies dL_dimage and dL_dalpha as they are read (NULL = 1): the upstream gradient of
     a scalar loss whose image gradient was produced unscaled by the loss kernel (gh_l1_loss / gh_fit_loss), so that the
     autograd product `dL/dimage * dL/dloss` needs no pass over the images of its own. */
  const float* upstream_scale;
  float* dL_dcov3D;           /* (P,6) with GhInputs.cov3D_precomp: d(loss)/d(xx, xy, xz, yy, yz, zz) of the symmetric storage (an
                                 off-diagonal entry carries both of its matrix positions), summed over views */
  /* v0.8, NULL = none: where the backward writes the fused image loss whose final sum the forward deferred (GH_FLAG_DEFER_LOSS_SUM
     in that forward's GhDims.flags; the value GhOutputs.l1_loss / fit_loss->loss would have received, up to the order of the
     fixed-order float32 sum — bitwise reproducible). Written by the render stage (gh_backward, gh_backward_refresh,
     gh_backward_stages with GH_BWD_RENDER); not with GH_FLAG_SPLIT_STREAMS or gh_backward_shared (GH_ERR_UNSUPPORTED: those
     forwards fuse no loss). */
  float* deferred_loss;
} GhGrads;

/* Byte offsets of the internal arrays inside the workspace (public so tests can inspect every stage). */
typedef struct GhLayout {
  size_t total_bytes;
  size_t counters;       /* GhCounters */
  size_t geom;           /* float4[n_views*P][4]: one 64-byte line per Gaussian:
                            (px, py, conicA, conicB) (conicC, opacity, r, g) (b, rect bits, tile hit mask lo, hi)
                            (tiles_touched, view-space depth, 3-sigma tile rect, -): the last float4 is written for every
                            Gaussian, the first three for those that are listed in a tile;
                            rect = minx | miny<<8 | maxx<<16 | maxy<<24 (tile units), 0 = none (v0.8: depth and rect were arrays
                            of their own; every 4-byte store of the projection kernel at a view-major address costs 0.85 us) */
  size_t clamped;        /* uint8 [n_views*P]  SH colour clamp flags (bit ch) */
  size_t tiles_touched;  /* uint32[n_views*P]  tiles of the rect the alpha >= 1/255 ellipse reaches (exact tile culling); 0 = none */
  size_t slot_begin;     /* uint32[n_views*P]  first RECORD slot of the (view, Gaussian): the backward's sub-records of its instances are
                            [begin, begin+tiles), row-major over the hit tiles of its rect. Record slots are numbered in the order
                            the per-Gaussian kernels walk the (view, Gaussian) pairs — Gaussian-major, the views of a row adjacent
                            (pose batch: row-major) — so that the chain rule's wave reads one contiguous stretch of sub-records */
  size_t depth_keys_a, depth_keys_b; /* uint32[n_views*P] level-1 sort: depth bits (0xFFFFFFFF when culled); result in _a */
  size_t depth_vals_a, depth_vals_b; /* uint32[n_views*P] level-1 payload: view*P + gaussian; depth order in _a */
  size_t block_sums;     /* uint32[...]        scan scratch */
  size_t keys_a, keys_b; /* uint32[max_instances] level-3 sort: global tile id (one view) or tile id inside the view (two or more views: the
                            partition runs per view); sorted result in keys_a */
  size_t vals_a, vals_b; /* uint32[max_instances] payload view*P + gaussian; sorted result in vals_a */
  size_t sorted_slot;    /* uint32[max_instances] sorted position -> record slot (where the backward puts its sub-records): a permutation
                            of 0 .. D-1 */
  size_t inst_r0;        /* float4[max_instances] sorted per-instance render record (px, py, -conicA/2, conicB) */
  size_t inst_r1;        /* float4[max_instances]                                   (-conicC/2, opacity, r, g): the render kernels'
                            power is (A' dx dx + C' dy dy) - B dx dy — App. A.3's value bit for bit, one multiply cheaper (v0.8) */
  size_t inst_r2;        /* float2[max_instances]                                   (b, bits: 4x4-block mask of the tile) */
  size_t sort_tables;    /* uint32[...]        per-pass digit tables */
  size_t ranges;         /* uint2 [n_views*tiles] [start,end) into the sorted list */
  size_t tile_walk;      /* uint32[12][n_views*tiles]: [0] list entries actually walked by the forward (max n_contrib of the tile); [1] forward
                            waves that have finished the tile (the last one appends the tile's backward items); [2] stop positions
                            (tile_depth_seen). SCHEDULING HINTS, never cleared by the library's entry points, never affecting results:
                            [3] calls of at most 8,192 tiles — what the PREVIOUS forward over this workspace measured per tile (up to 3,072
                            tiles: the most entries one 4x4-pixel block took; above: half the entries walked): the next launch order;
                            [4..12) calls of 2,049 .. 8,192 tiles — uint32[n_views*tiles][8], cycles / 256 the previous backward's slowest
                            workgroup spent on (tile, depth segment min(k, 7)): the order of the next backward's work list */
  size_t tile_order;     /* uint32[n_views*tiles] forward launch order of the render blocks: longest tile lists first */
  size_t bwd_items;      /* uint2 [16][n_views*tiles + max_instances/GH_SEGMENT + 2] backward work items (tile, depth segment): sixteen regions
                            of the list's full capacity, counts in render_guard[1..17). The backward takes the regions from the highest
                            down, each from its end. Calls of more than 8,192 tiles use region 0 only, filled in the order the forward
                            finished the tiles (long tiles finish last); smaller calls: see tile_walk */
  size_t ckpt_rgb;       /* float4[slots][256] forward state (T, C0, C1, C2) of every pixel of a tile at list positions that are
                            multiples of GH_SEGMENT; slots = max_instances/GH_SEGMENT + n_views*tiles + 2 */
  size_t final_C;        /* float4[n_views*H*W] colour accumulated by the forward, without background */
  size_t final_T;        /* float [n_views*H*W] */
  size_t n_contrib;      /* uint32[n_views*H*W] */
  size_t inst_grad;      /* float[max_instances][4][9] per-(instance, quadrant) gradient sub-records (backward scratch) */
  size_t inst_flag;      /* uint8[max_instances][4]     1 where the quadrant wrote its sub-record (zeroed per forward) */
  size_t sh_rgb;         /* float4[n_views*P] SH colour stage output (r, g, b, clamp-flag bits); unused with colors_precomp */
  size_t dmean_sh;       /* float4[n_views*P] d(loss)/d(mean) through the SH view direction (backward scratch) */
  size_t sh_scratch;     /* float[ceil(P/16)][64] block partials of the global colour-weight gradient (SH mode) */
  size_t grad_sums;      /* float4[n_views*P][3] per-(view,Gaussian) sums of the sub-records: dpx dpy dA dB | dC do dr dg | db
                            (a kernel of its own writes them for SH colours and above half a megapixel per view; otherwise the
                            chain-rule kernel keeps them in registers) */
  size_t bwd_scratch;    /* blend-parameter reduction scratch */
  size_t cull_bound;     /* float [n_views*P]  opacity every (view, Gaussian)'s tiles were culled with (+inf: no rect); the guard of
                            gh_forward_refresh compares the current opacity with it */
  size_t inst_c;         /* float [max_instances] conic C of every sorted instance (the static part of inst_r1), read by
                            gh_forward_refresh */
  size_t attr;           /* float4[n_views*P]  (opacity, r, g, b) of the CURRENT step per (view, Gaussian): gh_forward_refresh */
  size_t half_counters;  /* GhCounters[2], 256 bytes apart: the counters of the two halves of a GH_FLAG_SPLIT_STREAMS call (every
                            per-view / per-tile / per-pixel array keeps its place; a half's per-instance arrays start at its
                            share of max_instances) */
  size_t key_bits;       /* uint2[projection blocks] (OR, AND) over the depth-key bits of the block's visible Gaussians: a depth-sort
                            pass whose digit is the same in every key (OR & ~AND has no bit in it) degenerates to a copy */
  size_t tile_bound;     /* float[n_views*tiles] the effective occlusion bound of this call (GhInputs.tile_depth_bound after the
                            neighbourhood test; +inf = unbounded), read by every kernel that decides list membership */
  size_t block_tiles;    /* uint32[projection blocks] instances counted by each block of the projection kernel (record-slot scan) */
  size_t render_guard;   /* uint32: the error bits of GhCounters.overflow as they stood BEFORE the render kernel of this call — written by
                            the kernel in front of it, read by every render wave through the scalar cache (the counters' own line
                            takes the render kernel's atomics); uint32[16] behind it: items per region of bwd_items */
  size_t loss_partials;  /* float[n_views*tiles][4] + 1: the fused image loss's sums of |image - target| per 8x8-pixel quadrant
                            (GhOutputs.l1_target), added up in index order by a one-workgroup kernel behind the render (or by the
                            backward: GH_FLAG_DEFER_LOSS_SUM); the last float is the factor of the final sum. Calls of at most 3,072
                            tiles: [n_views*tiles][16] + 1, one sum per 4x4-pixel block (the same whichever form walked the tile) */
  size_t view_start;     /* uint32[n_views + 1] first emit slot of every view (and D): the segments of the per-view tile partition (v0.8;
                            from two views on keys_a / keys_b hold tile ids INSIDE the view, see gh_partition_per_view) */
} GhLayout;

/* Library version: major<<16 | minor. */
int gh_version(void);

/* Fills `out` with the workspace layout for `dims`; returns GH_OK or an error. Pure host arithmetic. */
int gh_workspace_layout(const GhDims* dims, GhLayout* out);

/* Convenience: total workspace bytes (0 on invalid dims). */
size_t gh_workspace_bytes(const GhDims* dims);

/* 1 when a forward with these dims partitions its tile instances PER VIEW (GhLayout.keys_a then holds tile ids inside the view and
   GhLayout.view_start the views' first slots), 0 when by global tile id, negative GhStatus on invalid dims. For callers that
   inspect the binning arrays (tests, tools); nothing on the render path needs it. */
int gh_partition_is_per_view(const GhDims* dims);

/*
 * Forward: blend -> project -> conic -> bin -> sort -> composite. Replaces
 * GaussianRasterizer.forward (call sites renderer_one_shot.py:338-346, :372-379).
 * The workspace must be kept intact until the matching gh_backward has run.
 */
int gh_forward(const GhDims* dims, const GhInputs* in, const GhOutputs* out,
               void* workspace, size_t ws_bytes, void* hip_stream);

/*
 * Backward: per-pixel reverse walk -> per-instance records -> per-Gaussian chain rule.
 * Replaces the autograd backward of the reference's rasteriser call. Output gradient arrays are
 * fully overwritten (not accumulated into).
 */
int gh_backward(const GhDims* dims, const GhInputs* in, const GhGrads* grads,
                void* workspace, size_t ws_bytes, void* hip_stream);

/*
 * Stage-selective variants (same contracts as above). They let a caller time, overlap or re-run single
 * stages; gh_forward == gh_forward_stages(GH_FWD_ALL), gh_backward == gh_backward_stages(GH_BWD_ALL).
 * Stages must be run in order on one stream for a given workspace.
 */
#define GH_FWD_PREPROCESS 1u /* blend + projection + conic + tile rect (per Gaussian)          */
#define GH_FWD_BINNING    2u /* scan + emit + radix sort + ranges                               */
#define GH_FWD_RENDER     4u /* per-tile compositing                                            */
#define GH_FWD_ALL        7u
#define GH_BWD_RENDER     1u /* per-pixel reverse walk -> per-instance records                  */
#define GH_BWD_PREPROCESS 2u /* per-Gaussian record sum + chain rule + blend-parameter gradients */
#define GH_BWD_ALL        3u

int gh_forward_stages(const GhDims* dims, const GhInputs* in, const GhOutputs* out,
                      void* workspace, size_t ws_bytes, void* hip_stream, uint32_t stages);
int gh_backward_stages(const GhDims* dims, const GhInputs* in, const GhGrads* grads,
                       void* workspace, size_t ws_bytes, void* hip_stream, uint32_t stages);

/*
 * A second rasteriser call over the SAME geometry. The reference renders every view twice with identical means3D /
 * opacities / scales / rotations / camera: the RGB pass (renderer_one_shot.py:338-346) and the mask pass (:355-379,
 * colors_precomp = 1, bg = 0). `geometry_ws` is the workspace of a completed gh_forward with the same dims (P, n_views,
 * H, W, max_instances, flags) and the same geometry inputs; `workspace` (gh_workspace_bytes(dims), a different buffer)
 * receives this call's own state only: colour records, image state, backward scratch. Projection, both sorts, emit and
 * the record gather are skipped; results are bit-identical to a full gh_forward / gh_backward with the same inputs.
 * Colours must be colors_precomp (GH_ERR_UNSUPPORTED with shs); out->radii may be NULL (the first call's are the same).
 * The geometry workspace must stay intact until the matching gh_backward_shared has run.
 */
int gh_forward_shared(const GhDims* dims, const GhInputs* in, const GhOutputs* out, const void* geometry_ws,
                      void* workspace, size_t ws_bytes, void* hip_stream);
int gh_backward_shared(const GhDims* dims, const GhInputs* in, const GhGrads* grads, const void* geometry_ws,
                       void* workspace, size_t ws_bytes, void* hip_stream);

/*
 * A later step over STATIC geometry. The one-shot fit (infer_one_shot.py:489-524) renders the same Gaussians from the same
 * cameras every step; only the blend parameters it trains move — colours (color_w, color_b) and opacities (opacity_b),
 * renderer_one_shot.py:306-334. `geometry_ws` is the workspace of a completed gh_forward with GH_FLAG_STATIC_LISTS and the
 * same dims and means3D / scales / rotations / xyz_b / cameras; this call skips projection, both sorts, emit and the record
 * gather and only (a) [shs given] re-evaluates the SH colours, (b) refreshes the per-instance render records — opacity,
 * colour and the 4x4-block mask of the CURRENT opacity — in one streaming pass, (c) walks the lists. Forward images are those
 * of a full gh_forward bit for bit; gradients agree to rounding (another partition of the lists into depth segments).
 * Guard: a Gaussian whose opacity (+ opacity_b) has risen above the bound its tiles were culled with (max(2, twice the
 * opacity at build time)) might reach a tile that is not listed: the call then sets GhCounters.overflow |= 2 and the image is NaN,
 * exactly like an instance overflow — rebuild the lists with a full gh_forward.
 * gh_backward_refresh: gradients w.r.t. whatever `grads` asks for; with every geometry gradient pointer (dL_dmeans3D,
 * dL_dmeans2D, dL_dscales, dL_drotations, dL_dblend_xyz_b) NULL the per-Gaussian chain rule reduces to sums over the views
 * (the fit trains colour / opacity biases only), and — with precomputed colours, in gh_backward as well — the list walk leaves
 * the position / conic moments out of its sub-records.
 * Colour mode: the refresh call may bring other colours than the build, but in the SAME mode (both shs, or both
 * colors_precomp): the library reads the build's workspace with this call's layout, and the arrays behind the SH stage's
 * scratch move with M == 0 / M != 0 — a caller that switches modes builds the lists again.
 */
int gh_forward_refresh(const GhDims* dims, const GhInputs* in, const GhOutputs* out, const void* geometry_ws,
                       void* workspace, size_t ws_bytes, void* hip_stream);
int gh_backward_refresh(const GhDims* dims, const GhInputs* in, const GhGrads* grads, const void* geometry_ws,
                        void* workspace, size_t ws_bytes, void* hip_stream);

/*
 * Per-Gaussian bilinear lookup of a learnable UV map and its backward. Replaces
 * F.grid_sample(..., align_corners=True, mode="bilinear") of query_triplane_texture (renderer_one_shot.py:420-446)
 * at the call sites :489-492. `map` is CHANNEL-LAST (Hm, Wm, C) fp32 (the reference parameter (C,Hm,Wm) permuted
 * once); `uv` is (P,2) in [-1,1] (x = u indexes Wm, y = v indexes Hm); texels outside the map read as zero.
 * gh_uv_sample_backward ACCUMULATES into dL_dmap (the caller zeroes it) with float atomics: run-to-run order noise. It is
 * the one entry point of this header that is not bitwise reproducible; it needs no index built beforehand. The host side of
 * this repository (uvmap.py) does not call it: it builds the texel lists once and uses gh_uv_scatter_sorted below.
 */
int gh_uv_sample_forward(const float* map, const float* uv, float* out /* (P,C) */, int P, int C, int Hm, int Wm,
                         void* hip_stream);
int gh_uv_sample_backward(const float* uv, const float* dL_dout /* (P,C) */, float* dL_dmap /* (Hm,Wm,C) */, int P, int C,
                          int Hm, int Wm, void* hip_stream);

/*
 * Active-texel form of the same lookup for the one-shot fit loop (infer_one_shot.py:489-524). The Gaussians' UVs are
 * fixed during the fit, so only the texels under their bilinear footprints ever change (all others keep gradient 0
 * under the regularisers of :514-518 and stay 0 under Adam). `texels` holds those U texels compacted as (U, C);
 * slot (P,4) int32 = compact row of the nw, ne, sw, se corner (-1 = outside the map, reads as zero), w (P,4) = the
 * bilinear weights. gh_uv_gather_backward ACCUMULATES into dL_dtexels with float atomics (order noise, see above).
 * gh_uv_scatter_sorted is the deterministic backward of both lookups: the caller lists, once per set of UVs, the
 * (Gaussian, corner) pairs under every active texel — CSR, row_ptr (U+1) int32, pairs (nnz) int32 = 4 * gaussian + corner in
 * ascending order inside a texel — and the kernel GATHERS: one lane per (texel, channel) adds dL_dout[gaussian][c] * w[pair]
 * in list order and ACCUMULATES the sum into dL_dtexels (U,C). No atomics, bitwise reproducible.
 */
int gh_uv_gather_forward(const float* texels, const int32_t* slot, const float* w, float* out /* (P,C) */, int P, int C,
                         void* hip_stream);
int gh_uv_gather_backward(const int32_t* slot, const float* w, const float* dL_dout /* (P,C) */, float* dL_dtexels /* (U,C) */,
                          int P, int C, void* hip_stream);
int gh_uv_scatter_sorted(const int32_t* row_ptr /* (U+1) */, const int32_t* pairs /* (nnz) */, const float* w /* (P,4) */,
                         const float* dL_dout /* (P,C) */, float* dL_dtexels /* (U,C) */, int U, int C, void* hip_stream);

/*
 * One fused pass over a parameter array of n floats: torch.optim.Adam's update (infer_one_shot.py:345; step >= 1 is
 * the 1-based step count, no weight decay / amsgrad) on grad + reg_l1*sign(param) + reg_l2*2*param, i.e. with the
 * gradient of reg_l1*sum|param| + reg_l2*sum(param^2) (the regularisers of :514-518) folded in. `grad` is cleared
 * for the next accumulation. partials (n_partials,2) receives per-block sums of |param| and param^2 of the
 * PRE-update values (the regulariser value the loss of this step reports); n_partials is also the grid size.
 *
 * Device-side overflow guard (optional, NULL = none): `guard` points at the GhCounters of the forward whose gradients feed
 * this step (the first bytes of its workspace). When guard->overflow is set the step is a no-op for param / exp_avg /
 * exp_avg_sq (grad is still cleared, partials still written), so a sync-free or graph-replayed fit never steps on the
 * invalid gradients of an overflowed render. With a guard the bias-correction step count must live on the device as well:
 * `step_state` (two int32, zero-initialised by the caller, NULL = use `step`): [0] = the number of steps actually applied, read
 * by every block as it starts and advanced by the block that finishes last ([1] is that launch's ticket counter, 0 between
 * launches); `step` is then ignored apart from the >= 1 check, so a captured (hipGraph) step replays with the right bias
 * correction.
 */
int gh_adam_reg_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int step, float lr, float beta1,
                     float beta2, float eps, float reg_l1, float reg_l2, float* partials, int n_partials,
                     const GhCounters* guard, int32_t* step_state, void* hip_stream);

/* The same update for up to four tensors in ONE launch (the fit steps color_w, color_b and opacity_b together,
 * infer_one_shot.py:345): `tensors` is a HOST array of n_tensors descriptors; results are those of n_tensors
 * gh_adam_reg_step calls with the same step / lr / betas / eps / guard, bit for bit. */
typedef struct GhAdamTensor {
  float* param; float* grad; float* exp_avg; float* exp_avg_sq;
  size_t n;
  float reg_l1, reg_l2;
  float* partials; int n_partials;    /* (n_partials, 2) block sums of |param| and param^2 of the pre-update values */
  int32_t* step_state;                /* [2] device-side step count of the bias correction, or NULL (use `step`) */
} GhAdamTensor;
int gh_adam_reg_step_group(const GhAdamTensor* tensors, int n_tensors, int step, float lr, float beta1, float beta2, float eps,
                           const GhCounters* guard, void* hip_stream);

/* Two maps looked up at the same UVs in one launch (the fit's colour-bias and opacity-bias maps, renderer_one_shot.py:489-492)
 * and the deterministic backward of both: bit-identical to two gh_uv_gather_forward / gh_uv_scatter_sorted calls. */
int gh_uv_gather_forward2(const float* texels_a, int Ca, const float* texels_b, int Cb, const int32_t* slot, const float* w,
                          float* out_a /* (P,Ca) */, float* out_b /* (P,Cb) */, int P, void* hip_stream);
int gh_uv_scatter_sorted2(const int32_t* row_ptr, const int32_t* pairs, const float* w, const float* dL_dout_a /* (P,Ca) */, int Ca,
                          float* dL_dtexels_a /* (U,Ca) */, const float* dL_dout_b /* (P,Cb) */, int Cb, float* dL_dtexels_b /* (U,Cb) */,
                          int U, void* hip_stream);

/*
 * Loss assembly of the fit step (infer_one_shot.py:514-519: loss = image loss + 100 * mean|color_b| + mean(opacity_b^2)):
 * out2[1] = k_a * sum_i partials_a[i][col_a] + k_b * sum_i partials_b[i][col_b] — fixed-order sums over the (n, 2) block
 * partials gh_adam_reg_step left (col 0 = sum|param|, col 1 = sum param^2) — and out2[0] = base[0] + out2[1] (base may be
 * NULL). One block; bitwise reproducible.
 */
int gh_reg_total(const float* partials_a, int n_a, int col_a, float k_a, const float* partials_b, int n_b, int col_b, float k_b,
                 const float* base, float* out2, void* hip_stream);

/*
 * Image-loss consumer (the L1 term of utils.py:282-294 as bench.py / the fit loop use it):
 * loss_out[0] = mean|image - target| over n floats and dL_dimage = sign(image - target) / n (sign(0) = 0, as
 * torch.abs' backward) in one pass. image / target / dL_dimage must be 16-byte aligned; partials holds n_partials
 * floats of scratch (n_partials = grid size, e.g. 1024). Fixed-order sums: bitwise reproducible.
 * guard (optional, NULL = none): GhCounters of the forward that produced `image`; when its overflow flag is set the image
 * is invalid, so loss_out[0] = NaN and dL_dimage = 0 — nothing downstream can step on garbage (also gh_fit_loss).
 */
int gh_l1_loss(const float* image, const float* target, size_t n, float* loss_out, float* dL_dimage, float* partials,
               int n_partials, const GhCounters* guard, void* hip_stream);

/*
 * Image part of the one-shot fit loss for a stack of views (utils.py:180-252, :282-294; infer_one_shot.py:497, :507-510):
 *   loss = scale * sum_v [ lambda_l1 * mean|bbox*rgb - gt_rgb| + lambda_mask * mean((clip(alpha, -0.001, 1) - gt_mask)^2) ]
 * with its gradients w.r.t. the rasteriser outputs, in one pass. image (n_views,3,H,W) and alpha (n_views,H,W) are
 * GhOutputs' layouts; gt_rgb (n_views,H,W,3) and gt_mask (n_views,H,W) the reference's; bbox (n_views,H,W) fp32 or
 * NULL (colour is zeroed where bbox == 0). dL_dimage / dL_dalpha are what GhGrads takes. partials: n_partials floats.
 */
int gh_fit_loss(const float* image, const float* alpha, const float* gt_rgb, const float* gt_mask, const float* bbox,
                int n_views, int H, int W, float lambda_l1, float lambda_mask, float scale, float* loss_out,
                float* dL_dimage, float* dL_dalpha, float* partials, int n_partials, const GhCounters* guard, void* hip_stream);

/*
 * Interaction mask of the interaction-aware step (infer_one_shot.py:247-250):
 *     _, idx_world, _ = knn_points(pointclouds, pointclouds, K=100)
 *     _, idx_tpose, _ = knn_points(t_point, t_point, K=100)
 *     mask = (idx_world == idx_tpose).sum(-1) < 10
 * knn_points is pytorch3d.ops (third-party, not in the reference tree; environment.yml pins pytorch3d 0.7.x): exact
 * brute-force K nearest neighbours under squared L2 distance, returned sorted by ascending distance. gh_knn_indices
 * computes the same lists for one point set against itself (points (N,3) fp32, idx_out (N,K) int32, optional
 * dist_out (N,K) squared distances); equal distances are ordered by ascending index. 1 <= K <= 128, K <= N.
 * `workspace` must hold gh_knn_workspace_bytes(N) bytes. gh_knn_mismatch_mask writes mask_out[i] = 1 when fewer than
 * `min_same` ranks of the two lists of point i hold the same index. All work is enqueued on `hip_stream`.
 */
size_t gh_knn_workspace_bytes(int N);
int gh_knn_indices(const float* points, int N, int K, int32_t* idx_out, float* dist_out, void* workspace, size_t ws_bytes,
                   void* hip_stream);
int gh_knn_mismatch_mask(const int32_t* idx_a, const int32_t* idx_b, int N, int K, int min_same, uint8_t* mask_out,
                         void* hip_stream);

/*
 * Gaussian selection of forward_single_batch (renderer_one_shot.py:468-477): with s = if_gs_valid.squeeze(1),
 *     query_points_valid  = query_points[s > threshold_low]        gs_hidden_features_valid  = gs_hidden_features[s > threshold_low]
 *     query_points_copied = query_points[s > threshold_high]       gs_hidden_features_copied = gs_hidden_features[s > threshold_high]
 * (the reference then refines the copied positions with a network and concatenates valid ++ copied, :474-477).
 * score (N,), points (N,3), features (N,C) fp32 row-major. The kept rows are written in index order to valid_points /
 * valid_features and copied_points / copied_features (each with room for N rows; only the first counts[0] / counts[1] rows
 * are written), their source indices to valid_index / copied_index (int32, optional, NULL = skip), and counts[0..1]
 * (device memory) receive the two row counts: one read-back where the reference's four boolean-mask indexings make four,
 * or none for a caller that works on the padded outputs. NaN scores are dropped (the comparison is false), as in the reference.
 * workspace: gh_select_workspace_bytes(N) bytes. All work is enqueued on hip_stream.
 */
size_t gh_select_workspace_bytes(int N);
int gh_select_rows(const float* score, int N, float threshold_low, float threshold_high, const float* points,
                   const float* features, int C, float* valid_points, float* valid_features, float* copied_points,
                   float* copied_features, int32_t* valid_index, int32_t* copied_index, uint32_t* counts, void* workspace,
                   size_t ws_bytes, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* GH_RASTER_H */
