// gh_preprocess.hip — per-Gaussian stages: fused attribute blend + projection + covariance->conic
// (forward) and the per-Gaussian chain rule incl. blend-parameter gradients (backward).
// Restates SURVEY.md App. A.1 / A.5; the blend is tgs/models/renderer_one_shot.py:298-334.
// HBM-streaming kernels: one thread per (view, Gaussian), outputs packed as float4 records so the
// render kernels gather each Gaussian with two 16-byte loads + one 4-byte load.
#include "gh_internal.h"
GH_WG_TIMER_TU(pre)

// ------------------------------------------------------------------------------------------------
// Effective occlusion bound of this call from what the previous call reported (GhInputs.tile_depth_bound: (depth, block mask)
// per tile): a tile keeps its depth when all of its own pixels stopped (a finite depth says so) and every 4x4-pixel block of
// the eight neighbouring tiles that touches it did too; a neighbour outside the image does not count. One thread per tile.
__global__ __launch_bounds__(GH_BLOCK) void gh_tile_bound_kernel(const float2* __restrict__ seen, int NV, int gx, int gy,
                                                                  float* __restrict__ bound) {
  const int t = blockIdx.x * GH_BLOCK + threadIdx.x;
  const int tiles = gx * gy;
  if (t >= NV * tiles) return;
  const int v = t / tiles, tl = t - v * tiles, ty = tl / gx, tx = tl - ty * gx;
  const float2* sv = seen + (size_t)v * tiles;
  float d = sv[tl].x;
  auto need = [&](int nx, int ny, uint32_t bits) {      // the neighbour's blocks `bits` must all have stopped
    if (nx < 0 || ny < 0 || nx >= gx || ny >= gy) return true;
    return (__float_as_uint(sv[ny * gx + nx].y) & bits) == bits;
  };
  const bool ok = need(tx, ty - 1, 0xF000u) && need(tx, ty + 1, 0x000Fu) && need(tx - 1, ty, 0x8888u) && need(tx + 1, ty, 0x1111u) &&
                  need(tx - 1, ty - 1, 0x8000u) && need(tx + 1, ty - 1, 0x1000u) && need(tx - 1, ty + 1, 0x0008u) && need(tx + 1, ty + 1, 0x0001u);
  bound[t] = ok ? d : __uint_as_float(0x7F800000u);
}

template <bool COV>
__global__ __launch_bounds__(GH_BLOCK) void gh_preprocess_fwd_kernel(
    GhInputs in, int P, int NV, int N, int H, int W, int gx, int gy, int sh_degree, int M, float mod, uint32_t flags,
    const float4* __restrict__ sh_rgb, float4* __restrict__ geom,
    uint8_t* __restrict__ clamped, uint32_t* __restrict__ tiles_touched,
    uint32_t* __restrict__ depth_key, uint32_t* __restrict__ depth_val, GhCounters* __restrict__ ctr,
    int32_t* __restrict__ radii, int T, uint2* __restrict__ ranges, uint32_t* __restrict__ tile_walk,
    uint2* __restrict__ key_bits, float rdiv, float* __restrict__ cull_bound_out, const float* __restrict__ tile_bound,
    uint32_t* __restrict__ block_tiles, int n_proj_blocks, uint32_t* __restrict__ tile_order, int tiles_per_view) {
  GH_WG_TIMER(2);
  // Small launches (gh_fwd_fine_launch): the forward's launch order comes from what the PREVIOUS forward over this workspace measured
  // per tile (tile_walk[3]), so it does not wait for this call's lists: one spare workgroup per view ranks the tiles here, in the
  // shadow of the projection, instead of a kernel of its own between the binning and the render (4.5 us of a one-view step).
  // (they are the FIRST workgroups of the grid: at its end they started last and their 3 us were the kernel's tail — 35.0 -> 37.5 us)
  const int n_rank = (int)gridDim.x - n_proj_blocks;
  if ((int)blockIdx.x < n_rank) {
    gh_rank_tiles(nullptr, tiles_per_view, T / tiles_per_view, (int)blockIdx.x, tile_order, tile_walk + 3 * (size_t)T);
    return;
  }
  const int bid = (int)blockIdx.x - n_rank;
  __shared__ uint2 s_bits[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_tiles[GH_BLOCK / GH_WAVE];
  const int t = bid * GH_BLOCK + threadIdx.x;
  unsigned tiles = 0;
  uint32_t k_or = 0u, k_and = 0xFFFFFFFFu;              // bits of this thread's depth key if its Gaussian emits instances
  if (t == 0) {                                        // counters: reserved[0] = element count of the level-1 (depth) sort
    ctr->num_rendered = 0; ctr->overflow = 0; ctr->reserved[0] = (uint32_t)N; ctr->reserved[1] = 0;
  }
  if (t < T) { ranges[t] = make_uint2(0u, 0u); tile_walk[t] = 0u; tile_walk[T + t] = 0u; tile_walk[2 * T + t] = 0u; }   // per-tile state of the later stages
  // Per-lane state that crosses the wave-cooperative tile count below
  uint32_t dkey = 0xFFFFFFFFu;                          // culled Gaussians sort behind everything and emit nothing
  unsigned rect_bits = 0;
  int n = 0, i = 0, radius = 0;
  float cull_bound = __uint_as_float(0x7F800000u);      // +inf: no tile can be missing for a Gaussian without a rect
  float4 g0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g1 = g0;   // (px, py, A, B), (C, culling opacity): operands of the tile test
  float op = 0.0f, tz = 0.0f;
  int minx = 0, miny = 0, maxx = 0, maxy = 0;
  unsigned long long hitmask = 0ull;
  bool big = false;                                     // rect of more than 64 tiles: counted by the wave
  if (t < N) {
    // Thread -> (Gaussian row i, view v). Shared Gaussians: GAUSSIAN-major, the NV views of a row in adjacent lanes, so the
    // row's attributes (and its 192-byte blend rows) reach the wave once per Gaussian instead of once per (view, Gaussian)
    // — the views' loads of one row coalesce into one request. Every per-(view, Gaussian) output is a whole 64-byte line or
    // a 4-byte element of a view-major array (n = v * P + i): consecutive rows of a view stay adjacent in memory.
    // Pose batch (own rows per view): n = t, row = n.
    int v;
    // (rdiv = 1 / divisor where gh_div_small's range allows it: a run-time integer division costs ~40 instructions)
    if (flags & GH_FLAG_PER_VIEW_GAUSSIANS) { v = rdiv > 0.0f ? (int)gh_div_small((uint32_t)t, (uint32_t)P, rdiv) : t / P; i = t; }
    else { i = rdiv > 0.0f ? (int)gh_div_small((uint32_t)t, (uint32_t)NV, rdiv) : t / NV; v = t - i * NV; }
    n = (flags & GH_FLAG_PER_VIEW_GAUSSIANS) ? t : v * P + i;
    const float* cam = in.cams + (size_t)v * GH_CAM_FLOATS;
    GhGeo e;
    gh_geo_forward<COV>(in, cam, i, mod, H, W, e);
    const bool ok = (e.tz > 0.2f) && (e.det != 0.0f);
    if (ok) {
      float dinv = 1.0f / e.det;
      float mid = 0.5f * (e.a + e.c);
      float sq = sqrtf(fmaxf(0.1f, fmaf(mid, mid, -e.det)));
      float lam1 = mid + sq, lam2 = mid - sq;
      int rad = (int)ceilf(3.0f * sqrtf(fmaxf(lam1, lam2)));
      float ndcx = e.hx * e.winv, ndcy = e.hy * e.winv;
      float px = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
      float py = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
      minx = (int)((px - (float)rad) / (float)GH_TILE); minx = minx < 0 ? 0 : (minx > gx ? gx : minx);
      miny = (int)((py - (float)rad) / (float)GH_TILE); miny = miny < 0 ? 0 : (miny > gy ? gy : miny);
      maxx = (int)((px + (float)rad + (float)(GH_TILE - 1)) / (float)GH_TILE); maxx = maxx < 0 ? 0 : (maxx > gx ? gx : maxx);
      maxy = (int)((py + (float)rad + (float)(GH_TILE - 1)) / (float)GH_TILE); maxy = maxy < 0 ? 0 : (maxy > gy ? gy : maxy);
      int cnt = (maxx - minx) * (maxy - miny);
      if (cnt > 0) {
        rect_bits = (unsigned)minx | ((unsigned)miny << 8) | ((unsigned)maxx << 16) | ((unsigned)maxy << 24);
        radius = rad;                                   // API output: the reference's 3-sigma radius (App. A.1-6)
        op = in.opacities[i];
        if (in.blend_opacity_b) op = op + in.blend_opacity_b[i];
        // GH_FLAG_STATIC_LISTS: the lists outlive this call's opacities (gh_forward_refresh): cull with a bound above them
        const float op_cull = (flags & GH_FLAG_STATIC_LISTS) ? gh_static_cull_opacity(op) : op;
        cull_bound = op_cull;
        tz = e.tz;
        // Exact tile culling: of the tiles in the 3-sigma rect only those are instanced in which the alpha >= 1/255
        // ellipse reaches a pixel centre (gh_block_hit, conservative within its margin). A dropped tile holds no pixel
        // that would blend this Gaussian, so images and gradients are unchanged; gh_emit_kernel repeats this test.
        g0 = make_float4(px, py, e.c * dinv, -e.b * dinv); g1 = make_float4(e.a * dinv, op_cull, 0.0f, 0.0f);
        // Speculative occlusion bound (GhInputs.tile_depth_bound): an instance BEHIND its tile's bound is not listed. The
        // comparison is `depth > bound` in every kernel that decides membership (here, gh_emit_kernel and gh_ranges_kernel for
        // rects without a hit mask), so the lists stay consistent; the forward verifies the speculation per pixel.
        const float* tb = tile_bound ? tile_bound + (size_t)v * (gx * gy) : nullptr;
        if (cnt <= 64) {
          int bit = 0;                                  // row-major position in the rect; rects of <= 64 tiles keep the
          for (int ty = miny; ty < maxy; ++ty)          // hit mask so that the emit kernel does not repeat the tests
            for (int tx = minx; tx < maxx; ++tx, ++bit) {
              bool h = gh_block_hit(g0, g1, (float)(tx * GH_TILE), (float)(ty * GH_TILE), (float)(GH_TILE - 1));
              if (tb) h = h && !(tz > tb[ty * gx + tx]);
              tiles += h ? 1u : 0u;
              if (h) hitmask |= 1ull << bit;
            }
        } else big = true;
      }
    }
  }
  // Rects of more than 64 tiles (huge footprints; no hit mask is kept for them): counted by the WAVE, one such Gaussian at a
  // time, 64 tiles per trip — its own lane would take one tile per trip with the other 63 waiting.
  for (uint64_t m = gh_ballot(big); m != 0ull; m &= m - 1ull) {
    const int src = (int)__builtin_ctzll(m), lane = threadIdx.x & 63;
    auto bf = [&](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src)); };
    const float4 s0 = make_float4(bf(g0.x), bf(g0.y), bf(g0.z), bf(g0.w)), s1 = make_float4(bf(g1.x), bf(g1.y), 0.0f, 0.0f);
    const int sminx = __builtin_amdgcn_readlane(minx, src), sminy = __builtin_amdgcn_readlane(miny, src);
    const int sw = __builtin_amdgcn_readlane(maxx, src) - sminx, sn = sw * (__builtin_amdgcn_readlane(maxy, src) - sminy);
    const float stz = bf(tz);
    const int sn_idx = __builtin_amdgcn_readlane(n, src);
    const float* tb = tile_bound ? tile_bound + (size_t)(sn_idx / P) * (gx * gy) : nullptr;
    unsigned cnt = 0;
    for (int base = 0; base < sn; base += GH_WAVE) {
      const int k = base + lane;
      bool h = false;
      if (k < sn) {
        const int dy = k / sw, dx = k - dy * sw;
        h = gh_block_hit(s0, s1, (float)((sminx + dx) * GH_TILE), (float)((sminy + dy) * GH_TILE), (float)(GH_TILE - 1));
        if (tb) h = h && !(stz > tb[(sminy + dy) * gx + (sminx + dx)]);
      }
      cnt += (unsigned)__popcll(gh_ballot(h));
    }
    if (lane == src) tiles = cnt;
  }
  if (t < N) {
    if (tiles > 0) {
      float rgb[3];
      unsigned cl = 0;
      if (in.colors_precomp) {
        gh_blended_rgb(in, flags, i, rgb);
      } else {                                       // SH colours: evaluated by gh_sh_colour_fwd_kernel (gh_sh.hip)
        const float4 c4 = sh_rgb[n];
        rgb[0] = c4.x; rgb[1] = c4.y; rgb[2] = c4.z; cl = __float_as_uint(c4.w);
      }
      float4* grec = geom + (size_t)n * 4;           // one 64-byte line per Gaussian
      grec[0] = g0;
      grec[1] = make_float4(g1.x, op, rgb[0], rgb[1]);
      // .y = packed tile rect, .zw = tile hit mask: the post-sort gather reads one line
      grec[2] = make_float4(rgb[2], __uint_as_float(rect_bits), __uint_as_float((unsigned)hitmask), __uint_as_float((unsigned)(hitmask >> 32)));
      dkey = __float_as_uint(tz);                     // tz > 0.2: positive floats order like their bit patterns
      if (!in.colors_precomp) clamped[n] = (uint8_t)cl;   // (SH mode only: with precomputed colours nothing is clamped, nothing reads it)
    }
    // .x = instance count: the emit kernel, which walks the Gaussians in depth order, finds it in the line it reads anyway;
    // .y = view-space depth, .z = the 3-sigma tile rect (0 = none) of EVERY projected Gaussian, listed or not
    // (written for every Gaussian: a culled one keeps a stale line from an earlier call apart from this float4).
    // Round 6: this kernel's nine 4-byte stores per thread at view-major addresses (a wave = 8 Gaussians x 8 views: eight 32-byte
    // pieces per store) cost 0.85 us EACH at 8 views — depth and rect moved into this float4, the clamp flags are written in SH mode
    // only and the culling opacity for lists that a refresh will re-use only: 37.9 -> 34.5 us (same-box A/B).
    geom[(size_t)n * 4 + 3] = make_float4(__uint_as_float(tiles), tz, __uint_as_float(rect_bits), 0.0f);
    if (flags & GH_FLAG_STATIC_LISTS) cull_bound_out[n] = cull_bound;   // the opacity its tiles were culled with (guard of gh_forward_refresh)
    tiles_touched[n] = tiles;
    depth_key[n] = dkey;
    depth_val[n] = (uint32_t)n;
    if (radii) radii[n] = radius;
    if (tiles > 0) { k_or = dkey; k_and = dkey; }     // only Gaussians that emit instances need to be in depth order
  }
  // (OR, AND) of the block's keys: the depth sort skips a digit no two keys differ in (views at ~1 m: the top byte); and the
  // block's instance count: record slots are numbered in THIS kernel's thread order (gh_count_sorted_kernel scans the sums)
  uint32_t tsum = t < N ? tiles : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { k_or |= __shfl_xor(k_or, o); k_and &= __shfl_xor(k_and, o); tsum += __shfl_xor(tsum, o); }
  if ((threadIdx.x & 63) == 0) { s_bits[threadIdx.x >> 6] = make_uint2(k_or, k_and); s_tiles[threadIdx.x >> 6] = tsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    key_bits[bid] = make_uint2(s_bits[0].x | s_bits[1].x | s_bits[2].x | s_bits[3].x,
                                      s_bits[0].y & s_bits[1].y & s_bits[2].y & s_bits[3].y);
    block_tiles[bid] = (s_tiles[0] + s_tiles[1]) + (s_tiles[2] + s_tiles[3]);
  }
}

void gh_launch_preprocess_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, int32_t* radii, char* ws,
                              const GhLayout& L, hipStream_t s) {
  const int T = g.NV * g.tiles;
  if (g.N == 0) {                                      // nothing to project: only the counters and per-tile state
    (void)hipMemsetAsync(ws + L.counters, 0, sizeof(GhCounters), s);
    (void)hipMemsetAsync(ws + L.ranges, 0, L.tile_order - L.ranges, s);
    (void)hipMemsetAsync(ws + L.render_guard, 0, 4 + 4 * GH_BWD_CLASSES, s);      // (a refresh call over nothing goes straight to the render)
    return;
  }
  int nblk = ((g.N > T ? g.N : T) + GH_BLOCK - 1) / GH_BLOCK;
  const float* tile_bound = nullptr;
  if (in->tile_depth_bound) {                          // speculative occlusion bound: neighbourhood test first (one tiny kernel)
    hipLaunchKernelGGL(gh_tile_bound_kernel, dim3((T + GH_BLOCK - 1) / GH_BLOCK), dim3(GH_BLOCK), 0, s, (const float2*)in->tile_depth_bound,
                       g.NV, g.gx, g.gy, (float*)(ws + L.tile_bound));
    tile_bound = (const float*)(ws + L.tile_bound);
  }
  auto kern = in->cov3D_precomp ? gh_preprocess_fwd_kernel<true> : gh_preprocess_fwd_kernel<false>;
  const int n_rank = gh_order_in_projection(g) ? g.NV : 0;        // spare workgroups: the forward's launch order (see the kernel)
  hipLaunchKernelGGL(kern, dim3(nblk + n_rank), dim3(GH_BLOCK), 0, s, *in, g.P, g.NV, g.N, g.H, g.W, g.gx, g.gy,
                     d->sh_degree, d->M, d->scale_modifier, d->flags, (const float4*)(ws + L.sh_rgb), (float4*)(ws + L.geom),
                     (uint8_t*)(ws + L.clamped), (uint32_t*)(ws + L.tiles_touched),
                     (uint32_t*)(ws + L.depth_keys_a), (uint32_t*)(ws + L.depth_vals_a), (GhCounters*)(ws + L.counters), radii,
                     T, (uint2*)(ws + L.ranges), (uint32_t*)(ws + L.tile_walk), (uint2*)(ws + L.key_bits),
                     g.N < (1 << 24) ? 1.0f / (float)((d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) ? g.P : g.NV) : 0.0f,
                     (float*)(ws + L.cull_bound), tile_bound, (uint32_t*)(ws + L.block_tiles), nblk, (uint32_t*)(ws + L.tile_order), g.tiles);
}

// ------------------------------------------------------------------------------------------------
// Deterministic block-level accumulation of blend-parameter gradients: wave DPP sum, lane 63 adds the
// wave's total to its own LDS row (single writer per row), rows are combined in fixed order at the end.
__device__ __forceinline__ void gh_block_acc(float (*s_part)[64], int slot, float v) {
  float t = gh_wave_sum_to63(v);
  if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6][slot] += t;
}

// One thread per Gaussian; loops the views so gradients w.r.t. view-independent attributes are summed
// in registers/own memory in a fixed order (no atomics, bitwise reproducible).
// (SH colours only — with precomputed colours the chain-rule kernel sums the records itself, gh_sum_records.)
// Fixed-order sum of every Gaussian's per-(instance, quadrant) gradient sub-records; they sit at the consecutive record slots
// [slot_begin, slot_begin + tiles). Few registers, so the reads run at full occupancy; the chain-rule kernel then reads the 9 sums
// coalesced. The (view, Gaussian) pairs are taken in the order their record slots are numbered (round 5: the projection kernel's
// thread order, gh_count_sorted_kernel), so neighbouring quads read neighbouring stretches of sub-records (1024x1024 SH3: 111 ->
// 104 us; with slots in emit = depth order the stretches were scattered).
__global__ __launch_bounds__(GH_BLOCK) void gh_record_sum_kernel(int N, int P, int NV, float rdiv, uint32_t cap,
                                                                  const uint32_t* __restrict__ slot_begin,
                                                                  const uint32_t* __restrict__ tiles_touched,
                                                                  const float* __restrict__ inst_grad,
                                                                  const uint32_t* __restrict__ inst_flag, float4* __restrict__ gsum) {
  // FOUR lanes per (view, Gaussian), one per quadrant: the quad reads the 144 contiguous bytes of an instance's four
  // sub-records.
  // Quads walk the (view, Gaussian) pairs in the order the record slots are numbered (the projection kernel's thread order:
  // Gaussian-major, views adjacent; NV == 0: row-major, the pose batch), so neighbouring quads read neighbouring slots.
  const int t = blockIdx.x * GH_BLOCK + threadIdx.x;
  const int j = t >> 2, q = t & 3;
  const bool live = j < N;                              // whole quads are live or not; no early return (DPP below)
  uint32_t n = 0u;
  if (live) {
    if (NV == 0) n = (uint32_t)j;
    else { const uint32_t i = rdiv > 0.0f ? gh_div_small((uint32_t)j, (uint32_t)NV, rdiv) : (uint32_t)j / (uint32_t)NV; n = ((uint32_t)j - i * (uint32_t)NV) * (uint32_t)P + i; }
  }
  uint32_t o0 = 0, o1 = 0;
  if (live) {
    o0 = slot_begin[n]; o1 = o0 + tiles_touched[n];
    if (o1 > cap) o1 = cap;
    if (o0 > o1) o0 = o1;
  }
  // Compensated (Kahan) fixed-order sums: a footprint of hundreds of tiles adds thousands of signed sub-records that
  // largely cancel, and a plain fp32 running sum loses the result's low bits to the large intermediate values (VERDICT r1
  // item 9: whole-image Gaussians missed the 1e-3 gradient bar against the oracle's double accumulation). The kernel is
  // bound by its scattered reads, the 27 extra adds per record are free. Bitwise reproducible as before.
  float s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, c9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // Four instances per trip: their flag words (4 quadrant flag bytes each), then the flagged sub-records, are all requested
  // before the first is used — the loop is a chain of dependent scattered reads otherwise (flag -> record, instance after
  // instance), and most Gaussians have no more than four instances.
  for (uint32_t base = o0; base < o1; base += 4) {
    bool has[4];
    GhF3 r[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) has[i] = base + i < o1 && ((inst_flag[base + i] >> (8 * q)) & 1u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (has[i]) {
        const GhF3* p = (const GhF3*)(inst_grad + ((size_t)(base + i) * 4 + q) * GH_REC_G);
        r[i][0] = p[0]; r[i][1] = p[1]; r[i][2] = p[2];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (has[i]) {
        const float v9[9] = {r[i][0].x, r[i][0].y, r[i][0].z, r[i][1].x, r[i][1].y, r[i][1].z, r[i][2].x, r[i][2].y, r[i][2].z};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float y = v9[k] - c9[k];
          const float t = s9[k] + y;
          c9[k] = (t - s9[k]) - y;
          s9[k] = t;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) s9[k] -= c9[k];               // fold the last compensation in
  // quadrants combined in fixed order, (q0 + q1) + (q2 + q3), in every lane of the quad
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    s9[k] += gh_dpp<0xB1>(s9[k]);                            // quad_perm [1,0,3,2]
    s9[k] += gh_dpp<0x4E>(s9[k]);                            // quad_perm [2,3,0,1]
  }
  if (live && q < 3) {                                       // three lanes write the 48-byte sum record
    float4 o;
    if (q == 0) o = make_float4(s9[0], s9[1], s9[2], s9[3]);
    else if (q == 1) o = make_float4(s9[4], s9[5], s9[6], s9[7]);
    else o = make_float4(s9[8], 0.0f, 0.0f, 0.0f);
    gsum[(size_t)n * 3 + q] = o;
  }
}

// Sum over the aligned group of G = 2^lg lanes this lane belongs to, in every lane of the group; fixed order, so the
// result is bitwise reproducible. Steps 4 and 8 use the row mirrors: the quads / 8-lane groups are uniform by then, so
// the mirrored lane holds exactly the other half's sum. All lanes of the wave must be active.
__device__ __forceinline__ float gh_group_sum(float v, int lg) {
  if (lg > 0) v += gh_dpp<0xB1>(v);          // quad_perm [1,0,3,2]
  if (lg > 1) v += gh_dpp<0x4E>(v);          // quad_perm [2,3,0,1]
  if (lg > 2) v += gh_dpp<0x141>(v);         // row_half_mirror
  if (lg > 3) v += gh_dpp<0x140>(v);         // row_mirror
  if (lg > 4) v += __shfl_xor(v, 16);
  if (lg > 5) v += __shfl_xor(v, 32);
  return v;
}

// RGB_MODE is a template parameter so the colours-precomputed path does not pay the registers of the SH path.
// The record sum inside the chain-rule kernel (colours precomputed: nothing else reads the sums): the lane of a (view,
// Gaussian) adds up the sub-records of its instances itself — emit slots [o0, o1), four quadrant flags per slot, quadrants in
// order — with one float64 accumulator per moment. No 48-byte sum record is written and read back and one launch less;
// a lane has four times the reads of gh_record_sum_kernel's quad lanes, measured 55 -> 66 us for the sums alone at the
// chain rule's occupancy, against the 37 us kernel that now runs inside those waits.
__device__ __forceinline__ void gh_sum_records(uint32_t o0, uint32_t o1, const float* __restrict__ inst_grad,
                                               const uint32_t* __restrict__ inst_flag, float* s9) {
  // float64 accumulators (9 adds per sub-record where the compensated float32 sum of gh_record_sum_kernel takes 36): a
  // footprint of hundreds of tiles adds thousands of signed sub-records that largely cancel
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // one emit slot per trip (two slots' records in flight: 143 VGPRs, three waves per SIMD, no faster)
  uint32_t fl = o0 < o1 ? inst_flag[o0] : 0u;
  for (uint32_t sl = o0; sl < o1; ++sl) {
    const uint32_t f = fl;
    if (sl + 1 < o1) fl = inst_flag[sl + 1];             // the next slot's flags travel while this slot's records are summed
    GhF3 r[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((f >> (8 * q)) & 1u) {
        const GhF3* p = (const GhF3*)(inst_grad + ((size_t)sl * 4 + q) * GH_REC_G);
        r[q][0] = p[0]; r[q][1] = p[1]; r[q][2] = p[2];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((f >> (8 * q)) & 1u) {
        acc[0] += (double)r[q][0].x; acc[1] += (double)r[q][0].y; acc[2] += (double)r[q][0].z;
        acc[3] += (double)r[q][1].x; acc[4] += (double)r[q][1].y; acc[5] += (double)r[q][1].z;
        acc[6] += (double)r[q][2].x; acc[7] += (double)r[q][2].y; acc[8] += (double)r[q][2].z;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) s9[k] = (float)acc[k];
}

// Chain rule per (view, Gaussian): a group of G = 2^lg adjacent lanes shares one row of the attribute arrays and
// splits its views (lane vv takes views vv, vv + G, ...), so 8 views run 8-wide instead of as a loop of 8 in one thread
// (P threads cannot fill 256 CUs); the per-view results are combined over the group in fixed order (gh_group_sum), no
// atomics, bitwise reproducible. Pose batch (GH_FLAG_PER_VIEW_GAUSSIANS): NV*P rows, row i is seen by view i / P only
// and G = 1.
// GEOM = false (no gradient w.r.t. means / scales / rotations / means2D / xyz_b is asked for — the one-shot fit trains colour
// and opacity biases only): the chain rule through the projection is skipped and the kernel only sums over the views.
// The same for a call that wants no geometry gradient (the one-shot fit: colour and opacity only): of the nine moments only
// sum h (opacity) and the three colour moments are read — the last 16 bytes of a 36-byte sub-record, one load instead of three —
// two emit slots per trip.
struct GhF4u { float x, y, z, w; };                      // 16-byte access at 4-byte alignment (global_load_dwordx4)
__device__ __forceinline__ void gh_sum_records_colour(uint32_t o0, uint32_t o1, const float* __restrict__ inst_grad,
                                                      const uint32_t* __restrict__ inst_flag, float* s9) {
  double acc[4] = {0, 0, 0, 0};
  uint32_t fl0 = o0 < o1 ? inst_flag[o0] : 0u, fl1 = o0 + 1 < o1 ? inst_flag[o0 + 1] : 0u;
  for (uint32_t sl = o0; sl < o1; sl += 2) {
    const uint32_t f0 = fl0, f1 = fl1;
    fl0 = sl + 2 < o1 ? inst_flag[sl + 2] : 0u;
    fl1 = sl + 3 < o1 ? inst_flag[sl + 3] : 0u;
    GhF4u r[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t f = i ? f1 : f0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((f >> (8 * q)) & 1u) r[i][q] = *(const GhF4u*)(inst_grad + ((size_t)(sl + i) * 4 + q) * GH_REC_G + 5);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t f = i ? f1 : f0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if ((f >> (8 * q)) & 1u) {
          acc[0] += (double)r[i][q].x; acc[1] += (double)r[i][q].y; acc[2] += (double)r[i][q].z; acc[3] += (double)r[i][q].w;
        }
      }
    }
  }
  s9[5] = (float)acc[0]; s9[6] = (float)acc[1]; s9[7] = (float)acc[2]; s9[8] = (float)acc[3];
}

// FUSED (RGB_MODE only): the sums come straight from the render backward's sub-records (gh_sum_records). A split call's second
// half keeps its per-instance arrays cap_a entries further on (views >= v_split; unsplit: v_split = NV).
// COV (GEOM only): Sigma3D was given (GhInputs.cov3D_precomp): the chain ends at dL/dSigma (GhGrads.dL_dcov3D) instead of running on
// to scales and rotations — an instantiation of its own, so that the six extra accumulators never cost the common kernels a register.
template <bool RGB_MODE, bool GEOM, bool FUSED, bool COV = false>
__global__ __launch_bounds__(GH_BLOCK) void gh_preprocess_bwd_kernel(
    GhInputs in, GhGrads gr, int P, int NV, int H, int W, int sh_degree, int M, float mod, uint32_t flags, int lg,
    const uint32_t* __restrict__ tiles_touched, const float4* __restrict__ dmean_sh, const float4* __restrict__ gsum,
    float* __restrict__ scratch, const uint32_t* __restrict__ slot_begin, const float* __restrict__ inst_grad,
    const uint32_t* __restrict__ inst_flag, int v_split, uint32_t cap_a, uint32_t cap_b) {
  GH_WG_TIMER(3);
  __shared__ float s_part[GH_BLOCK / GH_WAVE][64];
  const bool per_view = (flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;
  const int G = 1 << lg;
  const int slot = blockIdx.x * GH_BLOCK + threadIdx.x;
  const int i = slot >> lg, vv = slot & (G - 1);        // row of the attribute arrays, lane inside its view group
  const bool live = i < (per_view ? NV * P : P);
  const int n_rounds = per_view ? 1 : (NV + G - 1) / G;  // uniform over the wave: the block sums below need every lane
  constexpr bool rgb_mode = RGB_MODE;
  const bool wpg = (flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
  const bool red_w = rgb_mode && in.blend_color_w && !wpg && gr.dL_dblend_color_w;   // global (48,) weights: block reduce
  const bool red_x = GEOM && in.blend_xyz_b && gr.dL_dblend_xyz_b;
  if (threadIdx.x < 64) {
#pragma unroll
    for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) s_part[w][threadIdx.x] = 0.0f;
  }
  __syncthreads();

  float am[3] = {0, 0, 0}, as[3] = {0, 0, 0}, aq[4] = {0, 0, 0, 0}, araw[3] = {0, 0, 0}, ao = 0.0f;
  float acov[6] = {0, 0, 0, 0, 0, 0};                   // COV only
  for (int k = 0; k < n_rounds; ++k) {
    const int v_raw = per_view ? (live ? i / P : 0) : vv + k * G;
    const bool vok = per_view || v_raw < NV;
    const int v = vok ? v_raw : 0;
    const size_t n = per_view ? (size_t)(live ? i : 0) : (size_t)v * P + (live ? i : 0);
    const uint32_t tt = live && vok ? tiles_touched[n] : 0u;
    const bool vis = tt != 0u;
    float s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (FUSED) {
      if (vis) {
        const bool hb = v >= v_split;
        const uint32_t capv = hb ? cap_b : cap_a, off = hb ? cap_a : 0u;
        uint32_t o0 = slot_begin[n], o1 = o0 + tt;
        if (o1 > capv) o1 = capv;
        if (o0 > o1) o0 = o1;
        if (GEOM) gh_sum_records(off + o0, off + o1, inst_grad, inst_flag, s9);
        else gh_sum_records_colour(off + o0, off + o1, inst_grad, inst_flag, s9);
      }
    } else if (vis) {
      const float4* r = gsum + n * 3;
      const float4 r0 = r[0], r1 = r[1]; const float r2 = r[2].x;
      s9[0] = r0.x; s9[1] = r0.y; s9[2] = r0.z; s9[3] = r0.w; s9[4] = r1.x; s9[5] = r1.y; s9[6] = r1.z; s9[7] = r1.w; s9[8] = r2;
    }
    float g_px = 0.0f, g_py = 0.0f;
    float dm[3] = {0, 0, 0};
    const float* cam = in.cams + (size_t)v * GH_CAM_FLOATS;
    const float* V = cam; const float* PM = cam + 16;
    if (!GEOM) {
      if (vis) {
        if (rgb_mode) { araw[0] += s9[6]; araw[1] += s9[7]; araw[2] += s9[8]; }
        ao += s9[5];
      }
    } else if (vis) {
      GhGeo e;
      gh_geo_forward<COV>(in, cam, i, mod, H, W, e);
      // The render backward sums the raw pixel moments of h = G * dL/dalpha per instance:
      //   s9[0..5] = sum h dx, sum h dy, sum h dx^2, sum h dx dy, sum h dy^2, sum h     (dx = px_gaussian - px_pixel)
      // the factors that are constant per (view, Gaussian) — opacity and the conic — are applied here, once, instead of
      // per pixel and list entry:  dL/dG = o * dL/dalpha,  dL/d(px,py) = -o (A Shx + B Shy, C Shy + B Shx),
      // dL/d(A,B,C) = -o (Shxx / 2, Shxy, Shyy / 2),  dL/do = Sh.
      float op = in.opacities[i];
      if (in.blend_opacity_b) op = op + in.blend_opacity_b[i];
      const float dinv = 1.0f / e.det;
      const float cA = e.c * dinv, cB = -e.b * dinv, cC = e.a * dinv;
      g_px = -op * (cA * s9[0] + cB * s9[1]);
      g_py = -op * (cC * s9[1] + cB * s9[0]);
      const float gA = -0.5f * op * s9[2], gB = -op * s9[3], gC = -0.5f * op * s9[4], g_o = s9[5];
      // ---- colour ----
      if (rgb_mode) {
        araw[0] += s9[6]; araw[1] += s9[7]; araw[2] += s9[8];
      } else {                                         // SH colours: gh_sh_colour_bwd_kernel did the colour chain
        const float4 d4 = dmean_sh[n];
        dm[0] = d4.x; dm[1] = d4.y; dm[2] = d4.z;
      }
      // ---- conic -> dilated cov2D (a,b,c) ----
      float a = e.a, b = e.b, cc = e.c, det = e.det;
      float det2inv = 1.0f / (det * det);
      float dL_da = (-cc * cc * gA + b * cc * gB - b * b * gC) * det2inv;
      float dL_dc = (-b * b * gA + a * b * gB - a * a * gC) * det2inv;
      float dL_db = (2.0f * b * cc * gA - (a * cc + b * b) * gB + 2.0f * a * b * gC) * det2inv;
      // G2 = [[da, db/2],[db/2, dc]];  dSigma(full) = T^T G2 T ; dT = 2 G2 T Sigma
      float g00 = dL_da, g01 = 0.5f * dL_db, g11 = dL_dc;
      const float* T = e.T;
      float GT[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) { GT[k] = g00 * T[k] + g01 * T[3 + k]; GT[3 + k] = g01 * T[k] + g11 * T[3 + k]; }
      float dS[9];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) dS[3 * p + q] = T[p] * GT[q] + T[3 + p] * GT[3 + q];
      float Sf[9] = {e.S[0], e.S[1], e.S[2], e.S[1], e.S[3], e.S[4], e.S[2], e.S[4], e.S[5]};
      float dT[6];
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          dT[3 * r2 + q] = 2.0f * (GT[3 * r2] * Sf[q] + GT[3 * r2 + 1] * Sf[3 + q] + GT[3 * r2 + 2] * Sf[6 + q]);
      if (COV) {
        // Sigma itself is the input: its symmetric storage (xx xy xz yy yz zz) receives both matrix positions of an off-diagonal entry
        acov[0] += dS[0]; acov[1] += dS[1] + dS[3]; acov[2] += dS[2] + dS[6]; acov[3] += dS[4]; acov[4] += dS[5] + dS[7]; acov[5] += dS[8];
      } else {
      // Sigma = M M^T, M = R diag(s): dM = 2 dSigma M
      float Mm[9];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 3; ++j) Mm[3 * p + j] = e.R[3 * p + j] * e.s[j];
      float dM[9];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          dM[3 * p + j] = 2.0f * (dS[3 * p] * Mm[j] + dS[3 * p + 1] * Mm[3 + j] + dS[3 * p + 2] * Mm[6 + j]);
      float dR[9];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float ds = dM[j] * e.R[j] + dM[3 + j] * e.R[3 + j] + dM[6 + j] * e.R[6 + j];
        as[j] += ds * mod;
#pragma unroll
        for (int p = 0; p < 3; ++p) dR[3 * p + j] = dM[3 * p + j] * e.s[j];
      }
      {
        float r = in.rotations[4 * i], x = in.rotations[4 * i + 1], y = in.rotations[4 * i + 2], z = in.rotations[4 * i + 3];
        aq[0] += 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        aq[1] += 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
        aq[2] += 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
        aq[3] += 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
      }
      }
      // ---- T = J W (the 1.3*tanfov clamp freezes the clamped view-space x / y, App. A.4-3) ----
      float dJ00 = dT[0] * V[0] + dT[1] * V[4] + dT[2] * V[8];
      float dJ02 = dT[0] * V[2] + dT[1] * V[6] + dT[2] * V[10];
      float dJ11 = dT[3] * V[1] + dT[4] * V[5] + dT[5] * V[9];
      float dJ12 = dT[3] * V[2] + dT[4] * V[6] + dT[5] * V[10];
      float tzi = 1.0f / e.tz, tz2i = tzi * tzi, tz3i = tz2i * tzi;
      float dtx = e.xclamped ? 0.0f : -e.fx * tz2i * dJ02;
      float dty = e.yclamped ? 0.0f : -e.fy * tz2i * dJ12;
      float dtz = -e.fx * tz2i * dJ00 - e.fy * tz2i * dJ11 + 2.0f * e.fx * e.cx * tz3i * dJ02 + 2.0f * e.fy * e.cy * tz3i * dJ12;
#pragma unroll
      for (int a2 = 0; a2 < 3; ++a2) dm[a2] += dtx * V[4 * a2] + dty * V[4 * a2 + 1] + dtz * V[4 * a2 + 2];
      // ---- projection path ----
      float dndcx = g_px * 0.5f * (float)W, dndcy = g_py * 0.5f * (float)H;
      float dhx = dndcx * e.winv, dhy = dndcy * e.winv;
      float dhw = -(dndcx * e.hx + dndcy * e.hy) * e.winv * e.winv;
#pragma unroll
      for (int a2 = 0; a2 < 3; ++a2) dm[a2] += dhx * PM[4 * a2] + dhy * PM[4 * a2 + 1] + dhw * PM[4 * a2 + 3];
      am[0] += dm[0]; am[1] += dm[1]; am[2] += dm[2];
      ao += g_o;
    }
    if (GEOM && live && vok && gr.dL_dmeans2D) {
      gr.dL_dmeans2D[3 * n] = vis ? g_px * 0.5f * (float)W : 0.0f;
      gr.dL_dmeans2D[3 * n + 1] = vis ? g_py * 0.5f * (float)H : 0.0f;
      gr.dL_dmeans2D[3 * n + 2] = 0.0f;
    }
    if (GEOM && red_x) { gh_block_acc(s_part, 48, dm[0]); gh_block_acc(s_part, 49, dm[1]); gh_block_acc(s_part, 50, dm[2]); }
  }

  // per-view results -> per-Gaussian sums, in every lane of the group
  if (GEOM) {
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) { am[c3] = gh_group_sum(am[c3], lg); as[c3] = gh_group_sum(as[c3], lg); }
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) aq[c4] = gh_group_sum(aq[c4], lg);
    if (COV) {
#pragma unroll
      for (int c6 = 0; c6 < 6; ++c6) acov[c6] = gh_group_sum(acov[c6], lg);
    }
  }
  ao = gh_group_sum(ao, lg);
  if (rgb_mode) {
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) araw[c3] = gh_group_sum(araw[c3], lg);
  }
  const bool writer = live && vv == 0;                  // one lane per row writes / contributes the per-Gaussian terms
  if (writer) {
    if (GEOM && gr.dL_dmeans3D) { gr.dL_dmeans3D[3 * i] = am[0]; gr.dL_dmeans3D[3 * i + 1] = am[1]; gr.dL_dmeans3D[3 * i + 2] = am[2]; }
    if (gr.dL_dopacities) gr.dL_dopacities[i] = ao;
    if (gr.dL_dblend_opacity_b && in.blend_opacity_b) gr.dL_dblend_opacity_b[i] = ao;
    if (GEOM && !COV && gr.dL_dscales) { gr.dL_dscales[3 * i] = as[0]; gr.dL_dscales[3 * i + 1] = as[1]; gr.dL_dscales[3 * i + 2] = as[2]; }
    if (GEOM && !COV && gr.dL_drotations) { gr.dL_drotations[4 * i] = aq[0]; gr.dL_drotations[4 * i + 1] = aq[1]; gr.dL_drotations[4 * i + 2] = aq[2]; gr.dL_drotations[4 * i + 3] = aq[3]; }
    if (GEOM && COV && gr.dL_dcov3D) {
#pragma unroll
      for (int c6 = 0; c6 < 6; ++c6) gr.dL_dcov3D[6 * i + c6] = acov[c6];
    }
  }
  if (rgb_mode) {
    // c' = ((c*w0 + w1) - 1) + b0  (renderer_one_shot.py:324,328): araw = sum over views of dL/dc'
    const float* w = in.blend_color_w ? in.blend_color_w + (wpg ? (size_t)(live ? i : 0) * 48 : 0) : nullptr;
    float c[3] = {0, 0, 0};
    if (live) { c[0] = in.colors_precomp[3 * i]; c[1] = in.colors_precomp[3 * i + 1]; c[2] = in.colors_precomp[3 * i + 2]; }
    if (writer) {
      if (gr.dL_dcolors) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) gr.dL_dcolors[3 * i + ch] = w ? araw[ch] * w[ch] : araw[ch];
      }
      if (in.blend_color_b && gr.dL_dblend_color_b && (flags & GH_FLAG_BLEND_COLOR_B_RGB)) {
        float* o = gr.dL_dblend_color_b + (size_t)i * 3;
        o[0] = araw[0]; o[1] = araw[1]; o[2] = araw[2];
      }
    }
    // 48-wide rows (the reference's (P,48) bias / weight samples, of which RGB mode reads 3 / 6 entries): the lanes of
    // the view group write the row together as float4s, one coalesced 192-byte row instead of 48 scalar stores per thread
    if (live) {
      if (in.blend_color_b && gr.dL_dblend_color_b && !(flags & GH_FLAG_BLEND_COLOR_B_RGB)) {
        float4* o = (float4*)(gr.dL_dblend_color_b + (size_t)i * 48);
        for (int k = vv; k < 12; k += G) o[k] = k == 0 ? make_float4(araw[0], araw[1], araw[2], 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
      if (w && wpg && gr.dL_dblend_color_w) {
        float4* o = (float4*)(gr.dL_dblend_color_w + (size_t)i * 48);
        for (int k = vv; k < 12; k += G)
          o[k] = k == 0 ? make_float4(araw[0] * c[0], araw[1] * c[1], araw[2] * c[2], araw[0])
                        : (k == 1 ? make_float4(araw[1], araw[2], 0.0f, 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
      }
    }
    if (red_w) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        gh_block_acc(s_part, ch, writer ? araw[ch] * c[ch] : 0.0f);
        gh_block_acc(s_part, 3 + ch, writer ? araw[ch] : 0.0f);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64 && (red_w || red_x))
    scratch[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
        s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
}

// Second stage of the blend-parameter reduction: one block per slot, fixed-order sum over the per-block
// partials (strided per thread, DPP per wave, 4 waves in order) => bitwise reproducible. scratch: [64 slots][nblk] (the writers'
// grid size = nblk).
__global__ __launch_bounds__(GH_BLOCK) void gh_blend_reduce_kernel(const float* __restrict__ scratch, int nblk,
                                                                    float* __restrict__ d_w, float* __restrict__ d_xyz) {
  __shared__ float s_w[GH_BLOCK / GH_WAVE];
  const int t = blockIdx.x;      // slot 0..50
  float s = 0.0f;
  // (slot-major partials, scratch[slot][block]: this block's row is contiguous — with the block-major layout of rounds 1-5 every
  //  load of the wave touched 64 different lines, 20 MB of line traffic for 0.6 MB of partials; the per-thread order of the sum is
  //  unchanged, so the result is the same bit for bit)
  for (int b = threadIdx.x; b < nblk; b += GH_BLOCK) s += scratch[(size_t)t * nblk + b];
  s = gh_wave_sum_to63(s);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
    if (d_w && t < 48) d_w[t] = tot;
    if (d_xyz && t >= 48) d_xyz[t - 48] = tot;
  }
}

void gh_launch_preprocess_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const GhGrads* gr, const char* wg, char* ws,
                              const GhLayout& L, hipStream_t s, int parts, int v_split, size_t cap_a) {
  if (g.P == 0) return;
  const bool per_view = (d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;
  const int rows = per_view ? g.N : g.P;
  int lg = 0;                                            // lanes per row: the smallest power of two >= n_views, at most 64
  if (!per_view) while ((1 << lg) < g.NV && lg < 6) ++lg;
  int nblk = (int)((((size_t)rows << lg) + GH_BLOCK - 1) / GH_BLOCK);   // <= 2 N / 256 + 1: bwd_scratch holds 64 floats per block
  const bool geom = gh_wants_geometry(in, gr);
  // colours precomputed: only the chain-rule kernel reads the record sums, and it takes them itself (FUSED); SH colours:
  // gh_sh_colour_bwd2_kernel reads the colour moments first, so the sums are a kernel of their own
  // — up to half a megapixel per view: a lane of the fused form reads all the instances of its (view, Gaussian) one after the
  // other, and their number grows with the image (3.8 on average at 512x334: 93.6 -> 80.9 us fused; 8.6 at 1024x1024: 156 ->
  // 164 us). The image size decides, so every call of a shape (and both halves of a split call) takes the same path.
  const bool rgb = in->colors_precomp != nullptr;
  const bool fused = rgb && (size_t)g.H * (size_t)g.W <= ((size_t)1 << 19);
  const bool halves = v_split >= 0;                      // a split call: the second half's instance arrays start cap_a entries in
  auto kern = rgb ? (fused ? (geom ? gh_preprocess_bwd_kernel<true, true, true> : gh_preprocess_bwd_kernel<true, false, true>)
                           : (geom ? gh_preprocess_bwd_kernel<true, true, false> : gh_preprocess_bwd_kernel<true, false, false>))
                  : (geom ? gh_preprocess_bwd_kernel<false, true, false> : gh_preprocess_bwd_kernel<false, false, false>);
  if (in->cov3D_precomp && geom)                         // Sigma3D given: the chain ends at dL/dSigma
    kern = rgb ? (fused ? gh_preprocess_bwd_kernel<true, true, true, true> : gh_preprocess_bwd_kernel<true, true, false, true>)
               : gh_preprocess_bwd_kernel<false, true, false, true>;
  const int nblk_n = (int)(((size_t)g.N * 4 + GH_BLOCK - 1) / GH_BLOCK);
  if ((parts & GH_PBWD_RECORD_SUM) && !fused)
    hipLaunchKernelGGL(gh_record_sum_kernel, dim3(nblk_n), dim3(GH_BLOCK), 0, s, g.N, g.P, per_view ? 0 : g.NV,
                       g.N < (1 << 24) && !per_view ? 1.0f / (float)g.NV : 0.0f, (uint32_t)g.cap,
                       (const uint32_t*)(wg + L.slot_begin), (const uint32_t*)(wg + L.tiles_touched),
                       (const float*)(ws + L.inst_grad), (const uint32_t*)(ws + L.inst_flag), (float4*)(ws + L.grad_sums));
  if (!(parts & GH_PBWD_CHAIN)) return;
  const int nblk_sh = gh_launch_sh_colour_bwd(d, g, in, gr, wg, ws, L, s);     // SH mode only; no-op with colors_precomp
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(GH_BLOCK), 0, s, *in, *gr, g.P, g.NV, g.H, g.W,
                     d->sh_degree, d->M, d->scale_modifier, d->flags, lg,
                     (const uint32_t*)(wg + L.tiles_touched), (const float4*)(ws + L.dmean_sh),
                     (const float4*)(ws + L.grad_sums), (float*)(ws + L.bwd_scratch),
                     (const uint32_t*)(wg + L.slot_begin), (const float*)(ws + L.inst_grad), (const uint32_t*)(ws + L.inst_flag),
                     halves ? v_split : g.NV, (uint32_t)(halves ? cap_a : (size_t)g.cap), (uint32_t)(halves ? (size_t)g.cap - cap_a : 0));
  const bool wpg = (d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
  float* dw = (in->blend_color_w && !wpg) ? gr->dL_dblend_color_w : nullptr;
  float* dx = (geom && in->blend_xyz_b) ? gr->dL_dblend_xyz_b : nullptr;
  if (nblk_sh > 0) {                     // SH mode: the colour-weight partials come from the SH kernel's own scratch
    hipLaunchKernelGGL(gh_blend_reduce_kernel, dim3(48), dim3(GH_BLOCK), 0, s, (const float*)(ws + L.sh_scratch), nblk_sh, dw, (float*)nullptr);
    dw = nullptr;
  }
  if (dw || dx)
    hipLaunchKernelGGL(gh_blend_reduce_kernel, dim3(51), dim3(GH_BLOCK), 0, s, (const float*)(ws + L.bwd_scratch), nblk, dw, dx);
}
