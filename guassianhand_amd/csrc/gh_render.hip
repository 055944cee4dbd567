// gh_render.hip — per-tile alpha compositing and its backward (App. A.4).
//
// Work decomposition (wave64-native): one 256-thread workgroup per 16x16 tile, one wave per 8x8 pixel
// quadrant, one pixel per lane. Each wave walks the tile's depth-sorted list on its own:
//   * 64 list entries at a time are staged in REGISTERS (lane l holds entry base+l; the next 64 are
//     prefetched while the current ones are consumed),
//   * every lane tests "its" Gaussian against the wave's quadrant with a conservative bounding box of the
//     alpha >= 1/255 ellipse; a ballot turns that into a 64-bit hit mask (wavefront compaction),
//   * the wave iterates the set bits only; the selected Gaussian's 9 floats are broadcast with
//     v_readlane (SGPR operands, no LDS traffic) and the per-pixel blend is branch-free.
// The forward pass therefore uses no LDS and no barriers, and a quadrant retires as soon as its 64 pixels
// are saturated. Culling never changes results: the exact per-pixel tests of App. A.3 still decide.
//
// Backward: each pixel replays its list back to front (same staging / culling); the 9 per-Gaussian partial
// gradients are summed over the 64 lanes with DPP and stored by the quadrant's wave as ITS sub-record of the
// (tile, Gaussian) instance, at the instance's emit slot, plus a flag byte. No LDS, no barriers, no atomics:
// the per-Gaussian kernel adds each Gaussian's flagged sub-records (contiguous slots) in fixed order, so the
// gradients are bitwise reproducible.
#include "gh_internal.h"

__device__ __forceinline__ void gh_tile_coords(int blk, int gx, int tiles, int& v, int& tx, int& ty) {
  v = blk / tiles;
  int t = blk - v * tiles;
  ty = t / gx; tx = t - ty * gx;
}

__device__ __forceinline__ float gh_bcast(float v, int lane) {   // lane is wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Conservative test: can Gaussian (g0 = px,py,A,B; g1 = C,opacity,..) reach alpha >= 1/255 anywhere in the
// pixel block [qx0, qx0+ext] x [qy0, qy0+ext]? alpha >= 1/255  <=>  d^T Q d <= 2 ln(255 o); the axis-aligned
// extent of that ellipse is sqrt(2 tau Q^-1_xx), sqrt(2 tau Q^-1_yy). Margins absorb the approximate log/sqrt;
// anything non-finite answers "hit". A false "hit" only costs time, never changes a pixel.
__device__ __forceinline__ bool gh_block_hit(const float4& g0, const float4& g1, float qx0, float qy0, float ext) {
  const float o = g1.y;
  if (!(o >= 1.0f / 255.0f)) return false;          // alpha = min(.99, o*exp(p<=0)) <= o < 1/255 everywhere
  const float tau = __logf(255.0f * o) * 1.0001f + 1e-3f;
  const float det = g0.z * g1.x - g0.w * g0.w;
  const float k = 2.0f * tau / det;
  const float hx = __builtin_amdgcn_sqrtf(k * g1.x) * 1.001f + 0.02f;
  const float hy = __builtin_amdgcn_sqrtf(k * g0.z) * 1.001f + 0.02f;
  const bool miss = (g0.x + hx < qx0) || (g0.x - hx > qx0 + ext) || (g0.y + hy < qy0) || (g0.y - hy > qy0 + ext);
  return !miss;                                       // NaN extents compare false -> hit
}
__device__ __forceinline__ bool gh_quadrant_hit(const float4& g0, const float4& g1, float qx0, float qy0) {
  return gh_block_hit(g0, g1, qx0, qy0, 7.0f);
}

// ------------------------------------------------------------------------------------------------
struct GhBatch {          // 64 list entries staged in registers: lane l holds entry base+l
  float4 a, b;            // (px, py, A, B), (C, opacity, r, g)
  float cb;               // b
};

__device__ __forceinline__ void gh_load_batch(GhBatch& t, const float4* __restrict__ r0, const float4* __restrict__ r1,
                                              const float* __restrict__ r2, int idx, int total) {
  // unconditional loads from a clamped index (total >= 1): no exec-mask branch, so the compiler can wait
  // for exactly this batch (counted vmcnt) while the next one stays in flight. Entries >= total are
  // masked out of the hit ballot by the callers.
  const int i = idx < total ? idx : total - 1;
  t.a = r0[i]; t.b = r1[i]; t.cb = r2[i];
}

// ---- forward: wave = 4x4 pixels x 4 depth slots ---------------------------------------------------------
// lane = 4*pixel + slot. Per trip the wave takes the next (up to) four surviving list entries; slot s of every
// pixel evaluates entry s (alpha evaluation is the expensive, state-independent part and runs 4-wide), then the
// T / colour recurrence is applied in list order by walking the quad: step s computes the blend for every lane
// and quad_perm-broadcasts slot s's result, so all four lanes of a pixel always hold the pixel's current state.
// Arithmetic per pixel is exactly the sequential recurrence of App. A.3 (same operations, same order).
struct GhPixelFwd {
  float T, C0, C1, C2;
  uint32_t last;
  int done;               // 0 / 1 (kept as int so it can travel through DPP)
};

template <int S>
__device__ __forceinline__ float gh_quad_bcast(float v) {     // value of the quad's lane S, in all 4 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), S * 0x55, 0xF, 0xF, false));
}
template <int S>
__device__ __forceinline__ int gh_quad_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, S * 0x55, 0xF, 0xF, false);
}

__device__ __forceinline__ float gh_lane_fetch(float v, int src_lane_x4) {   // per-lane source (LDS crossbar, no LDS memory)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane_x4, __builtin_bit_cast(int, v)));
}

template <int S>
__device__ __forceinline__ void gh_fwd_chain_step(GhPixelFwd& p, float alpha, bool ok, float r, float g, float b, uint32_t pos1) {
  const bool valid = (p.done == 0) && ok;
  const float test_T = p.T * (1.0f - alpha);
  const bool stop = valid && (test_T < 0.0001f);
  const bool blend = valid && !stop;
  const float w = blend ? alpha * p.T : 0.0f;       // fma(c, 0, C) == C exactly: masked lanes keep their bits
  const float nC0 = fmaf(r, w, p.C0), nC1 = fmaf(g, w, p.C1), nC2 = fmaf(b, w, p.C2);
  const float nT = blend ? test_T : p.T;
  const uint32_t nlast = blend ? pos1 : p.last;
  const int ndone = (p.done != 0 || stop) ? 1 : 0;
  p.T = gh_quad_bcast<S>(nT);
  p.C0 = gh_quad_bcast<S>(nC0); p.C1 = gh_quad_bcast<S>(nC1); p.C2 = gh_quad_bcast<S>(nC2);
  p.last = (uint32_t)gh_quad_bcast_i<S>((int)nlast);
  p.done = gh_quad_bcast_i<S>(ndone);
}

// Consume one staged batch front to back, four entries per trip. Returns true when all 16 pixels are finished.
__device__ __forceinline__ bool gh_fwd_consume(const GhBatch& t, int base, int total, int lane, int slot, float fbx0, float fby0,
                                               float pxf, float pyf, GhPixelFwd& p) {
  const bool hit = (base + lane < total) && gh_block_hit(t.a, t.b, fbx0, fby0, 3.0f);
  uint64_t mask = __ballot(hit);
  while (mask) {
    // next four set bits, ascending (wave-uniform scalar work)
    const int j0 = __builtin_ctzll(mask); mask &= mask - 1;
    const int n1 = mask != 0; const int j1 = n1 ? __builtin_ctzll(mask) : j0; mask &= mask - 1;
    const int n2 = mask != 0; const int j2 = n2 ? __builtin_ctzll(mask) : j0; mask &= mask - 1;
    const int n3 = mask != 0; const int j3 = n3 ? __builtin_ctzll(mask) : j0; mask &= mask - 1;
    const int myj = slot == 0 ? j0 : (slot == 1 ? j1 : (slot == 2 ? j2 : j3));
    const bool have = slot == 0 || (slot == 1 && n1) || (slot == 2 && n2) || (slot == 3 && n3);
    const int src = myj << 2;
    const float gpx = gh_lane_fetch(t.a.x, src), gpy = gh_lane_fetch(t.a.y, src), cA = gh_lane_fetch(t.a.z, src);
    const float cB = gh_lane_fetch(t.a.w, src), cC = gh_lane_fetch(t.b.x, src), op = gh_lane_fetch(t.b.y, src);
    const float r = gh_lane_fetch(t.b.z, src), g = gh_lane_fetch(t.b.w, src), bl = gh_lane_fetch(t.cb, src);
    const float dx = gpx - pxf, dy = gpy - pyf;
    const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
    const float alpha = fminf(0.99f, op * gh_exp(fminf(power, 0.0f)));
    const bool ok = have && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
    const uint32_t pos1 = (uint32_t)(base + myj + 1);
    gh_fwd_chain_step<0>(p, alpha, ok, r, g, bl, pos1);
    gh_fwd_chain_step<1>(p, alpha, ok, r, g, bl, pos1);
    gh_fwd_chain_step<2>(p, alpha, ok, r, g, bl, pos1);
    gh_fwd_chain_step<3>(p, alpha, ok, r, g, bl, pos1);
    if (__all(p.done != 0)) return true;
  }
  return false;
}

// grid = 4 blocks per tile (one per 8x8 quadrant), 4 waves per block (one per 4x4 pixel block); no LDS, no barriers.
__global__ __launch_bounds__(GH_BLOCK) void gh_render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const float4* __restrict__ r0,
    const float4* __restrict__ r1, const float* __restrict__ r2, const float* __restrict__ cams, int H, int W, int gx,
    int tiles, float* __restrict__ image, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    uint32_t* __restrict__ tile_walk) {
  int v, tx, ty;
  const int tile = (int)tile_order[blockIdx.x >> 2];      // heaviest tiles are launched first
  const int quad = blockIdx.x & 3;
  gh_tile_coords(tile, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int slot = lane & 3, pi = lane >> 2;
  const int bx0 = tx * GH_TILE + (quad & 1) * 8 + (wid & 1) * 4, by0 = ty * GH_TILE + (quad >> 1) * 8 + (wid >> 1) * 4;
  const int x = bx0 + (pi & 3), y = by0 + (pi >> 2);
  const bool inside = x < W && y < H;
  const float pxf = (float)x, pyf = (float)y, fbx0 = (float)bx0, fby0 = (float)by0;
  const uint2 range = ranges[tile];
  const int total = (int)(range.y - range.x);
  r0 += range.x; r1 += range.x; r2 += range.x;

  GhPixelFwd p;
  p.T = 1.0f; p.C0 = p.C1 = p.C2 = 0.0f; p.last = 0; p.done = inside ? 0 : 1;
  if (total > 0 && !__all(p.done != 0)) {
    // two register sets in flight: while one batch is consumed the next one is already being loaded
    GhBatch A, B;
    gh_load_batch(A, r0, r1, r2, lane, total);
    for (int base = 0; base < total; base += 2 * GH_WAVE) {
      gh_load_batch(B, r0, r1, r2, base + GH_WAVE + lane, total);
      if (gh_fwd_consume(A, base, total, lane, slot, fbx0, fby0, pxf, pyf, p)) break;
      if (base + GH_WAVE >= total) break;
      gh_load_batch(A, r0, r1, r2, base + 2 * GH_WAVE + lane, total);
      if (gh_fwd_consume(B, base + GH_WAVE, total, lane, slot, fbx0, fby0, pxf, pyf, p)) break;
    }
  }
  if (total > 0) {                                   // walked length of the tile = max n_contrib (orders the backward)
    uint32_t m = p.last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m; }
    if (lane == 0 && m > 0) atomicMax(&tile_walk[tile], m);
  }
  if (inside && slot == 0) {
    const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
    const size_t pix = ((size_t)v * H + y) * W + x;
    final_T[pix] = p.T;
    n_contrib[pix] = p.last;
    float* img = image + (size_t)v * 3 * H * W + (size_t)y * W + x;
    img[0] = fmaf(p.T, bg[0], p.C0);
    img[(size_t)H * W] = fmaf(p.T, bg[1], p.C1);
    img[(size_t)2 * H * W] = fmaf(p.T, bg[2], p.C2);
  }
}

void gh_launch_render_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, float* image, char* ws, const GhLayout& L,
                          hipStream_t s) {
  hipLaunchKernelGGL(gh_render_fwd_kernel, dim3(4 * g.NV * g.tiles), dim3(GH_BLOCK), 0, s, (const uint2*)(ws + L.ranges),
                     (const uint32_t*)(ws + L.tile_order), (const float4*)(ws + L.inst_r0), (const float4*)(ws + L.inst_r1), (const float*)(ws + L.inst_r2),
                     in->cams, g.H, g.W, g.gx, g.tiles, image, (float*)(ws + L.final_T), (uint32_t*)(ws + L.n_contrib),
                     (uint32_t*)(ws + L.tile_walk));
}

// ------------------------------------------------------------------------------------------------
struct GhPixelBwd {
  float T, last_alpha, lc0, lc1, lc2, ar0, ar1, ar2;
};

struct GhBwdCtx {          // per-pixel constants of the reverse walk
  float pxf, pyf, T_final, bg_dot, d0, d1, d2;
  int last;
};

struct GhBwdEval {         // state-independent part of one (pixel, Gaussian) step
  float cA, cB, cC, op, cr, cg, cbl, dx, dy, G, alpha, inv1ma;
  bool contrib;
};

__device__ __forceinline__ GhBwdEval gh_bwd_eval(const GhBatch& t, int j, int pos, const GhBwdCtx& c) {
  GhBwdEval e;
  const float gpx = gh_bcast(t.a.x, j), gpy = gh_bcast(t.a.y, j);
  e.cA = gh_bcast(t.a.z, j); e.cB = gh_bcast(t.a.w, j); e.cC = gh_bcast(t.b.x, j); e.op = gh_bcast(t.b.y, j);
  e.cr = gh_bcast(t.b.z, j); e.cg = gh_bcast(t.b.w, j); e.cbl = gh_bcast(t.cb, j);
  e.dx = gpx - c.pxf; e.dy = gpy - c.pyf;
  const float power = -0.5f * (e.cA * e.dx * e.dx + e.cC * e.dy * e.dy) - e.cB * e.dx * e.dy;
  e.G = gh_exp(fminf(power, 0.0f));
  e.alpha = fminf(0.99f, e.op * e.G);
  e.contrib = (pos < c.last) && (power <= 0.0f) && (e.alpha >= 1.0f / 255.0f);
  // 1/(1-alpha): v_rcp_f32 (<= 1 ulp). Gradients carry a 1e-3 rtol; only the forward is bit-exact.
  e.inv1ma = __builtin_amdgcn_rcpf(1.0f - e.alpha);
  return e;
}

// State-dependent part: advance the pixel's (T, colour-behind) recurrence, reduce the 9 partials over the
// wave and let lane 63 store the quadrant's sub-record + flag byte.
__device__ __forceinline__ void gh_bwd_apply(const GhBwdEval& e, const GhBwdCtx& c, GhPixelBwd& p, uint32_t slot, int lane,
                                             float* __restrict__ my_rec, uint8_t* __restrict__ my_flag) {
  const float Tn = p.T * e.inv1ma;
  const float n0 = p.last_alpha * p.lc0 + (1.0f - p.last_alpha) * p.ar0;
  const float n1 = p.last_alpha * p.lc1 + (1.0f - p.last_alpha) * p.ar1;
  const float n2 = p.last_alpha * p.lc2 + (1.0f - p.last_alpha) * p.ar2;
  float dL_dalpha = (e.cr - n0) * c.d0 + (e.cg - n1) * c.d1 + (e.cbl - n2) * c.d2;
  dL_dalpha *= Tn;
  dL_dalpha += (-c.T_final * e.inv1ma) * c.bg_dot;
  const float dL_dG = e.op * dL_dalpha;                 // straight-through the 0.99 clamp (App. A.4-2)
  const float gdx = e.G * e.dx, gdy = e.G * e.dy;
  const float dchannel_dcolor = e.alpha * Tn;
  float r[9];
  r[0] = dL_dG * (-gdx * e.cA - gdy * e.cB);
  r[1] = dL_dG * (-gdy * e.cC - gdx * e.cB);
  r[2] = -0.5f * gdx * e.dx * dL_dG;
  r[3] = -gdx * e.dy * dL_dG;
  r[4] = -0.5f * gdy * e.dy * dL_dG;
  r[5] = e.G * dL_dalpha;
  r[6] = dchannel_dcolor * c.d0; r[7] = dchannel_dcolor * c.d1; r[8] = dchannel_dcolor * c.d2;
#pragma unroll
  for (int q = 0; q < 9; ++q) r[q] = gh_wave_sum_to63(e.contrib ? r[q] : 0.0f);
  // per-lane state advances only where the pixel really blended this Gaussian
  p.T = e.contrib ? Tn : p.T;
  p.ar0 = e.contrib ? n0 : p.ar0; p.ar1 = e.contrib ? n1 : p.ar1; p.ar2 = e.contrib ? n2 : p.ar2;
  p.lc0 = e.contrib ? e.cr : p.lc0; p.lc1 = e.contrib ? e.cg : p.lc1; p.lc2 = e.contrib ? e.cbl : p.lc2;
  p.last_alpha = e.contrib ? e.alpha : p.last_alpha;
  if (lane == 63) {
    float4* rec = (float4*)(my_rec + (size_t)slot * (4 * GH_REC));
    rec[0] = make_float4(r[0], r[1], r[2], r[3]);
    rec[1] = make_float4(r[4], r[5], r[6], r[7]);
    rec[2] = make_float4(r[8], 0.0f, 0.0f, 0.0f);
    my_flag[(size_t)slot * 4] = 1;
  }
}

// Consume one staged batch back to front, two hits per trip (independent alpha evaluations overlap).
__device__ __forceinline__ void gh_bwd_consume(const GhBatch& t, int sbase, int wave_last, int lane, float fqx0, float fqy0,
                                               const GhBwdCtx& c, GhPixelBwd& p, const uint32_t* __restrict__ slots,
                                               float* __restrict__ my_rec, uint8_t* __restrict__ my_flag) {
  const int idx = sbase + lane;
  uint64_t mask = __ballot((idx < wave_last) && gh_quadrant_hit(t.a, t.b, fqx0, fqy0));
  if (mask == 0) return;
  const uint32_t slot_l = slots[idx < wave_last ? idx : wave_last - 1];
  while (mask) {
    const int j0 = 63 - __builtin_clzll(mask);          // back to front
    mask &= ~(1ull << j0);
    const bool two = mask != 0;
    const int j1 = two ? 63 - __builtin_clzll(mask) : j0;
    if (two) mask &= ~(1ull << j1);
    const GhBwdEval e0 = gh_bwd_eval(t, j0, sbase + j0, c);
    GhBwdEval e1 = gh_bwd_eval(t, j1, sbase + j1, c);
    e1.contrib = e1.contrib && two;
    if (__any(e0.contrib))                               // wave-uniform
      gh_bwd_apply(e0, c, p, (uint32_t)__builtin_amdgcn_readlane((int)slot_l, j0), lane, my_rec, my_flag);
    if (__any(e1.contrib))
      gh_bwd_apply(e1, c, p, (uint32_t)__builtin_amdgcn_readlane((int)slot_l, j1), lane, my_rec, my_flag);
  }
}

// One wave per 8x8 quadrant, fully autonomous (no LDS, no barriers): the quadrant's partial record of
// instance `slot` goes to inst_grad[slot][quadrant][0..8] and inst_flag[slot][quadrant] = 1 (flags are
// zeroed per call). The per-Gaussian kernel adds the flagged sub-records in fixed (slot, quadrant) order.
__global__ __launch_bounds__(GH_BLOCK) void gh_render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const uint32_t* __restrict__ sorted_slot,
    const float4* __restrict__ r0, const float4* __restrict__ r1, const float* __restrict__ r2, const float* __restrict__ cams,
    int H, int W, int gx, int tiles, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dimage, float* __restrict__ inst_grad, uint8_t* __restrict__ inst_flag) {
  int v, tx, ty;
  const int tile = (int)tile_order[blockIdx.x];
  gh_tile_coords(tile, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int qx0 = tx * GH_TILE + (wid & 1) * 8, qy0 = ty * GH_TILE + (wid >> 1) * 8;
  const int x = qx0 + (lane & 7), y = qy0 + (lane >> 3);
  const bool inside = x < W && y < H;
  const float pxf = (float)x, pyf = (float)y, fqx0 = (float)qx0, fqy0 = (float)qy0;
  const uint2 range = ranges[tile];
  if (range.y == range.x) return;
  r0 += range.x; r1 += range.x; r2 += range.x;
  const uint32_t* slots = sorted_slot + range.x;

  const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
  float T_final = 1.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
  int last = 0;
  if (inside) {
    const size_t pix = ((size_t)v * H + y) * W + x;
    T_final = final_T[pix];
    last = (int)n_contrib[pix];
    const float* dimg = dL_dimage + (size_t)v * 3 * H * W + (size_t)y * W + x;
    d0 = dimg[0]; d1 = dimg[(size_t)H * W]; d2 = dimg[(size_t)2 * H * W];
  }
  GhBwdCtx c;
  c.pxf = pxf; c.pyf = pyf; c.T_final = T_final; c.d0 = d0; c.d1 = d1; c.d2 = d2; c.last = last;
  c.bg_dot = bg[0] * d0 + bg[1] * d1 + bg[2] * d2;
  int wave_last = last;                    // list positions >= wave_last were blended by no pixel of this quadrant
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(wave_last, o); wave_last = t > wave_last ? t : wave_last; }
  if (wave_last == 0) return;

  GhPixelBwd p;
  p.T = T_final; p.last_alpha = 0.0f; p.lc0 = p.lc1 = p.lc2 = 0.0f; p.ar0 = p.ar1 = p.ar2 = 0.0f;
  float* my_rec = inst_grad + wid * GH_REC;
  uint8_t* my_flag = inst_flag + wid;
  // batches of 64 from the back; two register sets keep the next batch in flight
  const int nb = (wave_last + GH_WAVE - 1) / GH_WAVE;
  GhBatch A, B;
  gh_load_batch(A, r0, r1, r2, (nb - 1) * GH_WAVE + lane, wave_last);
  for (int k = nb - 1; k >= 0; k -= 2) {
    if (k >= 1) gh_load_batch(B, r0, r1, r2, (k - 1) * GH_WAVE + lane, wave_last);
    gh_bwd_consume(A, k * GH_WAVE, wave_last, lane, fqx0, fqy0, c, p, slots, my_rec, my_flag);
    if (k < 1) break;
    if (k >= 2) gh_load_batch(A, r0, r1, r2, (k - 2) * GH_WAVE + lane, wave_last);
    gh_bwd_consume(B, (k - 1) * GH_WAVE, wave_last, lane, fqx0, fqy0, c, p, slots, my_rec, my_flag);
  }
}

void gh_launch_render_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const float* dL_dimage, char* ws,
                          const GhLayout& L, hipStream_t s) {
  if (g.cap == 0) return;
  (void)hipMemsetAsync(ws + L.inst_flag, 0, (size_t)g.cap * 4, s);
  gh_launch_tile_order_bwd(g, ws, L, s);
  hipLaunchKernelGGL(gh_render_bwd_kernel, dim3(g.NV * g.tiles), dim3(GH_BLOCK), 0, s, (const uint2*)(ws + L.ranges),
                     (const uint32_t*)(ws + L.tile_order_bwd), (const uint32_t*)(ws + L.vals_a), (const float4*)(ws + L.inst_r0), (const float4*)(ws + L.inst_r1),
                     (const float*)(ws + L.inst_r2), in->cams, g.H, g.W, g.gx, g.tiles,
                     (const float*)(ws + L.final_T), (const uint32_t*)(ws + L.n_contrib), dL_dimage,
                     (float*)(ws + L.inst_grad), (uint8_t*)(ws + L.inst_flag));
}
