// gh_render.hip — per-tile alpha compositing and its backward (App. A.4).
//
// Work decomposition (wave64-native): four 256-thread workgroups per 16x16 tile (one per 8x8 quadrant), one wave per
// 4x4 pixel block, lane = 4*pixel + depth slot. Each wave walks the tile's depth-sorted list on its own:
//   * 64 list entries at a time are staged in REGISTERS (lane l holds entry base+l; the next 64 are prefetched while
//     the current ones are consumed),
//   * every lane tests the bit of the wave's block in "its" entry's 16-bit block mask (precomputed once per instance
//     by gh_ranges_kernel with the exact ellipse/rectangle test); a ballot compacts the survivors,
//   * each trip takes the next four survivors: slot s of every pixel evaluates entry s (ds_bpermute fetch), and the
//     T / colour recurrence runs in exact list order over the quad as DPP-fused prefix products and sums.
// The forward uses no LDS memory and no barriers, and a block retires as soon as its 16 pixels are saturated.
// Culling never changes results: the exact per-pixel tests of App. A.3 still decide.
//
// Backward: same mapping back to front; the nine per-Gaussian partial gradients are summed over the 16 pixels of a
// slot with DPP + the LDS crossbar, the four waves of a quadrant are combined through a double-buffered LDS stage
// (one barrier per 64 entries) and stored as the quadrant's sub-record of the (tile, Gaussian) instance at the
// instance's emit slot, plus a flag byte. No atomics: the per-Gaussian kernel adds each Gaussian's flagged
// sub-records (contiguous slots) in fixed order, so the gradients are bitwise reproducible.
#include "gh_internal.h"

// blockIdx -> (work item, quadrant) with the four quadrant workgroups of an item on ONE XCD: workgroups are dealt
// round-robin over the 8 XCDs (b and b + 8 share one), so inside every run of 32 workgroups item = r & 7, quadrant = r >> 3.
// The four read the same list records: one L2 fetches them instead of four. A speed choice only (placement is not a
// contract); the items left over when the count is not a multiple of 8 use the plain mapping.
__device__ __forceinline__ void gh_item_quad(uint32_t b, uint32_t n_items_grid, uint32_t& item, uint32_t& quad) {
  const uint32_t full = (n_items_grid >> 3) << 5;        // workgroups covered by whole groups of 8 items
  if (b < full) { const uint32_t r = b & 31u; item = ((b >> 5) << 3) + (r & 7u); quad = r >> 3; }
  else { item = b >> 2; quad = b & 3u; }
}

__device__ __forceinline__ void gh_tile_coords(int blk, int gx, int tiles, int& v, int& tx, int& ty) {
  v = blk / tiles;
  int t = blk - v * tiles;
  ty = t / gx; tx = t - ty * gx;
}

// ------------------------------------------------------------------------------------------------
struct GhBatch {          // 64 list entries staged in registers: lane l holds entry base+l
  float4 a, b;            // (px, py, A, B), (C, opacity, r, g)
  float cb;               // b
  uint32_t blocks;        // 16-bit mask: 4x4-pixel blocks of the tile the entry can reach (gh_ranges_kernel)
};

__device__ __forceinline__ void gh_load_batch(GhBatch& t, const float4* __restrict__ r0, const float4* __restrict__ r1,
                                              const float2* __restrict__ r2, int idx, int total) {
  // unconditional loads from a clamped index (total >= 1): no exec-mask branch, so the compiler can wait
  // for exactly this batch (counted vmcnt) while the next one stays in flight. Entries >= total are
  // masked out of the hit ballot by the callers.
  const int i = idx < total ? idx : total - 1;
  t.a = r0[i]; t.b = r1[i];
  const float2 c = r2[i];
  t.cb = c.x; t.blocks = __float_as_uint(c.y);
}

// ---- forward: wave = 4x4 pixels x 4 depth slots ---------------------------------------------------------
// lane = 4*pixel + slot. Per trip the wave takes the next (up to) four surviving list entries; slot s of every
// pixel evaluates entry s (alpha evaluation is the expensive, state-independent part and runs 4-wide), then the
// T / colour recurrence is applied in list order by walking the quad: step s computes the blend for every lane
// and quad_perm-broadcasts slot s's result, so all four lanes of a pixel always hold the pixel's current state.
// Arithmetic per pixel is exactly the sequential recurrence of App. A.3 (same operations, same order).
struct GhPixelFwd {
  float T, C0, C1, C2, A;  // A: accumulated alpha = the mask channel (colour 1, bg 0)
  uint32_t last;
  int done;               // 0 / 1 (kept as int so it can travel through DPP)
};

template <int S>
__device__ __forceinline__ float gh_quad_bcast(float v) {     // value of the quad's lane S, in all 4 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), S * 0x55, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int gh_quad_perm_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int S>
__device__ __forceinline__ int gh_quad_bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, S * 0x55, 0xF, 0xF, false);
}

// acc = (((acc + m[slot 0]) + m[slot 1]) + m[slot 2]) + m[slot 3] over the lane's quad: four DPP-fused adds. Written as
// assembly because the compiler lowers the equivalent intrinsics to v_mov 0 + v_mov_dpp + v_add per term (3x the
// instructions). The s_nop covers the VALU-write -> DPP-read hazard on m (2 wait states); acc is the non-DPP operand.
__device__ __forceinline__ void gh_quad_accumulate(float& acc, float m) {
  asm("s_nop 1\n\t"
               "v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %1, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %1, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
               : "+v"(acc) : "v"(m));
}

// b1 = B*m[0] + a[0], b2 = b1*m[1] + a[1], b3 = b2*m[2] + a[2], b4 = b3*m[3] + a[3]  (x[k] = the quad's lane k): the backward's
// affine recurrence over the four slots as eight DPP-fused instructions.
__device__ __forceinline__ void gh_quad_affine4(float B, float m, float a, float& b1, float& b2, float& b3, float& b4) {
  float t;
  asm("s_nop 1\n\t"
      "v_mul_f32_dpp %4, %5, %7 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %6, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %4, %5, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %6, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %4, %5, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %6, %4 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %4, %5, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %6, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"
      : "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4), "=&v"(t) : "v"(m), "v"(a), "v"(B));
}

// Value for this lane's slot: a0 in slot-0 lanes, a1 in slot-1 lanes, ... (lane = 4*pixel + slot, so the lane sets are
// the constant masks 0x1111.., 0x2222.., ...). Three v_cndmask with literal lane masks; the compiler otherwise turns
// the nested ?: into exec-mask branches.
__device__ __forceinline__ float gh_slot_select(float a0, float a1, float a2, float a3) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %5\n\t"
               "v_cndmask_b32_e64 %0, %0, %3, %6\n\t"
               "v_cndmask_b32_e64 %0, %0, %4, %7"
               : "=&v"(r) : "v"(a0), "v"(a1), "v"(a2), "v"(a3),
                 "s"(0x2222222222222222ull), "s"(0x4444444444444444ull), "s"(0x8888888888888888ull));
  return r;
}

// P1 = T * f[slot 0], P2 = P1 * f[slot 1], P3 = P2 * f[slot 2] (f[slot k] = the quad's lane k): the sequential prefix
// products of the recurrence as three DPP-fused multiplies (assembly for the same reason as above).
__device__ __forceinline__ void gh_quad_prefix3(float T, float f, float& P1, float& P2, float& P3) {
  asm("s_nop 1\n\t"
               "v_mul_f32_dpp %0, %3, %4 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
               "v_mul_f32_dpp %1, %3, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
               "v_mul_f32_dpp %2, %3, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf"
               : "=&v"(P1), "=&v"(P2), "=&v"(P3) : "v"(f), "v"(T));
}

__device__ __forceinline__ float gh_lane_fetch(float v, int src_lane_x4) {   // per-lane source (LDS crossbar, no LDS memory)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane_x4, __builtin_bit_cast(int, v)));
}

// The (up to) four lowest / highest set bits of a wave-uniform hit mask, removed from the mask, as lane numbers.
// s_ff1 / s_flbit + s_bitset0 (which uses the low 6 bits of its operand, so the -1 of an empty mask is harmless): two to
// three scalar instructions per pick instead of the six of `ctz`, `mask &= mask - 1`, compare and select.
// With fewer than four bits set the trailing picks are -1 (low) / -64 (high): callers gate on the count taken beforehand,
// and in the packed form (j0 | j1 << 8 | j2 << 16 | j3 << 24) << 2 such a value only spills into the bytes of LATER picks,
// which are invalid too, so no masking is needed.
__device__ __forceinline__ void gh_pop4_low(uint64_t& mask, int& j0, int& j1, int& j2, int& j3) {
  asm("s_ff1_i32_b64 %1, %0\n\ts_bitset0_b64 %0, %1\n\t"
      "s_ff1_i32_b64 %2, %0\n\ts_bitset0_b64 %0, %2\n\t"
      "s_ff1_i32_b64 %3, %0\n\ts_bitset0_b64 %0, %3\n\t"
      "s_ff1_i32_b64 %4, %0\n\ts_bitset0_b64 %0, %4"
      : "+s"(mask), "=&s"(j0), "=&s"(j1), "=&s"(j2), "=&s"(j3));
}
__device__ __forceinline__ void gh_pop4_high(uint64_t& mask, int& j0, int& j1, int& j2, int& j3) {
  asm("s_flbit_i32_b64 %1, %0\n\ts_xor_b32 %1, %1, 63\n\ts_bitset0_b64 %0, %1\n\t"
      "s_flbit_i32_b64 %2, %0\n\ts_xor_b32 %2, %2, 63\n\ts_bitset0_b64 %0, %2\n\t"
      "s_flbit_i32_b64 %3, %0\n\ts_xor_b32 %3, %3, 63\n\ts_bitset0_b64 %0, %3\n\t"
      "s_flbit_i32_b64 %4, %0\n\ts_xor_b32 %4, %4, 63\n\ts_bitset0_b64 %0, %4"
      : "+s"(mask), "=&s"(j0), "=&s"(j1), "=&s"(j2), "=&s"(j3) : : "scc");
}

// Consume one staged batch front to back, four entries per trip. Returns true when all 16 pixels are finished.
template <bool ALPHA>
__device__ __forceinline__ bool gh_fwd_consume(const GhBatch& t, int base, int total, int lane, int slot, int blk,
                                               float pxf, float pyf, GhPixelFwd& p) {
  const uint32_t slot8 = (uint32_t)slot * 8u;
  const bool hit = (base + lane < total) && ((t.blocks >> blk) & 1u);
  uint64_t mask = gh_ballot(hit);
  bool finished = false;                                             // set (and the mask cleared) inside the rare stop branch, so
  while (mask) {                                                     // the common path's loop control is one scalar compare
    // next four set bits, ascending (wave-uniform scalar work)
    const int nh = __builtin_popcountll(mask);                      // entries left in this batch (>= 1)
    int j0, j1, j2, j3;
    gh_pop4_low(mask, j0, j1, j2, j3);
    // the four entry lanes travel as bytes of one scalar: a lane extracts its slot's with a single v_bfe
    const uint32_t packed4 = ((uint32_t)j0 | ((uint32_t)j1 << 8) | ((uint32_t)j2 << 16) | ((uint32_t)j3 << 24)) << 2;
    const int src = (int)((packed4 >> slot8) & 0xFFu);              // 4 * entry lane = ds_bpermute address
    const bool have = slot < nh;
    const float gpx = gh_lane_fetch(t.a.x, src), gpy = gh_lane_fetch(t.a.y, src), cA = gh_lane_fetch(t.a.z, src);
    const float cB = gh_lane_fetch(t.a.w, src), cC = gh_lane_fetch(t.b.x, src), op = gh_lane_fetch(t.b.y, src);
    const float r = gh_lane_fetch(t.b.z, src), g = gh_lane_fetch(t.b.w, src), bl = gh_lane_fetch(t.cb, src);
    const float dx = gpx - pxf, dy = gpy - pyf;
    const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
    const float alpha = fminf(0.99f, op * gh_exp(fminf(power, 0.0f)));
    const bool ok = have && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
    // The recurrence collapses to DPP-fused prefix products / sums over the quad, in exact list order.
    const bool valid = (p.done == 0) && ok;
    const float f = valid ? 1.0f - alpha : 1.0f;                   // x*1 == x: skipped entries leave T bit-identical
    float P1, P2, P3;                                              // T before slot 1, 2, 3: three DPP-fused multiplies
    gh_quad_prefix3(p.T, f, P1, P2, P3);
    const float Pm = gh_slot_select(p.T, P1, P2, P3);             // T seen by this lane's entry
    const float Pn = Pm * f;                 // ... and right after it: the same product as P_{slot+1} (own f == its broadcast)
    const float P4 = gh_quad_bcast<3>(Pn);   // T after the trip
    // Early stop (App. A.3): the FIRST entry of a pixel with T(1-alpha) < 1e-4 is not blended and ends the pixel.
    // Up to and including that entry the prefix products above are exactly the sequential ones, so the stop slot, the
    // entries blended before it and the T they leave behind are all read off the same values; flags of later slots
    // (computed from products that never happen) are masked by the first one.
    bool blend = valid;
    float Tn = P4;
    const uint64_t sb = gh_ballot(valid && Pn < 0.0001f);
    if (sb) {                                                      // wave-uniform, rare
      const uint32_t qb = (uint32_t)(sb >> (lane & 60)) & 0xFu;   // stop flags of this pixel's four slots
      blend = valid && ((qb & ((2u << slot) - 1u)) == 0u);         // no stop at or before this slot
      Tn = (qb & 1u) ? p.T : ((qb & 2u) ? P1 : ((qb & 4u) ? P2 : ((qb & 8u) ? P3 : P4)));   // T right before the stop
      if (qb) p.done = 1;
      if (__all(p.done != 0)) { finished = true; mask = 0; }        // every pixel of the block is saturated: last trip
    }
    const float w = blend ? alpha * Pm : 0.0f;                     // C + c*0 == C exactly
    const float m0 = r * w, m1 = g * w, m2 = bl * w;
    gh_quad_accumulate(p.C0, m0);              // C = (((C + m[slot 0]) + m[slot 1]) + m[slot 2]) + m[slot 3], in list order
    gh_quad_accumulate(p.C1, m1);
    gh_quad_accumulate(p.C2, m2);
    if (ALPHA) gh_quad_accumulate(p.A, w);
    p.T = Tn;
    // n_contrib: every lane remembers the last entry of ITS slot that was blended; the pixel's value is the maximum
    // over its four lanes, taken once after the walk.
    p.last = blend ? (uint32_t)(base + 1) + (uint32_t)(src >> 2) : p.last;   // per LANE (positions ascend); quad max at the end
  }
  return finished;
}

// grid = 4 blocks per tile (one per 8x8 quadrant), 4 waves per block (one per 4x4 pixel block); no LDS, no barriers.
// ALPHA: also accumulate the mask channel (colour 1, bg 0) — SURVEY §8 f-2.
template <bool ALPHA>
__global__ __launch_bounds__(GH_BLOCK) void gh_render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const float4* __restrict__ r0,
    const float4* __restrict__ r1, const float2* __restrict__ r2, const float* __restrict__ cams, int H, int W, int gx,
    int tiles, float* __restrict__ image, float* __restrict__ alpha_img, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_walk, float4* __restrict__ ckpt_rgb,
    float4* __restrict__ final_C, uint2* __restrict__ items, GhCounters* __restrict__ ctr) {
  int v, tx, ty;
  uint32_t item_idx, quad_u;
  gh_item_quad(blockIdx.x, gridDim.x >> 2, item_idx, quad_u);
  const int tile = (int)tile_order[item_idx];             // heaviest tiles are launched first
  const int quad = (int)quad_u;
  gh_tile_coords(tile, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int slot = lane & 3, pi = lane >> 2;
  const int bx0 = tx * GH_TILE + (quad & 1) * 8 + (wid & 1) * 4, by0 = ty * GH_TILE + (quad >> 1) * 8 + (wid >> 1) * 4;
  const int x = bx0 + (pi & 3), y = by0 + (pi >> 2);
  const bool inside = x < W && y < H;
  const float pxf = (float)x, pyf = (float)y;
  const int blk = ((quad >> 1) * 2 + (wid >> 1)) * 4 + (quad & 1) * 2 + (wid & 1);      // this wave's bit in the block masks
  const uint2 range = ranges[tile];
  const int total = (int)(range.y - range.x);
  r0 += range.x; r1 += range.x; r2 += range.x;
  // state checkpoints for the segmented backward: slot of (tile, position m*GH_SEGMENT) = range.x/GH_SEGMENT + tile + m - 1
  // (disjoint between tiles because floor(a+b) >= floor(a) + floor(b)); pixel = row-major index inside the tile
  const size_t ck0 = ((size_t)(range.x / GH_SEGMENT) + (size_t)tile) * 256 + (size_t)((y - ty * GH_TILE) * GH_TILE + (x - tx * GH_TILE));

  GhPixelFwd p;
  p.T = 1.0f; p.C0 = p.C1 = p.C2 = p.A = 0.0f; p.last = 0; p.done = inside ? 0 : 1;
  if (total > 0 && !__all(p.done != 0)) {
    // two register sets in flight: while one batch is consumed the next one is already being loaded
    GhBatch A, B;
    gh_load_batch(A, r0, r1, r2, lane, total);
    for (int base = 0; base < total; base += 2 * GH_WAVE) {
      gh_load_batch(B, r0, r1, r2, base + GH_WAVE + lane, total);
      if (gh_fwd_consume<ALPHA>(A, base, total, lane, slot, blk, pxf, pyf, p)) break;
      if (base + GH_WAVE >= total) break;
      gh_load_batch(A, r0, r1, r2, base + 2 * GH_WAVE + lane, total);
      if (gh_fwd_consume<ALPHA>(B, base + GH_WAVE, total, lane, slot, blk, pxf, pyf, p)) break;
      const int next = base + 2 * GH_WAVE;              // wave-uniform: a checkpoint every GH_SEGMENT entries (rare)
      if ((next % GH_SEGMENT) == 0 && next < total && inside && slot == 0) {
        const size_t ck = ck0 + (size_t)(next / GH_SEGMENT - 1) * 256;
        ckpt_rgb[ck] = make_float4(p.T, p.C0, p.C1, p.C2);
      }
    }
  }
  {                                                  // n_contrib of the pixel = max over the four slot lanes of its quad
    const uint32_t a = (uint32_t)gh_quad_perm_i<0xB1>((int)p.last);      // quad_perm [1,0,3,2]
    p.last = a > p.last ? a : p.last;
    const uint32_t b = (uint32_t)gh_quad_perm_i<0x4E>((int)p.last);      // quad_perm [2,3,0,1]
    p.last = b > p.last ? b : p.last;
  }
  if (total > 0) {                                   // walked length of the tile = max n_contrib over its 16 waves
    uint32_t m = p.last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m; }
    if (lane == 0) {
      // The LAST of the tile's 16 waves (4 quadrant blocks x 4) appends the tile's backward work items, one per depth
      // segment of the walked prefix: the list is in the order the forward finished the tiles. The backward takes it from
      // the end, so the tiles that ran longest start first; the order only affects scheduling, never results.
      // Ordering without fences (an agent-scope release would write back the whole L2): only relaxed agent-scope RMW
      // atomics carry the data; each returns its old value, so waiting for the return means it has been performed.
      const uint32_t prev_max = __hip_atomic_fetch_max(&tile_walk[tile], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" :: "v"(prev_max) : "memory");
      uint32_t* done = tile_walk + (size_t)gridDim.x / 4;          // completion counters follow the T walk entries
      const uint32_t prev_done = __hip_atomic_fetch_add(&done[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev_done == 4u * (GH_BLOCK / GH_WAVE) - 1u) {
        const uint32_t w = __hip_atomic_fetch_max(&tile_walk[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // final value
        const uint32_t nseg = (w + GH_SEGMENT - 1u) / GH_SEGMENT;
        if (nseg) {
          const uint32_t pos = __hip_atomic_fetch_add(&ctr->reserved[1], nseg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (uint32_t k = 0; k < nseg; ++k) items[pos + k] = make_uint2((uint32_t)tile, k);
        }
      }
    }
  }
  if (inside && slot == 0) {
    const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
    const size_t pix = ((size_t)v * H + y) * W + x;
    final_T[pix] = p.T;
    n_contrib[pix] = p.last;
    final_C[pix] = make_float4(p.C0, p.C1, p.C2, 0.0f);
    float* img = image + (size_t)v * 3 * H * W + (size_t)y * W + x;
    img[0] = fmaf(p.T, bg[0], p.C0);
    img[(size_t)H * W] = fmaf(p.T, bg[1], p.C1);
    img[(size_t)2 * H * W] = fmaf(p.T, bg[2], p.C2);
    if (ALPHA) alpha_img[pix] = fmaf(p.T, 0.0f, p.A);
  }
}

void gh_launch_render_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, float* image, float* alpha, char* ws,
                          const GhLayout& L, hipStream_t s) {
  const dim3 grid(4 * g.NV * g.tiles), block(GH_BLOCK);
  const uint2* ranges = (const uint2*)(ws + L.ranges);
  const uint32_t* order = (const uint32_t*)(ws + L.tile_order);
  const float4* r0 = (const float4*)(ws + L.inst_r0); const float4* r1 = (const float4*)(ws + L.inst_r1);
  const float2* r2 = (const float2*)(ws + L.inst_r2);
  float* fT = (float*)(ws + L.final_T); uint32_t* nc = (uint32_t*)(ws + L.n_contrib); uint32_t* tw = (uint32_t*)(ws + L.tile_walk);
  float4* ck = (float4*)(ws + L.ckpt_rgb); float4* fC = (float4*)(ws + L.final_C);
  uint2* items = (uint2*)(ws + L.bwd_items); GhCounters* ctr = (GhCounters*)(ws + L.counters);
  if (alpha)
    hipLaunchKernelGGL(gh_render_fwd_kernel<true>, grid, block, 0, s, ranges, order, r0, r1, r2, in->cams, g.H, g.W, g.gx,
                       g.tiles, image, alpha, fT, nc, tw, ck, fC, items, ctr);
  else
    hipLaunchKernelGGL(gh_render_fwd_kernel<false>, grid, block, 0, s, ranges, order, r0, r1, r2, in->cams, g.H, g.W, g.gx,
                       g.tiles, image, alpha, fT, nc, tw, ck, fC, items, ctr);
}

// ------------------------------------------------------------------------------------------------
// ---- backward: wave = 4x4 pixels x 4 depth slots, block = one 8x8 quadrant ----------------------------------
// Same lane mapping as the forward (lane = 4*pixel + slot); entries are taken four per trip from the back.
// Per pixel the reverse recurrence carries only (T, Bs): Bs = dL/dpixel . (colour behind). The oracle's colour behind
// obeys B <- alpha_k*c_k + (1-alpha_k)*B, which is linear, and dL/dalpha only needs d . B, so one scalar chain
// Bs <- alpha_k*(d . c_k) + (1-alpha_k)*Bs replaces the per-channel ones; the mask channel (A = 1 - T_final) rides on the
// background term. The recurrence walks the quad with quad_perm broadcasts; every lane keeps the state that
// was current at ITS slot and then evaluates its nine partial gradients once. They are summed over the 16 pixels
// of the slot (row_shr:4, row_shr:8, then xor-16 / xor-32 through the LDS crossbar) and the four waves of the
// quadrant are combined through a double-buffered LDS stage, one barrier per 64 list entries, in fixed wave
// order: lane l of the flushing wave owns entry l and writes the quadrant's sub-record + flag byte.
struct GhStateBwd { float T, Bs; };   // Bs = dL/dpixel . (colour behind), the only form the colour behind is needed in

__device__ __forceinline__ float gh_slot_sum16(float v, int lane) {   // sum over the 16 pixels of this lane's slot
  v += gh_dpp<0x114>(v);                                              // row_shr:4
  v += gh_dpp<0x118>(v);                                              // row_shr:8 -> lanes 12..15 of each row: row sums
  v += gh_lane_fetch(v, (lane ^ 16) << 2);
  v += gh_lane_fetch(v, (lane ^ 32) << 2);                            // lanes 12..15 (+16k): totals per slot
  return v;
}

// r[q] += row_shr:4, then += row_shr:8, for nine values: lanes 12..15 of every row end up with the row's sums per slot.
// Assembly because the compiler splits four of the eighteen fused DPP adds into v_mov_dpp + v_add. The two steps of a
// value are nine instructions apart (VALU write -> DPP read needs two wait states); s_nop covers the producer of r[8].
__device__ __forceinline__ void gh_row_sum9(float (&r)[9]) {
#define GH_RS(N, SH) "v_add_f32_dpp %" #N ", %" #N ", %" #N " row_shr:" #SH " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
  asm("s_nop 1\n\t"
      GH_RS(0, 4) GH_RS(1, 4) GH_RS(2, 4) GH_RS(3, 4) GH_RS(4, 4) GH_RS(5, 4) GH_RS(6, 4) GH_RS(7, 4) GH_RS(8, 4)
      GH_RS(0, 8) GH_RS(1, 8) GH_RS(2, 8) GH_RS(3, 8) GH_RS(4, 8) GH_RS(5, 8) GH_RS(6, 8) GH_RS(7, 8) GH_RS(8, 8)
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]));
#undef GH_RS
}

template <bool ALPHA>
__global__ __launch_bounds__(GH_BLOCK) void gh_render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint2* __restrict__ items, const GhCounters* __restrict__ ctr,
    const uint32_t* __restrict__ sorted_slot,
    const float4* __restrict__ r0, const float4* __restrict__ r1, const float2* __restrict__ r2, const float* __restrict__ cams,
    int H, int W, int gx, int tiles, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float4* __restrict__ ckpt_rgb, const float4* __restrict__ final_C,
    const float* __restrict__ dL_dimage, const float* __restrict__ dL_dalpha_img, const float* __restrict__ upstream_scale,
    float* __restrict__ inst_grad, uint8_t* __restrict__ inst_flag) {
  __shared__ float s_part[2][GH_BLOCK / GH_WAVE][GH_WAVE][GH_REC];    // [buffer][wave][entry][9]  (18 KB: 8 blocks per CU)
  __shared__ uint64_t s_mask[2][GH_BLOCK / GH_WAVE];                  // entries a wave wrote
  __shared__ int s_qlast;
  int v, tx, ty;
  const uint32_t n_items = ctr->reserved[1];             // written by the forward; the grid is sized for the list's capacity
  uint32_t item_idx, quad_u;
  gh_item_quad(blockIdx.x, gridDim.x >> 2, item_idx, quad_u);
  if (item_idx >= n_items) return;
  const uint2 item = items[n_items - 1u - item_idx];     // (tile, depth segment): the tiles the forward finished last go first
  const int tile = (int)item.x;
  const int seg_lo = (int)item.y * GH_SEGMENT, seg_hi = seg_lo + GH_SEGMENT;
  const int quad = (int)quad_u;
  gh_tile_coords(tile, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int slot = lane & 3, pi = lane >> 2;
  const uint32_t slot8 = (uint32_t)slot * 8u;
  const int bx0 = tx * GH_TILE + (quad & 1) * 8 + (wid & 1) * 4, by0 = ty * GH_TILE + (quad >> 1) * 8 + (wid >> 1) * 4;
  const int x = bx0 + (pi & 3), y = by0 + (pi >> 2);
  const bool inside = x < W && y < H;
  const float pxf = (float)x, pyf = (float)y;
  const int blk = ((quad >> 1) * 2 + (wid >> 1)) * 4 + (quad & 1) * 2 + (wid & 1);
  const uint2 range = ranges[tile];
  if (range.y == range.x) return;
  r0 += range.x; r1 += range.x; r2 += range.x;
  const uint32_t* slots = sorted_slot + range.x;

  const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
  float T_final = 1.0f, d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, dM = 0.0f;
  int last = 0;
  if (inside) {
    const size_t pix = ((size_t)v * H + y) * W + x;
    T_final = final_T[pix];
    last = (int)n_contrib[pix];
    const float* dimg = dL_dimage + (size_t)v * 3 * H * W + (size_t)y * W + x;
    d0 = dimg[0]; d1 = dimg[(size_t)H * W]; d2 = dimg[(size_t)2 * H * W];
    if (ALPHA) dM = dL_dalpha_img[pix];             // fused mask channel: colour 1, background 0
    if (upstream_scale) {                           // dL/dloss of a scalar loss, applied here instead of in a pass of its own
      const float us = upstream_scale[0];
      d0 *= us; d1 *= us; d2 *= us; dM *= us;
    }
  }
  // background term and (ALPHA) the mask channel: both are multiples of T_final / (1 - alpha_k)
  const float bg_dot = (bg[0] * d0 + bg[1] * d1 + bg[2] * d2) - dM;
  int wave_last = last;                    // list positions >= wave_last were blended by no pixel of this 4x4 block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(wave_last, o); wave_last = t > wave_last ? t : wave_last; }
  if (tid == 0) s_qlast = 0;
  __syncthreads();
  if (lane == 0) atomicMax(&s_qlast, wave_last);
  __syncthreads();
  const int qlast = s_qlast;               // ... by no pixel of the quadrant
  if (qlast <= seg_lo) return;             // block-uniform: the quadrant blended nothing inside this depth segment
  const int qend = qlast < seg_hi ? qlast : seg_hi;

  // State behind the segment. A pixel whose last blended entry lies inside (or before) the segment starts from
  // (T_final, nothing behind), as the unsegmented walk does. A pixel that blends entries behind the cut starts from
  // the forward's own state at the cut: T = transmittance in front of entry seg_hi, colour behind it =
  // (final colour - colour accumulated in front of the cut) / T  — exact T instead of T_final divided back up.
  GhStateBwd st;
  st.T = T_final; st.Bs = 0.0f;
  if (inside && last > seg_hi) {
    const size_t ck = ((size_t)(range.x / GH_SEGMENT) + (size_t)tile + (size_t)item.y) * 256 +
                      (size_t)((y - ty * GH_TILE) * GH_TILE + (x - tx * GH_TILE));
    const float4 c = ckpt_rgb[ck];
    const float4 fc = final_C[((size_t)v * H + y) * W + x];
    const float iT = 1.0f / c.x;           // > 1e-4: the pixel was still open at the cut
    st.T = c.x;
    st.Bs = fmaf(d2, (fc.z - c.w) * iT, fmaf(d1, (fc.y - c.z) * iT, d0 * ((fc.x - c.y) * iT)));
  }
  const int nb = (qend - seg_lo + GH_WAVE - 1) / GH_WAVE;
  const uint32_t row3_slot_bit = lane >= 48 ? 1u << slot : 0u;     // the lanes that hold a slot's totals after the 16-pixel sums
  GhBatch cur, nxt;
  gh_load_batch(cur, r0, r1, r2, seg_lo + (nb - 1) * GH_WAVE + lane, qend);
  for (int k = nb - 1; k >= 0; --k) {
    const int buf = k & 1;
    const int sbase = seg_lo + k * GH_WAVE;
    gh_load_batch(nxt, r0, r1, r2, seg_lo + (k > 0 ? (k - 1) * GH_WAVE : 0) + lane, qend);   // next batch in flight
    uint64_t processed = 0;
    uint64_t mask = gh_ballot((sbase + lane < wave_last) && ((cur.blocks >> blk) & 1u));
    while (mask) {
      // next four set bits, descending (back to front); wave-uniform scalar work
      const int nh = __builtin_popcountll(mask);                     // entries left in this batch (>= 1)
      int j0, j1, j2, j3;
      const uint64_t before = mask;
      gh_pop4_high(mask, j0, j1, j2, j3);
      const uint64_t picked = before ^ mask;                             // the (up to) four entries of this trip
      const uint32_t packed4 = ((uint32_t)j0 | ((uint32_t)j1 << 8) | ((uint32_t)j2 << 16) | ((uint32_t)j3 << 24)) << 2;
      const int src = (int)((packed4 >> slot8) & 0xFFu);            // one v_bfe: 4 * this slot's entry lane
      const int myj = src >> 2;
      const bool have = slot < nh;
      const float gpx = gh_lane_fetch(cur.a.x, src), gpy = gh_lane_fetch(cur.a.y, src), cA = gh_lane_fetch(cur.a.z, src);
      const float cB = gh_lane_fetch(cur.a.w, src), cC = gh_lane_fetch(cur.b.x, src), op = gh_lane_fetch(cur.b.y, src);
      const float cr = gh_lane_fetch(cur.b.z, src), cg = gh_lane_fetch(cur.b.w, src), cbl = gh_lane_fetch(cur.cb, src);
      const float dx = gpx - pxf, dy = gpy - pyf;
      const float power = -0.5f * (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
      const float G = gh_exp(fminf(power, 0.0f));
      const float alpha = fminf(0.99f, op * G);
      const bool contrib = have && (sbase + myj < last) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
      const uint64_t cm = gh_ballot(contrib);
      if (cm == 0) continue;                                           // wave-uniform: nothing blended by this block
      // 1/(1-alpha): v_rcp_f32 (<= 1 ulp). Gradients carry a 1e-3 rtol; only the forward is bit-exact.
      // Reverse recurrence over the quad as DPP-fused prefix products / affine updates, in exact processing order:
      //   T <- T * f,  B <- B * m + a   with (f, m, a) = (1/(1-alpha), 1-alpha, alpha*c) where the pixel blended the
      //   entry and (1, 1, 0) otherwise (x*1 and x+0 leave the state bit-identical). One select does it all: with the
      //   EFFECTIVE alpha = 0 for pixels that did not blend the entry, m = 1, f = rcp(1) = 1 and a = 0 fall out exactly.
      const float ae = contrib ? alpha : 0.0f;
      const float m = 1.0f - ae;
      const float inv1ma = __builtin_amdgcn_rcpf(m);      // v_rcp_f32 (<= 1 ulp; exact for 1). Gradients carry a 1e-3 rtol.
      const float f = inv1ma;
      // The three colour channels only ever enter dL/dalpha through their dot product with dL/dpixel, and the recurrence
      // of the colour behind is linear, so ONE scalar chain carries Bs = d . B:  Bs <- Bs * m + ae * (d . c).
      const float e = fmaf(d2, cbl, fmaf(d1, cg, d0 * cr));                             // d . c of this lane's entry
      const float ace = ae * e;
      const float T1 = st.T * gh_quad_bcast<0>(f), T2 = T1 * gh_quad_bcast<1>(f), T3 = T2 * gh_quad_bcast<2>(f),
                  T4 = T3 * gh_quad_bcast<3>(f);
      const float mTn = gh_slot_select(T1, T2, T3, T4);                                  // T right after this lane's entry
      float mBs;
      {
        float b1, b2, b3, b4;
        gh_quad_affine4(st.Bs, m, ace, b1, b2, b3, b4);
        mBs = gh_slot_select(st.Bs, b1, b2, b3);                                         // d . (colour behind this lane's entry)
        st.Bs = b4;
      }
      st.T = T4;
      // the mask channel A = 1 - T_final has dA/dalpha_k = T_final / (1 - alpha_k): it rides on the background term
      float dL_dalpha = (e - mBs) * mTn;
      dL_dalpha += (-T_final * inv1ma) * bg_dot;
      // pixels that did not blend the entry contribute nothing: every partial below is a product with dL_dalpha or
      // dchannel_dcolor (all other factors are finite), so zeroing these two replaces nine selects
      dL_dalpha = contrib ? dL_dalpha : 0.0f;
      // Raw pixel moments of h = G * dL/dalpha; opacity and conic factors are constant per instance and are applied
      // after all sums, once per (view, Gaussian), by gh_preprocess_bwd_kernel (the 0.99 clamp is straight-through,
      // App. A.4-2): 6 multiplies here instead of 20.
      const float h = G * dL_dalpha;
      const float hx = h * dx, hy = h * dy;
      const float dchannel_dcolor = ae * mTn;
      float r[9];
      r[0] = hx; r[1] = hy;
      r[2] = hx * dx; r[3] = hx * dy; r[4] = hy * dy;
      r[5] = h;
      r[6] = dchannel_dcolor * d0; r[7] = dchannel_dcolor * d1; r[8] = dchannel_dcolor * d2;
      // Sum every value over the 16 pixels of each slot. In-row half with DPP: lanes 12..15 of each row then hold the row's
      // sums for slots 0..3. Before the cross-row half (the LDS crossbar, by far the most expensive step) four values are
      // PACKED into one register — value 4g+1 / +2 / +3 moves to lanes 8..11 / 4..7 / 0..3 of its row with bank-masked
      // row shifts — so the xor-16 / xor-32 butterfly runs 3 times per trip instead of 9.
      gh_row_sum9(r);
      float R[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        int pk = __builtin_bit_cast(int, r[4 * g]);
        if (g < 2) {
          pk = __builtin_amdgcn_update_dpp(pk, __builtin_bit_cast(int, r[4 * g + 1]), 0x104, 0xF, 0x4, false);   // row_shl:4  -> bank 2
          pk = __builtin_amdgcn_update_dpp(pk, __builtin_bit_cast(int, r[4 * g + 2]), 0x108, 0xF, 0x2, false);   // row_shl:8  -> bank 1
          pk = __builtin_amdgcn_update_dpp(pk, __builtin_bit_cast(int, r[4 * g + 3]), 0x10C, 0xF, 0x1, false);   // row_shl:12 -> bank 0
        }
        float v = __builtin_bit_cast(float, pk);
        v += gh_lane_fetch(v, (lane ^ 16) << 2);
        v += gh_lane_fetch(v, (lane ^ 32) << 2);             // every row: bank b holds the total of value 4g + 3 - b, slot = lane & 3
        R[g] = v;
      }
      // slots in which at least one pixel blended its entry get a partial record (wave-uniform bookkeeping, scalar):
      // act = OR of the contribution ballot over the 16 pixels -> one bit per slot. Nearly always every picked entry is
      // active, and then the processed set simply grows by the picked bits.
      uint32_t act = (uint32_t)cm | (uint32_t)(cm >> 32);
      act |= act >> 16; act |= act >> 8; act |= act >> 4; act &= 15u;
      if (act == (nh >= 4 ? 15u : (1u << nh) - 1u)) processed |= picked;
      else {
        if (act & 1u) processed |= 1ull << j0;
        if (act & 2u) processed |= 1ull << j1;
        if (act & 4u) processed |= 1ull << j2;
        if (act & 8u) processed |= 1ull << j3;
      }
      if (act & row3_slot_bit) {                                        // row 3: lane 48 + 4b + slot holds values 3-b, 7-b (and 8 for b = 3)
        float* pr = &s_part[buf][wid][myj][0];
        const int q0 = 3 - ((lane >> 2) & 3);
        pr[q0] = R[0];
        pr[4 + q0] = R[1];
        if (lane >= 60) pr[8] = R[2];
      }
    }
    if (lane == 0) s_mask[buf][wid] = processed;
    __syncthreads();
    if (wid == (k & 3)) {                       // flush batch k: lane l owns entry l; fixed wave order => reproducible
      const int pos = sbase + lane;
      if (pos < qend) {
        float s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        bool any = false;
#pragma unroll
        for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) {
          if ((s_mask[buf][w] >> lane) & 1ull) {
            any = true;
#pragma unroll
            for (int q = 0; q < 9; ++q) s9[q] += s_part[buf][w][lane][q];
          }
        }
        if (any) {
          const uint32_t sl = slots[pos];
          GhF3* rec = (GhF3*)(inst_grad + ((size_t)sl * 4 + quad) * GH_REC_G);
          rec[0] = GhF3{s9[0], s9[1], s9[2]};
          rec[1] = GhF3{s9[3], s9[4], s9[5]};
          rec[2] = GhF3{s9[6], s9[7], s9[8]};
          inst_flag[(size_t)sl * 4 + quad] = 1;
        }
      }
    }
    cur = nxt;
  }
}

void gh_launch_render_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const float* dL_dimage,
                          const float* dL_dalpha, const float* upstream_scale, char* ws, const GhLayout& L, hipStream_t s) {
  if (g.cap == 0) return;
  // inst_flag was cleared by gh_ranges_kernel; repeated backwards set the same flags again (same n_contrib).
  // The work list (tile, depth segment) was written by the forward's last wave of every tile.
  const dim3 grid(4 * (unsigned)g.n_items), block(GH_BLOCK);      // capacity of the work list; surplus blocks exit at once
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, (const uint2*)(ws + L.ranges), (const uint2*)(ws + L.bwd_items),
                       (const GhCounters*)(ws + L.counters),
                       (const uint32_t*)(ws + L.sorted_slot), (const float4*)(ws + L.inst_r0), (const float4*)(ws + L.inst_r1),
                       (const float2*)(ws + L.inst_r2), in->cams, g.H, g.W, g.gx, g.tiles, (const float*)(ws + L.final_T),
                       (const uint32_t*)(ws + L.n_contrib), (const float4*)(ws + L.ckpt_rgb),
                       (const float4*)(ws + L.final_C), dL_dimage, dL_dalpha, upstream_scale, (float*)(ws + L.inst_grad),
                       (uint8_t*)(ws + L.inst_flag));
  };
  if (dL_dalpha) launch(gh_render_bwd_kernel<true>); else launch(gh_render_bwd_kernel<false>);
}
