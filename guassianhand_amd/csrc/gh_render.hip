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
// Backward (rewritten in round 2): the transposed mapping — lane = list entry. A wave owns one 8x8 quadrant of a (tile,
// 256-entry depth segment) work item and loops over the pixels; the reverse recurrence is two prefix scans per pixel (DPP);
// every lane accumulates the nine gradient moments of ITS entry in registers and the per-(entry, block) sums meet in
// wave-private LDS rows. One wave per workgroup, no barriers; the wave stores the quadrant's 36-byte sub-record of every
// blended entry at the instance's emit slot, plus a flag byte. No atomics: gh_record_sum_kernel adds each Gaussian's
// flagged sub-records (contiguous slots) in fixed order, so the gradients are bitwise reproducible. (Round 3 built the
// four quadrant waves of an item as ONE workgroup that writes one record per instance: the quadrants' loads differ by 2.1x
// on average, the barrier made the kernel 58 us slower for 25 us saved in the record sum —
// tools/experiments/r3_bwd_coupled_quadrants.patch, profiles/r3_bwd_coupled_ab.txt.)
#include "gh_internal.h"
#include <stdlib.h>

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
  // SEEN only (GhOutputs.tile_depth_seen): the walk goes on VIRTUALLY behind the stop — nothing is blended — until the
  // transmittance is below GH_SEEN_T as well: a pixel that ended just under the stop threshold is one rounding away from needing
  // more of its list, a pixel that also passes the stricter threshold is not.
  float vT;               // virtual transmittance behind the stop
  uint32_t stopq;         // 4 * (list position + 1) of the entry that took vT below GH_SEEN_T (this lane's slot only), 0 = none
  int vdone;              // the virtual walk has ended too
};
#define GH_SEEN_T 0.00005f

// (quad_perm gives every lane a source lane, so no `old` value is needed: mov_dpp, not update_dpp(0, ..) — the latter costs a
// v_mov 0 in front of every use)
template <int S>
__device__ __forceinline__ float gh_quad_bcast(float v) {     // value of the quad's lane S, in all 4 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), S * 0x55, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int gh_quad_perm_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false);
}
template <int S>
__device__ __forceinline__ int gh_quad_bcast_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, S * 0x55, 0xF, 0xF, false);
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
template <bool ALPHA, bool SEEN>
__device__ __forceinline__ bool gh_fwd_consume(const GhBatch& t, int base, int total, int lane, int slot, int blk,
                                               float& pxf, float pyf, GhPixelFwd& p, float4* __restrict__ s_col, int& hits) {
  const uint32_t slot8 = (uint32_t)slot * 8u;
  // the batch's colours and opacities go through wave-private LDS memory (one 16-byte store per lane and batch, one
  // broadcast 16-byte load per trip) instead of four crossbar fetches per trip: they are needed after the exponential,
  // off the front of the trip's dependency chain, where the geometry's five fetches stay
  s_col[lane] = make_float4(t.b.z, t.b.w, t.cb, t.b.y);            // (r, g, b, opacity)
  const bool hit = (base + lane < total) && ((t.blocks >> blk) & 1u);
  uint64_t mask = gh_ballot(hit);
  hits += __builtin_popcountll(mask);                                // (FINE launches: how heavy this wave's walk was, for the next launch order)
  bool finished = false;                                             // set (and the mask cleared) inside the rare stop branch, so
  while (mask) {                                                     // the common path's loop control is one scalar compare
    // next four set bits, ascending (wave-uniform scalar work)
    const int nh = __builtin_popcountll(mask);                      // entries left in this batch (>= 1)
    int j0, j1, j2, j3;
    gh_pop4_low(mask, j0, j1, j2, j3);
    // the four entry lanes travel as bytes of one scalar: a lane extracts its slot's with a single v_bfe
    // (& 0xFCFC..: with fewer than four hits the trailing picks are -1 and spill set bits into the later bytes; every byte must
    // stay a valid 4 * lane, because the colour record is read at byte address 4 * src — a scalar instruction, not a vector one)
    const uint32_t packed4 = (((uint32_t)j0 | ((uint32_t)j1 << 8) | ((uint32_t)j2 << 16) | ((uint32_t)j3 << 24)) << 2) & 0xFCFCFCFCu;
    const int src = (int)((packed4 >> slot8) & 0xFFu);              // 4 * entry lane = ds_bpermute address
    const bool have = slot < nh;             // (round 3: `have` and `p.done == 0` as scalar lane masks through an inverse ballot
                                             // instead of two v_cmp per trip: +6 us — the 64-bit scalar arithmetic lengthens every trip)
    const float gpx = gh_lane_fetch(t.a.x, src), gpy = gh_lane_fetch(t.a.y, src), cA = gh_lane_fetch(t.a.z, src);
    const float cB = gh_lane_fetch(t.a.w, src), cC = gh_lane_fetch(t.b.x, src);
    const float4 col = *(const float4*)((const char*)s_col + 4 * src);   // s_col[src >> 2]: src is 4 * lane, one shift-add
    const float r = col.x, g = col.y, bl = col.z, op = col.w;
    const float dx = gpx - pxf, dy = gpy - pyf;
    // cA, cC: the conic's A and C times -0.5, scaled once per instance by gh_ranges_kernel. Scaling by a power of two commutes with
    // every rounding, so this IS -0.5 (A dx dx + C dy dy) - B dx dy of App. A.3 bit for bit, one multiply per (entry, pixel) cheaper.
    const float power = (cA * dx * dx + cC * dy * dy) - cB * dx * dy;
    const float alpha = fminf(0.99f, op * gh_exp(power));          // (power > 0: rejected below, whatever this evaluates to)
    // A finished pixel (and one outside the image) has NaN coordinates: its power is NaN, `power <= 0` fails — no test of `done`
    // in the common path (the select below keeps the NaN out of every accumulator).
    const bool ok = have && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
    // The recurrence collapses to DPP-fused prefix products / sums over the quad, in exact list order.
    const bool valid = SEEN ? (p.done == 0) && ok : ok;     // (SEEN: a stopped pixel walks on virtually with its real coordinates)
    // alpha of the entries that count, 0 for the others: ONE select serves the transmittance factor (1 - 0 == 1 exactly:
    // skipped entries leave T bit-identical) and the blend weight (0 * T == +0, C + c * 0 == C exactly)
    const float ae = valid ? alpha : 0.0f;
    const float f = 1.0f - ae;
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
    float w = ae * Pm;                                             // blend weight alpha * T of the entries that count
    const bool stopc = valid && Pn < 0.0001f;
    if (gh_ballot(stopc)) {                                        // wave-uniform, rare
      // stop flags of this pixel's four slots, gathered over the quad HERE: taking them from the ballot's value made the
      // compiler rebuild the mask with two vector instructions in every trip of the common path
      const int sf = stopc ? 1 : 0;
      const uint32_t qb = (uint32_t)(gh_quad_bcast_i<0>(sf) | (gh_quad_bcast_i<1>(sf) << 1) | (gh_quad_bcast_i<2>(sf) << 2) |
                                     (gh_quad_bcast_i<3>(sf) << 3));
      blend = valid && ((qb & ((2u << slot) - 1u)) == 0u);         // no stop at or before this slot
      w = blend ? w : 0.0f;
      Tn = (qb & 1u) ? p.T : ((qb & 2u) ? P1 : ((qb & 4u) ? P2 : ((qb & 8u) ? P3 : P4)));   // T right before the stop
      if (SEEN && qb && p.done == 0) {
        // the virtual walk starts at the stop: transmittance right behind the stopping entry (the later slots of this trip
        // are left out of it: the virtual stop can only come later for that, never earlier)
        p.vT = (qb & 1u) ? P1 : ((qb & 2u) ? P2 : ((qb & 4u) ? P3 : P4));
        if (p.vT < GH_SEEN_T) {
          p.vdone = 1;
          if (stopc && (qb & ((1u << slot) - 1u)) == 0u) p.stopq = (uint32_t)(4 * (base + 1)) + (uint32_t)src;
        }
      }
      if (qb) { p.done = 1; if (!SEEN) pxf = __uint_as_float(0x7FC00000u); }
      if (__all((SEEN ? p.vdone : p.done) != 0)) { finished = true; mask = 0; }   // every pixel of the block is saturated: last trip
    } else if (SEEN) {
      // pixels between their stop and their virtual stop (wave-uniform test; a few trips per pixel)
      const bool vlive = p.done != 0 && p.vdone == 0;
      if (gh_ballot(vlive)) {
        const float vae = (vlive && ok) ? alpha : 0.0f;
        const float vf = 1.0f - vae;
        float V1, V2, V3;
        gh_quad_prefix3(p.vT, vf, V1, V2, V3);
        const float Vn = gh_slot_select(p.vT, V1, V2, V3) * vf;      // virtual transmittance right behind this lane's entry
        const float V4 = gh_quad_bcast<3>(Vn);
        const int vs = (vae > 0.0f && Vn < GH_SEEN_T) ? 1 : 0;
        const uint32_t qv = (uint32_t)(gh_quad_bcast_i<0>(vs) | (gh_quad_bcast_i<1>(vs) << 1) | (gh_quad_bcast_i<2>(vs) << 2) |
                                       (gh_quad_bcast_i<3>(vs) << 3));
        if (vlive) {
          if (qv) {
            p.vdone = 1;
            if (vs && (qv & ((1u << slot) - 1u)) == 0u) p.stopq = (uint32_t)(4 * (base + 1)) + (uint32_t)src;
          } else p.vT = V4;
        }
        if (__all(p.vdone != 0)) { finished = true; mask = 0; }
      }
    }
    const float m0 = r * w, m1 = g * w, m2 = bl * w;
    gh_quad_accumulate(p.C0, m0);              // C = (((C + m[slot 0]) + m[slot 1]) + m[slot 2]) + m[slot 3], in list order
    gh_quad_accumulate(p.C1, m1);
    gh_quad_accumulate(p.C2, m2);
    if (ALPHA) gh_quad_accumulate(p.A, w);
    p.T = Tn;
    // n_contrib: every lane remembers the last entry of ITS slot that was blended; the pixel's value is the maximum
    // over its four lanes, taken once after the walk.
    // per LANE (positions ascend), kept in units of a quarter entry — 4 * (position + 1) = 4 * (base + 1) + src — so that the
    // byte the lane extracted serves as it stands; converted once after the walk (quad max at the end)
    p.last = blend ? (uint32_t)(4 * (base + 1)) + (uint32_t)src : p.last;
  }
  return finished;
}

// ---- forward, fine-grained form: wave = 2x2 pixels x 16 depth slots -----------------------------------------------
// lane = 16 * pixel + slot: a pixel's slots are one DPP row. Per trip the wave takes the next (up to) SIXTEEN surviving entries of the
// batch — a quarter of the trips of the 4-slot form for the same 4x4-pixel block, split over four waves. The entry lanes come from a
// wave-private byte table (hit lanes store their lane number at their rank among the hits; slot s of trip t reads byte 16 t + s),
// not from scalar picks. The recurrence stays the sequential one of App. A.3, as FIXED-POINT iterations along the row:
//     I_s <- I_{s-1} * f_s   (v_mul_f32_dpp row_shr:1, in place; lane 0 of the row has no source lane and keeps T * f_0)
// applied 15 times leaves ((T f_0) f_1) ... f_s in lane s — after step k lanes 0..k hold their final value and every later step
// recomputes exactly the same product from the same operands. The colour sums run the same way with v_add_f32_dpp. State between
// trips (T, C) is the value in the row's LAST lane: the next trip's lane 0 takes it with row_ror:1 (fused into its first operation).
__device__ __forceinline__ float gh_row_ror1(float v) {       // lane s of a row takes lane s - 1, lane 0 takes lane 15
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
}
#define GH_ROW_STEP(OP, R, X) "s_nop 1\n\t" OP " " R ", " R ", " X " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define GH_X5(a) a a a a a
#define GH_X15(a) GH_X5(a) GH_X5(a) GH_X5(a)

template <bool ALPHA>
__device__ __forceinline__ bool gh_fwd_consume_fine(const GhBatch& t, int base, int total, int lane, int slot, int blk,
                                                    float& pxf, float pyf, GhPixelFwd& p, float4* __restrict__ s_col,
                                                    uint8_t* __restrict__ s_idx, int& hits) {
  s_col[lane] = make_float4(t.b.z, t.b.w, t.cb, t.b.y);            // (r, g, b, opacity), as gh_fwd_consume
  const bool hit = (base + lane < total) && ((t.blocks >> blk) & 1u);
  const uint64_t mask = gh_ballot(hit);
  const int nh = __builtin_popcountll(mask);
  hits += nh;
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  if (hit) s_idx[rank] = (uint8_t)(lane * 4);                      // 4 * entry lane = ds_bpermute address
  bool finished = false;
  for (int t0 = 0; t0 < nh; t0 += 16) {
    const bool have = t0 + slot < nh;
    // (bytes past the batch's hits are stale: & 0xFC keeps them a valid 4 * lane — the records behind it are finite, their
    //  weight is zero)
    const int src = (int)s_idx[t0 + slot] & 0xFC;
    const float gpx = gh_lane_fetch(t.a.x, src), gpy = gh_lane_fetch(t.a.y, src), cA = gh_lane_fetch(t.a.z, src);
    const float cB = gh_lane_fetch(t.a.w, src), cC = gh_lane_fetch(t.b.x, src);
    const float4 col = *(const float4*)((const char*)s_col + 4 * src);
    const float dx = gpx - pxf, dy = gpy - pyf;
    const float power = (cA * dx * dx + cC * dy * dy) - cB * dx * dy;      // the same expression as gh_fwd_consume, bit for bit
    const float alpha = fminf(0.99f, col.w * gh_exp(power));
    const bool ok = have && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);   // (a finished pixel has NaN coordinates: never ok)
    const float ae = ok ? alpha : 0.0f;
    const float f = 1.0f - ae;
    // T right behind every slot's entry, in list order along the row
    const float Tin = gh_row_ror1(p.T);                                    // lane 0: T in front of the trip
    float I = Tin * f, Pm = Tin;
    asm(GH_X15(GH_ROW_STEP("v_mul_f32_dpp", "%0", "%2"))
        "s_nop 1\n\tv_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(I), "+v"(Pm) : "v"(f));                                      // Pm: T seen by this lane's entry (lane 0 keeps Tin)
    // Early stop (App. A.3): the FIRST entry of a pixel with T (1 - alpha) < 1e-4 is not blended and ends the pixel; up to and
    // including that entry the products above are the sequential ones (see gh_fwd_consume).
    bool blend = ok;
    float w = ae * Pm;
    const bool stopc = ok && I < 0.0001f;
    const uint64_t sm = gh_ballot(stopc);
    if (sm) {                                                              // wave-uniform, rare
      const uint32_t qb = (uint32_t)(sm >> (lane & 48)) & 0xFFFFu;         // stop flags of this pixel's sixteen slots
      if (qb) {
        const int fs = __builtin_ctz(qb);
        blend = ok && slot < fs;                                           // no stop at or before this slot
        w = blend ? w : 0.0f;
        p.done = 1; pxf = __uint_as_float(0x7FC00000u);
      }
      // T right before the stop = what the stopping entry saw (rows without a stop fetch their own lane: unchanged)
      const float Tstop = gh_lane_fetch(Pm, ((lane & 48) + (qb ? __builtin_ctz(qb) : 0)) * 4);
      I = qb ? Tstop : I;
      if (__all(p.done != 0)) finished = true;                             // every pixel of the block is saturated: last trip
    }
    // C = (((C + m[slot 0]) + m[slot 1]) + ...) + m[slot 15] with m = colour * w, in list order; the mask channel's m is w
    {
      float m0, m1, m2;
      if (ALPHA)
        asm("v_mul_f32 %4, %7, %10\n\tv_mul_f32 %5, %8, %10\n\tv_mul_f32 %6, %9, %10\n\t"
            "v_add_f32_dpp %0, %0, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %2, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %10 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            GH_X15("v_add_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %10 row_shr:1 row_mask:0xf bank_mask:0xf\n\t")
            : "+v"(p.C0), "+v"(p.C1), "+v"(p.C2), "+v"(p.A), "=&v"(m0), "=&v"(m1), "=&v"(m2)
            : "v"(col.x), "v"(col.y), "v"(col.z), "v"(w));
      else
        asm("v_mul_f32 %3, %6, %9\n\tv_mul_f32 %4, %7, %9\n\tv_mul_f32 %5, %8, %9\n\t"
            "v_add_f32_dpp %0, %0, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32_dpp %2, %2, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            GH_X15("v_add_f32_dpp %0, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t")
            : "+v"(p.C0), "+v"(p.C1), "+v"(p.C2), "=&v"(m0), "=&v"(m1), "=&v"(m2)
            : "v"(col.x), "v"(col.y), "v"(col.z), "v"(w));
    }
    p.T = I;                                                               // (the row's last lane: T behind the trip)
    p.last = blend ? (uint32_t)(4 * (base + 1)) + (uint32_t)src : p.last;
    if (finished) break;
  }
  return finished;
}

// The tile's backward work items into the work list (one lane of the tile's last wave; kept out of line: it runs once per tile and
// the render kernel's code should stay as small as its walk) — see the call site for what the regions are.
__device__ __attribute__((noinline)) void gh_append_items(int tile, uint32_t nseg, uint32_t item_idx, uint32_t n_tiles_call, uint32_t use_classes,
                                                          uint32_t* __restrict__ class_count, uint2* __restrict__ items, uint32_t n_items_cap,
                                                          uint32_t* __restrict__ cost) {
  uint4 hist_lo = make_uint4(0u, 0u, 0u, 0u), hist_hi = hist_lo;
  if (use_classes == 1u) { hist_lo = ((const uint4*)cost)[0]; hist_hi = ((const uint4*)cost)[1]; }
  const uint32_t hist[GH_BWD_COST_SLOTS] = {hist_lo.x, hist_lo.y, hist_lo.z, hist_lo.w, hist_hi.x, hist_hi.y, hist_hi.z, hist_hi.w};
  if (use_classes != 1u) {                                 // modes 0 and 2: one atomic per tile
    const uint32_t c = (use_classes == 2u && item_idx < (n_tiles_call >> 2)) ? 1u : 0u;
    const uint32_t pos = __hip_atomic_fetch_add(&class_count[c], nseg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t j = 0; j < nseg; ++j)
      items[(size_t)c * n_items_cap + pos + (c ? nseg - 1u - j : j)] = make_uint2((uint32_t)tile, j);    // (class 1: segment 0 first)
  }
  // one atomic per RUN of segments of one class (a tile's segments get cheaper towards the back: one to three runs)
  uint32_t k = use_classes == 1u ? 0u : nseg;
  while (k < nseg) {
    uint32_t c = hist[k < GH_BWD_COST_SLOTS ? k : GH_BWD_COST_SLOTS - 1u] >> GH_BWD_CLASS_SHIFT;
    c = c < GH_BWD_CLASSES ? c : GH_BWD_CLASSES - 1u;
    uint32_t e = k + 1u;
    while (e < nseg) {
      uint32_t c2 = hist[e < GH_BWD_COST_SLOTS ? e : GH_BWD_COST_SLOTS - 1u] >> GH_BWD_CLASS_SHIFT;
      c2 = c2 < GH_BWD_CLASSES ? c2 : GH_BWD_CLASSES - 1u;
      if (c2 != c) break;
      ++e;
    }
    const uint32_t pos = __hip_atomic_fetch_add(&class_count[c], e - k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the backward reads a region from its END: a run's segments are stored back to front, so that a tile's segment 0 — all
    //  its pixels still live: the expensive one — goes first; the one-region form of large launches keeps rounds 2-5's order)
    for (uint32_t j = k; j < e; ++j)
      items[(size_t)c * n_items_cap + pos + (use_classes ? e - 1u - j : j - k)] = make_uint2((uint32_t)tile, j);
    k = e;
  }
  if (use_classes == 1u) { ((uint4*)cost)[0] = make_uint4(0u, 0u, 0u, 0u); ((uint4*)cost)[1] = make_uint4(0u, 0u, 0u, 0u); }
}

// grid = 4 blocks per tile (one per 8x8 quadrant), 4 waves per block (one per 4x4 pixel block); no LDS, no barriers.
// ALPHA: also accumulate the mask channel (colour 1, bg 0) — SURVEY §8 f-2.
// LOSS: fused image loss. 1 = GhOutputs.l1_target: every wave leaves the gradient sign(img - gt) / n of its 16 pixels, every workgroup
// ONE partial sum of |img - gt| at a place fixed by (tile, quadrant) — plain stores, no hand-off between workgroups inside this
// kernel (a last-arriver tree over the tiles was costed: every level is a drained store + an RMW + dependent loads on the kernel's
// tail, as long as the small sum kernel that follows). gh_launch_partials_sum adds the partials up in index order: bitwise
// reproducible. 2 = GhOutputs.fit_loss (with ALPHA): the fit's image loss k_l1 |bbox * rgb - gt_rgb| + k_m (clip(alpha) - gt_mask)^2
// and its gradients w.r.t. image and alpha — gh_fit_loss_kernel's expressions, pixel for pixel.
struct GhFusedLoss {
  const float* target;       // LOSS 1: (n_views,3,H,W); LOSS 2: gt_rgb (n_views,H,W,3)
  float* dL;                 // (n_views,3,H,W)
  float inv_n;               // LOSS 1: 1 / (n_views*3*H*W)
  float* part;               // [T][4]: one sum per 8x8-pixel quadrant
  const float* gt_mask;      // LOSS 2: (n_views,H,W)
  const float* bbox;         // LOSS 2: (n_views,H,W) or NULL
  float* dalpha;             // LOSS 2: (n_views,H,W)
  float k_l1, k_m;           // LOSS 2: scale * lambda_l1 / (3 HW), scale * lambda_mask / HW
  float sum_scale;           // factor of the final sum over the partials (LOSS 1: inv_n, LOSS 2: 1); left behind the partials
};

// FINE (launches of at most GH_FWD_FINE_TILES tiles, never with SEEN): the grid carries 12 extra workgroups for each of the first
// `fine_k` tiles of the launch order; such a tile, if its list holds at least `fine_min` entries, is walked by 16 workgroups — one per
// 4x4-pixel block, its four waves 2x2 pixels x 16 slots each (gh_fwd_consume_fine) — instead of four; the others run as ever and
// their extra workgroups exit at once. Results are the same bit for bit: only which wave owns a pixel changes.
template <bool ALPHA, bool SEEN, int LOSS, bool FINE = false>
__global__ __launch_bounds__(GH_BLOCK) void gh_render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_order, const float4* __restrict__ r0,
    const float4* __restrict__ r1, const float2* __restrict__ r2, const float* __restrict__ cams, int H, int W, int gx,
    int tiles, float* __restrict__ image, float* __restrict__ alpha_img, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_walk, float4* __restrict__ ckpt_rgb,
    float4* __restrict__ final_C, uint2* __restrict__ items, GhCounters* __restrict__ ctr, const uint32_t* __restrict__ render_guard,
    uint32_t guard_mask, const float* __restrict__ tile_depth_bound, float* __restrict__ tile_depth_seen, float seen_scale, uint32_t seen_slack,
    const uint32_t* __restrict__ sorted_gid, const float4* __restrict__ geom, const GhFusedLoss l1,
    uint32_t n_tiles_call, uint32_t fine_k, uint32_t fine_min, uint32_t n_items_cap, uint32_t* __restrict__ class_count, uint32_t* __restrict__ heavy_out, uint32_t heavy_hits, uint32_t use_classes) {
  // LOSS: the quadrant's four wave sums meet in LDS (the last wave to arrive adds them up in block order); the arrival counter is
  // cleared behind the one barrier of the kernel, which the four waves reach as they start — before any load is in flight
  __shared__ float s_l1[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_l1_n;
  if (LOSS) {
    if (threadIdx.x == 0) {
      s_l1_n = 0u;
      // the final sum's factor, behind the last partial: a deferred sum (GhGrads.deferred_loss) finds it there
      if (blockIdx.x == 0) l1.part[(size_t)n_tiles_call * (FINE ? 16 : 4)] = l1.sum_scale;
    }
    __syncthreads();
  }
  int v, tx, ty;
  uint32_t item_idx, quad_u, sub = 0u;
  // FINE: the first 16 fine_k workgroups belong to the fine_k tiles at the head of the launch order, sixteen each (one per 4x4-pixel
  // block: they must START first — behind the other tiles' workgroups they would begin a round late); the rest as ever, four per tile
  bool extra = false;                                                // a workgroup that only exists for the fine form of its tile
  if (FINE && blockIdx.x < 16u * fine_k) { item_idx = blockIdx.x >> 4; quad_u = (blockIdx.x >> 2) & 3u; sub = blockIdx.x & 3u; extra = sub != 0u; }
  else { gh_item_quad(blockIdx.x - (FINE ? 16u * fine_k : 0u), n_tiles_call - (FINE ? fine_k : 0u), item_idx, quad_u); item_idx += FINE ? fine_k : 0u; }
  // (wave-uniform values are moved to scalar registers by hand: the compiler keeps what comes out of a global load in vector
  //  registers, and with it the tile coordinates, the list range and the three 64-bit record pointers — per lane, through the walk)
  const int tile = __builtin_amdgcn_readfirstlane((int)tile_order[item_idx]);             // heaviest tiles are launched first
  const int quad = (int)quad_u;
  gh_tile_coords(tile, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint2 range_v = ranges[tile];
  const uint2 range = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.y));
  const int total = (int)(range.y - range.x);
  // fine form: this workgroup owns the 4x4-pixel block `sub` of its quadrant, its waves the block's 2x2-pixel quarters
  const bool fine = FINE && item_idx < fine_k && (uint32_t)total >= fine_min;
  if (FINE && extra && !fine) return;                               // (no barrier between here and the kernel's end for this workgroup's waves)
  const int blk4 = fine ? (int)sub : wid;                            // the 4x4 block of the quadrant this wave works in
  const int slot = fine ? (lane & 15) : (lane & 3), pi = fine ? (lane >> 4) : (lane >> 2);
  const int bx0 = tx * GH_TILE + (quad & 1) * 8 + (blk4 & 1) * 4, by0 = ty * GH_TILE + (quad >> 1) * 8 + (blk4 >> 1) * 4;
  const int x = fine ? bx0 + (wid & 1) * 2 + (pi & 1) : bx0 + (pi & 3), y = fine ? by0 + (wid >> 1) * 2 + (pi >> 1) : by0 + (pi >> 2);
  const bool inside = x < W && y < H;
  float pxf = inside ? (float)x : __uint_as_float(0x7FC00000u);      // NaN = the pixel takes nothing (any more): see gh_fwd_consume
  const float pyf = (float)y;
  // Addressing of everything per pixel: a wave-uniform base (the view's plane: scalar registers) + ONE 32-bit offset per lane. With
  // 64-bit per-lane addresses the epilogue's nine stores held eighteen address registers at once — the kernel's register peak,
  // which cost the fused-loss variant a wave per SIMD (71 VGPRs).
  const size_t hw = (size_t)H * W;
  const uint32_t poff = (uint32_t)y * (uint32_t)W + (uint32_t)x;
  const int blk = ((quad >> 1) * 2 + (blk4 >> 1)) * 4 + (quad & 1) * 2 + (blk4 & 1);    // this wave's bit in the block masks
  r0 += range.x; r1 += range.x; r2 += range.x;
  // state checkpoints for the segmented backward: slot of (tile, position m*GH_SEGMENT) = range.x/GH_SEGMENT + tile + m - 1
  // (disjoint between tiles because floor(a+b) >= floor(a) + floor(b)); pixel = row-major index inside the tile
  const size_t ck0 = ((size_t)(range.x / GH_SEGMENT) + (size_t)tile) * 256 + (size_t)((y - ty * GH_TILE) * GH_TILE + (x - tx * GH_TILE));

  __shared__ float4 s_col_all[GH_BLOCK / GH_WAVE][GH_WAVE];
  float4* s_col = s_col_all[wid];
  __shared__ uint8_t s_idx_all[FINE ? GH_BLOCK / GH_WAVE : 1][80];          // fine form: entry lanes of a batch's hits, in list order
  uint8_t* s_idx = s_idx_all[FINE ? wid : 0];
  // LOSS: the pixel's target is fetched NOW (three registers through the walk) — at the end of the wave the loads' latency would be
  // exposed time of a finished wave's slot
  float tg0 = 0.0f, tg1 = 0.0f, tg2 = 0.0f, tgm = 0.0f;
  bool in_box = true;
  if (LOSS == 1 && inside && slot == 0) {
    const float* tg = l1.target + (size_t)v * 3 * hw;
    tg0 = tg[poff]; tg1 = (tg + hw)[poff]; tg2 = (tg + 2 * hw)[poff];
  }
  if (LOSS == 2 && inside && slot == 0) {
    const float* tg = l1.target + (size_t)v * 3 * hw;
    tg0 = tg[3 * poff]; tg1 = tg[3 * poff + 1]; tg2 = tg[3 * poff + 2];     // channel-last, as the reference holds it
    tgm = (l1.gt_mask + (size_t)v * hw)[poff];
    if (l1.bbox) in_box = (l1.bbox + (size_t)v * hw)[poff] != 0.0f;
  }
  int hits = 0;                                      // list entries this wave's block mask let through (batches it consumed)
  GhPixelFwd p;
  p.T = 1.0f; p.C0 = p.C1 = p.C2 = p.A = 0.0f; p.last = 0; p.stopq = 0; p.done = inside ? 0 : 1;
  p.vT = 1.0f; p.vdone = p.done;
  if (FINE && fine) {
    if (total > 0 && !__all(p.done != 0)) {
      GhBatch A, B;
      gh_load_batch(A, r0, r1, r2, lane, total);
      for (int base = 0; base < total; base += 2 * GH_WAVE) {
        gh_load_batch(B, r0, r1, r2, base + GH_WAVE + lane, total);
        if (gh_fwd_consume_fine<ALPHA>(A, base, total, lane, slot, blk, pxf, pyf, p, s_col, s_idx, hits)) break;
        if (base + GH_WAVE >= total) break;
        gh_load_batch(A, r0, r1, r2, base + 2 * GH_WAVE + lane, total);
        if (gh_fwd_consume_fine<ALPHA>(B, base + GH_WAVE, total, lane, slot, blk, pxf, pyf, p, s_col, s_idx, hits)) break;
        const int next = base + 2 * GH_WAVE;
        if ((next % GH_SEGMENT) == 0 && next < total) {                 // (wave-uniform, rare; the state sits in the rows' last lanes)
          const float cT = gh_row_ror1(p.T), c0 = gh_row_ror1(p.C0), c1 = gh_row_ror1(p.C1), c2 = gh_row_ror1(p.C2);
          if (inside && slot == 0) ckpt_rgb[ck0 + (size_t)(next / GH_SEGMENT - 1) * 256] = make_float4(cT, c0, c1, c2);
        }
      }
    }
    // the walk's state from the rows' last lanes to their first (the epilogue's lanes); n_contrib = max over the row
    p.T = gh_row_ror1(p.T); p.C0 = gh_row_ror1(p.C0); p.C1 = gh_row_ror1(p.C1); p.C2 = gh_row_ror1(p.C2);
    if (ALPHA) p.A = gh_row_ror1(p.A);
    p.last >>= 2;
    {
      uint32_t a = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.last, 0x128, 0xF, 0xF, false); p.last = a > p.last ? a : p.last;   // row_ror:8
      a = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.last, 0x124, 0xF, 0xF, false); p.last = a > p.last ? a : p.last;            // row_ror:4
      a = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.last, 0x122, 0xF, 0xF, false); p.last = a > p.last ? a : p.last;            // row_ror:2
      a = (uint32_t)__builtin_amdgcn_mov_dpp((int)p.last, 0x121, 0xF, 0xF, false); p.last = a > p.last ? a : p.last;            // row_ror:1
    }
  } else {
  if (total > 0 && !__all(p.done != 0)) {
    // two register sets in flight: while one batch is consumed the next one is already being loaded
    GhBatch A, B;
    gh_load_batch(A, r0, r1, r2, lane, total);
    for (int base = 0; base < total; base += 2 * GH_WAVE) {
      gh_load_batch(B, r0, r1, r2, base + GH_WAVE + lane, total);
      if (gh_fwd_consume<ALPHA, SEEN>(A, base, total, lane, slot, blk, pxf, pyf, p, s_col, hits)) break;
      if (base + GH_WAVE >= total) break;
      gh_load_batch(A, r0, r1, r2, base + 2 * GH_WAVE + lane, total);
      if (gh_fwd_consume<ALPHA, SEEN>(B, base + GH_WAVE, total, lane, slot, blk, pxf, pyf, p, s_col, hits)) break;
      const int next = base + 2 * GH_WAVE;              // wave-uniform: a checkpoint every GH_SEGMENT entries (rare)
      if ((next % GH_SEGMENT) == 0 && next < total && inside && slot == 0) {
        const size_t ck = ck0 + (size_t)(next / GH_SEGMENT - 1) * 256;
        ckpt_rgb[ck] = make_float4(p.T, p.C0, p.C1, p.C2);
      }
    }
  }
  p.last >>= 2;                                      // quarter-entry units -> list position + 1 (gh_fwd_consume)
  {                                                  // n_contrib of the pixel = max over the four slot lanes of its quad
    const uint32_t a = (uint32_t)gh_quad_perm_i<0xB1>((int)p.last);      // quad_perm [1,0,3,2]
    p.last = a > p.last ? a : p.last;
    const uint32_t b = (uint32_t)gh_quad_perm_i<0x4E>((int)p.last);      // quad_perm [2,3,0,1]
    p.last = b > p.last ? b : p.last;
  }
  }
  // Speculative occlusion bound (GhInputs.tile_depth_bound): the tile's list holds nothing behind the bound. A pixel that reached
  // the early stop inside it looked at nothing further back — its result is the unbounded call's bit for bit; a pixel that ran
  // off the end of a truncated list may miss entries: it is poisoned below and the call is flagged (overflow bit 2).
  const bool bounded = tile_depth_bound != nullptr && tile_depth_bound[tile] < __uint_as_float(0x7F800000u);
  const bool pixel_miss = bounded && inside && p.done == 0;
  const bool wave_miss = !__all(p.done != 0);       // some in-image pixel of this block never reached the stop
  // what the NEXT call may rely on: every pixel of the block passed the stricter virtual threshold inside this list
  const bool wave_unsat = SEEN ? !__all(p.vdone != 0) : wave_miss;
  if (bounded && wave_miss && lane == 0) atomicOr(&ctr->overflow, 4u);
  if (SEEN && total == 0 && tid == 0 && quad == 0) {   // empty list: no bound, no block of it stops anything
    tile_depth_seen[2 * tile] = __uint_as_float(0x7F800000u); tile_depth_seen[2 * tile + 1] = __uint_as_float(0u);
  }
  if (LOSS) {
    // (a fused loss never comes with an occlusion bound — gh_forward rejects the pair — so pixel_miss is false here and the
    // pixel values below are the ones stored at the end of the kernel)
    float labs = 0.0f;
    if (inside && slot == 0) {
      const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
      const float poison = (*render_guard & guard_mask) ? __uint_as_float(0x7FC00000u) : 0.0f;
      float* dl = l1.dL + (size_t)v * 3 * hw;
      const float c0 = fmaf(p.T, bg[0], p.C0) + poison, c1 = fmaf(p.T, bg[1], p.C1) + poison, c2 = fmaf(p.T, bg[2], p.C2) + poison;
      if (LOSS == 1) {
        const float d0 = c0 - tg0, d1 = c1 - tg1, d2 = c2 - tg2;
        labs = (fabsf(d0) + fabsf(d1)) + fabsf(d2);
        // torch.sign: sign(0) = 0; a NaN pixel (poisoned call) leaves no gradient, as gh_l1_loss's guard
        dl[poff] = d0 > 0.0f ? l1.inv_n : (d0 < 0.0f ? -l1.inv_n : 0.0f);
        (dl + hw)[poff] = d1 > 0.0f ? l1.inv_n : (d1 < 0.0f ? -l1.inv_n : 0.0f);
        (dl + 2 * hw)[poff] = d2 > 0.0f ? l1.inv_n : (d2 < 0.0f ? -l1.inv_n : 0.0f);
      } else {
        // gh_fit_loss_kernel's expressions: colour zeroed outside the box, clip(alpha, -0.001, 1) passes the gradient on [min, max]
        const float d0 = (in_box ? c0 : 0.0f) - tg0, d1 = (in_box ? c1 : 0.0f) - tg1, d2 = (in_box ? c2 : 0.0f) - tg2;
        labs = l1.k_l1 * fabsf(d0); labs += l1.k_l1 * fabsf(d1); labs += l1.k_l1 * fabsf(d2);
        dl[poff] = in_box ? (d0 > 0.0f ? l1.k_l1 : (d0 < 0.0f ? -l1.k_l1 : 0.0f)) : 0.0f;
        (dl + hw)[poff] = in_box ? (d1 > 0.0f ? l1.k_l1 : (d1 < 0.0f ? -l1.k_l1 : 0.0f)) : 0.0f;
        (dl + 2 * hw)[poff] = in_box ? (d2 > 0.0f ? l1.k_l1 : (d2 < 0.0f ? -l1.k_l1 : 0.0f)) : 0.0f;
        const float a = fmaf(p.T, 0.0f, p.A) + poison;                 // the mask channel as it is stored below
        const float ac = fminf(fmaxf(a, -0.001f), 1.0f);
        const float e = ac - tgm;
        labs += l1.k_m * e * e;
        (l1.dalpha + (size_t)v * hw)[poff] = (a >= -0.001f && a <= 1.0f) ? 2.0f * l1.k_m * e : 0.0f;
      }
    }
    if (FINE && fine) {
      // FINE launches keep one partial per 4x4-pixel block, and it must not depend on the form the tile was walked in (which tiles
      // go fine is a scheduling decision: the loss may not change with it): the block's sixteen pixel terms meet in LDS and the last
      // wave to arrive adds them up in the lanes and the order of the coarse form's wave sum.
      __shared__ float s_px[16];
      if (slot == 0) s_px[(y - by0) * 4 + (x - bx0)] = labs;
      uint32_t k = 0u;
      if (lane == 63) k = __hip_atomic_fetch_add(&s_l1_n, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      if ((uint32_t)__builtin_amdgcn_readlane((int)k, 63) == GH_BLOCK / GH_WAVE - 1) {
        const float bsum = gh_wave_sum_to63((lane & 3) == 0 ? s_px[lane >> 2] : 0.0f);
        if (lane == 63) l1.part[(size_t)tile * 16 + quad * 4 + sub] = bsum;
      }
    } else {
    const float wsum = gh_wave_sum_to63(labs);
    if (lane == 63) {
      s_l1[wid] = wsum;
      const uint32_t k = __hip_atomic_fetch_add(&s_l1_n, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (k == GH_BLOCK / GH_WAVE - 1) {
        if (!FINE) l1.part[(size_t)tile * 4 + quad] = (s_l1[0] + s_l1[1]) + (s_l1[2] + s_l1[3]);
        else ((float4*)l1.part)[(size_t)tile * 4 + quad] = make_float4(s_l1[0], s_l1[1], s_l1[2], s_l1[3]);   // (wave = block)
      }
    }
    }
  }
  if (total > 0) {                                   // walked length of the tile = max n_contrib over its 16 waves
    uint32_t m = p.last;
    uint32_t sq = p.stopq >> 2;                      // position + 1 of the entry that stopped the lane's pixel
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(m, o); m = t > m ? t : m;
      const uint32_t u = __shfl_xor(sq, o); sq = u > sq ? u : sq;
    }
    if (lane == 0) {
      // The LAST of the tile's waves (16, or 64 of a fine tile) appends the tile's backward work items, one per depth segment of the
      // walked prefix (the order of the list only affects scheduling, never results: see below).
      // Ordering without fences (an agent-scope release would write back the whole L2): only relaxed agent-scope RMW
      // atomics carry the data; each returns its old value, so waiting for the return means it has been performed.
      // Small launches (heavy_hits): the most entries any 4x4 block of the tile took = what the tile's longest wave does; the NEXT
      // call's launch order and choice of fine tiles go by it (gh_rank_tiles; a scheduling hint only — stale or missing values cost
      // time). Every variant a small launch can run leaves it, SEEN included: a call that left nothing would leave the next one
      // without an order.
      if ((FINE || SEEN) && heavy_hits) __hip_atomic_fetch_max(&heavy_out[tile], (uint32_t)hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t* cost = tile_walk + 4 * (size_t)n_tiles_call + (size_t)tile * GH_BWD_COST_SLOTS;      // the tile's backward history (below)
      const uint32_t prev_max = __hip_atomic_fetch_max(&tile_walk[tile], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" :: "v"(prev_max) : "memory");
      uint32_t* done = tile_walk + (size_t)n_tiles_call;          // completion counters follow the T walk entries
      if (SEEN) {                                                  // (same ordering discipline: relaxed RMWs, results awaited)
        uint32_t* stop = done + (size_t)n_tiles_call;             // ... and the stop positions follow those
        const uint32_t ps = __hip_atomic_fetch_max(&stop[tile], sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(ps) : "memory");
      }
      // bits 0..7 of the completion word count the tile's waves; bit 8 + b = every in-image pixel of block b reached the stop
      // (each wave adds its own bit once: the sum is the OR)
      const uint32_t my_bit = (SEEN && !wave_unsat) ? (0x100u << blk) : 0u;
      const uint32_t prev_done = __hip_atomic_fetch_add(&done[tile], 1u + my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((prev_done & 0xFFu) == (fine ? 16u : 4u) * (GH_BLOCK / GH_WAVE) - 1u) {     // (a fine tile: 16 workgroups)
        const uint32_t w = __hip_atomic_fetch_max(&tile_walk[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // final value
        // large launches: the NEXT call's launch order goes by the entries this tile actually walked (one plain store per tile, by
        // its last wave; small launches keep the finer measure above) — gh_rank_tiles in the next projection kernel
        if (!heavy_hits && heavy_out) heavy_out[tile] = (w + 1u) >> 1;      // (the list LENGTH as the key instead: forward +1.4 us at 8 views)
        const uint32_t nseg = (w + GH_SEGMENT - 1u) / GH_SEGMENT;
        if (nseg) {
          // The work list is kept in GH_BWD_CLASSES regions, and the backward takes them from the highest class down (inside a region
          // from its end). use_classes (gh_bwd_class_mode):
          //  1 (2,049 .. 8,192 tiles): the class is what the item COST the previous backward over this workspace — cycles of its slowest
          //    quadrant, left per (tile, segment) by gh_render_bwd_kernel in tile_walk[4..12) — so the longest items go first. In the order
          //    of completion (rounds 2-5) the long items sat anywhere in the list and the kernel ended on a few of them with most wave
          //    slots idle (profiles/r6_bwd_order.txt: 8 views 187 -> 163 us). A hint only: no history = class 0 = the old order. The
          //    history is cleared behind the read (this call's backward writes it again) and fetched HERE, by the tile's last wave only:
          //    every wave prefetching it under its completion atomics measured 1-2 us slower.
          //  2 (up to 2,048 tiles, whose backward runs four waves per workgroup): no history — the first quarter of the forward's launch
          //    order (the heaviest tiles by ITS measure) is class 1, the rest class 0; the measured classes lost to this there.
          //  0 (larger launches): one region, the order of completion: ten rounds of workgroups leave a short tail whatever the order,
          //    and what concurrent workgroups share in the L2 weighs more (1024^2 x 8 views: 429 -> 439 us with classes).
          gh_append_items(tile, nseg, item_idx, n_tiles_call, use_classes, class_count, items, n_items_cap, cost);
        }
        if (SEEN) {
          // every pixel of the tile stopped (and passed the virtual threshold): nothing behind the last entry any of them looked at can matter next time either
          // (up to the motion margin seen_scale); otherwise no bound for this tile
          uint32_t* stop = done + (size_t)n_tiles_call;
          const uint32_t sp = __hip_atomic_fetch_max(&stop[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t sat_mask = ((prev_done | my_bit) >> 8) & 0xFFFFu;
          const bool unsat = sat_mask != 0xFFFFu || sp == 0u;
          // ... plus `seen_slack` entries: what a pixel that stopped only just finds when it needs a little more next time
          const uint32_t want = sp + seen_slack;
          float seen = __uint_as_float(0x7F800000u);
          if (!unsat) {
            if (want <= (uint32_t)total) seen = geom[(size_t)sorted_gid[range.x + want - 1u] * 4 + 3].y * seen_scale;       // (view-space depth of the Gaussian: the line's last float4)
            else if (bounded) seen = tile_depth_bound[tile] * seen_scale;     // this list was cut short itself: widen its bound
            else seen = geom[(size_t)sorted_gid[range.x + (uint32_t)total - 1u] * 4 + 3].y * seen_scale;
          }
          tile_depth_seen[2 * tile] = seen;
          tile_depth_seen[2 * tile + 1] = __uint_as_float(sat_mask);
        }
      }
    }
  }
  if (inside && slot == 0) {
    const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
    (final_T + (size_t)v * hw)[poff] = p.T;
    (n_contrib + (size_t)v * hw)[poff] = p.last;
    // the backward reads final_C only for a pixel that blends entries BEHIND a segment cut (last > seg_hi >= GH_SEGMENT,
    // gh_render_bwd_kernel): most pixels — the background, every pixel of a short list — never get there
    if (p.last > (uint32_t)GH_SEGMENT) (final_C + (size_t)v * hw)[poff] = make_float4(p.C0, p.C1, p.C2, 0.0f);
    float* img = image + (size_t)v * 3 * hw;
    // Device-side overflow guard: with D > max_instances the lists are truncated, so a sync-free caller must never see a
    // plausible image — it gets NaN (and GhCounters.overflow for the host to read whenever it chooses).
    // (a depth-bound miss elsewhere — bit 2 — does not touch this pixel: tiles are independent; this pixel's own miss does)
    // The capacity / stale-list / depth-key bits were set by kernels BEFORE this one; the kernel in front of this one left them in
    // a word of their own (GhLayout.render_guard), read here through the scalar cache. The only write this kernel makes to the
    // counters' line is the atomic OR of bit 2 above and the work-list atomics (reserved[1]): a plain load of THAT line cannot be a
    // scalar one, and 86 k waves fetching it with vector loads cost the kernel 100 us (same-box A/B, round 4: 144 -> 242 us).
    // guard_mask: bits 0, 1, 3 (the whole call is invalid); a second call over shared lists also takes bit 2 of the call that
    // built them — it has no bound of its own to verify, and lists a miss truncated are not its lists either.
    const float poison = ((*render_guard & guard_mask) || pixel_miss) ? __uint_as_float(0x7FC00000u) : 0.0f;
    img[poff] = fmaf(p.T, bg[0], p.C0) + poison;
    (img + hw)[poff] = fmaf(p.T, bg[1], p.C1) + poison;
    (img + 2 * hw)[poff] = fmaf(p.T, bg[2], p.C2) + poison;
    if (ALPHA) (alpha_img + (size_t)v * hw)[poff] = fmaf(p.T, 0.0f, p.A) + poison;
  }
}

void gh_launch_render_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, float* image, float* alpha, const char* wg, char* ws,
                          const GhLayout& L, hipStream_t s, float* seen, float seen_scale, uint32_t seen_slack, const GhOutputs* fused,
                          bool own_order) {
  // Small launches (resident all at once: as long as their longest wave) run their heaviest tiles in the fine-grained form: see the
  // kernel. GH_FWD_FINE_K / GH_FWD_FINE_MIN in the environment override the two thresholds (0 tiles = the coarse form only; tests
  // walk EVERY tile of their small scenes in the fine form with K = a large number, MIN = 0).
  static const long env_k = getenv("GH_FWD_FINE_K") ? atol(getenv("GH_FWD_FINE_K")) : -1;
  static const long env_min = getenv("GH_FWD_FINE_MIN") ? atol(getenv("GH_FWD_FINE_MIN")) : -1;
  const uint32_t n_tiles_call = (uint32_t)(g.NV * g.tiles);
  const bool fine_launch = gh_fwd_fine_launch(g) && !seen;
  uint32_t fine_k = fine_launch ? (uint32_t)(env_k >= 0 ? env_k : GH_FWD_FINE_K) : 0u;
  if (fine_k > n_tiles_call) fine_k = n_tiles_call;
  const uint32_t fine_min = (uint32_t)(env_min >= 0 ? env_min : GH_FWD_FINE_MIN);

  const dim3 grid(4 * n_tiles_call + 12 * fine_k), block(GH_BLOCK);
  const uint2* ranges = (const uint2*)(wg + L.ranges);
  const uint32_t* order = (const uint32_t*)((own_order ? (const char*)ws : wg) + L.tile_order);
  const float4* r0 = (const float4*)(wg + L.inst_r0); const float4* r1 = (const float4*)(ws + L.inst_r1);
  const float2* r2 = (const float2*)(ws + L.inst_r2);
  float* fT = (float*)(ws + L.final_T); uint32_t* nc = (uint32_t*)(ws + L.n_contrib); uint32_t* tw = (uint32_t*)(ws + L.tile_walk);
  float4* ck = (float4*)(ws + L.ckpt_rgb); float4* fC = (float4*)(ws + L.final_C);
  uint2* items = (uint2*)(ws + L.bwd_items); GhCounters* ctr = (GhCounters*)(ws + L.counters);
  // the occlusion bound belongs to the call that BUILT the lists (wg == ws); a second call over shared lists has none of its own
  // (and with nothing to project the bound kernel never ran: there is no bound to read)
  const float* bound = (wg == ws && in->tile_depth_bound && g.N > 0) ? (const float*)(ws + L.tile_bound) : nullptr;
  if (wg != ws) seen = nullptr;
  const uint32_t* gid = (const uint32_t*)(wg + L.vals_a); const float4* geom = (const float4*)(wg + L.geom);
  GhFusedLoss l1 = {};
  int loss_kind = 0;
  if (fused && fused->l1_target) {                   // GhOutputs.l1_*: the image loss from the kernel's own epilogue
    loss_kind = 1;
    l1.target = fused->l1_target; l1.dL = fused->l1_dL_dimage;
    l1.inv_n = (float)(1.0 / ((double)g.NV * 3.0 * (double)g.H * (double)g.W));
    l1.sum_scale = l1.inv_n;
  } else if (fused && fused->fit_loss) {             // GhOutputs.fit_loss: the fit's image loss (needs the mask channel)
    const GhFitLoss* f = fused->fit_loss;
    loss_kind = 2;
    l1.target = f->gt_rgb; l1.gt_mask = f->gt_mask; l1.bbox = f->bbox; l1.dL = f->dL_dimage; l1.dalpha = f->dL_dalpha;
    const float HW = (float)(g.H * g.W);
    l1.k_l1 = f->scale * f->lambda_l1 / (3.0f * HW); l1.k_m = f->scale * f->lambda_mask / HW;     // as gh_fit_loss
    l1.sum_scale = 1.0f;
  }
  l1.part = (float*)(ws + L.loss_partials);
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, ranges, order, r0, r1, r2, in->cams, g.H, g.W, g.gx,
                       g.tiles, image, alpha, fT, nc, tw, ck, fC, items, ctr, (const uint32_t*)(ws + L.render_guard),
                       wg == ws ? 11u : GH_COUNTER_ERROR_MASK, bound, seen, seen_scale, seen_slack, gid, geom, l1,
                       n_tiles_call, fine_k, fine_min, (uint32_t)g.n_items, (uint32_t*)(ws + L.render_guard) + 1,
                       g.total_tiles <= GH_ORDER_TILES ? tw + 3 * (size_t)n_tiles_call : nullptr, gh_fwd_fine_launch(g) ? 1u : 0u, gh_bwd_class_mode(g));
  };
  if (loss_kind == 1) {                              // (the entry point has ruled out alpha / seen / a bound)
    if (fine_launch) launch(gh_render_fwd_kernel<false, false, 1, true>); else launch(gh_render_fwd_kernel<false, false, 1>);
    // GH_FLAG_DEFER_LOSS_SUM: the sum is a spare workgroup of the backward's render kernel (GhGrads.deferred_loss)
    if (!(d->flags & GH_FLAG_DEFER_LOSS_SUM)) gh_launch_partials_sum(l1.part, gh_loss_partial_count(g), l1.inv_n, fused->l1_loss, s);
    return;
  }
  if (loss_kind == 2) {                              // (... seen / a bound, and required alpha)
    if (fine_launch) launch(gh_render_fwd_kernel<true, false, 2, true>); else launch(gh_render_fwd_kernel<true, false, 2>);
    if (!(d->flags & GH_FLAG_DEFER_LOSS_SUM)) gh_launch_partials_sum(l1.part, gh_loss_partial_count(g), 1.0f, fused->fit_loss->loss, s);
    return;
  }
  // SEEN (GhOutputs.tile_depth_seen wanted): the variant that walks on virtually behind the stop; the plain kernels are untouched
  if (seen) { if (alpha) launch(gh_render_fwd_kernel<true, true, 0>); else launch(gh_render_fwd_kernel<false, true, 0>); }
  else if (fine_launch) { if (alpha) launch(gh_render_fwd_kernel<true, false, 0, true>); else launch(gh_render_fwd_kernel<false, false, 0, true>); }
  else { if (alpha) launch(gh_render_fwd_kernel<true, false, 0>); else launch(gh_render_fwd_kernel<false, false, 0>); }
}

// ------------------------------------------------------------------------------------------------
// ---- backward: wave = one 8x8 quadrant, lane = LIST ENTRY ------------------------------------------------
// The forward's mapping (lane = 4*pixel + slot) pays for the sequential order of its recurrence with a crossbar fetch
// of every entry value per trip and a 16-pixel reduction tree per gradient. The backward carries a 1e-3 tolerance, not
// bit-exactness, so it uses the transposed mapping: the entries of a depth segment that can reach a 4x4 pixel block are
// COMPACTED into the lanes (from the back of the list to the front), and the wave loops over the block's pixels:
//   * the entry's record sits in the lane's own registers (no fetch at all); the pixel's values come from a 32-byte LDS
//     record read with a broadcast address;
//   * the reverse recurrence over the entries becomes two prefix scans per pixel (DPP row shifts + row broadcasts, no
//     LDS): Q = running product of (1 - alpha) from the back, S = running sum of alpha*T*(d.c);
//   * every lane accumulates the nine gradient moments of ITS entry over the pixels in registers: no cross-lane
//     reduction, no atomics on HBM.
// Batches are shape-adaptive: 64 entries x 1 pixel per iteration, or — for the remainder of a list — 32 entries x 2
// pixels or 16 entries x 4 pixels (the entries replicated over the lane groups, the scans cut at the group borders), so a
// short list does not idle three quarters of the wave (measured lane fill on the bench workload 62 % -> 88 %).
// A wave owns one quadrant's four 4x4 blocks and walks them one after the other; the per-(entry, block) sums meet in a
// wave-private LDS array (plain read-modify-write in program order, so the result is bitwise reproducible), and the wave
// writes the quadrant's sub-record of every blended entry at the end. One wave per workgroup: no barriers, no cross-wave
// traffic, a wave's registers and LDS are free the moment it is done.

#define GH_BWD_ACC 128     // compact accumulator rows (entries of the segment that reach the quadrant) held in LDS at a time
#define GH_PIX_DUMMY 64    // pixel record that blends nothing (padding of the 2- / 4-pixel iterations)

// Inclusive prefix product / sum in lane order over groups of L = 64, 32 or 16 lanes: Kogge-Stone inside each row of 16
// with DPP row shifts, then the row totals with row_bcast15 (L >= 32) and row_bcast31 (L = 64). Lanes without a source
// lane keep their value (in place, bound_ctrl off). A VALU write followed by a DPP read needs two wait states (s_nop 1).
#define GH_NOP "s_nop 1\n\t"
#define GH_SCAN_ROW(OP) \
  GH_NOP OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
  GH_NOP OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
  GH_NOP OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
  GH_NOP OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
#define GH_SCAN_B15(OP) "\n\t" GH_NOP OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
#define GH_SCAN_B31(OP) "\n\t" GH_NOP OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
// The multiplicative scan works in place (a lane without a source lane must keep its value, and 0 is not the identity here);
// its caller needs 1 / input afterwards: the reciprocal is taken as the block's first instruction, so that the input register
// itself can be scanned (the compiler otherwise sinks the v_rcp below the scan and scans a copy).
template <int L>
__device__ __forceinline__ float gh_scan_mul_rcp(float v, float& rcp_in) {
  float r;
  if (L == 64) asm("v_rcp_f32 %1, %0\n\t" GH_SCAN_ROW("v_mul_f32_dpp") GH_SCAN_B15("v_mul_f32_dpp") GH_SCAN_B31("v_mul_f32_dpp") : "+v"(v), "=&v"(r));
  else if (L == 32) asm("v_rcp_f32 %1, %0\n\t" GH_SCAN_ROW("v_mul_f32_dpp") GH_SCAN_B15("v_mul_f32_dpp") : "+v"(v), "=&v"(r));
  else asm("v_rcp_f32 %1, %0\n\t" GH_SCAN_ROW("v_mul_f32_dpp") : "+v"(v), "=&v"(r));
  rcp_in = r;
  return v;
}
// The additive scan leaves its INPUT alive (the caller needs it again): the first step writes a new register — a lane without
// a source lane reads 0 (bound_ctrl), 0 + x = x — instead of working in place on a copy (one v_mov per pixel iteration).
#define GH_SCAN_ROW_ADD_OUT \
  GH_NOP "v_add_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
  GH_NOP "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
  GH_NOP "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
  GH_NOP "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
template <int L>
__device__ __forceinline__ float gh_scan_add(float x) {
  float v;
  if (L == 64) asm(GH_SCAN_ROW_ADD_OUT GH_SCAN_B15("v_add_f32_dpp") GH_SCAN_B31("v_add_f32_dpp") : "=&v"(v) : "v"(x));
  else if (L == 32) asm(GH_SCAN_ROW_ADD_OUT GH_SCAN_B15("v_add_f32_dpp") : "=&v"(v) : "v"(x));
  else asm(GH_SCAN_ROW_ADD_OUT : "=&v"(v) : "v"(x));
  return v;
}

struct GhBwdEntry {            // one list entry in the lane's registers
  float4 a, b;                 // (px, py, A, B), (C, opacity, r, g)
  float cb;                    // b
  int pos;                     // list position inside the tile
};

// One batch: cnt <= L entries (lane % L = index from the back), 64 / L pixels per iteration. am: the block's pixels that
// blend into the batch (bit i = pixel i of the block, wave-uniform); pix_base: LDS record index of the block's pixel 0.
// Adds the lane's nine sums into acc[9]; returns whether some pixel blended the lane's entry.
// GEOM = false (no gradient w.r.t. the geometry is wanted — the one-shot fit): the five position / conic moments are left out.
template <int L, bool GEOM, bool UNROLL2>
__device__ __forceinline__ bool gh_bwd_batch(const GhBwdEntry& e, bool valid, uint32_t am_in, int pix_base, int lane,
                                             float4* s_pix, float (&acc)[9]) {
  constexpr int NPX = GH_WAVE / L;
  uint32_t am = (uint32_t)__builtin_amdgcn_readfirstlane((int)am_in);
  const uint32_t slot8 = (uint32_t)(lane / L) * 8u;
  const bool seg_last = (lane % L) == L - 1;
  bool any = false;
  // record index of this lane's next pixel: the next NPX set bits of am, padded with the dummy. Scalar instructions written
  // out (s_ff1 + s_bitset0, which uses the low 5 bits of its operand, so the -1 of an empty mask is harmless): the mask
  // stays in an SGPR across the loop instead of being walked with vector instructions.
  auto pick = [&]() -> uint32_t {
    int j[NPX];
    if (NPX == 1) asm("s_ff1_i32_b32 %1, %0\n\ts_bitset0_b32 %0, %1" : "+s"(am), "=&s"(j[0]));
    else if (NPX == 2) asm("s_ff1_i32_b32 %1, %0\n\ts_bitset0_b32 %0, %1\n\ts_ff1_i32_b32 %2, %0\n\ts_bitset0_b32 %0, %2"
                           : "+s"(am), "=&s"(j[0]), "=&s"(j[NPX > 1 ? 1 : 0]));
    else asm("s_ff1_i32_b32 %1, %0\n\ts_bitset0_b32 %0, %1\n\ts_ff1_i32_b32 %2, %0\n\ts_bitset0_b32 %0, %2\n\t"
             "s_ff1_i32_b32 %3, %0\n\ts_bitset0_b32 %0, %3\n\ts_ff1_i32_b32 %4, %0\n\ts_bitset0_b32 %0, %4"
             : "+s"(am), "=&s"(j[0]), "=&s"(j[NPX > 1 ? 1 : 0]), "=&s"(j[NPX > 2 ? 2 : 0]), "=&s"(j[NPX > 3 ? 3 : 0]));
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < NPX; ++i) packed |= (uint32_t)(j[i] < 0 ? GH_PIX_DUMMY : pix_base + j[i]) << (8 * i);
    return NPX == 1 ? packed : ((packed >> slot8) & 0xFFu);
  };
  uint32_t pidx = pick();
  float4 pa0 = s_pix[2 * pidx], pa1 = s_pix[2 * pidx + 1], pb0, pb1;   // (T, B, d0, d1), (d2, last, px, py) of this lane's pixel
  bool more;
  // One pixel of the lane's sub-list; the records of the next one are loaded into (n0, n1) half way through. UNROLL2 (the one-wave
  // form): the loop below runs it with the two register sets swapped every other time — written as `p = next` the compiler copies
  // five registers per pixel (72 -> 70 vector instructions per pixel). Measured nothing while the kernel ended on a few long
  // workgroups (round 6, before its work list went longest-first); with its wave slots full 162.4 -> 161.3 us at 8 views, the step
  // -2 us; the four-wave form of small launches lost 1.7 us to it (its longest workgroup IS the kernel there) and keeps the plain loop.
  auto one_pixel = [&](const float4& p0, const float4& p1, float4& n0, float4& n1) {
    const uint32_t pcur = pidx;
    // alpha exactly as the forward evaluated it (same expression, same gh_exp): the same entries count as blended
    const float dx = e.a.x - p1.z, dy = e.a.y - p1.w;
    const float power = (e.a.z * dx * dx + e.b.x * dy * dy) - e.a.w * dx * dy;      // (A, C carry their -0.5: see the forward)
    const float G = gh_exp(power);                              // (power > 0: `contrib` is false, G is never used)
    const float alpha = fminf(0.99f, e.b.y * G);
    const bool contrib = valid && (e.pos < __float_as_int(p1.y)) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
    any |= contrib;
    const float ae = contrib ? alpha : 0.0f;                    // entries the pixel did not blend: factor 1, weight 0
    const float m1 = 1.0f - ae;
    // Q_l = product of (1 - alpha) over this entry and everything behind it in the batch:
    // T in front of the entry = (T behind the batch) / Q_l;   rm1 = 1 / (1 - alpha_l)
    float rm1;
    const float Q = gh_scan_mul_rcp<L>(m1, rm1);
    const float Tk = p0.x * __builtin_amdgcn_rcpf(Q);
    const float ec = fmaf(p1.x, e.cb, fmaf(p0.w, e.b.w, p0.z * e.b.z));      // d . c of this entry
    const float w = ae * Tk;                                               // the forward's blend weight alpha * T
    const float we = w * ec;
    acc[6] = fmaf(w, p0.z, acc[6]); acc[7] = fmaf(w, p0.w, acc[7]); acc[8] = fmaf(w, p1.x, acc[8]);
    const float sB = p0.y;
    // the pixel records of the next iteration land while the second half of this one is computed
    more = am != 0u;
    pidx = pick();                                               // all dummy once the mask is empty
    n0 = s_pix[2 * pidx]; n1 = s_pix[2 * pidx + 1];
    const float S = gh_scan_add<L>(we) + sB;          // d . (colour blended at or behind this entry) + background / mask term
    // dL/dalpha_k = T_k (d . c_k) - (d . colour strictly behind + background / mask term) / (1 - alpha_k)
    const float dLda = Tk * ec - rm1 * (S - we);
    const float h = contrib ? G * dLda : 0.0f;        // raw moments of h = G dL/dalpha; opacity / conic factors are applied
    const float hx = h * dx, hy = h * dy;             // once per (view, Gaussian) by gh_preprocess_bwd_kernel
    if (GEOM) {
      acc[0] += hx; acc[1] += hy;
      acc[2] = fmaf(hx, dx, acc[2]); acc[3] = fmaf(hx, dy, acc[3]); acc[4] = fmaf(hy, dy, acc[4]);
    }
    acc[5] += h;
    // state in front of the batch: the group's last lane holds the totals (lanes past the count: factor 1, weight 0)
    if (seg_last) *(float2*)&s_pix[2 * pcur] = make_float2(Tk, S);
  };
  if (UNROLL2) {
    for (;;) {
      one_pixel(pa0, pa1, pb0, pb1);
      if (!more) break;
      one_pixel(pb0, pb1, pa0, pa1);
      if (!more) break;
    }
  } else {
    do { one_pixel(pa0, pa1, pb0, pb1); pa0 = pb0; pa1 = pb1; } while (more);
  }
  return any;
}

// NW = waves per workgroup. 1: a wave walks the quadrant's four blocks one after the other. 4 (small launches: one or two
// views leave most of the 256 CUs idle and the longest quadrant IS the kernel): the quadrant's four blocks run in four
// waves of one workgroup, each with its own accumulator rows, and are combined in fixed wave order behind one barrier.
// GEOM = false: sub-records carry sum h and the three colour moments only (their last 16 bytes; the rest is not written).
// Leaves the workgroup's duration behind when it ends, wherever it returns (a destructor: the kernel has many exits).
// (ON = the one-wave form only: launches small enough for the four-wave form never record — gh_bwd_class_mode — and the five
//  registers this held there cost that kernel a workgroup per CU: 79 -> 84 VGPRs, one view 51 -> 55 us)
template <bool ON>
struct GhBwdCost {
  uint32_t t0; uint32_t* slot;
  __device__ ~GhBwdCost() {
    if (ON) {
      const uint32_t dt = ((uint32_t)__builtin_readcyclecounter() - t0) >> 8;       // (a workgroup lives far less than 2^32 cycles)
      if (threadIdx.x == 0 && slot) __hip_atomic_fetch_max(slot, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
};
#ifdef GH_EXP_BWD_TIME
// (experiment: per-workgroup cycle counters of the backward, read back by gh_exp_read_bwd_times: tools/experiments/bwd_lpt_sim.py)
__device__ uint4 gh_exp_bwd_times[1 << 18];
struct GhExpTimer {
  uint64_t t0; uint32_t slot, item, tile;
  __device__ ~GhExpTimer() {
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && slot < (1u << 18)) gh_exp_bwd_times[slot] = make_uint4((uint32_t)(t1 - t0), item, tile, (uint32_t)t0);
  }
};
extern "C" int gh_exp_read_bwd_times(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gh_exp_bwd_times), bytes); }
extern "C" int gh_exp_clear_bwd_times() { void* p; if (hipGetSymbolAddress(&p, HIP_SYMBOL(gh_exp_bwd_times)) != hipSuccess) return -1; return (int)hipMemset(p, 0, sizeof(uint4) << 18); }
#endif
template <bool ALPHA, int NW, bool GEOM>
__global__ __launch_bounds__(NW * GH_WAVE) void gh_render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint2* __restrict__ items, const GhCounters* __restrict__ ctr,
    const uint32_t* __restrict__ sorted_slot,
    const float4* __restrict__ r0, const float4* __restrict__ r1, const float2* __restrict__ r2, const float* __restrict__ cams,
    int H, int W, int gx, int tiles, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float4* __restrict__ ckpt_rgb, const float4* __restrict__ final_C,
    const float* __restrict__ dL_dimage, const float* __restrict__ dL_dalpha_img, const float* __restrict__ upstream_scale,
    float* __restrict__ inst_grad, uint8_t* __restrict__ inst_flag,
    const float4* __restrict__ loss_part, uint32_t loss_n4, float* __restrict__ loss_out,
    const uint32_t* __restrict__ class_count, uint32_t n_items_cap, uint32_t* __restrict__ cost, uint32_t use_classes) {
  // GhGrads.deferred_loss: the fused image loss's final sum (GH_FLAG_DEFER_LOSS_SUM left the forward without its one-workgroup sum
  // kernel) by workgroup 0 of this launch — dispatched first, done long before the kernel's tail; every other workgroup's
  // index moves down by one. Fixed order: bitwise reproducible.
  uint32_t bid = blockIdx.x, nblk_items = gridDim.x;
  if (loss_out) {
    if (bid == 0u) {
      __shared__ float s_ls[NW];
      float a = 0.0f, b = 0.0f, c = 0.0f, e = 0.0f;
      for (uint32_t i = threadIdx.x; i < loss_n4; i += NW * GH_WAVE) { const float4 v = loss_part[i]; a += v.x; b += v.y; c += v.z; e += v.w; }
      const float sw = gh_wave_sum_to63((a + b) + (c + e));
      const float scale = ((const float*)loss_part)[(size_t)loss_n4 * 4];          // left by the forward behind the partials
      if (NW == 1) { if (threadIdx.x == 63) loss_out[0] = scale * sw; }
      else {
        if ((threadIdx.x & 63) == 63) s_ls[threadIdx.x >> 6] = sw;
        __syncthreads();
        if (threadIdx.x == 0) { float t = 0.0f;
#pragma unroll
          for (int k = 0; k < NW; ++k) t += s_ls[k];
          loss_out[0] = scale * t; }
      }
      return;
    }
    bid -= 1u; nblk_items -= 1u;
  }
  __shared__ float s_acc_all[NW][GH_BWD_ACC * GH_REC];    // per wave: [compact quadrant entry][9]
  __shared__ float4 s_pix[2 * (GH_WAVE + 1)];             // per pixel (T, B, d0, d1), (d2, last, px, py); last record = dummy
  __shared__ uint16_t s_q_all[NW][GH_SEGMENT];            // per wave: compacted entry list of the block being walked: raw | compact << 8
  __shared__ uint8_t s_f_all[NW][GH_BWD_ACC];             // per wave: 1 where some pixel of its block(s) blended the entry
  const int wv = NW == 1 ? 0 : (int)(threadIdx.x >> 6);
  float* s_acc = s_acc_all[wv];
  uint16_t* s_q = s_q_all[wv];
  uint8_t* s_f = s_f_all[wv];
  // The work list, written by the forward (the grid is sized for its capacity): GH_BWD_CLASSES regions of n_items_cap entries, an item
  // in the region of what it cost the previous backward (see gh_render_fwd_kernel); the most expensive class goes first, inside a class
  // the tiles the forward finished last. (use_classes = 0 — large launches, or the A/B switch: the forward put everything into class 0.)
  uint32_t item_idx, quad_u;
  gh_item_quad(bid, nblk_items >> 2, item_idx, quad_u);       // the four quadrants of an item share an XCD (L2)
  // (wave-uniform scalar work: a running total from the top class down; the item sits in the first class whose running total passes
  //  its index, at that total - 1 - index from the region's start = the region read from its end)
  // (written as the five scalar instructions per class it is: the compiler built 64-bit lane masks and vector adds out of the
  //  C form — 270 instructions in front of every workgroup's first load, 2 us of a small launch)
  static_assert(GH_BWD_CLASSES == 16, "sixteen counts, one 64-byte line");
  uint32_t acc = 0u, before = 0u, first_end = 0xFFFFFFFFu, tmp;
  // (s_cselect reads the compare's SCC before s_addc consumes AND overwrites it)
#define GH_CLS_STEP(N) "s_add_u32 %2, %2, %" #N "\n\ts_cmp_le_u32 %2, %4\n\ts_cselect_b32 %3, -1, %2\n\ts_addc_u32 %0, %0, 0\n\ts_min_u32 %1, %1, %3\n\t"
  asm(GH_CLS_STEP(20) GH_CLS_STEP(19) GH_CLS_STEP(18) GH_CLS_STEP(17) GH_CLS_STEP(16) GH_CLS_STEP(15) GH_CLS_STEP(14) GH_CLS_STEP(13)
      GH_CLS_STEP(12) GH_CLS_STEP(11) GH_CLS_STEP(10) GH_CLS_STEP(9) GH_CLS_STEP(8) GH_CLS_STEP(7) GH_CLS_STEP(6) GH_CLS_STEP(5)
      : "+s"(before), "+s"(first_end), "+s"(acc), "=&s"(tmp)
      : "s"(item_idx), "s"(class_count[0]), "s"(class_count[1]), "s"(class_count[2]), "s"(class_count[3]), "s"(class_count[4]),
        "s"(class_count[5]), "s"(class_count[6]), "s"(class_count[7]), "s"(class_count[8]), "s"(class_count[9]), "s"(class_count[10]),
        "s"(class_count[11]), "s"(class_count[12]), "s"(class_count[13]), "s"(class_count[14]), "s"(class_count[15])
      : "scc");
#undef GH_CLS_STEP
  if (before == GH_BWD_CLASSES) return;                         // past the end of the list
  const uint32_t cls = GH_BWD_CLASSES - 1u - before;
  const uint2 item_v = items[(size_t)cls * n_items_cap + (first_end - 1u - item_idx)];
  const uint2 item = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)item_v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)item_v.y));
  const int tile = (int)item.x, quad = (int)quad_u;       // (wave-uniform values in scalar registers by hand: see the forward)
#ifdef GH_EXP_BWD_TIME
  GhExpTimer exp_timer = {__builtin_readcyclecounter(), bid, item_idx * 4u + quad_u, item.x | (item.y << 24)};
#endif
  // what this workgroup costs, for the next call's work list: cycles / 256 of the item's slowest quadrant, per (tile, segment)
  GhBwdCost<NW == 1> cost_rec = {NW == 1 ? (uint32_t)__builtin_readcyclecounter() : 0u,
                                 NW == 1 && use_classes ? cost + (size_t)item.x * GH_BWD_COST_SLOTS + (item.y < GH_BWD_COST_SLOTS ? item.y : GH_BWD_COST_SLOTS - 1u) : nullptr};
  const int seg_lo = (int)item.y * GH_SEGMENT, seg_hi = seg_lo + GH_SEGMENT;
  int v, tx, ty;
  gh_tile_coords(tile, gx, tiles, v, tx, ty);
  const int lane = threadIdx.x & 63;
  // pixel state: one pixel per lane, lane = 16 * block + 4 * row + column
  const int lb = lane >> 4, lp = lane & 15;
  const int x = tx * GH_TILE + (quad & 1) * 8 + (lb & 1) * 4 + (lp & 3), y = ty * GH_TILE + (quad >> 1) * 8 + (lb >> 1) * 4 + (lp >> 2);
  const bool inside = x < W && y < H;
  const uint2 range_v = ranges[tile];
  const uint2 range = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.y));
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
  int qlast = last;                        // list positions >= qlast were blended by no pixel of this quadrant
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(qlast, o); qlast = t > qlast ? t : qlast; }
  if (qlast <= seg_lo) return;             // wave-uniform: the quadrant blended nothing inside this depth segment
  const int qend = qlast < seg_hi ? qlast : seg_hi;
  const int nb = (qend - seg_lo + GH_WAVE - 1) / GH_WAVE;             // raw batches of 64 list entries in the segment (1..4)
  int blast = last;                        // ... by no pixel of the lane's 4x4 block (max over the DPP row)
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { int t = __shfl_xor(blast, o); blast = t > blast ? t : blast; }

  // Per-pixel state behind the segment, ABSOLUTE form: T = transmittance in front of list position `cut`, B = d . (colour
  // blended at positions >= cut) + the background / mask term T_final * (bg . d - dM); both enter dL/dalpha_k as
  // -(...) / (1 - alpha_k). A pixel whose last blended entry lies inside (or before) the segment starts from (T_final,
  // nothing behind); one that blends entries behind the cut starts from the forward's own state at the cut (exact T).
  float vT = T_final;
  float vB = T_final * ((bg[0] * d0 + bg[1] * d1 + bg[2] * d2) - dM);
  if (inside && last > seg_hi) {
    const size_t ck = ((size_t)(range.x / GH_SEGMENT) + (size_t)tile + (size_t)item.y) * 256 +
                      (size_t)((y - ty * GH_TILE) * GH_TILE + (x - tx * GH_TILE));
    const float4 c = ckpt_rgb[ck];
    const float4 fc = final_C[((size_t)v * H + y) * W + x];
    vT = c.x;
    vB += fmaf(d2, fc.z - c.w, fmaf(d1, fc.y - c.z, d0 * (fc.x - c.y)));
  }
  if (NW == 1 || lb == wv) {                           // (NW = 4: every wave owns the records of its own block's pixels)
    s_pix[2 * lane] = make_float4(vT, vB, d0, d1);
    s_pix[2 * lane + 1] = make_float4(d2, __int_as_float(last), (float)x, (float)y);
  }
  if (lane == 0) {
    s_pix[2 * GH_PIX_DUMMY] = make_float4(1.0f, 0.0f, 0.0f, 0.0f);
    s_pix[2 * GH_PIX_DUMMY + 1] = make_float4(0.0f, __int_as_float(0), 0.0f, 0.0f);       // last = 0: blends nothing
  }

  // 4x4-block masks of the segment's entries: lane l holds entries l, l + 64, l + 128, l + 192 of the segment
  const uint32_t quad_bits = 0x33u << (8 * (quad >> 1) + 2 * (quad & 1));          // the quadrant's four blocks in the entry masks
  uint32_t mk[GH_SEGMENT / GH_WAVE];
  int qcnt[GH_SEGMENT / GH_WAVE];                                                   // entries per raw batch that reach the quadrant
#pragma unroll
  for (int k = 0; k < GH_SEGMENT / GH_WAVE; ++k) {
    const int pos = seg_lo + k * GH_WAVE + lane;
    mk[k] = (k < nb && pos < qend) ? (__float_as_uint(r2[pos].y) & quad_bits) : 0u;
    qcnt[k] = __popcll(gh_ballot(mk[k] != 0u));
  }
  __builtin_amdgcn_wave_barrier();

  // Chunks of consecutive raw batches, from the back, whose entries fit the accumulator rows (nearly always the whole segment)
  int k_hi = nb - 1;
  while (k_hi >= 0) {
    int k_lo = k_hi, rows = 0;
#pragma unroll
    for (int k = GH_SEGMENT / GH_WAVE - 1; k >= 0; --k)
      if (k <= k_hi && k == k_lo && (k == k_hi || rows + qcnt[k] <= GH_BWD_ACC)) { rows += qcnt[k]; k_lo = k - 1; }
    k_lo += 1;                                                  // chunk = raw batches [k_lo, k_hi], `rows` accumulator rows
    // compact row of each of the lane's entries inside the chunk
    int crow[GH_SEGMENT / GH_WAVE];
    {
      int base = 0;
#pragma unroll
      for (int k = 0; k < GH_SEGMENT / GH_WAVE; ++k) {
        crow[k] = 0;
        if (k >= k_lo && k <= k_hi) {
          const uint64_t qm = gh_ballot(mk[k] != 0u);
          crow[k] = base + __popcll(qm & ((1ull << lane) - 1ull));
          base += __popcll(qm);
        }
      }
    }
    for (int i = lane; i < rows * GH_REC; i += GH_WAVE) s_acc[i] = 0.0f;
    for (int i = lane; i < rows; i += GH_WAVE) s_f[i] = 0;
    __builtin_amdgcn_wave_barrier();

#pragma unroll 1
    for (int b = NW == 1 ? 0 : wv; b < (NW == 1 ? 4 : wv + 1); ++b) {
      const int bl = __builtin_amdgcn_readlane(blast, 16 * b);          // first list position no pixel of the block blended
      if (bl <= seg_lo + k_lo * GH_WAVE) continue;                     // the block blended nothing inside this chunk
      const int bend = bl < seg_hi ? bl : seg_hi;
      const int bit = ((quad >> 1) * 2 + (b >> 1)) * 4 + (quad & 1) * 2 + (b & 1);      // the block's bit in the entry masks
      // ---- compaction: the entries that can reach the block, from the back of the list to the front
      int nhit = 0;
#pragma unroll
      for (int k = GH_SEGMENT / GH_WAVE - 1; k >= 0; --k) {
        if (k < k_lo || k > k_hi) continue;
        const bool hit = ((mk[k] >> bit) & 1u) && (seg_lo + k * GH_WAVE + lane < bend);
        const uint64_t hm = gh_ballot(hit);
        if (hit) ((volatile uint16_t*)s_q)[nhit + __popcll((hm >> lane) >> 1)] = (uint16_t)((k * GH_WAVE + lane) | (crow[k] << 8));
        nhit += __popcll(hm);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      for (int j0 = 0; j0 < nhit;) {
        const int rem = nhit - j0;
        // batch shape: 64 lanes x 1 pixel, or for short remainders 32 x 2 / 16 x 4 (rem in (32, 48] goes as 32 + a 16-lane batch)
        const int L = rem > 48 ? 64 : (rem > 16 ? 32 : 16);
        const int cnt = rem < L ? rem : L;
        const int el = lane & (L - 1);
        const bool valid = el < cnt;
        const uint32_t qe = valid ? (uint32_t)((volatile uint16_t*)s_q)[j0 + el] : 0u;
        const int ci = (int)(qe >> 8);
        GhBwdEntry e;
        e.pos = seg_lo + (int)(qe & 0xFFu);
        e.a = r0[e.pos]; e.b = r1[e.pos]; e.cb = r2[e.pos].x;
        const int minpos = __builtin_amdgcn_readlane(e.pos, cnt - 1);     // front-most entry of the batch
        const uint32_t am = (uint32_t)(gh_ballot(last > minpos) >> (16 * b)) & 0xFFFFu;     // pixels of the block that blend into the batch
        float acc[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        bool any = false;
        if (am) {
          if (L == 64) any = gh_bwd_batch<64, GEOM, NW == 1>(e, valid, am, 16 * b, lane, s_pix, acc);
          else if (L == 32) any = gh_bwd_batch<32, GEOM, NW == 1>(e, valid, am, 16 * b, lane, s_pix, acc);
          else any = gh_bwd_batch<16, GEOM, NW == 1>(e, valid, am, 16 * b, lane, s_pix, acc);
        }
        // The lane groups of a 32- / 16-lane batch hold partial sums of the SAME entries (different pixels): fold them into
        // group 0 (fixed order). Then plain read-modify-write of the entry's accumulator row: the wave owns the rows and the
        // entries of a batch are distinct (LDS float atomics cost 0.19 ms of a 0.36 ms kernel when this was ds_add_f32).
        int anyi = any ? 1 : 0;
        if (L < 64) {
#pragma unroll
          for (int q = GEOM ? 0 : 5; q < GH_REC; ++q) acc[q] += __shfl_xor(acc[q], 32);
          anyi |= __shfl_xor(anyi, 32);
          if (L < 32) {
#pragma unroll
            for (int q = GEOM ? 0 : 5; q < GH_REC; ++q) acc[q] += __shfl_xor(acc[q], 16);
            anyi |= __shfl_xor(anyi, 16);
          }
        }
        if (lane < L && anyi) {                                       // lane's entry was blended by some pixel of the block
          float* d = s_acc + ci * GH_REC;
#pragma unroll
          for (int q = GEOM ? 0 : 5; q < GH_REC; ++q) d[q] += acc[q];
          ((volatile uint8_t*)s_f)[ci] = 1;
        }
        __builtin_amdgcn_wave_barrier();
        j0 += cnt;
      }
    }
    // ---- the quadrant's sub-record of every entry some pixel blended, at the instance's emit slot
    if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
#pragma unroll
    for (int k = 0; k < GH_SEGMENT / GH_WAVE; ++k) {
      if (k < k_lo || k > k_hi || (NW > 1 && (k & (NW - 1)) != wv)) continue;      // NW = 4: raw batch k is flushed by wave k
      if (mk[k] == 0u) continue;
      float s9[GH_REC] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      bool any = false;
#pragma unroll
      for (int w = 0; w < NW; ++w) {                     // fixed wave order: bitwise reproducible
        if (((volatile uint8_t*)s_f_all[w])[crow[k]]) {
          any = true;
          const float* a9 = s_acc_all[w] + crow[k] * GH_REC;
#pragma unroll
          for (int q = GEOM ? 0 : 5; q < GH_REC; ++q) s9[q] += a9[q];
        }
      }
      if (any) {
        const uint32_t sl = slots[seg_lo + k * GH_WAVE + lane];
        GhF3* rec = (GhF3*)(inst_grad + ((size_t)sl * 4 + quad) * GH_REC_G);
        if (GEOM) {
          rec[0] = GhF3{s9[0], s9[1], s9[2]};
          rec[1] = GhF3{s9[3], s9[4], s9[5]};
        } else ((float*)rec)[5] = s9[5];
        rec[2] = GhF3{s9[6], s9[7], s9[8]};
        inst_flag[(size_t)sl * 4 + quad] = 1;
      }
    }
    if (NW == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    k_hi = k_lo - 1;
  }
}

void gh_launch_render_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const float* dL_dimage,
                          const float* dL_dalpha, const float* upstream_scale, const char* wg, char* ws, const GhLayout& L, hipStream_t s,
                          bool geom, float* deferred_loss) {
  const size_t n_part = gh_loss_partial_count(g);          // (as the forward's fused loss left them)
  if (g.cap == 0) {                                       // nothing was listed: no render backward; a deferred sum still has to run
    if (deferred_loss) gh_launch_partials_sum((const float*)(ws + L.loss_partials), n_part, 1.0f, deferred_loss, s, true);
    return;
  }
  // inst_flag was cleared by gh_ranges_kernel; repeated backwards set the same flags again (they depend on the forward's
  // n_contrib only). The work list (tile, depth segment) was written by the forward's last wave of every tile.
  const bool small = g.total_tiles <= GH_FINE_TILES;    // few tiles: four waves per quadrant (one per 4x4 block)

  const dim3 grid(4 * (unsigned)g.n_items + (deferred_loss ? 1u : 0u)), block(small ? 4 * GH_WAVE : GH_WAVE);   // capacity of the work list x 4 quadrants; surplus workgroups exit at once
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, block, 0, s, (const uint2*)(wg + L.ranges), (const uint2*)(ws + L.bwd_items),
                       (const GhCounters*)(ws + L.counters),
                       (const uint32_t*)(wg + L.sorted_slot), (const float4*)(wg + L.inst_r0), (const float4*)(ws + L.inst_r1),
                       (const float2*)(ws + L.inst_r2), in->cams, g.H, g.W, g.gx, g.tiles, (const float*)(ws + L.final_T),
                       (const uint32_t*)(ws + L.n_contrib), (const float4*)(ws + L.ckpt_rgb),
                       (const float4*)(ws + L.final_C), dL_dimage, dL_dalpha, upstream_scale, (float*)(ws + L.inst_grad),
                       (uint8_t*)(ws + L.inst_flag), (const float4*)(ws + L.loss_partials), (uint32_t)(n_part / 4), deferred_loss,
                       (const uint32_t*)(ws + L.render_guard) + 1, (uint32_t)g.n_items,
                       (uint32_t*)(ws + L.tile_walk) + 4 * (size_t)g.NV * g.tiles, gh_bwd_class_mode(g) == 1u ? 1u : 0u);
  };
  // geom = false (precomputed colours and no geometry gradient wanted): colour / opacity moments only, see the kernel
  if (small) {
    if (geom) { if (dL_dalpha) launch(gh_render_bwd_kernel<true, 4, true>); else launch(gh_render_bwd_kernel<false, 4, true>); }
    else { if (dL_dalpha) launch(gh_render_bwd_kernel<true, 4, false>); else launch(gh_render_bwd_kernel<false, 4, false>); }
  } else {
    if (geom) { if (dL_dalpha) launch(gh_render_bwd_kernel<true, 1, true>); else launch(gh_render_bwd_kernel<false, 1, true>); }
    else { if (dL_dalpha) launch(gh_render_bwd_kernel<true, 1, false>); else launch(gh_render_bwd_kernel<false, 1, false>); }
  }
}

// ------------------------------------------------------------------------------------------------
// Second call over the same geometry (gh_forward_shared): the per-instance colour part of the render records
// (inst_r1.zw, inst_r2.x) is rebuilt from this call's colours; conic / opacity / block mask are copied from the geometry
// owner. Also resets this call's own per-tile walk state, backward flags and counters (D and the overflow flag are the
// geometry owner's).
__global__ __launch_bounds__(GH_BLOCK) void gh_recolour_kernel(GhInputs in, uint32_t flags, int P, int T, uint32_t cap,
                                                                const GhCounters* __restrict__ gctr, const uint32_t* __restrict__ vals,
                                                                const float4* __restrict__ g_r1, const float2* __restrict__ g_r2,
                                                                float4* __restrict__ r1, float2* __restrict__ r2,
                                                                uint32_t* __restrict__ inst_flag, uint32_t* __restrict__ tile_walk,
                                                                GhCounters* __restrict__ ctr, uint32_t* __restrict__ render_guard) {
  const uint32_t i = blockIdx.x * GH_BLOCK + threadIdx.x;
  const uint32_t D = gctr->num_rendered;
  if (i == 0) {
    ctr->num_rendered = D; ctr->overflow = gctr->overflow; ctr->reserved[0] = gctr->reserved[0]; ctr->reserved[1] = 0;
    *render_guard = gctr->overflow & GH_COUNTER_ERROR_MASK;        // (this is the kernel in front of the render, see gh_render_fwd_kernel)
    for (int c = 0; c < GH_BWD_CLASSES; ++c) render_guard[1 + c] = 0u;       // (the work list's class counts: gh_render_fwd_kernel)
  }
  if (i < (uint32_t)T) { tile_walk[i] = 0u; tile_walk[T + i] = 0u; tile_walk[2 * T + i] = 0u; }
  const uint32_t n = D < cap ? D : cap;
  if (i >= n) return;
  inst_flag[i] = 0;
  const uint32_t gid = vals[i];
  const int row = (flags & GH_FLAG_PER_VIEW_GAUSSIANS) ? (int)gid : (int)(gid % (uint32_t)P);
  float rgb[3];
  gh_blended_rgb(in, flags, row, rgb);
  const float4 a = g_r1[i];
  const float2 b = g_r2[i];
  r1[i] = make_float4(a.x, a.y, rgb[0], rgb[1]);
  r2[i] = make_float2(rgb[2], b.y);
}

void gh_launch_recolour(const GhDims* d, const GhGrid& g, const GhInputs* in, const char* wg, char* ws, const GhLayout& L, hipStream_t s) {
  const int T = g.NV * g.tiles;
  const size_t n = (size_t)g.cap > (size_t)T ? (size_t)g.cap : (size_t)T;
  const int nblk = (int)((n + GH_BLOCK - 1) / GH_BLOCK);
  hipLaunchKernelGGL(gh_recolour_kernel, dim3(nblk > 0 ? nblk : 1), dim3(GH_BLOCK), 0, s, *in, d->flags, g.P, T, (uint32_t)g.cap,
                     (const GhCounters*)(wg + L.counters), (const uint32_t*)(wg + L.vals_a), (const float4*)(wg + L.inst_r1),
                     (const float2*)(wg + L.inst_r2), (float4*)(ws + L.inst_r1), (float2*)(ws + L.inst_r2),
                     (uint32_t*)(ws + L.inst_flag), (uint32_t*)(ws + L.tile_walk), (GhCounters*)(ws + L.counters),
                     (uint32_t*)(ws + L.render_guard));
}

// ------------------------------------------------------------------------------------------------
// A later step over static geometry (gh_forward_refresh): the lists of `wg` were built with GH_FLAG_STATIC_LISTS (tiles culled
// with gh_static_cull_opacity: max(2, twice the opacity of the build call)); this call's opacities and colours differ from the
// build call's. Two streaming kernels rebuild the part of the render records that moved:
//   gh_refresh_attr_kernel      one thread per (view, Gaussian): (opacity, r, g, b) of THIS step as one 16-byte record, and the
//                               GUARD — above the opacity its tiles were culled with (cull_bound) a tile may be missing from the
//                               lists, so the call is poisoned like an instance overflow (GhCounters.overflow |= 2 -> NaN image);
//                               also this call's per-tile walk state. (The counters are written by a one-wave kernel in front —
//                               NOT a memset: inside a replayed HIP graph a 16-byte memset node was seen to land after the
//                               kernels that follow it on ROCm 7.2, and a store by thread 0 of this kernel could land after
//                               another block's atomicOr);
//   gh_refresh_instance_kernel  one thread per sorted instance: its Gaussian's record (one 16-byte gather), the static centre /
//                               conic (inst_r0, inst_c of the geometry owner) and tile id (its sorted keys) -> opacity, colour and
//                               the 4x4-block mask OF THE CURRENT OPACITY, so that the render kernels skip, with one bit test,
//                               the instances the exact culling of a full call would not have listed.
__global__ void gh_refresh_init_kernel(const GhCounters* __restrict__ gctr, GhCounters* __restrict__ ctr) {
  if (threadIdx.x == 0) {
    ctr->num_rendered = gctr->num_rendered; ctr->overflow = gctr->overflow; ctr->reserved[0] = gctr->reserved[0]; ctr->reserved[1] = 0u;
  }
}

__global__ __launch_bounds__(GH_BLOCK) void gh_refresh_attr_kernel(GhInputs in, uint32_t flags, int P, int NV, int N, int T,
                                                                    const GhCounters* __restrict__ gctr, const float* __restrict__ cull_bound,
                                                                    const float4* __restrict__ sh_rgb, float4* __restrict__ attr,
                                                                    uint32_t* __restrict__ tile_walk, GhCounters* __restrict__ ctr,
                                                                    int n_attr_blocks, const uint2* __restrict__ ranges, uint32_t* __restrict__ order,
                                                                    int tiles_per_view) {
  // spare workgroups (launches of at most GH_ORDER_TILES tiles): the forward's launch order, ranked anew from what the PREVIOUS step's
  // forward measured per tile (this call's tile_walk[3]; a tile without a measurement: by its list length) into this call's own
  // tile_order — the build call's order is by list length, and its choice of fine-grained tiles with it
  // (the grid's FIRST workgroups: they run in the shadow of the others instead of being the kernel's tail)
  const int n_rank = (int)gridDim.x - n_attr_blocks;
  if ((int)blockIdx.x < n_rank) {
    gh_rank_tiles(ranges, tiles_per_view, NV, (int)blockIdx.x, order, tile_walk + 3 * (size_t)T);
    return;
  }
  const int t = ((int)blockIdx.x - n_rank) * GH_BLOCK + threadIdx.x;
  if (t < T) { tile_walk[t] = 0u; tile_walk[T + t] = 0u; tile_walk[2 * T + t] = 0u; }
  if (t >= N) return;
  const int row = (flags & GH_FLAG_PER_VIEW_GAUSSIANS) ? t : t % P;       // n = view * P + row (view-major, as the lists' payload)
  float op = in.opacities[row];
  if (in.blend_opacity_b) op = op + in.blend_opacity_b[row];
  float rgb[3];
  if (in.colors_precomp) gh_blended_rgb(in, flags, row, rgb);
  else { const float4 c4 = sh_rgb[t]; rgb[0] = c4.x; rgb[1] = c4.y; rgb[2] = c4.z; }
  attr[t] = make_float4(op, rgb[0], rgb[1], rgb[2]);
  if (op > cull_bound[t]) atomicOr(&ctr->overflow, 2u);
}

__global__ __launch_bounds__(GH_BLOCK) void gh_refresh_instance_kernel(uint32_t cap, int gx, int tiles, float rtiles, float rgx,
                                                                        const GhCounters* __restrict__ gctr,
                                                                        const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                        const float4* __restrict__ g_r0, const float* __restrict__ g_c,
                                                                        const float4* __restrict__ attr, float4* __restrict__ r1,
                                                                        float2* __restrict__ r2, uint32_t* __restrict__ inst_flag,
                                                                        const GhCounters* __restrict__ ctr, uint32_t* __restrict__ render_guard,
                                                                        int local_keys) {
  const uint32_t i = blockIdx.x * GH_BLOCK + threadIdx.x;
  // the kernel in front of the render (gh_refresh_attr_kernel, complete by now, may have raised bit 1): the error bits in their own word
  if (i == 0) { *render_guard = ctr->overflow & GH_COUNTER_ERROR_MASK; for (int c = 0; c < GH_BWD_CLASSES; ++c) render_guard[1 + c] = 0u; }
  const uint32_t D = gctr->num_rendered;
  const uint32_t n = D < cap ? D : cap;
  if (i >= n) return;
  inst_flag[i] = 0;
  const float4 at = attr[vals[i]];                      // (opacity, r, g, b) of the instance's (view, Gaussian)
  const float4 a = g_r0[i];
  const float cC = g_c[i];
  const uint32_t t = keys[i];                            // global tile id, or (per-view partition: local_keys) the tile id inside the view
  uint32_t tl, ty;
  if (local_keys) { tl = t; ty = gh_div_small(tl, (uint32_t)gx, rgx); }
  else if (rtiles > 0.0f) { tl = t - gh_div_small(t, (uint32_t)tiles, rtiles) * (uint32_t)tiles; ty = gh_div_small(tl, (uint32_t)gx, rgx); }
  else { tl = t % (uint32_t)tiles; ty = tl / (uint32_t)gx; }
  const uint32_t tx = tl - ty * (uint32_t)gx;
  // (inst_r0 holds -A/2 for the render kernels' power; the mask test takes the conic itself: * -2 is exact)
  const float4 a_t = make_float4(a.x, a.y, -2.0f * a.z, a.w);
  const float4 b = make_float4(cC, at.x, at.y, at.z);
  const uint32_t m = gh_block_mask16(a_t, b, (float)(tx * GH_TILE), (float)(ty * GH_TILE));
  r1[i] = make_float4(-0.5f * cC, at.x, at.y, at.z);
  r2[i] = make_float2(at.w, __uint_as_float(m));
}

bool gh_launch_refresh(const GhDims* d, const GhGrid& g, const GhInputs* in, const char* wg, char* ws, const GhLayout& L, hipStream_t s) {
  const int T = g.NV * g.tiles;
  const int na = g.N > T ? g.N : T;
  const bool rank = g.total_tiles <= GH_ORDER_TILES && gh_heavy_order_enabled() && !(g.flags & GH_FLAG_FRESH_ORDER);
  const int n_attr_blocks = (na + GH_BLOCK - 1) / GH_BLOCK;
  hipLaunchKernelGGL(gh_refresh_init_kernel, dim3(1), dim3(GH_WAVE), 0, s, (const GhCounters*)(wg + L.counters), (GhCounters*)(ws + L.counters));
  hipLaunchKernelGGL(gh_refresh_attr_kernel, dim3(n_attr_blocks + (rank ? g.NV : 0)), dim3(GH_BLOCK), 0, s, *in, d->flags, g.P, g.NV, g.N, T,
                     (const GhCounters*)(wg + L.counters), (const float*)(wg + L.cull_bound), (const float4*)(ws + L.sh_rgb),
                     (float4*)(ws + L.attr), (uint32_t*)(ws + L.tile_walk), (GhCounters*)(ws + L.counters),
                     n_attr_blocks, (const uint2*)(wg + L.ranges), (uint32_t*)(ws + L.tile_order), g.tiles);
  const int nblk = (int)(((size_t)g.cap + GH_BLOCK - 1) / GH_BLOCK);
  hipLaunchKernelGGL(gh_refresh_instance_kernel, dim3(nblk > 0 ? nblk : 1), dim3(GH_BLOCK), 0, s, (uint32_t)g.cap, g.gx, g.tiles,
                     (long long)g.NV * g.tiles < (1ll << 24) ? 1.0f / (float)g.tiles : 0.0f, 1.0f / (float)g.gx,
                     (const GhCounters*)(wg + L.counters), (const uint32_t*)(wg + L.keys_a), (const uint32_t*)(wg + L.vals_a),
                     (const float4*)(wg + L.inst_r0), (const float*)(wg + L.inst_c), (const float4*)(ws + L.attr),
                     (float4*)(ws + L.inst_r1), (float2*)(ws + L.inst_r2), (uint32_t*)(ws + L.inst_flag),
                     (const GhCounters*)(ws + L.counters), (uint32_t*)(ws + L.render_guard), gh_partition_per_view(g) ? 1 : 0);
  return rank;
}
