// gh_binning.hip — tile binning for the rasteriser, hand-written for wave64.
//
// The published algorithm sorts D tile instances by the 64-bit key (tile << 32 | depth bits). The same ordering
// is produced here in two levels, which moves 3x fewer bytes:
//   1. every view's P Gaussians are stably sorted by depth among themselves (32-bit keys, views = independent
//      segments of one batched sort; ties keep index order, culled ones go last in their view),
//   2. instances are emitted in that order (view-major, depth-minor; one per touched tile, row-major in the rect),
//   3. a STABLE radix partition by global tile id (32-bit keys, D elements, ceil(tile_bits/8) passes) gathers each
//      tile's entries without disturbing their depth order  ==  stable sort by (tile, depth, index).
// (A partition by the tile index inside the view alone — one 1024-digit pass instead of two 8-bit ones — was built and
// measured: the wide pass cost what the two narrow ones cost, and the tile-major / view-minor list layout it produces
// made the record gather 16 us slower: neighbouring lists no longer share Gaussians.)
// Then per-tile [start,end) ranges, the per-instance render records in sorted order and the launch order.
// The instance count D never leaves the device: kernels read their element count from device memory and grids
// are sized from the caller's capacity (max_instances), so the whole stage is sync-free / graph-capturable.
#include "gh_internal.h"
GH_WG_TIMER_TU(bin)
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// LSD radix sort engine: 32-bit keys + 32-bit payload, digits of up to 8 bits (MAXD = 256 digit slots).
// Each block owns GH_BLOCK * ITEMS consecutive keys of ONE segment. Two shapes:
//   * one segment whose element count is read from device memory (*n_ptr, clamped to cap): seg_len = 0, gridDim.y = 1;
//   * gridDim.y independent segments of exactly seg_len elements each, segment g = elements [g*seg_len, (g+1)*seg_len):
//     the per-view depth sort (every view's P Gaussians are sorted among themselves, in place in their segment).
// Tables: table[(seg * ndig + digit) * nblk + block], tot[seg * ndig + digit]; nblk = gridDim.x = blocks per segment.
__device__ __forceinline__ uint32_t gh_clamp_n(const uint32_t* n_ptr, uint32_t cap) {
  const uint32_t n = *n_ptr;
  return n < cap ? n : cap;
}

__device__ __forceinline__ uint32_t gh_seg_count(const uint32_t* n_ptr, uint32_t cap, uint32_t seg_len) {
  return seg_len ? seg_len : gh_clamp_n(n_ptr, cap);
}

// Third shape (round 6): `nseg` segments of VARIABLE length whose bounds live in device memory — vstart[0 .. nseg], ascending,
// clamped to cap here — the per-view tile partition: view v's instances are emit slots [vstart[v], vstart[v+1]). The grid is one
// row of ceil(cap / tile) + nseg blocks; block b finds its segment by walking the bounds (wave-uniform scalar loads, nseg <= a few
// dozen); the digit tables are indexed by the GLOBAL block number b, the per-digit totals by (segment, digit).
struct GhVarSeg { uint32_t seg, lb, n, s0, gb0, nb; };   // segment, block inside it, its element count / first element / first global block / blocks
__device__ __forceinline__ bool gh_var_seg_of_block(const uint32_t* __restrict__ vstart, int nseg, uint32_t cap, uint32_t tile,
                                                    uint32_t b, GhVarSeg& o) {
  uint32_t acc = 0, a = vstart[0] < cap ? vstart[0] : cap;
  for (int v = 0; v < nseg; ++v) {
    uint32_t e = vstart[v + 1]; e = e < cap ? e : cap; e = e < a ? a : e;
    const uint32_t nb = (e - a + tile - 1u) / tile;
    if (b < acc + nb) { o.seg = (uint32_t)v; o.lb = b - acc; o.n = e - a; o.s0 = a; o.gb0 = acc; o.nb = nb; return true; }
    acc += nb; a = e;
  }
  return false;
}
__device__ __forceinline__ void gh_var_seg_of_index(const uint32_t* __restrict__ vstart, int nseg, uint32_t cap, uint32_t tile,
                                                    uint32_t seg, GhVarSeg& o) {
  uint32_t acc = 0, a = vstart[0] < cap ? vstart[0] : cap;
  for (int v = 0; v <= (int)seg; ++v) {
    uint32_t e = vstart[v + 1]; e = e < cap ? e : cap; e = e < a ? a : e;
    const uint32_t nb = (e - a + tile - 1u) / tile;
    if (v == (int)seg) { o.seg = seg; o.lb = 0; o.n = e - a; o.s0 = a; o.gb0 = acc; o.nb = nb; return; }
    acc += nb; a = e;
  }
}

// Bits in which at least two (relevant) keys differ, from the per-producer-block (OR, AND) of the key bits
// (gh_preprocess_fwd_kernel; key_bits[n_bits] receives the word). Computed ONCE, by block (0, 0) of the first pass's histogram
// kernel; every later pass reads the word: a pass whose digit has no varying bit is the identity (its histogram kernel exits,
// its scatter kernel copies). Keys that were left out of the (OR, AND) — Gaussians that emit no instance — may land anywhere.
__device__ __forceinline__ void gh_store_varying_bits(uint2* __restrict__ key_bits, int n_bits, uint32_t* __restrict__ wide_flag, bool nbits24) {
  __shared__ uint32_t s_or[GH_BLOCK / GH_WAVE], s_and[GH_BLOCK / GH_WAVE];
  uint32_t o = 0u, a = 0xFFFFFFFFu;
  for (int i = threadIdx.x; i < n_bits; i += GH_BLOCK) { const uint2 b = key_bits[i]; o |= b.x; a &= b.y; }
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) { o |= __shfl_xor(o, k); a &= __shfl_xor(a, k); }
  if ((threadIdx.x & 63) == 0) { s_or[threadIdx.x >> 6] = o; s_and[threadIdx.x >> 6] = a; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t varying = (s_or[0] | s_or[1] | s_or[2] | s_or[3]) & ~(s_and[0] & s_and[1] & s_and[2] & s_and[3]);
    key_bits[n_bits].x = varying;
    // The top byte of the keys: when it does not vary the caller learns that GH_FLAG_DEPTH24 holds for this call (information bit);
    // when it does and this sort has no pass for it (nbits24), the call is invalid (one thread per call: no contention)
    if (wide_flag) {
      if ((varying >> 24) == 0u) atomicOr(wide_flag, GH_COUNTER_DEPTH24_OK);
      else if (nbits24) atomicOr(wide_flag, 8u);
    }
  }
  __syncthreads();
}
__device__ __forceinline__ bool gh_digit_varies(const uint2* __restrict__ key_bits, int n_bits, int shift, uint32_t dmask) {
  return !key_bits || shift == 0 || ((key_bits[n_bits].x >> shift) & dmask) != 0u;      // shift 0 is the producing pass
}

// Pass part 1: per-block digit histogram -> table[digit][block]; SELF: -> table[block][digit] (the scatter kernel sums the
// rows of the blocks before it itself, see there).
template <int ITEMS, int MAXD, bool SELF>
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr,
                                                                  uint32_t cap, uint32_t seg_len, int shift, uint32_t dmask,
                                                                  uint32_t* __restrict__ table, const uint2* __restrict__ key_bits,
                                                                  int n_bits, uint32_t* __restrict__ wide_flag, int total_bits,
                                                                  const uint32_t* __restrict__ vstart, int nseg) {
  __shared__ uint32_t s_hist[MAXD];
  if (key_bits && shift == 0 && blockIdx.x == 0 && blockIdx.y == 0) gh_store_varying_bits((uint2*)key_bits, n_bits, wide_flag, total_bits <= 24);
  if (!gh_digit_varies(key_bits, n_bits, shift, dmask)) return;       // the scatter of this pass is a plain copy
  uint32_t n = gh_seg_count(n_ptr, cap, seg_len);
  uint32_t seg = blockIdx.y;
  const uint32_t nblk = gridDim.x, ndig = dmask + 1u;
  uint32_t base = blockIdx.x * (uint32_t)(GH_BLOCK * ITEMS);
  if (vstart) {                                          // variable segments: this block's segment; tables by global block number
    GhVarSeg vs;
    if (!gh_var_seg_of_block(vstart, nseg, cap, (uint32_t)(GH_BLOCK * ITEMS), blockIdx.x, vs)) return;
    n = vs.n; base = vs.lb * (uint32_t)(GH_BLOCK * ITEMS); keys += vs.s0; seg = 0u;
  }
  if (base >= n) return;
  keys += (size_t)seg * seg_len;
  uint32_t k[ITEMS];                                       // loads in flight while the counters are cleared
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const uint32_t idx = base + j * GH_BLOCK + threadIdx.x;
    k[j] = idx < n ? keys[idx] : 0u;
  }
  for (uint32_t d = threadIdx.x; d < ndig; d += GH_BLOCK) s_hist[d] = 0;
  __syncthreads();
  // One LDS atomic per RUN of equal digits in lane order, not per key: after an earlier pass (or with keys that arrive grouped,
  // like the tile partition's high digit: view-major instances) neighbouring lanes carry the same digit and 64 same-address LDS
  // atomics serialise (the second tile pass's histogram took 12.3 us where the first took 5.6).
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const uint32_t idx = base + j * GH_BLOCK + threadIdx.x;
    const uint32_t dg = idx < n ? ((k[j] >> shift) & dmask) : 0xFFFFFFFFu;       // the tail's lanes form a run of their own
    const uint32_t prev = (uint32_t)__shfl_up((int)dg, 1);
    const bool head = lane == 0 || dg != prev;
    const uint64_t after = (gh_ballot(head) >> lane) >> 1;                       // run heads behind this lane
    if (head && idx < n) atomicAdd(&s_hist[dg], after ? (uint32_t)__builtin_ctzll(after) + 1u : (uint32_t)(64 - lane));
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < ndig; d += GH_BLOCK) {
    if (SELF) table[((size_t)seg * nblk + blockIdx.x) * ndig + d] = s_hist[d];
    else table[((size_t)seg * ndig + d) * nblk + blockIdx.x] = s_hist[d];
  }
}

// Pass part 2: one block per (digit, segment): exclusive scan of its row over the active blocks, row total -> tot.
// Four consecutive row entries per thread: 1024 blocks per sweep (one sweep for the depth level, two for 3 M instances).
template <int ITEMS>
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_scan_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ tot,
                                                                  const uint32_t* __restrict__ n_ptr, uint32_t cap, uint32_t seg_len,
                                                                  int nblk_cap, const uint32_t* __restrict__ vstart, int nseg) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_carry;
  const uint32_t n = gh_seg_count(n_ptr, cap, seg_len);
  int nblk = (int)((n + (GH_BLOCK * ITEMS) - 1) / (GH_BLOCK * ITEMS));
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const size_t rowi = (size_t)blockIdx.y * gridDim.x + blockIdx.x;       // seg * ndig + digit
  uint32_t* row = table + rowi * nblk_cap;
  if (vstart) {                                          // variable segments: the digit's row over the segment's own blocks
    GhVarSeg vs;
    gh_var_seg_of_index(vstart, nseg, cap, (uint32_t)(GH_BLOCK * ITEMS), blockIdx.y, vs);
    row = table + (size_t)blockIdx.x * nblk_cap + vs.gb0;
    nblk = (int)vs.nb;
  }
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 4 * GH_BLOCK) {
    const int i0 = base + tid * 4;
    uint32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + j < nblk ? row[i0 + j] : 0u;
    const uint32_t mine = (v[0] + v[1]) + (v[2] + v[3]);
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    const uint32_t carry = s_carry;
    uint32_t run = carry + woff + x - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j) { if (i0 + j < nblk) row[i0 + j] = run; run += v[j]; }
    __syncthreads();
    if (tid == GH_BLOCK - 1) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) tot[rowi] = s_carry;
}

// Pass part 3: stable scatter. Ranking is per wave with ballot matching (one ballot per digit bit), waves are
// ordered through an LDS prefix over their digit counts, so equal digits keep their input order. The tile is first
// sorted into LDS and then written out, so that each digit run leaves as contiguous global segments.
// Thread t owns the DPT = MAXD / GH_BLOCK consecutive digits [t*DPT, (t+1)*DPT) in the digit-indexed phases.
// SELF (segments of at most 128 blocks, e.g. the per-view depth sort): there is no scan kernel; the histogram table is
// block-major and every block adds up the rows of the blocks before it (its prefix) and of all blocks (the digit totals)
// with coalesced reads — one launch less per pass where the row scan was nothing but launch latency.
#ifndef GH_SELF_LOADS
#define GH_SELF_LOADS 13u
#endif
template <int ITEMS, int MAXD, bool SELF>
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t cap, uint32_t seg_len, int shift, uint32_t dmask,
    int nbit, const uint32_t* __restrict__ table, const uint32_t* __restrict__ tot, const uint2* __restrict__ key_bits, int n_bits,
    const uint32_t* __restrict__ vstart, int nseg) {
  constexpr int DPT = MAXD / GH_BLOCK;
  constexpr int NW = GH_BLOCK / GH_WAVE;
  __shared__ uint32_t s_base[MAXD];                        // global base of (digit, this block)
  __shared__ uint32_t s_cnt[NW][MAXD];                     // per-wave digit counters -> per-wave bases
  __shared__ uint32_t s_w[NW];
  __shared__ uint32_t s_lbase[MAXD];                       // first position of each digit in the locally sorted tile
  __shared__ uint32_t s_key[(GH_BLOCK * ITEMS)], s_val[(GH_BLOCK * ITEMS)];
  uint32_t n = gh_seg_count(n_ptr, cap, seg_len);
  uint32_t seg = blockIdx.y, tseg = blockIdx.y;            // segment of the per-digit totals / of the table rows
  const uint32_t nblk = gridDim.x, ndig = dmask + 1u;
  uint32_t blk_base = blockIdx.x * (uint32_t)(GH_BLOCK * ITEMS);
  size_t seg_off = (size_t)seg * seg_len;
  if (!SELF && vstart) {                                 // variable segments (see gh_var_seg_of_block)
    GhVarSeg vs;
    if (!gh_var_seg_of_block(vstart, nseg, cap, (uint32_t)(GH_BLOCK * ITEMS), blockIdx.x, vs)) return;
    n = vs.n; blk_base = vs.lb * (uint32_t)(GH_BLOCK * ITEMS); seg_off = vs.s0; seg = vs.seg; tseg = 0u;
  }
  const bool varies = gh_digit_varies(key_bits, n_bits, shift, dmask);
  if (blk_base >= n) return;
  keys_in += seg_off; vals_in += seg_off; keys_out += seg_off; vals_out += seg_off;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (!varies) {                                   // every key carries the same digit: the stable scatter is the identity
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const uint32_t idx = blk_base + r * GH_BLOCK + tid;
      if (idx < n) { keys_out[idx] = keys_in[idx]; vals_out[idx] = vals_in[idx]; }
    }
    return;
  }

  // the block's keys: loads issued first, they do not depend on the digit bases computed below.
  // Wave w owns keys [w*ITEMS*64, (w+1)*ITEMS*64) of the block's tile, visited as ITEMS rounds of 64 consecutive keys,
  // so (round, lane) order == memory order.
  uint32_t key[ITEMS], val[ITEMS];
  const uint32_t wave_base = blk_base + wid * (ITEMS * GH_WAVE);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t idx = wave_base + r * GH_WAVE + lane;
    key[r] = idx < n ? keys_in[idx] : ~0u;
    val[r] = idx < n ? vals_in[idx] : 0u;
  }

  // digit base = exclusive scan over digits of tot[] + this block's row prefix
  if (SELF) {
    // The segment's histogram rows [block][digit] are summed by the whole workgroup: every wave-load takes 256 consecutive
    // words (16 bytes per lane) of the table, so a lane always meets the same four digits, a few loads cover all the rows
    // (one per thread and row took ~100 dependent-latency-bound 4-byte loads), and the partial sums meet in LDS.
    static_assert(!SELF || DPT == 1, "SELF: one digit per thread");
    const uint32_t nact = (n + (uint32_t)(GH_BLOCK * ITEMS) - 1u) / (uint32_t)(GH_BLOCK * ITEMS);   // blocks that wrote a row
    const uint32_t* seg_tab = table + (size_t)seg * nblk * ndig;
    uint32_t* s_tall = s_key;                              // (the key / value staging area is not in use yet)
    uint32_t* s_tpre = s_key + MAXD;
    s_tall[tid] = 0u; s_tpre[tid] = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) s_cnt[w][tid] = 0;
    __syncthreads();
    if (ndig >= 4u) {
      const uint32_t nq = nact * ndig / 4u;                // 16-byte words in the rows
      const uint32_t lg = 31u - (uint32_t)__clz((int)ndig);
      uint32_t t_all[4] = {0u, 0u, 0u, 0u}, t_pre[4] = {0u, 0u, 0u, 0u};
      for (uint32_t q0 = (uint32_t)wid * GH_WAVE + lane; q0 < nq; q0 += GH_SELF_LOADS * GH_BLOCK) {
        uint4 c[GH_SELF_LOADS];
#pragma unroll
        for (uint32_t j = 0; j < GH_SELF_LOADS; ++j) {
          const uint32_t q = q0 + j * GH_BLOCK;
          c[j] = q < nq ? ((const uint4*)seg_tab)[q] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (uint32_t j = 0; j < GH_SELF_LOADS; ++j) {
          const bool pre = (((q0 + j * GH_BLOCK) * 4u) >> lg) < blockIdx.x;      // the row this word belongs to
          t_all[0] += c[j].x; t_all[1] += c[j].y; t_all[2] += c[j].z; t_all[3] += c[j].w;
          t_pre[0] += pre ? c[j].x : 0u; t_pre[1] += pre ? c[j].y : 0u; t_pre[2] += pre ? c[j].z : 0u; t_pre[3] += pre ? c[j].w : 0u;
        }
      }
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t d = ((uint32_t)lane * 4u + j) & dmask;
        atomicAdd(&s_tall[d], t_all[j]);
        atomicAdd(&s_tpre[d], t_pre[j]);
      }
    } else {
      const uint32_t d = (uint32_t)tid;
      if (d < ndig) {
        uint32_t ta = 0, tp = 0;
        for (uint32_t b = 0; b < nact; ++b) { const uint32_t c = seg_tab[(size_t)b * ndig + d]; ta += c; tp += b < blockIdx.x ? c : 0u; }
        s_tall[d] = ta; s_tpre[d] = tp;
      }
    }
    __syncthreads();
    const uint32_t v = (uint32_t)tid < ndig ? s_tall[tid] : 0u, pre = (uint32_t)tid < ndig ? s_tpre[tid] : 0u;
    uint32_t total;
    const uint32_t run = gh_block_excl_scan(v, s_w, &total);
    s_base[tid] = run + pre;
  } else {
    uint32_t v[DPT], pre[DPT], sum = 0;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const uint32_t d = (uint32_t)(tid * DPT + k);
      v[k] = d < ndig ? tot[(size_t)seg * ndig + d] : 0u;
      pre[k] = d < ndig ? table[((size_t)tseg * ndig + d) * nblk + blockIdx.x] : 0u;
      sum += v[k];
#pragma unroll
      for (int w = 0; w < NW; ++w) s_cnt[w][d] = 0;
    }
    uint32_t total;
    uint32_t run = gh_block_excl_scan(sum, s_w, &total);
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const uint32_t d = (uint32_t)(tid * DPT + k);
      s_base[d] = run + pre[k];
      run += v[k];
    }
  }
  __syncthreads();

  // Phase A: rank keys inside the wave.
  uint32_t rank[ITEMS];
  volatile uint32_t* cnt = s_cnt[wid];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t idx = wave_base + r * GH_WAVE + lane;
    const bool valid = idx < n;
    const uint32_t dg = (key[r] >> shift) & dmask;
    uint64_t peers = gh_ballot(valid);
    for (int b = 0; b < nbit; ++b) {              // wave-uniform: one ballot per bit of this pass's digit
      const bool bit = (dg >> b) & 1u;
      const uint64_t m = gh_ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    uint32_t prev = 0;
    if (valid) prev = cnt[dg];                // every peer reads the same counter (LDS broadcast)
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) cnt[dg] = prev + (uint32_t)__popcll(peers);   // one writer per digit
    __builtin_amdgcn_wave_barrier();
    rank[r] = prev + before;
  }
  __syncthreads();
  // Phase B: per-wave bases inside the block's LOCALLY sorted tile: block count per digit, exclusive scan over the
  // digits, waves in order.
  {
    uint32_t c[DPT][NW], tot_d[DPT], sum = 0;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const uint32_t d = (uint32_t)(tid * DPT + k);
      tot_d[k] = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { c[k][w] = s_cnt[w][d]; tot_d[k] += c[k][w]; }
      sum += tot_d[k];
    }
    uint32_t total;
    uint32_t run = gh_block_excl_scan(sum, s_w, &total);
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
      const uint32_t d = (uint32_t)(tid * DPT + k);
      s_lbase[d] = run;
#pragma unroll
      for (int w = 0; w < NW; ++w) { s_cnt[w][d] = run; run += c[k][w]; }
    }
  }
  __syncthreads();
  // Phase C: stage the tile in LDS in sorted order ...
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t idx = wave_base + r * GH_WAVE + lane;
    if (idx < n) {
      const uint32_t dg = (key[r] >> shift) & dmask;
      const uint32_t lpos = s_cnt[wid][dg] + rank[r];
      s_key[lpos] = key[r];
      s_val[lpos] = val[r];
    }
  }
  __syncthreads();
  // ... and write it out: consecutive threads hold consecutive elements of a digit run -> contiguous global segments
  const uint32_t nvalid = n - blk_base < (uint32_t)(GH_BLOCK * ITEMS) ? n - blk_base : (uint32_t)(GH_BLOCK * ITEMS);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t e = r * GH_BLOCK + tid;
    if (e < nvalid) {
      const uint32_t k = s_key[e];
      const uint32_t dg = (k >> shift) & dmask;
      const uint32_t dst = s_base[dg] + (e - s_lbase[dg]);
      keys_out[dst] = k;
      vals_out[dst] = s_val[e];
    }
  }
}

// Sorts (keys, vals) on bits [0, nbits); in/out ping-pong (pointers are swapped so that on return k_in / v_in hold the
// result). ITEMS keys per thread: 16 for large inputs (bandwidth), 8 / 4 for smaller ones (more, shorter blocks: the pass is
// latency-bound when it cannot fill the 256 CUs). seg_len / segs: see the top of this section (0 / 1 = one segment with a
// device-side count).

template <int ITEMS>
static void gh_radix_sort_t(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr,
                            uint32_t cap, int nbits, uint32_t seg_len, int segs, uint32_t* table, hipStream_t s,
                            const uint2* key_bits, int n_bits, uint32_t* wide_flag) {
  const size_t per_seg = seg_len ? seg_len : cap;
  const int nblk = (int)((per_seg + GH_BLOCK * ITEMS - 1) / (GH_BLOCK * ITEMS));
  if (nblk == 0 || segs == 0) return;
  // Small single-segment sorts of 9 or 10 key bits (the tile partition of ONE 512x334 view: 672 tiles) run as ONE pass with
  // 1024 digits: a small launch is bound by the latency of its kernels, not by bandwidth, and three launches beat six
  // (at 8 views the wide pass cost what the two narrow ones cost — DESIGN.md, round 1).
  if (ITEMS == 4 && segs == 1 && nbits > 8 && nbits <= 10 && !key_bits) {
    const uint32_t dmask = (1u << nbits) - 1u, ndig = dmask + 1u;
    uint32_t* tot = table + (size_t)ndig * nblk;
    const dim3 gb(nblk, 1), gs(ndig, 1), blk(GH_BLOCK);
    hipLaunchKernelGGL((gh_radix_hist_kernel<4, 1024, false>), gb, blk, 0, s, k_in, n_ptr, cap, seg_len, 0, dmask, table, key_bits, n_bits, wide_flag, nbits, nullptr, 0);
    hipLaunchKernelGGL(gh_radix_scan_kernel<4>, gs, blk, 0, s, table, tot, n_ptr, cap, seg_len, nblk, nullptr, 0);
    hipLaunchKernelGGL((gh_radix_scatter_kernel<4, 1024, false>), gb, blk, 0, s, k_in, v_in, k_out, v_out, n_ptr, cap, seg_len,
                       0, dmask, nbits, table, tot, key_bits, n_bits, nullptr, 0);
    uint32_t* t = k_in; k_in = k_out; k_out = t;
    t = v_in; v_in = v_out; v_out = t;
    return;
  }
  const int passes = (nbits + 7) / 8;
  for (int p = 0; p < passes; ++p) {
    // spread the bits evenly over the passes (e.g. 13 bits -> 6 + 7)
    const int lo = (nbits * p) / passes, hi = (nbits * (p + 1)) / passes;
    const uint32_t dmask = (1u << (hi - lo)) - 1u;
    const uint32_t ndig = dmask + 1u;
    uint32_t* tot = table + (size_t)segs * ndig * nblk;
    const dim3 gb(nblk, segs), gs(ndig, segs), blk(GH_BLOCK);
    if (nblk <= 128) {                                   // short segments: no scan kernel (see gh_radix_scatter_kernel)
      hipLaunchKernelGGL((gh_radix_hist_kernel<ITEMS, 256, true>), gb, blk, 0, s, k_in, n_ptr, cap, seg_len, lo, dmask, table, key_bits, n_bits, wide_flag, nbits, nullptr, 0);
      hipLaunchKernelGGL((gh_radix_scatter_kernel<ITEMS, 256, true>), gb, blk, 0, s, k_in, v_in, k_out, v_out, n_ptr, cap, seg_len,
                         lo, dmask, hi - lo, table, tot, key_bits, n_bits, nullptr, 0);
    } else {
      hipLaunchKernelGGL((gh_radix_hist_kernel<ITEMS, 256, false>), gb, blk, 0, s, k_in, n_ptr, cap, seg_len, lo, dmask, table, key_bits, n_bits, wide_flag, nbits, nullptr, 0);
      hipLaunchKernelGGL(gh_radix_scan_kernel<ITEMS>, gs, blk, 0, s, table, tot, n_ptr, cap, seg_len, nblk, nullptr, 0);
      hipLaunchKernelGGL((gh_radix_scatter_kernel<ITEMS, 256, false>), gb, blk, 0, s, k_in, v_in, k_out, v_out, n_ptr, cap, seg_len,
                         lo, dmask, hi - lo, table, tot, key_bits, n_bits, nullptr, 0);
    }
    uint32_t* t = k_in; k_in = k_out; k_out = t;
    t = v_in; v_in = v_out; v_out = t;
  }
}

// keys per thread: measured on the 8-view workload (98 k keys per segment / 3 M instances): 2 / 4 / 8 for the small class
// gave 0.296 / 0.287 / 0.294 ms of binning, 4 / 8 / 16 for the middle class 0.298 / 0.287 / 0.294. Fixed-length segment sorts
// (the per-view depth sort; every block sums the histogram rows of its segment, so fewer, larger blocks read less): 8 from
// half a million keys in all (8 views x 98 k: scatter pass 11.5 -> 10.3 us; 2 views: 7.4 -> 8.2 us, so 4 below that).
static int gh_radix_items(size_t per_segment, int segs, bool seg_sort) {
  if (seg_sort && per_segment <= ((size_t)1 << 21)) return per_segment * (size_t)segs >= ((size_t)1 << 19) ? 8 : 4;
  return per_segment <= ((size_t)1 << 21) ? 4 : (per_segment <= ((size_t)1 << 25) ? 8 : 16);
}

// Table words for sorting `segs` segments of `per_segment` elements (capacity).
size_t gh_radix_table_words(size_t per_segment, int segs) {
  // the smallest tile this segment length may be sorted with (keys per thread depend on the number of segments, and a call
  // split into two halves shares one table): linear in `segs`
  const size_t tile = (size_t)GH_BLOCK * (per_segment <= ((size_t)1 << 21) ? 4 : gh_radix_items(per_segment, segs, true));
  return (size_t)segs * 1024 * ((per_segment + tile - 1) / tile) + (size_t)segs * 1024;      // up to 1024 digits per pass
}

size_t gh_radix_table_words(size_t cap) {
  // (monotone in cap: the keys per thread double at 2^21 and 2^25 elements, which would halve the table just above a threshold —
  //  a capacity a little larger must never ask for a smaller workspace)
  size_t words = 0;
  const size_t steps[3] = {cap < ((size_t)1 << 21) ? cap : ((size_t)1 << 21), cap < ((size_t)1 << 25) ? cap : ((size_t)1 << 25), cap};
  for (int k = 0; k < 3; ++k) {
    const size_t tile = (size_t)GH_BLOCK * gh_radix_items(steps[k], 1, false);
    const size_t w = (size_t)1024 * ((steps[k] + tile - 1) / tile) + 1024;
    if (w > words) words = w;
  }
  return words;
}

// Passes gh_radix_sort runs for `nbits` key bits on one segment of capacity `cap` (callers pick the start buffer by its parity).
int gh_radix_passes(size_t cap, int nbits) {
  if (gh_radix_items(cap, 1, false) == 4 && nbits > 8 && nbits <= 10) return 1;
  return (nbits + 7) / 8;
}

void gh_radix_sort_ex(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr, uint32_t cap,
                      int nbits, uint32_t seg_len, int segs, uint32_t* table, hipStream_t s, const uint2* key_bits, int n_bits,
                      uint32_t* wide_flag) {
  const size_t per_seg = seg_len ? seg_len : cap;
  const int items = gh_radix_items(per_seg, segs, seg_len != 0u);
  if (items == 4) gh_radix_sort_t<4>(k_in, v_in, k_out, v_out, n_ptr, cap, nbits, seg_len, segs, table, s, key_bits, n_bits, wide_flag);
  else if (items == 8) gh_radix_sort_t<8>(k_in, v_in, k_out, v_out, n_ptr, cap, nbits, seg_len, segs, table, s, key_bits, n_bits, wide_flag);
  else gh_radix_sort_t<16>(k_in, v_in, k_out, v_out, n_ptr, cap, nbits, seg_len, segs, table, s, key_bits, n_bits, wide_flag);
}

// Variable-length segments (gh_var_seg_of_block): nseg segments [vstart[v], vstart[v+1]) of one array of capacity cap, each sorted
// stably on bits [0, nbits) among its own elements, in place in its range: ceil(nbits / 8) passes of (histogram, row scan, scatter).
template <int ITEMS>
static void gh_radix_sort_var_t(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* vstart, int nseg,
                                uint32_t cap, int nbits, uint32_t* table, hipStream_t s) {
  const int nblk = (int)(((size_t)cap + GH_BLOCK * ITEMS - 1) / (GH_BLOCK * ITEMS)) + nseg;
  // (ONE pass of 1024 digits per view instead of two of 32 — measured at 8 views: histogram 18.2 + scan 13.6 + scatter 31.0 = 63 us against
  //  51 for the two narrow passes, as round 1 found for the single-segment form: the 6 MB digit table costs more than the second pass)
  const int passes = (nbits + 7) / 8;
  for (int p = 0; p < passes; ++p) {
    const int lo = (nbits * p) / passes, hi = (nbits * (p + 1)) / passes;
    const uint32_t dmask = (1u << (hi - lo)) - 1u, ndig = dmask + 1u;
    uint32_t* tot = table + (size_t)ndig * nblk;
    const dim3 gb(nblk, 1), gs(ndig, nseg), blk(GH_BLOCK);
    hipLaunchKernelGGL((gh_radix_hist_kernel<ITEMS, 256, false>), gb, blk, 0, s, k_in, vstart, cap, 0u, lo, dmask, table, (const uint2*)nullptr, 0,
                       (uint32_t*)nullptr, nbits, vstart, nseg);
    hipLaunchKernelGGL(gh_radix_scan_kernel<ITEMS>, gs, blk, 0, s, table, tot, vstart, cap, 0u, nblk, vstart, nseg);
    hipLaunchKernelGGL((gh_radix_scatter_kernel<ITEMS, 256, false>), gb, blk, 0, s, k_in, v_in, k_out, v_out, vstart, cap, 0u, lo, dmask, hi - lo,
                       table, tot, (const uint2*)nullptr, 0, vstart, nseg);
    uint32_t* t = k_in; k_in = k_out; k_out = t;
    t = v_in; v_in = v_out; v_out = t;
  }
}

void gh_radix_sort_var(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* vstart, int nseg,
                       uint32_t cap, int nbits, uint32_t* table, hipStream_t s) {
  const int items = gh_radix_items((size_t)cap, 1, false);
  if (items == 4) gh_radix_sort_var_t<4>(k_in, v_in, k_out, v_out, vstart, nseg, cap, nbits, table, s);
  else if (items == 8) gh_radix_sort_var_t<8>(k_in, v_in, k_out, v_out, vstart, nseg, cap, nbits, table, s);
  else gh_radix_sort_var_t<16>(k_in, v_in, k_out, v_out, vstart, nseg, cap, nbits, table, s);
}

void gh_radix_sort(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr, uint32_t cap,
                   int nbits, uint32_t* table, hipStream_t s) {
  gh_radix_sort_ex(k_in, v_in, k_out, v_out, n_ptr, cap, nbits, 0u, 1, table, s, nullptr, 0, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Level 2: walk the Gaussians in depth order.
// Per-block sums of tiles-touched in depth-sorted order (the emit kernel's slot scan).
__global__ __launch_bounds__(GH_BLOCK) void gh_count_sorted_kernel(int N, const uint32_t* __restrict__ perm,
                                                                    const uint32_t* __restrict__ tiles_touched,
                                                                    uint32_t* __restrict__ block_sums) {
  __shared__ unsigned s_wsum[GH_BLOCK / GH_WAVE];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * GH_BLOCK + tid;
  const unsigned c = i < N ? tiles_touched[perm[i]] : 0u;
  const unsigned ws = gh_wave_sum_u32(c);
  if ((tid & 63) == 0) s_wsum[tid >> 6] = ws;
  __syncthreads();
  if (tid == 0) block_sums[blockIdx.x] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
}

// The RECORD slots (where the backward puts its sub-records): thread i here is thread i of the projection kernel (same block
// size), whose blocks left their instance counts in block_tiles; every block adds up the counts of the blocks before it (coalesced
// L2 reads) and scans its own 256 counts, so slot_begin[n] = number of instances of all (view, Gaussian) pairs in front of n in the
// projection kernel's thread order: the chain-rule kernel walks the pairs in that order and reads one contiguous stretch per wave.
// Runs as SPARE workgroups of gh_emit_kernel (round 6): nothing on the way to the emit reads the slots — their first reader is
// gh_ranges_kernel, four launches on — so the numbering costs gh_count_sorted_kernel's critical path nothing (10.6 -> 6.x us) and
// rides in the emit kernel's latency gaps. (Folding the scan into the projection kernel instead cost that kernel 3 us and
// gh_ranges_kernel 1.4-3.3 us for the extra gather: measured, App. R6.)
__device__ __forceinline__ void gh_number_record_slots(uint32_t blk, int N, int P, int NV, float rdiv,
                                                       const uint32_t* __restrict__ tiles_touched,
                                                       const uint32_t* __restrict__ block_tiles, uint32_t* __restrict__ slot_begin) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE], s_p[GH_BLOCK / GH_WAVE];
  const int tid = threadIdx.x;
  const int i = (int)blk * GH_BLOCK + tid;
  // (view, Gaussian) of projection thread i: Gaussian-major, views adjacent (NV == 0: row-major, the pose batch)
  uint32_t n = 0u;
  if (i < N) {
    if (NV == 0) n = (uint32_t)i;
    else { const uint32_t r = rdiv > 0.0f ? gh_div_small((uint32_t)i, (uint32_t)NV, rdiv) : (uint32_t)i / (uint32_t)NV; n = ((uint32_t)i - r * (uint32_t)NV) * (uint32_t)P + r; }
  }
  const uint32_t ct = i < N ? tiles_touched[n] : 0u;
  uint32_t part = 0;
  for (uint32_t b = tid; b < blk; b += 4 * GH_BLOCK) {
    const uint32_t b1 = b + GH_BLOCK, b2 = b + 2 * GH_BLOCK, b3 = b + 3 * GH_BLOCK;    // four loads in flight per trip
    const uint32_t v0 = block_tiles[b], v1 = b1 < blk ? block_tiles[b1] : 0u, v2 = b2 < blk ? block_tiles[b2] : 0u,
                   v3 = b3 < blk ? block_tiles[b3] : 0u;
    part += (v0 + v1) + (v2 + v3);
  }
  part = gh_wave_sum_u32(part);
  if ((tid & 63) == 0) s_p[tid >> 6] = part;
  uint32_t total;
  const uint32_t excl = gh_block_excl_scan(ct, s_w, &total);        // (two barriers: s_p is visible behind them)
  if (i < N) slot_begin[n] = ((s_p[0] + s_p[1]) + (s_p[2] + s_p[3])) + excl;
}

// One thread per (view, Gaussian) IN DEPTH ORDER: block-local scan -> first emit slot of the Gaussian, then one
// instance per tile of its rect that passes the exact ellipse/tile test (the same test that counted them in the
// projection kernel), row-major: key = global tile id, payload = view*P+gaussian.
// A Gaussian's instances occupy consecutive emit slots (the stable partition below reads them in this order); the backward's
// sub-records live at the RECORD slots gh_count_sorted_kernel numbered, not here.
// The instances are written by the WAVE, not by their Gaussian's lane: the 64 Gaussians of a wave own one contiguous run of
// slots; lane l of trip j takes slot 64 j + l of the run, finds its Gaussian (binary search over the wave's prefix sums in
// LDS) and the tile (the k-th set bit of the Gaussian's hit mask) — coalesced stores and ceil(run / 64) trips, where one
// lane per Gaussian ran max(tiles) trips with the other lanes idle. Rects larger than the 64-bit hit mask (rare, huge
// footprints) are still walked by their own lane.
__device__ __forceinline__ uint32_t gh_kth_set_bit(uint32_t lo, uint32_t hi, uint32_t k) {
  uint32_t c = (uint32_t)__popc(lo), w = lo, pos = 0;
  if (k >= c) { k -= c; w = hi; pos = 32; }
  c = (uint32_t)__popc(w & 0xFFFFu); if (k >= c) { k -= c; w >>= 16; pos += 16; }
  c = (uint32_t)__popc(w & 0xFFu);   if (k >= c) { k -= c; w >>= 8;  pos += 8; }
  c = (uint32_t)__popc(w & 0xFu);    if (k >= c) { k -= c; w >>= 4;  pos += 4; }
  c = (uint32_t)__popc(w & 0x3u);    if (k >= c) { k -= c; w >>= 2;  pos += 2; }
  return pos + ((k >= (w & 1u)) ? 1u : 0u);
}

#define GH_EMIT_MARKS 128                                // 32-bit words of run-end marks per wave: runs of up to 4,096 slots
__global__ __launch_bounds__(GH_BLOCK) void gh_emit_kernel(
    int N, int P, int gx, int tiles, uint32_t cap, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ tiles_touched,
    const uint32_t* __restrict__ block_sums, float4* __restrict__ geom,
    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, GhCounters* __restrict__ ctr, float rP, uint32_t flags,
    const float* __restrict__ tile_depth_bound, int n_emit_blocks, int NVs, float rdiv, const uint32_t* __restrict__ block_tiles,
    uint32_t* __restrict__ slot_begin, uint32_t* __restrict__ view_start, int n_views) {
  GH_WG_TIMER(4);
  constexpr int NW = GH_BLOCK / GH_WAVE;
  if ((int)blockIdx.x >= n_emit_blocks) {               // the spare workgroups: the record-slot numbering (see gh_number_record_slots)
    gh_number_record_slots(blockIdx.x - (uint32_t)n_emit_blocks, N, P, NVs, rdiv, tiles_touched, block_tiles, slot_begin);
    return;
  }
  __shared__ uint32_t s_w[NW], s_p[NW];
  __shared__ uint32_t s_end[NW][GH_WAVE];               // per wave: inclusive prefix of the lanes' instance counts
  __shared__ uint4 s_g[NW][GH_WAVE];                    // (rect, hit mask lo, hi, n); rect = 0: not written by the wave
  __shared__ uint32_t s_vb[NW][GH_WAVE];                // first global tile id of the Gaussian's view
  __shared__ __attribute__((aligned(8))) uint32_t s_marks[NW][GH_EMIT_MARKS];   // per wave: bit p = a lane's run ends at slot p of the wave's run
  __shared__ uint2 s_own[NW][GH_WAVE];                  // by rank among the lanes that own slots: (first slot of the run, view's first tile id)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x * GH_BLOCK + tid;
  const uint32_t n = i < N ? perm[i] : 0u;
  float4* grec = geom + (size_t)n * 4;
  const float4 g3 = grec[3];                            // (instance count, view-space depth, ..)
  const uint32_t cnt = i < N ? __float_as_uint(g3.x) : 0u;   // instance count, rect + tile hit mask: one 64-byte line
  const float4 g2 = grec[2];
  // first emit slot of this block = sum of the per-block instance counts of all blocks before it (gh_count_sorted_kernel):
  // every block adds them up itself (a few thousand coalesced L2 reads) instead of waiting for a one-block scan kernel
  uint32_t part = 0;
  const uint32_t blk = blockIdx.x;
  for (uint32_t b = tid; b < blk; b += 4 * GH_BLOCK) {
    const uint32_t b1 = b + GH_BLOCK, b2 = b + 2 * GH_BLOCK, b3 = b + 3 * GH_BLOCK;    // four loads in flight per trip
    const uint32_t v0 = block_sums[b], v1 = b1 < blk ? block_sums[b1] : 0u, v2 = b2 < blk ? block_sums[b2] : 0u,
                   v3 = b3 < blk ? block_sums[b3] : 0u;
    part += (v0 + v1) + (v2 + v3);
  }
  part = gh_wave_sum_u32(part);
  uint32_t x = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
  const uint32_t r = __float_as_uint(g2.y);
  const int minx = r & 255, miny = (r >> 8) & 255, maxx = (r >> 16) & 255, maxy = r >> 24;
  const bool small = (maxx - minx) * (maxy - miny) <= 64;               // the projection kernel kept the hit mask
  const uint32_t vtile0 = (rP > 0.0f ? gh_div_small(n, (uint32_t)P, rP) : n / (uint32_t)P) * (uint32_t)tiles;   // first global tile id of the view
  // keys: global tile ids, or (view_start given: the per-view partition) tile ids inside the view — the views are segments of the
  // emit order already, so the partition only has to order each view's instances by their ceil(log2 tiles) local bits
  const uint32_t vbase = view_start ? 0u : vtile0;
  const uint32_t wave_total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
  const bool marks = wave_total <= (uint32_t)(GH_EMIT_MARKS * 32);         // wave-uniform: which owner search the trips below use
  if (marks) {
    // Owner of a slot = the number of runs that END in front of it. The lanes that own slots mark the last slot of their run in a
    // bit array over the wave's run (one LDS atomic each) and leave their record at their RANK among those lanes; a trip then reads
    // the 64 marks of its window with one broadcast load, and a lane's owner is a running count + a popcount — where a binary
    // search over the lanes' prefix sums took six DEPENDENT LDS reads per trip (round 5; measured -0.8 us of 23.8, not the -4 costed).
    const bool owns = cnt != 0u;
    const uint32_t rank = (uint32_t)__popcll(gh_ballot(owns) & ((1ull << lane) - 1ull));
    s_marks[wid][lane] = 0u; s_marks[wid][lane + GH_WAVE] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (owns) {
      s_g[wid][rank] = make_uint4(small ? r : 0u, __float_as_uint(g2.z), __float_as_uint(g2.w), n);
      s_own[wid][rank] = make_uint2(x - cnt, vbase);                 // first slot of the run inside the wave's run, view's first tile id
      atomicOr(&s_marks[wid][(x - 1u) >> 5], 1u << ((x - 1u) & 31u));
    }
  } else {
    // (a wave that holds a rect of thousands of tiles: its run does not fit the mark array — a binary search over the prefix sums)
    s_end[wid][lane] = x;
    s_g[wid][lane] = make_uint4(small && cnt ? r : 0u, __float_as_uint(g2.z), __float_as_uint(g2.w), n);
    s_vb[wid][lane] = vbase;
  }
  if (lane == 63) s_w[wid] = x;
  if (lane == 0) s_p[wid] = part;
  __syncthreads();
  uint32_t woff = 0, blk_off = 0, blk_sum = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { if (w < wid) woff += s_w[w]; blk_sum += s_w[w]; blk_off += s_p[w]; }
  if ((int)blockIdx.x == n_emit_blocks - 1 && tid == 0) {        // the last emit block knows the instance total D
    const uint32_t total = blk_off + blk_sum;
    ctr->num_rendered = total;
    if (view_start) view_start[n_views] = total;
    // (the projection kernel cleared the word; bits 3 / 4 may already be set; a BINNING-only re-run clears a stale bit 0)
    if (total > cap) atomicOr(&ctr->overflow, 1u); else atomicAnd(&ctr->overflow, ~1u);
  }
  const uint32_t wave_base = blk_off + woff;
  // first emit slot of every view = that of the first Gaussian of its segment of the depth order (position v * P)
  if (view_start && i < N) {
    const uint32_t iv = rP > 0.0f ? gh_div_small((uint32_t)i, (uint32_t)P, rP) : (uint32_t)i / (uint32_t)P;
    if ((uint32_t)i == iv * (uint32_t)P) view_start[iv] = wave_base + x - cnt;
  }
  // the wave's run of slots, 64 per trip
  if (marks) {
    uint32_t done = 0;                                    // runs that end in front of this trip's window
    for (uint32_t j = 0; j < wave_total; j += GH_WAVE) {
      const uint2 w2 = *(const uint2*)&s_marks[wid][j >> 5];          // wave-uniform address: one broadcast read
      const uint64_t wm = ((uint64_t)w2.y << 32) | w2.x;
      const uint32_t o = done + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
      done += (uint32_t)__popcll(wm);
      const uint32_t sl = j + (uint32_t)lane;
      if (sl >= wave_total) break;                        // (only the last trip is partial)
      const uint4 g = s_g[wid][o];
      const uint2 ow = s_own[wid][o];
      if (g.x == 0u) continue;                            // a large rect: written by the wave below
      const uint32_t k = sl - ow.x;
      const uint32_t bit = gh_kth_set_bit(g.y, g.z, k);
      const uint32_t mnx = g.x & 255u, mny = (g.x >> 8) & 255u, wdt = ((g.x >> 16) & 255u) - mnx;
      const uint32_t dy = (uint32_t)(((float)bit + 0.5f) * __frcp_rn((float)wdt)), dx = bit - dy * wdt;
      const uint32_t slot = wave_base + sl;
      if (slot < cap) {
        keys[slot] = ow.y + (mny + dy) * (uint32_t)gx + (mnx + dx);
        vals[slot] = g.w;                                  // the emit slot is recomputed from (n, tile) after the sort
      }
    }
  } else {
    for (uint32_t j = 0; j < wave_total; j += GH_WAVE) {
      const uint32_t sl = j + (uint32_t)lane;
      if (sl >= wave_total) break;
      uint32_t o = 0;                                     // owner = number of lanes whose run ends at or before sl
#pragma unroll
      for (int step = 32; step >= 1; step >>= 1) if (s_end[wid][o + step - 1] <= sl) o += step;
      const uint4 g = s_g[wid][o];
      if (g.x == 0u) continue;
      const uint32_t k = sl - (o ? s_end[wid][o - 1] : 0u);
      const uint32_t bit = gh_kth_set_bit(g.y, g.z, k);
      const uint32_t mnx = g.x & 255u, mny = (g.x >> 8) & 255u, wdt = ((g.x >> 16) & 255u) - mnx;
      const uint32_t dy = (uint32_t)(((float)bit + 0.5f) * __frcp_rn((float)wdt)), dx = bit - dy * wdt;
      const uint32_t slot = wave_base + sl;
      if (slot < cap) {
        keys[slot] = s_vb[wid][o] + (mny + dy) * (uint32_t)gx + (mnx + dx);
        vals[slot] = g.w;
      }
    }
  }
  // Rects larger than the hit mask: the culling test again, by the WAVE for one such Gaussian at a time (64 tiles per trip,
  // the hits of a trip take consecutive slots in tile order), not by the Gaussian's own lane with the other 63 waiting.
  const bool bigl = i < N && cnt != 0 && !small;
  for (uint64_t m = gh_ballot(bigl); m != 0ull; m &= m - 1ull) {
    const int src = (int)__builtin_ctzll(m);
    auto bf = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
    const float4 g0v = grec[0], g1v = grec[1];          // (every lane reads its own line; the source lane's values are broadcast)
    const float4 s0 = make_float4(bf(g0v.x), bf(g0v.y), bf(g0v.z), bf(g0v.w));
    const float sop = bf(g1v.y);
    // (C, opacity): the projection kernel's operands — the opacity it culled with (GH_FLAG_STATIC_LISTS: gh_static_cull_opacity)
    const float4 s1 = make_float4(bf(g1v.x), (flags & GH_FLAG_STATIC_LISTS) ? gh_static_cull_opacity(sop) : sop, 0.0f, 0.0f);
    const uint32_t sr = (uint32_t)__builtin_amdgcn_readlane((int)r, src);
    const uint32_t sminx = sr & 255u, sminy = (sr >> 8) & 255u, sw = ((sr >> 16) & 255u) - sminx, sn = sw * ((sr >> 24) - sminy);
    const uint32_t sn_id = (uint32_t)__builtin_amdgcn_readlane((int)n, src), svb = (uint32_t)__builtin_amdgcn_readlane((int)vtile0, src);
    const uint32_t svk = (uint32_t)__builtin_amdgcn_readlane((int)vbase, src);      // base of the keys (0 with the per-view partition)
    uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)(wave_base + x - cnt), src);
    // (the speculative occlusion bound, exactly as the projection kernel applied it when it counted this Gaussian's tiles)
    const float stz = bf(g3.y);
    for (uint32_t base = 0; base < sn; base += GH_WAVE) {
      const uint32_t k = base + (uint32_t)lane;
      bool h = false;
      uint32_t tx = 0, ty = 0;
      if (k < sn) {
        const uint32_t dy = k / sw;
        ty = sminy + dy; tx = sminx + (k - dy * sw);
        h = gh_block_hit(s0, s1, (float)(tx * GH_TILE), (float)(ty * GH_TILE), (float)(GH_TILE - 1));
        if (tile_depth_bound) h = h && !(stz > tile_depth_bound[svb + ty * (uint32_t)gx + tx]);
      }
      const uint64_t hm = gh_ballot(h);
      if (h) {
        const uint32_t slot = off + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        if (slot < cap) {
          keys[slot] = svk + ty * (uint32_t)gx + tx;
          vals[slot] = sn_id;
        }
      }
      off += (uint32_t)__popcll(hm);
    }
  }
}

// Per sorted instance: tile ranges, the render record gathered from the Gaussian's 64-byte geometry line, and the
// 16-bit mask of the tile's 4x4-pixel blocks the alpha >= 1/255 ellipse can reach (gh_block_mask16): the render
// kernels test one bit instead of repeating the ellipse/rectangle test per wave, forward and backward.
__global__ __launch_bounds__(GH_BLOCK) void gh_ranges_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                              const GhCounters* __restrict__ ctr, uint32_t cap, int gx, int tiles,
                                                              const float4* __restrict__ geom, uint32_t* __restrict__ sorted_slot,
                                                              uint2* __restrict__ ranges, float4* __restrict__ r0,
                                                              float4* __restrict__ r1, float2* __restrict__ r2,
                                                              uint32_t* __restrict__ inst_flag, const uint32_t* __restrict__ slot_begin,
                                                              float rtiles, float rgx, uint32_t flags, float* __restrict__ inst_c,
                                                              const float* __restrict__ tile_depth_bound, uint32_t P_local, float rP,
                                                              uint32_t* __restrict__ render_guard) {
  GH_WG_TIMER(5);
  // render_guard given (small launches, whose launch order was ranked inside the projection kernel): this IS the last kernel in front
  // of the render — the error bits as they stand now (every kernel that raises one is complete), in their own word; see gh_tile_order_kernel
  if (render_guard && blockIdx.x == 0 && threadIdx.x == 0) { *render_guard = ctr->overflow & GH_COUNTER_ERROR_MASK; for (int c = 0; c < GH_BWD_CLASSES; ++c) render_guard[1 + c] = 0u; }
  const uint32_t n = gh_clamp_n(&ctr->num_rendered, cap);
  // Blocks b, b + 8, b + 16, .. share an XCD (round-robin dispatch): each of the 8 groups takes one CONTIGUOUS eighth of the
  // sorted instances. A Gaussian's instances sit in neighbouring tiles' lists — a list length apart for the tile to the
  // right, a tile row apart for the one below — so its geometry line is found in the XCD's own L2 again instead of being
  // fetched by whichever XCD the neighbouring block happened to land on. (Placement only matters for speed; the grid is a
  // multiple of 8 blocks, so every (group, position) below the chunk length exists.)
  const uint32_t chunk = ((n + GH_BLOCK - 1) / GH_BLOCK + 7u) >> 3;
  if ((blockIdx.x >> 3) >= chunk) return;
  const uint32_t i = ((blockIdx.x & 7u) * chunk + (blockIdx.x >> 3)) * GH_BLOCK + threadIdx.x;
  if (i >= n) return;
  gh_stream(&inst_flag[i], 0u);                        // quadrant flags of the backward's sub-records (emit slots 0 .. D-1)
  const uint32_t gid = vals[i];
  uint32_t t = keys[i];
  // P_local != 0: the per-view partition left tile ids INSIDE the view in the keys; the view is the payload's (view * P + row)
  auto view_tiles = [&](uint32_t g_) { return (rP > 0.0f ? gh_div_small(g_, P_local, rP) : g_ / P_local) * (uint32_t)tiles; };
  if (P_local) t += view_tiles(gid);
  if (i == 0) ranges[t].x = 0;
  else {
    uint32_t tp = keys[i - 1];
    if (P_local) tp += view_tiles(vals[i - 1]);
    if (tp != t) { ranges[tp].y = i; ranges[t].x = i; }
  }
  if (i == n - 1) ranges[t].y = n;
  const float4* grec = geom + (size_t)gid * 4;       // one 64-byte line: record, tile rect, tile hit mask
  const float4 a = grec[0], b = grec[1], c = grec[2];
  const uint32_t slot0 = slot_begin[gid];            // first record slot (4-byte gather from an L2-sized array)
  const float cb = c.x;
  uint32_t tl, ty;                                     // tile inside the view, its row
  if (rtiles > 0.0f) { tl = t - gh_div_small(t, (uint32_t)tiles, rtiles) * (uint32_t)tiles; ty = gh_div_small(tl, (uint32_t)gx, rgx); }
  else { tl = t % (uint32_t)tiles; ty = tl / (uint32_t)gx; }
  const uint32_t tx = tl - ty * (uint32_t)gx;
  // record slot of (gid, tile): the Gaussian's slots are numbered row-major over the HIT tiles of its rect
  const uint32_t r = __float_as_uint(c.y);
  const uint32_t minx = r & 255u, miny = (r >> 8) & 255u, maxx = (r >> 16) & 255u, maxy = r >> 24;
  const uint32_t bit = (ty - miny) * (maxx - minx) + (tx - minx);
  uint32_t before = 0;
  const bool small = (maxx - minx) * (maxy - miny) <= 64u;
  if (small) {
    const unsigned long long hm = ((unsigned long long)__float_as_uint(c.w) << 32) | __float_as_uint(c.z);
    before = (uint32_t)__popcll(hm & ((1ull << bit) - 1ull));
  }
  // Rects larger than the hit mask (huge footprints): the tiles in front of this one are counted again with the culling test,
  // by the WAVE for one such instance at a time — 64 tiles per trip — where the instance's own lane took one tile per trip with
  // the other 63 lanes waiting (at 1024x1024 0.04 % of the instances cost 8 % of the kernel that way).
  const uint64_t bigm = gh_ballot(!small);
  if (bigm != 0ull) {
    if (gh_ballot(true) == ~0ull) {
      const int lane = threadIdx.x & 63;
      uint64_t m = bigm;
      while (m != 0ull) {
        const int src = (int)__builtin_ctzll(m);
        m &= m - 1ull;
        auto bf = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
        const float4 sa = make_float4(bf(a.x), bf(a.y), bf(a.z), bf(a.w));
        const float sop = bf(b.y);
        const float4 sbc = make_float4(bf(b.x), (flags & GH_FLAG_STATIC_LISTS) ? gh_static_cull_opacity(sop) : sop, 0.0f, 0.0f);   // as culled
        const uint32_t sr = (uint32_t)__builtin_amdgcn_readlane((int)r, src), sbit = (uint32_t)__builtin_amdgcn_readlane((int)bit, src);
        const uint32_t sminx = sr & 255u, sminy = (sr >> 8) & 255u, sw = ((sr >> 16) & 255u) - sminx;
        // (speculative occlusion bound: the same `depth > bound` the projection and emit kernels applied)
        const float stz = tile_depth_bound ? bf(grec[3].y) : 0.0f;            // (view-space depth: the line's last float4)
        const uint32_t svb = (uint32_t)__builtin_amdgcn_readlane((int)(t - tl), src);      // first global tile id of the view
        uint32_t cnt = 0;
        for (uint32_t base = 0; base < sbit; base += GH_WAVE) {
          const uint32_t k = base + (uint32_t)lane;
          bool h = false;
          if (k < sbit) {
            const uint32_t dy = k / sw, dx = k - dy * sw;
            h = gh_block_hit(sa, sbc, (float)((sminx + dx) * GH_TILE), (float)((sminy + dy) * GH_TILE), (float)(GH_TILE - 1));
            if (tile_depth_bound) h = h && !(stz > tile_depth_bound[svb + (sminy + dy) * (uint32_t)gx + (sminx + dx)]);
          }
          cnt += (uint32_t)__popcll(gh_ballot(h));
        }
        if (lane == src) before = cnt;
      }
    } else if (!small) {                                // (the last, partial wave: lane by lane)
      uint32_t k = 0;
      const float4 bc = make_float4(b.x, (flags & GH_FLAG_STATIC_LISTS) ? gh_static_cull_opacity(b.y) : b.y, 0.0f, 0.0f);   // as culled
      const float tzl = tile_depth_bound ? grec[3].y : 0.0f;
      for (uint32_t yy = miny; yy < maxy && k < bit; ++yy)
        for (uint32_t xx = minx; xx < maxx && k < bit; ++xx, ++k) {
          bool h = gh_block_hit(a, bc, (float)(xx * GH_TILE), (float)(yy * GH_TILE), (float)(GH_TILE - 1));
          if (tile_depth_bound) h = h && !(tzl > tile_depth_bound[(t - tl) + yy * (uint32_t)gx + xx]);
          before += h ? 1u : 0u;
        }
    }
  }
  // (D > max_instances: the lists are a truncated, invalid set and record slots run up to D - 1: keep the backward's stores inside)
  const uint32_t rslot = slot0 + before;
  gh_stream(&sorted_slot[i], rslot < cap ? rslot : cap - 1u);           // (streamed: see gh_stream)
  const uint32_t m = gh_block_mask16(a, b, (float)(tx * GH_TILE), (float)(ty * GH_TILE));
  // the render kernels' records carry -A/2 and -C/2: their power = (A' dx dx + C' dy dy) - B dx dy is App. A.3's expression bit for
  // bit (a scaling by a power of two commutes with every rounding) and one multiply per (entry, pixel) cheaper, forward and backward
  gh_stream(&r0[i], make_float4(a.x, a.y, -0.5f * a.z, a.w)); gh_stream(&r1[i], make_float4(-0.5f * b.x, b.y, b.z, b.w));
  gh_stream(&r2[i], make_float2(cb, __uint_as_float(m)));
  if (flags & GH_FLAG_STATIC_LISTS) gh_stream(&inst_c[i], b.x);   // the conic's C again, compact: what gh_forward_refresh reads of r1
}

// Longest-processing-time-first launch order for the render kernels: a counting sort of the tiles by list length (256
// buckets of 16 entries, longest first). Workgroups are dispatched in blockIdx order, so the heavy tiles start at t = 0 and
// the light / empty ones fill in behind them instead of forming the tail. One block PER VIEW sorts that view's tiles; the
// global order interleaves the views rank by rank (order[r * NV + v] = view v's r-th heaviest tile), which is the merged
// order up to differences between the views' r-th tiles. The order only affects scheduling, never results; the order
// inside a bucket is arbitrary. (The backward's order comes from the forward itself: gh_render_fwd_kernel.)
__global__ __launch_bounds__(GH_BLOCK) void gh_tile_order_kernel(const uint2* __restrict__ ranges, int tiles, int NV,
                                                                  uint32_t* __restrict__ order, const GhCounters* __restrict__ ctr,
                                                                  uint32_t* __restrict__ render_guard, uint32_t* __restrict__ heavy, int use_heavy) {
  // the last kernel in front of the render: the error bits as they stand now, in a word of their own (the render kernel's waves
  // read it through the scalar cache; its own atomics go to the counters' line)
  if (blockIdx.x == 0 && threadIdx.x == 0) { *render_guard = ctr->overflow & GH_COUNTER_ERROR_MASK; for (int c = 0; c < GH_BWD_CLASSES; ++c) render_guard[1 + c] = 0u; }   // ([1..]: items per class of the backward's work list)
  gh_rank_tiles(ranges, tiles, NV, (int)blockIdx.x, order, heavy, use_heavy != 0);
}

bool gh_heavy_order_enabled() {
  static const bool by_hits = !(getenv("GH_FWD_HEAVY_ORDER") && atoi(getenv("GH_FWD_HEAVY_ORDER")) == 0);   // (A/B switch)
  return by_hits;
}

bool gh_bwd_classes_enabled() {
  static const bool on = !(getenv("GH_BWD_CLASSES") && atoi(getenv("GH_BWD_CLASSES")) == 0);               // (A/B switch)
  return on;
}

// (the rare paths of a small launch — nothing projected, nothing listed — and every large launch: the order by a kernel of its own)
static void gh_launch_tile_order(const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s) {
  uint32_t* heavy = g.total_tiles <= GH_ORDER_TILES && gh_heavy_order_enabled() ? (uint32_t*)(ws + L.tile_walk) + 3 * (size_t)g.NV * g.tiles : nullptr;
  hipLaunchKernelGGL(gh_tile_order_kernel, dim3(g.NV), dim3(GH_BLOCK), 0, s, (const uint2*)(ws + L.ranges), g.tiles, g.NV,
                     (uint32_t*)(ws + L.tile_order), (const GhCounters*)(ws + L.counters), (uint32_t*)(ws + L.render_guard), heavy,
                     (g.flags & GH_FLAG_FRESH_ORDER) ? 0 : 1);
}

void gh_launch_binning(const GhDims* d, const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s, const float* tile_depth_bound) {
  if (g.N == 0) { gh_launch_tile_order(g, ws, L, s); return; }   // ranges are all-empty (memset): any order
  const int nblk_pre = (g.N + GH_BLOCK - 1) / GH_BLOCK;
  GhCounters* ctr = (GhCounters*)(ws + L.counters);
  const uint32_t cap = (uint32_t)g.cap;
  uint32_t* table = (uint32_t*)(ws + L.sort_tables);

  // level 1: depth order of every view's Gaussians (keys / payload written by the preprocess kernel): NV segments of P
  uint32_t* dk_in = (uint32_t*)(ws + L.depth_keys_a); uint32_t* dk_out = (uint32_t*)(ws + L.depth_keys_b);
  uint32_t* dv_in = (uint32_t*)(ws + L.depth_vals_a); uint32_t* dv_out = (uint32_t*)(ws + L.depth_vals_b);
  const int T = g.NV * g.tiles;
  const int n_proj_blocks = ((g.N > T ? g.N : T) + GH_BLOCK - 1) / GH_BLOCK;          // grid of gh_preprocess_fwd_kernel
  // GH_FLAG_DEPTH24: three passes (the top byte is asserted constant and checked by the first histogram kernel), else four
  const bool d24 = (d->flags & GH_FLAG_DEPTH24) != 0;
  gh_radix_sort_ex(dk_in, dv_in, dk_out, dv_out, &ctr->reserved[0], (uint32_t)g.N, d24 ? 24 : 32, (uint32_t)g.P, g.NV, table, s,
                   (const uint2*)(ws + L.key_bits), n_proj_blocks, &ctr->overflow);
  const uint32_t* perm = dv_in;                       // (the sort swaps the pointers: four passes end in the *_a buffers, three in *_b)

  // level 2: emit in depth order
  const uint32_t* tiles_touched = (const uint32_t*)(ws + L.tiles_touched);
  const bool per_view = (d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;
  hipLaunchKernelGGL(gh_count_sorted_kernel, dim3(nblk_pre), dim3(GH_BLOCK), 0, s, g.N, perm, tiles_touched,
                     (uint32_t*)(ws + L.block_sums));
  // level 3: stable partition by tile id; an odd number of passes starts in the b buffers so the result is in *_a.
  // Two or more views: PER VIEW (round 6) — the views are contiguous segments of the emit order, so each is partitioned by the
  // ceil(log2 tiles) bits of the tile id INSIDE the view (8 views of 512x334: 2 passes of 5 bits instead of 6 + 7; 32 poses of
  // 1024x1024: 2 passes instead of 3); the segments' bounds (GhLayout.view_start) come from the emit kernel.
  const bool per_view_part = gh_partition_per_view(g);
  int vbits = 1; while ((1 << vbits) < g.tiles) ++vbits;
  const int tile_passes = per_view_part ? (vbits + 7) / 8 : gh_radix_passes((size_t)g.cap, g.tile_bits);
  uint32_t* view_start = per_view_part ? (uint32_t*)(ws + L.view_start) : nullptr;
  uint32_t* ka = (uint32_t*)(ws + L.keys_a); uint32_t* kb = (uint32_t*)(ws + L.keys_b);
  uint32_t* va = (uint32_t*)(ws + L.vals_a); uint32_t* vb = (uint32_t*)(ws + L.vals_b);
  const bool start_b = (tile_passes & 1) != 0;
  uint32_t* k_in = start_b ? kb : ka; uint32_t* k_out = start_b ? ka : kb;
  uint32_t* v_in = start_b ? vb : va; uint32_t* v_out = start_b ? va : vb;
  // (grid: the emit blocks, then as many spare blocks that number the record slots in the projection kernel's thread order)
  hipLaunchKernelGGL(gh_emit_kernel, dim3(2 * nblk_pre), dim3(GH_BLOCK), 0, s, g.N, g.P, g.gx, g.tiles, cap, perm, tiles_touched,
                     (const uint32_t*)(ws + L.block_sums), (float4*)(ws + L.geom), k_in, v_in, ctr,
                     g.N < (1 << 24) ? 1.0f / (float)g.P : 0.0f, d->flags, tile_depth_bound, nblk_pre, per_view ? 0 : g.NV,
                     g.N < (1 << 24) && !per_view ? 1.0f / (float)g.NV : 0.0f, (const uint32_t*)(ws + L.block_tiles),
                     (uint32_t*)(ws + L.slot_begin), view_start, g.NV);
  if (cap == 0) { gh_launch_tile_order(g, ws, L, s); return; }    // the emit kernel has written D (it stores nothing past cap)
  if (per_view_part) gh_radix_sort_var(k_in, v_in, k_out, v_out, view_start, g.NV, cap, vbits, table, s);
  else gh_radix_sort(k_in, v_in, k_out, v_out, &ctr->num_rendered, cap, g.tile_bits, table, s);

  const int nblk_d = (int)((g.cap + GH_BLOCK - 1) / GH_BLOCK);
  hipLaunchKernelGGL(gh_ranges_kernel, dim3((nblk_d + 7) & ~7), dim3(GH_BLOCK), 0, s, ka, va, ctr, cap, g.gx, g.tiles,
                     (const float4*)(ws + L.geom), (uint32_t*)(ws + L.sorted_slot),
                     (uint2*)(ws + L.ranges), (float4*)(ws + L.inst_r0),
                     (float4*)(ws + L.inst_r1), (float2*)(ws + L.inst_r2), (uint32_t*)(ws + L.inst_flag),
                     (const uint32_t*)(ws + L.slot_begin),
                     (long long)g.NV * g.tiles < (1ll << 24) ? 1.0f / (float)g.tiles : 0.0f, 1.0f / (float)g.gx,   // gh_div_small's range
                     d->flags, (float*)(ws + L.inst_c), tile_depth_bound, per_view_part ? (uint32_t)g.P : 0u,
                     g.N < (1 << 24) ? 1.0f / (float)g.P : 0.0f,
                     gh_order_in_projection(g) ? (uint32_t*)(ws + L.render_guard) : nullptr);
  if (!gh_order_in_projection(g)) gh_launch_tile_order(g, ws, L, s);
}
