// gh_binning.hip — tile binning for the rasteriser, hand-written for wave64.
//
// The published algorithm sorts D tile instances by the 64-bit key (tile << 32 | depth bits). The same ordering
// is produced here in two levels, which moves 3x fewer bytes:
//   1. the N = views*P Gaussians are stably sorted by depth ONCE (32-bit keys, N elements; ties keep index
//      order, culled ones go last),
//   2. instances are emitted in that order (one per touched tile, row-major in the rect), so the instance list
//      is already depth-ordered,
//   3. a STABLE radix partition by tile id (32-bit keys, D elements, ceil(tile_bits/8) passes) gathers each
//      tile's entries without disturbing their depth order  ==  stable sort by (tile, depth, index).
// Then per-tile [start,end) ranges, the per-instance render records in sorted order and the launch order.
// The instance count D never leaves the device: kernels read their element count from device memory and grids
// are sized from the caller's capacity (max_instances), so the whole stage is sync-free / graph-capturable.
#include "gh_internal.h"

// ------------------------------------------------------------------------------------------------
// LSD radix sort engine: 32-bit keys + 32-bit payload, <= 8-bit digits, element count read from device memory.
// Each block owns GH_BLOCK * ITEMS consecutive keys.
__device__ __forceinline__ uint32_t gh_clamp_n(const uint32_t* n_ptr, uint32_t cap) {
  const uint32_t n = *n_ptr;
  return n < cap ? n : cap;
}

// Pass part 1: per-block digit histogram -> table[digit][block].
template <int ITEMS>
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr,
                                                                  uint32_t cap, int shift, uint32_t dmask,
                                                                  uint32_t* __restrict__ table, int nblk_cap) {
  __shared__ uint32_t s_hist[256];
  const uint32_t n = gh_clamp_n(n_ptr, cap);
  const uint32_t base = blockIdx.x * (uint32_t)(GH_BLOCK * ITEMS);
  if (base >= n) return;
  s_hist[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const uint32_t idx = base + j * GH_BLOCK + threadIdx.x;
    if (idx < n) atomicAdd(&s_hist[(keys[idx] >> shift) & dmask], 1u);
  }
  __syncthreads();
  table[(size_t)threadIdx.x * nblk_cap + blockIdx.x] = s_hist[threadIdx.x];
}

// Pass part 2: one block per digit: exclusive scan of its row over the active blocks, row total -> tot[digit].
template <int ITEMS>
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_scan_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ tot,
                                                                  const uint32_t* __restrict__ n_ptr, uint32_t cap, int nblk_cap) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_carry;
  const uint32_t n = gh_clamp_n(n_ptr, cap);
  const int nblk = (int)((n + (GH_BLOCK * ITEMS) - 1) / (GH_BLOCK * ITEMS));
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t* row = table + (size_t)blockIdx.x * nblk_cap;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += GH_BLOCK) {
    const int idx = base + tid;
    const uint32_t v = idx < nblk ? row[idx] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    const uint32_t carry = s_carry;
    if (idx < nblk) row[idx] = carry + woff + x - v;
    __syncthreads();
    if (tid == GH_BLOCK - 1) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) tot[blockIdx.x] = s_carry;
}

// Pass part 3: stable scatter. Ranking is per wave with ballot matching (one ballot per digit bit), waves are
// ordered through an LDS prefix over their digit counts, so equal digits keep their input order. The tile is first
// sorted into LDS and then written out, so that each digit run leaves as contiguous global segments.
template <int ITEMS>
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t cap, int shift, uint32_t dmask,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ tot, int nblk_cap) {
  __shared__ uint32_t s_base[256];                         // global base of (digit, this block)
  __shared__ uint32_t s_cnt[GH_BLOCK / GH_WAVE][256];      // per-wave digit counters -> per-wave bases
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_lbase[256];                        // first position of each digit in the locally sorted tile
  __shared__ uint32_t s_key[(GH_BLOCK * ITEMS)], s_val[(GH_BLOCK * ITEMS)];
  const uint32_t n = gh_clamp_n(n_ptr, cap);
  const uint32_t blk_base = blockIdx.x * (uint32_t)(GH_BLOCK * ITEMS);
  if (blk_base >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // digit base = exclusive scan over digits of tot[] + this block's row prefix
  {
    const uint32_t v = (uint32_t)tid <= dmask ? tot[tid] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
#pragma unroll
    for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) s_cnt[w][tid] = 0;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    s_base[tid] = woff + x - v + ((uint32_t)tid <= dmask ? table[(size_t)tid * nblk_cap + blockIdx.x] : 0u);
  }
  __syncthreads();

  // Phase A: rank keys inside the wave. Wave w owns keys [w*1024, (w+1)*1024) of the block's tile,
  // visited as 16 rounds of 64 consecutive keys, so (round, lane) order == memory order.
  uint32_t key[ITEMS];
  uint32_t rank[ITEMS];
  const uint32_t wave_base = blk_base + wid * (ITEMS * GH_WAVE);
  volatile uint32_t* cnt = s_cnt[wid];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t idx = wave_base + r * GH_WAVE + lane;
    const bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : ~0u;
    const uint32_t dg = (key[r] >> shift) & dmask;
    uint64_t peers = gh_ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if ((dmask >> b) & 1u) {                  // wave-uniform: only the bits of this pass
        const bool bit = (dg >> b) & 1u;
        const uint64_t m = gh_ballot(bit);
        peers &= bit ? m : ~m;
      }
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
  // Phase B: per-wave bases inside the block's LOCALLY sorted tile (digit = tid): block count per digit, exclusive
  // scan over the digits, waves in order.
  {
    uint32_t c[GH_BLOCK / GH_WAVE];
    uint32_t tot_d = 0;
#pragma unroll
    for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) { c[w] = s_cnt[w][tid]; tot_d += c[w]; }
    uint32_t x = tot_d;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t run = x - tot_d;
    for (int w = 0; w < wid; ++w) run += s_w[w];
    s_lbase[tid] = run;
#pragma unroll
    for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) { s_cnt[w][tid] = run; run += c[w]; }
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
      s_val[lpos] = vals_in[idx];
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

// Sorts (keys, vals) on bits [0, nbits) with ceil(nbits/8) digits; in/out ping-pong (pointers are swapped so that
// on return k_in / v_in hold the result). ITEMS keys per thread: 16 for large inputs (bandwidth), 4 for small ones
// (more, shorter blocks: the pass is latency-bound when it cannot fill the 256 CUs).
template <int ITEMS>
static void gh_radix_sort_t(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr,
                            uint32_t cap, int nbits, uint32_t* table, hipStream_t s) {
  const int passes = (nbits + 7) / 8;
  const int nblk = (int)(((size_t)cap + GH_BLOCK * ITEMS - 1) / (GH_BLOCK * ITEMS));
  if (nblk == 0) return;
  uint32_t* tot = table + (size_t)256 * nblk;
  for (int p = 0; p < passes; ++p) {
    // spread the bits evenly over the passes (e.g. 13 bits -> 6 + 7)
    const int lo = (nbits * p) / passes, hi = (nbits * (p + 1)) / passes;
    const uint32_t dmask = (1u << (hi - lo)) - 1u;
    hipLaunchKernelGGL(gh_radix_hist_kernel<ITEMS>, dim3(nblk), dim3(GH_BLOCK), 0, s, k_in, n_ptr, cap, lo, dmask, table, nblk);
    hipLaunchKernelGGL(gh_radix_scan_kernel<ITEMS>, dim3(dmask + 1), dim3(GH_BLOCK), 0, s, table, tot, n_ptr, cap, nblk);
    hipLaunchKernelGGL(gh_radix_scatter_kernel<ITEMS>, dim3(nblk), dim3(GH_BLOCK), 0, s, k_in, v_in, k_out, v_out, n_ptr, cap, lo,
                       dmask, table, tot, nblk);
    uint32_t* t = k_in; k_in = k_out; k_out = t;
    t = v_in; v_in = v_out; v_out = t;
  }
}

int gh_radix_items(size_t cap) { return cap <= ((size_t)1 << 21) ? 4 : 16; }

size_t gh_radix_table_words(size_t cap) {
  const size_t tile = (size_t)GH_BLOCK * gh_radix_items(cap);
  return 256 * ((cap + tile - 1) / tile) + 256;
}

void gh_radix_sort(uint32_t*& k_in, uint32_t*& v_in, uint32_t*& k_out, uint32_t*& v_out, const uint32_t* n_ptr, uint32_t cap,
                   int nbits, uint32_t* table, hipStream_t s) {
  if (gh_radix_items(cap) == 4) gh_radix_sort_t<4>(k_in, v_in, k_out, v_out, n_ptr, cap, nbits, table, s);
  else gh_radix_sort_t<16>(k_in, v_in, k_out, v_out, n_ptr, cap, nbits, table, s);
}

// ------------------------------------------------------------------------------------------------
// Level 2: walk the Gaussians in depth order.
// Per-block sums of tiles-touched in depth-sorted order.
__global__ __launch_bounds__(GH_BLOCK) void gh_count_sorted_kernel(int N, const uint32_t* __restrict__ perm,
                                                                    const uint32_t* __restrict__ tiles_touched,
                                                                    uint32_t* __restrict__ block_sums) {
  __shared__ unsigned s_wsum[GH_BLOCK / GH_WAVE];
  const int i = blockIdx.x * GH_BLOCK + threadIdx.x;
  const unsigned c = i < N ? tiles_touched[perm[i]] : 0u;
  const unsigned ws = gh_wave_sum_u32(c);
  if ((threadIdx.x & 63) == 0) s_wsum[threadIdx.x >> 6] = ws;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
}

// Exclusive scan of the per-block tile counts (one block, carry loop) + instance total / overflow flag.
__global__ __launch_bounds__(1024) void gh_scan_blocksums_kernel(uint32_t* __restrict__ block_sums, int nblk,
                                                                  GhCounters* __restrict__ ctr, uint32_t cap) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    const int idx = base + tid;
    const uint32_t v = idx < nblk ? block_sums[idx] : 0u;
    uint32_t x = v;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    const uint32_t carry = s_carry;
    if (idx < nblk) block_sums[idx] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    const uint32_t total = s_carry;
    ctr->num_rendered = total;
    ctr->overflow = total > cap ? 1u : 0u;
  }
}

// One thread per (view, Gaussian) IN DEPTH ORDER: block-local scan -> first emit slot of the Gaussian, then one
// instance per tile of its rect that passes the exact ellipse/tile test (the same test that counted them in the
// projection kernel), row-major: key = global tile id, payload = view*P+gaussian.
// A Gaussian's instances occupy consecutive slots [slot_begin, slot_begin + tiles): the backward sums its records there.
__global__ __launch_bounds__(GH_BLOCK) void gh_emit_kernel(
    int N, int P, int gx, int tiles, uint32_t cap, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ tiles_touched,
    const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ slot_begin, float4* __restrict__ geom,
    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x * GH_BLOCK + tid;
  const uint32_t n = i < N ? perm[i] : 0u;
  float4* grec = geom + (size_t)n * 4;
  const uint32_t cnt = i < N ? tiles_touched[n] : 0u;
  const float4 g2 = grec[2];                            // rect + tile hit mask: issued before the scan, not after it
  uint32_t x = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) s_w[wid] = x;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += s_w[w];
  if (i >= N) return;
  uint32_t off = block_offsets[blockIdx.x] + woff + x - cnt;
  slot_begin[n] = off;
  if (cnt == 0) return;
  const uint32_t r = __float_as_uint(g2.y);
  const int minx = r & 255, miny = (r >> 8) & 255, maxx = (r >> 16) & 255, maxy = r >> 24;
  const uint32_t vbase = (n / (uint32_t)P) * (uint32_t)tiles;
  const bool small = (maxx - minx) * (maxy - miny) <= 64;               // the projection kernel kept the hit mask
  const unsigned long long hitmask = ((unsigned long long)__float_as_uint(g2.w) << 32) | __float_as_uint(g2.z);
  float4 g0 = make_float4(0, 0, 0, 0), g1 = g0;
  if (!small) { g0 = grec[0]; g1 = make_float4(grec[1].x, grec[1].y, 0.0f, 0.0f); }   // (C, opacity): the projection kernel's operands
  int bit = 0;
  for (int ty = miny; ty < maxy; ++ty)
    for (int tx = minx; tx < maxx; ++tx, ++bit) {
      const bool h = small ? ((hitmask >> bit) & 1ull) != 0
                           : gh_block_hit(g0, g1, (float)(tx * GH_TILE), (float)(ty * GH_TILE), (float)(GH_TILE - 1));
      if (!h) continue;
      if (off < cap) {
        keys[off] = vbase + (uint32_t)(ty * gx + tx);
        vals[off] = n;                                   // the emit slot is recomputed from (n, tile) after the sort
      }
      ++off;
    }
}

// Per sorted instance: tile ranges, the render record gathered from the Gaussian's 64-byte geometry line, and the
// 16-bit mask of the tile's 4x4-pixel blocks the alpha >= 1/255 ellipse can reach (gh_block_mask16): the render
// kernels test one bit instead of repeating the ellipse/rectangle test per wave, forward and backward.
__global__ __launch_bounds__(GH_BLOCK) void gh_ranges_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                              const GhCounters* __restrict__ ctr, uint32_t cap, int gx, int tiles,
                                                              const float4* __restrict__ geom, uint32_t* __restrict__ sorted_slot,
                                                              uint2* __restrict__ ranges, float4* __restrict__ r0,
                                                              float4* __restrict__ r1, float2* __restrict__ r2,
                                                              uint32_t* __restrict__ inst_flag, const uint32_t* __restrict__ slot_begin) {
  const uint32_t n = gh_clamp_n(&ctr->num_rendered, cap);
  const uint32_t i = blockIdx.x * GH_BLOCK + threadIdx.x;
  if (i >= n) return;
  inst_flag[i] = 0;                                    // quadrant flags of the backward's sub-records (emit slots 0 .. D-1)
  const uint32_t t = keys[i];
  if (i == 0) ranges[t].x = 0;
  else {
    const uint32_t tp = keys[i - 1];
    if (tp != t) { ranges[tp].y = i; ranges[t].x = i; }
  }
  if (i == n - 1) ranges[t].y = n;
  const uint32_t gid = vals[i];
  const float4* grec = geom + (size_t)gid * 4;       // one 64-byte line: record, tile rect, tile hit mask
  const float4 a = grec[0], b = grec[1], c = grec[2];
  const uint32_t slot0 = slot_begin[gid];            // first emit slot (4-byte gather from an L2-sized array; keeping it
                                                     // in the geometry line cost the emit kernel a scattered line write)
  const float cb = c.x;
  const uint32_t tl = t % (uint32_t)tiles, ty = tl / (uint32_t)gx, tx = tl - ty * (uint32_t)gx;
  // emit slot of (gid, tile): the Gaussian's instances were emitted row-major over the HIT tiles of its rect
  const uint32_t r = __float_as_uint(c.y);
  const uint32_t minx = r & 255u, miny = (r >> 8) & 255u, maxx = (r >> 16) & 255u, maxy = r >> 24;
  const uint32_t bit = (ty - miny) * (maxx - minx) + (tx - minx);
  uint32_t before;
  if ((maxx - minx) * (maxy - miny) <= 64u) {
    const unsigned long long hm = ((unsigned long long)__float_as_uint(c.w) << 32) | __float_as_uint(c.z);
    before = (uint32_t)__popcll(hm & ((1ull << bit) - 1ull));
  } else {                                              // rect larger than the mask: recount (rare, huge footprints)
    before = 0;
    uint32_t k = 0;
    for (uint32_t yy = miny; yy < maxy && k < bit; ++yy)
      for (uint32_t xx = minx; xx < maxx && k < bit; ++xx, ++k)
        before += gh_block_hit(a, b, (float)(xx * GH_TILE), (float)(yy * GH_TILE), (float)(GH_TILE - 1)) ? 1u : 0u;
  }
  sorted_slot[i] = slot0 + before;
  const uint32_t m = gh_block_mask16(a, b, (float)(tx * GH_TILE), (float)(ty * GH_TILE));
  r0[i] = a; r1[i] = b; r2[i] = make_float2(cb, __uint_as_float(m));
}

// Longest-processing-time-first launch order for the render kernels: a counting sort of the tiles by list
// length (256 buckets of 16 entries, longest first). Workgroups are dispatched in blockIdx order, so the heavy
// tiles start at t=0 and the light / empty ones fill in behind them instead of forming the tail.
// One block; the order inside a bucket is arbitrary (it only affects scheduling, never results).
// The key is the tile's list length. (The backward's order comes from the forward itself: gh_render_fwd_kernel.)
__global__ __launch_bounds__(1024) void gh_tile_order_kernel(const uint2* __restrict__ ranges, int ntiles,
                                                              uint32_t* __restrict__ order) {
  __shared__ uint32_t s_cnt[256];
  __shared__ uint32_t s_w[4];
  const int tid = threadIdx.x;
  if (tid < 256) s_cnt[tid] = 0;
  __syncthreads();
  for (int t = tid; t < ntiles; t += 1024) {
    const uint2 r = ranges[t];
    const uint32_t len = r.y - r.x;
    uint32_t b = (len + 15u) >> 4; b = b > 255u ? 255u : b;
    atomicAdd(&s_cnt[255u - b], 1u);                 // bucket 0 = longest lists
  }
  __syncthreads();
  // exclusive scan of the 256 counters by the first 4 waves
  const int lane = tid & 63, wid = tid >> 6;
  uint32_t v = 0, x = 0;
  if (tid < 256) {
    v = s_cnt[tid];
    x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    s_cnt[tid] = woff + x - v;
  }
  __syncthreads();
  for (int t = tid; t < ntiles; t += 1024) {
    const uint2 r = ranges[t];
    const uint32_t len = r.y - r.x;
    uint32_t b = (len + 15u) >> 4; b = b > 255u ? 255u : b;
    order[atomicAdd(&s_cnt[255u - b], 1u)] = (uint32_t)t;
  }
}

static void gh_launch_tile_order(const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s) {
  hipLaunchKernelGGL(gh_tile_order_kernel, dim3(1), dim3(1024), 0, s, (const uint2*)(ws + L.ranges), g.NV * g.tiles,
                     (uint32_t*)(ws + L.tile_order));
}

void gh_launch_binning(const GhDims* d, const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s) {
  if (g.N == 0) { gh_launch_tile_order(g, ws, L, s); return; }   // ranges are all-empty (memset): any order
  const int nblk_pre = (g.N + GH_BLOCK - 1) / GH_BLOCK;
  GhCounters* ctr = (GhCounters*)(ws + L.counters);
  const uint32_t cap = (uint32_t)g.cap;
  uint32_t* table = (uint32_t*)(ws + L.sort_tables);

  // level 1: depth order of the Gaussians (keys / payload written by the preprocess kernel; reserved[0] = N)
  uint32_t* dk_in = (uint32_t*)(ws + L.depth_keys_a); uint32_t* dk_out = (uint32_t*)(ws + L.depth_keys_b);
  uint32_t* dv_in = (uint32_t*)(ws + L.depth_vals_a); uint32_t* dv_out = (uint32_t*)(ws + L.depth_vals_b);
  gh_radix_sort(dk_in, dv_in, dk_out, dv_out, &ctr->reserved[0], (uint32_t)g.N, 32, table, s);
  const uint32_t* perm = dv_in;                       // 4 passes: the result is back in the *_a buffers

  // level 2: emit in depth order
  const uint32_t* tiles_touched = (const uint32_t*)(ws + L.tiles_touched);
  hipLaunchKernelGGL(gh_count_sorted_kernel, dim3(nblk_pre), dim3(GH_BLOCK), 0, s, g.N, perm, tiles_touched,
                     (uint32_t*)(ws + L.block_sums));
  hipLaunchKernelGGL(gh_scan_blocksums_kernel, dim3(1), dim3(1024), 0, s, (uint32_t*)(ws + L.block_sums), nblk_pre, ctr, cap);
  if (cap == 0) { gh_launch_tile_order(g, ws, L, s); return; }
  // level 3: stable partition by tile id; an odd number of passes starts in the b buffers so the result is in *_a
  const int tile_passes = (g.tile_bits + 7) / 8;
  uint32_t* ka = (uint32_t*)(ws + L.keys_a); uint32_t* kb = (uint32_t*)(ws + L.keys_b);
  uint32_t* va = (uint32_t*)(ws + L.vals_a); uint32_t* vb = (uint32_t*)(ws + L.vals_b);
  const bool start_b = (tile_passes & 1) != 0;
  uint32_t* k_in = start_b ? kb : ka; uint32_t* k_out = start_b ? ka : kb;
  uint32_t* v_in = start_b ? vb : va; uint32_t* v_out = start_b ? va : vb;
  hipLaunchKernelGGL(gh_emit_kernel, dim3(nblk_pre), dim3(GH_BLOCK), 0, s, g.N, g.P, g.gx, g.tiles, cap, perm, tiles_touched,
                     (const uint32_t*)(ws + L.block_sums), (uint32_t*)(ws + L.slot_begin), (float4*)(ws + L.geom), k_in, v_in);
  gh_radix_sort(k_in, v_in, k_out, v_out, &ctr->num_rendered, cap, g.tile_bits, table, s);

  const int nblk_d = (int)((g.cap + GH_BLOCK - 1) / GH_BLOCK);
  hipLaunchKernelGGL(gh_ranges_kernel, dim3(nblk_d), dim3(GH_BLOCK), 0, s, ka, va, ctr, cap, g.gx, g.tiles,
                     (const float4*)(ws + L.geom), (uint32_t*)(ws + L.sorted_slot),
                     (uint2*)(ws + L.ranges), (float4*)(ws + L.inst_r0),
                     (float4*)(ws + L.inst_r1), (float2*)(ws + L.inst_r2), (uint32_t*)(ws + L.inst_flag),
                     (const uint32_t*)(ws + L.slot_begin));
  gh_launch_tile_order(g, ws, L, s);
}
