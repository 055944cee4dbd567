// gh_binning.hip — tile binning for the rasteriser, hand-written for wave64:
//   scan of tiles-touched  ->  instance emit (key = tile<<32 | depth bits, payload = emit slot)
//   ->  stable LSD radix sort (8-bit digits, ballot match ranking)  ->  per-tile [start,end) ranges.
// The instance count D never leaves the device: every kernel reads it from GhCounters and grids are
// sized from the caller's capacity (max_instances), so the whole stage is sync-free / graph-capturable.
#include "gh_internal.h"

// ------------------------------------------------------------------------------------------------
// Exclusive scan of the per-block tile counts (one block, carry loop) + instance total / overflow flag.
__global__ __launch_bounds__(1024) void gh_scan_blocksums_kernel(uint32_t* __restrict__ block_sums, int nblk,
                                                                  GhCounters* __restrict__ ctr, uint32_t cap) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    int idx = base + tid;
    uint32_t v = idx < nblk ? block_sums[idx] : 0u;
    uint32_t x = v;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    uint32_t carry = s_carry;
    if (idx < nblk) block_sums[idx] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t total = s_carry;
    ctr->num_rendered = total;
    ctr->overflow = total > cap ? 1u : 0u;
  }
}

// One thread per (view, Gaussian): block-local inclusive scan -> global offsets, then emit its tiles in
// row-major rect order (the emit order is the stable tie-break of the sort, App. A.2).
__global__ __launch_bounds__(GH_BLOCK) void gh_emit_kernel(
    int N, int P, int gx, int tiles, uint32_t cap, uint32_t* __restrict__ offsets /* in: tiles touched, out: inclusive scan */,
    const uint32_t* __restrict__ block_offsets, const uint32_t* __restrict__ rect, const float* __restrict__ depth,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ slot_gid) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n = blockIdx.x * GH_BLOCK + tid;
  uint32_t cnt = n < N ? offsets[n] : 0u;
  uint32_t x = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
  if (lane == 63) s_w[wid] = x;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += s_w[w];
  const uint32_t incl = block_offsets[blockIdx.x] + woff + x;
  if (n >= N) return;
  offsets[n] = incl;
  if (cnt == 0) return;
  uint32_t off = incl - cnt;
  const uint32_t r = rect[n];
  const int minx = r & 255, miny = (r >> 8) & 255, maxx = (r >> 16) & 255, maxy = r >> 24;
  const uint32_t dbits = __float_as_uint(depth[n]);
  const uint64_t vbase = (uint64_t)(n / P) * (uint64_t)tiles;
  for (int ty = miny; ty < maxy; ++ty)
    for (int tx = minx; tx < maxx; ++tx) {
      if (off < cap) {
        uint64_t tile = vbase + (uint64_t)ty * gx + tx;
        keys[off] = (tile << 32) | dbits;
        vals[off] = off;
        slot_gid[off] = (uint32_t)n;
      }
      ++off;
    }
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort, one 8-bit digit per pass. Each block owns GH_SORT_TILE consecutive keys.
__device__ __forceinline__ uint32_t gh_sort_n(const GhCounters* ctr, uint32_t cap) {
  uint32_t n = ctr->num_rendered;
  return n < cap ? n : cap;
}

// Pass part 1: per-block digit histogram -> table[digit][block].
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_hist_kernel(const uint64_t* __restrict__ keys,
                                                                  const GhCounters* __restrict__ ctr, uint32_t cap,
                                                                  int shift, uint32_t* __restrict__ table, int nblk_cap) {
  __shared__ uint32_t s_hist[256];
  const uint32_t n = gh_sort_n(ctr, cap);
  const uint32_t base = blockIdx.x * (uint32_t)GH_SORT_TILE;
  if (base >= n) return;
  s_hist[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < GH_SORT_ITEMS; ++j) {
    uint32_t idx = base + j * GH_BLOCK + threadIdx.x;
    if (idx < n) atomicAdd(&s_hist[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
  }
  __syncthreads();
  table[(size_t)threadIdx.x * nblk_cap + blockIdx.x] = s_hist[threadIdx.x];
}

// Pass part 2: one block per digit: exclusive scan of its row over the active blocks, row total -> tot[digit].
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_scan_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ tot,
                                                                  const GhCounters* __restrict__ ctr, uint32_t cap, int nblk_cap) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_carry;
  const uint32_t n = gh_sort_n(ctr, cap);
  const int nblk = (int)((n + GH_SORT_TILE - 1) / GH_SORT_TILE);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  uint32_t* row = table + (size_t)blockIdx.x * nblk_cap;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += GH_BLOCK) {
    int idx = base + tid;
    uint32_t v = idx < nblk ? row[idx] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    uint32_t carry = s_carry;
    if (idx < nblk) row[idx] = carry + woff + x - v;
    __syncthreads();
    if (tid == GH_BLOCK - 1) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) tot[blockIdx.x] = s_carry;
}

// Pass part 3: stable scatter. Ranking is per wave with ballot matching (8 ballots per key), waves are
// ordered through an LDS prefix over their digit counts, so equal digits keep their input order.
__global__ __launch_bounds__(GH_BLOCK) void gh_radix_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const GhCounters* __restrict__ ctr, uint32_t cap, int shift,
    const uint32_t* __restrict__ table, const uint32_t* __restrict__ tot, int nblk_cap) {
  __shared__ uint32_t s_base[256];                         // global base of (digit, this block)
  __shared__ uint32_t s_cnt[GH_BLOCK / GH_WAVE][256];      // per-wave digit counters -> per-wave bases
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  const uint32_t n = gh_sort_n(ctr, cap);
  const uint32_t blk_base = blockIdx.x * (uint32_t)GH_SORT_TILE;
  if (blk_base >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // digit base = exclusive scan over digits of tot[] + this block's row prefix
  {
    uint32_t v = tot[tid];
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) s_w[wid] = x;
#pragma unroll
    for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) s_cnt[w][tid] = 0;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_w[w];
    s_base[tid] = woff + x - v + table[(size_t)tid * nblk_cap + blockIdx.x];
  }
  __syncthreads();

  // Phase A: rank keys inside the wave. Wave w owns keys [w*1024, (w+1)*1024) of the block's tile,
  // visited as 16 rounds of 64 consecutive keys, so (round, lane) order == memory order.
  uint64_t key[GH_SORT_ITEMS];
  uint32_t rank[GH_SORT_ITEMS];
  const uint32_t wave_base = blk_base + wid * (GH_SORT_ITEMS * GH_WAVE);
  volatile uint32_t* cnt = s_cnt[wid];
  const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < GH_SORT_ITEMS; ++r) {
    const uint32_t idx = wave_base + r * GH_WAVE + lane;
    const bool valid = idx < n;
    key[r] = valid ? keys_in[idx] : ~0ull;
    const uint32_t dg = (uint32_t)(key[r] >> shift) & 255u;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dg >> b) & 1u;
      const uint64_t m = __ballot(bit);
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
  // Phase B: turn per-wave counts into per-wave bases (digit = tid), in wave order.
  {
    uint32_t run = s_base[tid];
#pragma unroll
    for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) { uint32_t c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
  }
  __syncthreads();
  // Phase C: scatter.
#pragma unroll
  for (int r = 0; r < GH_SORT_ITEMS; ++r) {
    const uint32_t idx = wave_base + r * GH_WAVE + lane;
    if (idx < n) {
      const uint32_t dg = (uint32_t)(key[r] >> shift) & 255u;
      const uint32_t dst = s_cnt[wid][dg] + rank[r];
      keys_out[dst] = key[r];
      vals_out[dst] = vals_in[idx];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Per-tile ranges from the sorted keys + the per-instance render records in sorted order: the one gather
// of the pipeline happens here, massively parallel, so both render kernels stream contiguous records.
__global__ __launch_bounds__(GH_BLOCK) void gh_ranges_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                              const uint32_t* __restrict__ slot_gid, const GhCounters* __restrict__ ctr,
                                                              uint32_t cap, uint2* __restrict__ ranges, uint32_t* __restrict__ sorted_gid,
                                                              const float4* __restrict__ g0, const float4* __restrict__ g1,
                                                              const float* __restrict__ gb, float4* __restrict__ r0,
                                                              float4* __restrict__ r1, float* __restrict__ r2) {
  const uint32_t n = gh_sort_n(ctr, cap);
  const uint32_t i = blockIdx.x * GH_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = (uint32_t)(keys[i] >> 32);
  if (i == 0) ranges[t].x = 0;
  else {
    const uint32_t tp = (uint32_t)(keys[i - 1] >> 32);
    if (tp != t) { ranges[tp].y = i; ranges[t].x = i; }
  }
  if (i == n - 1) ranges[t].y = n;
  const uint32_t gid = slot_gid[vals[i]];
  sorted_gid[i] = gid;
  r0[i] = g0[gid]; r1[i] = g1[gid]; r2[i] = gb[gid];
}

// Longest-processing-time-first launch order for the render kernels: a counting sort of the tiles by list
// length (256 buckets of 16 entries, longest first). Workgroups are dispatched in blockIdx order, so the heavy
// tiles start at t=0 and the light / empty ones fill in behind them instead of forming the tail.
// One block; the order inside a bucket is arbitrary (it only affects scheduling, never results).
// The key is the tile's list length (forward) or, when `walk` is given, the number of list entries the forward
// actually walked before every pixel saturated (backward: exact work proxy).
__global__ __launch_bounds__(1024) void gh_tile_order_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ walk,
                                                              int ntiles, uint32_t* __restrict__ order) {
  __shared__ uint32_t s_cnt[256];
  __shared__ uint32_t s_w[4];
  const int tid = threadIdx.x;
  if (tid < 256) s_cnt[tid] = 0;
  __syncthreads();
  for (int t = tid; t < ntiles; t += 1024) {
    uint32_t len;
    if (walk) len = walk[t]; else { const uint2 r = ranges[t]; len = r.y - r.x; }
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
    for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
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
    uint32_t len;
    if (walk) len = walk[t]; else { const uint2 r = ranges[t]; len = r.y - r.x; }
    uint32_t b = (len + 15u) >> 4; b = b > 255u ? 255u : b;
    order[atomicAdd(&s_cnt[255u - b], 1u)] = (uint32_t)t;
  }
}

static void gh_launch_tile_order(const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s) {
  hipLaunchKernelGGL(gh_tile_order_kernel, dim3(1), dim3(1024), 0, s, (const uint2*)(ws + L.ranges), (const uint32_t*)nullptr,
                     g.NV * g.tiles, (uint32_t*)(ws + L.tile_order));
}

void gh_launch_tile_order_bwd(const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s) {
  hipLaunchKernelGGL(gh_tile_order_kernel, dim3(1), dim3(1024), 0, s, (const uint2*)(ws + L.ranges),
                     (const uint32_t*)(ws + L.tile_walk), g.NV * g.tiles, (uint32_t*)(ws + L.tile_order_bwd));
}

void gh_launch_binning(const GhDims* d, const GhGrid& g, char* ws, const GhLayout& L, hipStream_t s) {
  if (g.N == 0) { gh_launch_tile_order(g, ws, L, s); return; }   // ranges are all-empty (memset): any order
  const int nblk_pre = (g.N + GH_BLOCK - 1) / GH_BLOCK;
  GhCounters* ctr = (GhCounters*)(ws + L.counters);
  const uint32_t cap = (uint32_t)g.cap;
  uint64_t* ka = (uint64_t*)(ws + L.keys_a); uint64_t* kb = (uint64_t*)(ws + L.keys_b);
  uint32_t* va = (uint32_t*)(ws + L.vals_a); uint32_t* vb = (uint32_t*)(ws + L.vals_b);
  hipLaunchKernelGGL(gh_scan_blocksums_kernel, dim3(1), dim3(1024), 0, s, (uint32_t*)(ws + L.block_sums), nblk_pre, ctr, cap);
  if (cap == 0) { gh_launch_tile_order(g, ws, L, s); return; }
  // an odd number of passes starts in the b buffers so the result always lands in keys_a / vals_a
  const bool start_b = (g.n_pass & 1) != 0;
  hipLaunchKernelGGL(gh_emit_kernel, dim3(nblk_pre), dim3(GH_BLOCK), 0, s, g.N, g.P, g.gx, g.tiles, cap,
                     (uint32_t*)(ws + L.offsets), (const uint32_t*)(ws + L.block_sums), (const uint32_t*)(ws + L.rect),
                     (const float*)(ws + L.depth), start_b ? kb : ka, start_b ? vb : va, (uint32_t*)(ws + L.slot_gid));
  uint32_t* table = (uint32_t*)(ws + L.sort_tables);
  uint32_t* tot = table + (size_t)256 * g.nblk_sort;
  uint64_t* kin = start_b ? kb : ka; uint64_t* kout = start_b ? ka : kb;
  uint32_t* vin = start_b ? vb : va; uint32_t* vout = start_b ? va : vb;
  for (int p = 0; p < g.n_pass; ++p) {
    const int shift = 8 * p;
    hipLaunchKernelGGL(gh_radix_hist_kernel, dim3(g.nblk_sort), dim3(GH_BLOCK), 0, s, kin, ctr, cap, shift, table, g.nblk_sort);
    hipLaunchKernelGGL(gh_radix_scan_kernel, dim3(256), dim3(GH_BLOCK), 0, s, table, tot, ctr, cap, g.nblk_sort);
    hipLaunchKernelGGL(gh_radix_scatter_kernel, dim3(g.nblk_sort), dim3(GH_BLOCK), 0, s, kin, vin, kout, vout, ctr, cap, shift,
                       table, tot, g.nblk_sort);
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  const int nblk_d = (int)((g.cap + GH_BLOCK - 1) / GH_BLOCK);
  hipLaunchKernelGGL(gh_ranges_kernel, dim3(nblk_d), dim3(GH_BLOCK), 0, s, ka, va, (const uint32_t*)(ws + L.slot_gid), ctr, cap,
                     (uint2*)(ws + L.ranges), (uint32_t*)(ws + L.sorted_gid), (const float4*)(ws + L.geom_g0),
                     (const float4*)(ws + L.geom_g1), (const float*)(ws + L.geom_b), (float4*)(ws + L.inst_r0),
                     (float4*)(ws + L.inst_r1), (float*)(ws + L.inst_r2));
  gh_launch_tile_order(g, ws, L, s);
}
