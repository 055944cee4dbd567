// gh_select.hip — the Gaussian selection of forward_single_batch (tgs/models/renderer_one_shot.py:468-477) as a stable
// stream compaction on the device: rows whose validity score exceeds threshold_low are kept (:469-470), rows above
// threshold_high are copied a second time (:472-473; the reference then refines the copies' positions and concatenates,
// :474-477). The reference does this with four boolean-mask indexings, each a host round trip (nonzero); here the two
// row sets are produced by three small launches with the counts left on the device.
// HBM-bound byte moving: wave ballots rank the rows, a row is copied by the whole wave (lanes = columns).
#include "gh_internal.h"

// per block of 256 rows: how many pass each threshold
__global__ __launch_bounds__(GH_BLOCK) void gh_select_count_kernel(const float* __restrict__ score, int N, float lo, float hi,
                                                                    uint32_t* __restrict__ blk_cnt /* [2][nblk] */) {
  __shared__ uint32_t s_a[GH_BLOCK / GH_WAVE], s_b[GH_BLOCK / GH_WAVE];
  const int i = blockIdx.x * GH_BLOCK + threadIdx.x;
  const float sc = i < N ? score[i] : 0.0f;
  const bool a = i < N && sc > lo, b = i < N && sc > hi;          // NaN compares false: dropped, like the reference's mask
  const uint32_t ca = (uint32_t)__popcll(gh_ballot(a)), cb = (uint32_t)__popcll(gh_ballot(b));
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = ca; s_b[threadIdx.x >> 6] = cb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_cnt[blockIdx.x] = s_a[0] + s_a[1] + s_a[2] + s_a[3];
    blk_cnt[gridDim.x + blockIdx.x] = s_b[0] + s_b[1] + s_b[2] + s_b[3];
  }
}

// one block: exclusive scan of the two rows of block counts in place, totals -> counts[0..1]
__global__ __launch_bounds__(GH_BLOCK) void gh_select_scan_kernel(uint32_t* __restrict__ blk_cnt, int nblk, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_w[GH_BLOCK / GH_WAVE];
  __shared__ uint32_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int r = 0; r < 2; ++r) {
    uint32_t* row = blk_cnt + (size_t)r * nblk;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += GH_BLOCK) {
      const int i = base + tid;
      const uint32_t v = i < nblk ? row[i] : 0u;
      uint32_t x = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
      if (lane == 63) s_w[wid] = x;
      __syncthreads();
      uint32_t woff = 0;
      for (int w = 0; w < wid; ++w) woff += s_w[w];
      const uint32_t carry = s_carry;
      if (i < nblk) row[i] = carry + woff + x - v;
      __syncthreads();
      if (tid == GH_BLOCK - 1) s_carry = carry + woff + x;
      __syncthreads();
    }
    if (tid == 0) counts[r] = s_carry;
    __syncthreads();
  }
}

// Copies row `src_row` of the two arrays to row `dst_row`: lanes = columns (3 position floats, C feature floats).
__device__ __forceinline__ void gh_copy_row(const float* __restrict__ pts, const float* __restrict__ feat, int C, uint32_t src_row,
                                            float* __restrict__ out_pts, float* __restrict__ out_feat, uint32_t dst_row, int lane) {
  if (lane < 3) out_pts[(size_t)dst_row * 3 + lane] = pts[(size_t)src_row * 3 + lane];
  for (int c = lane; c < C; c += GH_WAVE) out_feat[(size_t)dst_row * C + c] = feat[(size_t)src_row * C + c];
}

// wave w of block b owns rows [b*256 + w*64, +64): the selected ones keep their order (stable compaction)
__global__ __launch_bounds__(GH_BLOCK) void gh_select_scatter_kernel(
    const float* __restrict__ score, int N, float lo, float hi, const float* __restrict__ pts, const float* __restrict__ feat, int C,
    const uint32_t* __restrict__ blk_off /* [2][nblk], scanned */, float* __restrict__ valid_pts, float* __restrict__ valid_feat,
    float* __restrict__ copy_pts, float* __restrict__ copy_feat, int32_t* __restrict__ valid_idx, int32_t* __restrict__ copy_idx) {
  __shared__ uint32_t s_a[GH_BLOCK / GH_WAVE], s_b[GH_BLOCK / GH_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = blockIdx.x * GH_BLOCK + tid;
  const float sc = i < N ? score[i] : 0.0f;
  const bool a = i < N && sc > lo, b = i < N && sc > hi;
  uint64_t ma = gh_ballot(a), mb = gh_ballot(b);
  if (lane == 0) { s_a[wid] = (uint32_t)__popcll(ma); s_b[wid] = (uint32_t)__popcll(mb); }
  __syncthreads();
  uint32_t oa = blk_off[blockIdx.x], ob = blk_off[gridDim.x + blockIdx.x];
  for (int w = 0; w < wid; ++w) { oa += s_a[w]; ob += s_b[w]; }
  const uint32_t row0 = (uint32_t)(blockIdx.x * GH_BLOCK + wid * GH_WAVE);
  // the wave walks its selected rows one by one (wave-uniform), every row copied by all lanes together
  while (ma) {
    const int j = __builtin_ctzll(ma);
    ma &= ma - 1;
    gh_copy_row(pts, feat, C, row0 + (uint32_t)j, valid_pts, valid_feat, oa, lane);
    if (valid_idx && lane == 0) valid_idx[oa] = (int32_t)(row0 + (uint32_t)j);
    ++oa;
  }
  while (mb) {
    const int j = __builtin_ctzll(mb);
    mb &= mb - 1;
    gh_copy_row(pts, feat, C, row0 + (uint32_t)j, copy_pts, copy_feat, ob, lane);
    if (copy_idx && lane == 0) copy_idx[ob] = (int32_t)(row0 + (uint32_t)j);
    ++ob;
  }
}

extern "C" size_t gh_select_workspace_bytes(int N) {
  if (N < 0) return 0;
  const size_t nblk = ((size_t)N + GH_BLOCK - 1) / GH_BLOCK;
  return (2 * nblk + 2) * sizeof(uint32_t);
}

extern "C" int gh_select_rows(const float* score, int N, float threshold_low, float threshold_high, const float* points,
                              const float* features, int C, float* valid_points, float* valid_features, float* copied_points,
                              float* copied_features, int32_t* valid_index, int32_t* copied_index, uint32_t* counts,
                              void* workspace, size_t ws_bytes, void* hip_stream) {
  if (N < 0 || C < 0 || !counts) return GH_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)hip_stream;
  (void)hipGetLastError();
  if (N == 0) {
    (void)hipMemsetAsync(counts, 0, 2 * sizeof(uint32_t), s);
    return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
  }
  if (!score || !points || (C > 0 && !features) || !valid_points || !copied_points || (C > 0 && (!valid_features || !copied_features)) ||
      !workspace)
    return GH_ERR_INVALID_ARG;
  if (ws_bytes < gh_select_workspace_bytes(N)) return GH_ERR_WORKSPACE_SMALL;
  const int nblk = (N + GH_BLOCK - 1) / GH_BLOCK;
  uint32_t* blk = (uint32_t*)workspace;
  hipLaunchKernelGGL(gh_select_count_kernel, dim3(nblk), dim3(GH_BLOCK), 0, s, score, N, threshold_low, threshold_high, blk);
  hipLaunchKernelGGL(gh_select_scan_kernel, dim3(1), dim3(GH_BLOCK), 0, s, blk, nblk, counts);
  hipLaunchKernelGGL(gh_select_scatter_kernel, dim3(nblk), dim3(GH_BLOCK), 0, s, score, N, threshold_low, threshold_high, points,
                     features, C, blk, valid_points, valid_features, copied_points, copied_features, valid_index, copied_index);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
