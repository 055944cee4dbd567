// gh_render.hip — per-tile alpha compositing and its backward (App. A.4).
//
// Work decomposition (wave64-native): one 256-thread workgroup per 16x16 tile, one wave per 8x8 pixel
// quadrant. The tile's depth-sorted Gaussian list is staged through LDS in chunks; every wave walks the
// chunk with wave-uniform control flow (ballot skip when no pixel of the quadrant is touched).
// Backward: each pixel replays its list back to front; the 9 per-Gaussian partial gradients are summed
// across the 64 lanes with DPP, across the 4 waves through LDS in fixed order, and written once per
// (tile, Gaussian) instance as a 48-byte record at the instance's emit slot. The per-Gaussian kernel
// then sums each Gaussian's contiguous records — no global atomics, bitwise reproducible gradients.
#include "gh_internal.h"

#define GH_CHUNK 256

__device__ __forceinline__ void gh_tile_coords(int blk, int gx, int tiles, int& v, int& tx, int& ty) {
  v = blk / tiles;
  int t = blk - v * tiles;
  ty = t / gx; tx = t - ty * gx;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GH_BLOCK) void gh_render_fwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ sorted_gid, const float4* __restrict__ g0,
    const float4* __restrict__ g1, const float* __restrict__ gb, const float* __restrict__ cams, int H, int W, int gx,
    int tiles, float* __restrict__ image, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  __shared__ float4 s_g0[GH_CHUNK];
  __shared__ float4 s_g1[GH_CHUNK];
  __shared__ float s_b[GH_CHUNK];
  int v, tx, ty;
  gh_tile_coords(blockIdx.x, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lx = (wid & 1) * 8 + (lane & 7), ly = (wid >> 1) * 8 + (lane >> 3);
  const int x = tx * GH_TILE + lx, y = ty * GH_TILE + ly;
  const bool inside = x < W && y < H;
  const float pxf = (float)x, pyf = (float)y;
  const uint2 range = ranges[blockIdx.x];
  const int total = (int)(range.y - range.x);
  const int rounds = (total + GH_CHUNK - 1) / GH_CHUNK;

  bool done = !inside;
  float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
  uint32_t contributor = 0, last = 0;
  int todo = total;
  for (int r = 0; r < rounds; ++r, todo -= GH_CHUNK) {
    if (__syncthreads_count(done) == GH_BLOCK) break;
    const int idx = r * GH_CHUNK + tid;
    if (idx < total) {
      const uint32_t gid = sorted_gid[range.x + idx];
      s_g0[tid] = g0[gid]; s_g1[tid] = g1[gid]; s_b[tid] = gb[gid];
    }
    __syncthreads();
    const int cnt = todo < GH_CHUNK ? todo : GH_CHUNK;
    for (int j = 0; j < cnt; ++j) {
      if (__all(done)) break;                       // wave-uniform: this quadrant is finished
      const float4 a = s_g0[j], b4 = s_g1[j];       // LDS broadcast reads
      if (!done) {
        ++contributor;
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float power = -0.5f * (a.z * dx * dx + b4.x * dy * dy) - a.w * dx * dy;
        if (power <= 0.0f) {
          const float alpha = fminf(0.99f, b4.y * gh_exp(power));
          if (alpha >= 1.0f / 255.0f) {
            const float test_T = T * (1.0f - alpha);
            if (test_T < 0.0001f) {
              done = true;
            } else {
              const float w = alpha * T;
              C0 = fmaf(b4.z, w, C0); C1 = fmaf(b4.w, w, C1); C2 = fmaf(s_b[j], w, C2);
              T = test_T; last = contributor;
            }
          }
        }
      }
    }
  }
  if (inside) {
    const float* bg = cams + (size_t)v * GH_CAM_FLOATS + 37;
    const size_t pix = ((size_t)v * H + y) * W + x;
    final_T[pix] = T;
    n_contrib[pix] = last;
    float* img = image + (size_t)v * 3 * H * W + (size_t)y * W + x;
    img[0] = fmaf(T, bg[0], C0);
    img[(size_t)H * W] = fmaf(T, bg[1], C1);
    img[(size_t)2 * H * W] = fmaf(T, bg[2], C2);
  }
}

void gh_launch_render_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, float* image, char* ws, const GhLayout& L,
                          hipStream_t s) {
  hipLaunchKernelGGL(gh_render_fwd_kernel, dim3(g.NV * g.tiles), dim3(GH_BLOCK), 0, s, (const uint2*)(ws + L.ranges),
                     (const uint32_t*)(ws + L.sorted_gid), (const float4*)(ws + L.geom_g0), (const float4*)(ws + L.geom_g1),
                     (const float*)(ws + L.geom_b), in->cams, g.H, g.W, g.gx, g.tiles, image, (float*)(ws + L.final_T),
                     (uint32_t*)(ws + L.n_contrib));
}

// ------------------------------------------------------------------------------------------------
#define GH_BCHUNK 128   // instances per LDS chunk in the backward walk

__global__ __launch_bounds__(GH_BLOCK) void gh_render_bwd_kernel(
    const uint2* __restrict__ ranges, const uint32_t* __restrict__ sorted_gid, const uint32_t* __restrict__ sorted_slot,
    const float4* __restrict__ g0, const float4* __restrict__ g1, const float* __restrict__ gb, const float* __restrict__ cams,
    int H, int W, int gx, int tiles, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dimage, float* __restrict__ inst_grad) {
  __shared__ float4 s_g0[GH_BCHUNK];
  __shared__ float4 s_g1[GH_BCHUNK];
  __shared__ float s_b[GH_BCHUNK];
  __shared__ uint32_t s_slot[GH_BCHUNK];
  __shared__ float s_part[GH_BLOCK / GH_WAVE][GH_BCHUNK][GH_REC];   // per-wave partial records
  __shared__ uint32_t s_touched[GH_BLOCK / GH_WAVE][GH_BCHUNK / 32];  // which records a wave wrote
  __shared__ int s_max;
  int v, tx, ty;
  gh_tile_coords(blockIdx.x, gx, tiles, v, tx, ty);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lx = (wid & 1) * 8 + (lane & 7), ly = (wid >> 1) * 8 + (lane >> 3);
  const int x = tx * GH_TILE + lx, y = ty * GH_TILE + ly;
  const bool inside = x < W && y < H;
  const float pxf = (float)x, pyf = (float)y;
  const uint2 range = ranges[blockIdx.x];
  const int total = (int)(range.y - range.x);
  if (total == 0) return;

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
  const float bg_dot = bg[0] * d0 + bg[1] * d1 + bg[2] * d2;
  if (tid == 0) s_max = 0;
  __syncthreads();
  {
    int m = last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(m, o); m = t > m ? t : m; }
    if (lane == 0) atomicMax(&s_max, m);
  }
  __syncthreads();
  const int max_last = s_max;   // instances at list positions >= max_last were reached by no pixel

  // zero records for the unreached tail of the list
  for (int k = max_last + tid; k < total; k += GH_BLOCK) {
    float4* r = (float4*)(inst_grad + (size_t)sorted_slot[range.x + k] * GH_REC);
    r[0] = make_float4(0, 0, 0, 0); r[1] = make_float4(0, 0, 0, 0); r[2] = make_float4(0, 0, 0, 0);
  }

  float T = T_final, last_alpha = 0.0f, lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f, ar0 = 0.0f, ar1 = 0.0f, ar2 = 0.0f;
  const int nchunks = (max_last + GH_BCHUNK - 1) / GH_BCHUNK;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int cbase = c * GH_BCHUNK;
    const int cnt = (max_last - cbase) < GH_BCHUNK ? (max_last - cbase) : GH_BCHUNK;
    if (tid < cnt) {
      const uint32_t gid = sorted_gid[range.x + cbase + tid];
      s_g0[tid] = g0[gid]; s_g1[tid] = g1[gid]; s_b[tid] = gb[gid];
      s_slot[tid] = sorted_slot[range.x + cbase + tid];
    }
    if (lane < GH_BCHUNK / 32) s_touched[wid][lane] = 0;
    __syncthreads();
    uint64_t t_lo = 0, t_hi = 0;                 // wave-uniform: which records of the chunk this wave wrote
    for (int j = cnt - 1; j >= 0; --j) {
      const int pos = cbase + j;                 // 0-based list position; pixel blended it iff pos < last && tests pass
      const float4 a = s_g0[j], b4 = s_g1[j];
      const float cb = s_b[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b4.x * dy * dy) - a.w * dx * dy;
      const float G = gh_exp(fminf(power, 0.0f));
      const float alpha = fminf(0.99f, b4.y * G);
      const bool contrib = (pos < last) && (power <= 0.0f) && (alpha >= 1.0f / 255.0f);
      if (!__any(contrib)) continue;             // wave-uniform skip: quadrant untouched by this Gaussian
      float r[9];
      if (contrib) {
        T = T / (1.0f - alpha);
        const float dchannel_dcolor = alpha * T;
        ar0 = last_alpha * lc0 + (1.0f - last_alpha) * ar0;
        ar1 = last_alpha * lc1 + (1.0f - last_alpha) * ar1;
        ar2 = last_alpha * lc2 + (1.0f - last_alpha) * ar2;
        lc0 = b4.z; lc1 = b4.w; lc2 = cb;
        float dL_dalpha = (b4.z - ar0) * d0 + (b4.w - ar1) * d1 + (cb - ar2) * d2;
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
        const float dL_dG = b4.y * dL_dalpha;    // straight-through the 0.99 clamp (App. A.4-2)
        const float gdx = G * dx, gdy = G * dy;
        r[0] = dL_dG * (-gdx * a.z - gdy * a.w);
        r[1] = dL_dG * (-gdy * b4.x - gdx * a.w);
        r[2] = -0.5f * gdx * dx * dL_dG;
        r[3] = -gdx * dy * dL_dG;
        r[4] = -0.5f * gdy * dy * dL_dG;
        r[5] = G * dL_dalpha;
        r[6] = dchannel_dcolor * d0; r[7] = dchannel_dcolor * d1; r[8] = dchannel_dcolor * d2;
      } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) r[q] = 0.0f;
      }
#pragma unroll
      for (int q = 0; q < 9; ++q) r[q] = gh_wave_sum_to63(r[q]);
      if (lane == 63) {
        float4* p = (float4*)&s_part[wid][j][0];
        p[0] = make_float4(r[0], r[1], r[2], r[3]);
        p[1] = make_float4(r[4], r[5], r[6], r[7]);
        s_part[wid][j][8] = r[8];
      }
      if (j < 64) t_lo |= 1ull << j; else t_hi |= 1ull << (j - 64);
    }
    if (lane == 0) {
      s_touched[wid][0] = (uint32_t)t_lo; s_touched[wid][1] = (uint32_t)(t_lo >> 32);
      s_touched[wid][2] = (uint32_t)t_hi; s_touched[wid][3] = (uint32_t)(t_hi >> 32);
    }
    __syncthreads();
    if (tid < cnt) {
      float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int w = 0; w < GH_BLOCK / GH_WAVE; ++w) {     // fixed wave order => reproducible sums
        if (s_touched[w][tid >> 5] & (1u << (tid & 31))) {
#pragma unroll
          for (int q = 0; q < 9; ++q) s[q] += s_part[w][tid][q];
        }
      }
      float4* rec = (float4*)(inst_grad + (size_t)s_slot[tid] * GH_REC);
      rec[0] = make_float4(s[0], s[1], s[2], s[3]);
      rec[1] = make_float4(s[4], s[5], s[6], s[7]);
      rec[2] = make_float4(s[8], 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
  }
}

void gh_launch_render_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const float* dL_dimage, char* ws,
                          const GhLayout& L, hipStream_t s) {
  hipLaunchKernelGGL(gh_render_bwd_kernel, dim3(g.NV * g.tiles), dim3(GH_BLOCK), 0, s, (const uint2*)(ws + L.ranges),
                     (const uint32_t*)(ws + L.sorted_gid), (const uint32_t*)(ws + L.vals_a), (const float4*)(ws + L.geom_g0),
                     (const float4*)(ws + L.geom_g1), (const float*)(ws + L.geom_b), in->cams, g.H, g.W, g.gx, g.tiles,
                     (const float*)(ws + L.final_T), (const uint32_t*)(ws + L.n_contrib), dL_dimage,
                     (float*)(ws + L.inst_grad));
}
