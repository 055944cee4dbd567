// gh_uv.hip — per-Gaussian bilinear lookup of the learnable UV maps and its scatter-add backward.
// Replaces F.grid_sample(align_corners=True, mode="bilinear", zeros padding) as used by query_triplane_texture
// (tgs/models/renderer_one_shot.py:420-446) at the call sites :489-492 (color_b 48x1024x2048, opacity_b 1x1024x2048).
// MI355X layout: the map is CHANNEL-LAST (Hm, Wm, C) so the C channels of a texel are contiguous (48 floats = 3
// cache lines) and lanes = channels: every gather is one contiguous segment per texel, and the backward's float
// atomics hit 4*C-byte contiguous runs (the access shape global atomics want, MI355X_MICROARCH.md).
#include "gh_internal.h"

__device__ __forceinline__ void gh_bilinear(float u, float v, int Hm, int Wm, int& x0, int& y0, float& wx1, float& wy1) {
  const float ix = ((u + 1.0f) * 0.5f) * (float)(Wm - 1);        // align_corners=True unnormalisation
  const float iy = ((v + 1.0f) * 0.5f) * (float)(Hm - 1);
  const float fx = floorf(ix), fy = floorf(iy);
  x0 = (int)fx; y0 = (int)fy;
  wx1 = ix - fx; wy1 = iy - fy;                                   // weight of the east / south neighbour
}

__global__ __launch_bounds__(GH_BLOCK) void gh_uv_sample_fwd_kernel(const float* __restrict__ map, const float* __restrict__ uv,
                                                                     float* __restrict__ out, int P, int C, int Hm, int Wm) {
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)P * C) return;
  const int i = (int)(idx / C), c = (int)(idx - (size_t)i * C);
  int x0, y0; float wx1, wy1;
  gh_bilinear(uv[2 * i], uv[2 * i + 1], Hm, Wm, x0, y0, wx1, wy1);
  const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const bool xa = x0 >= 0 && x0 < Wm, xb = x0 + 1 >= 0 && x0 + 1 < Wm, ya = y0 >= 0 && y0 < Hm, yb = y0 + 1 >= 0 && y0 + 1 < Hm;
  const size_t row0 = ((size_t)y0 * Wm + x0) * C + c, row1 = row0 + (size_t)Wm * C;
  float acc = 0.0f;                                               // zeros padding outside the map; nw, ne, sw, se order
  if (ya && xa) acc += map[row0] * (wx0 * wy0);
  if (ya && xb) acc += map[row0 + C] * (wx1 * wy0);
  if (yb && xa) acc += map[row1] * (wx0 * wy1);
  if (yb && xb) acc += map[row1 + C] * (wx1 * wy1);
  out[idx] = acc;
}

__global__ __launch_bounds__(GH_BLOCK) void gh_uv_sample_bwd_kernel(const float* __restrict__ uv, const float* __restrict__ dout,
                                                                     float* __restrict__ dmap, int P, int C, int Hm, int Wm) {
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)P * C) return;
  const int i = (int)(idx / C), c = (int)(idx - (size_t)i * C);
  int x0, y0; float wx1, wy1;
  gh_bilinear(uv[2 * i], uv[2 * i + 1], Hm, Wm, x0, y0, wx1, wy1);
  const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
  const bool xa = x0 >= 0 && x0 < Wm, xb = x0 + 1 >= 0 && x0 + 1 < Wm, ya = y0 >= 0 && y0 < Hm, yb = y0 + 1 >= 0 && y0 + 1 < Hm;
  const size_t row0 = ((size_t)y0 * Wm + x0) * C + c, row1 = row0 + (size_t)Wm * C;
  const float g = dout[idx];
  if (ya && xa) atomicAdd(&dmap[row0], g * (wx0 * wy0));
  if (ya && xb) atomicAdd(&dmap[row0 + C], g * (wx1 * wy0));
  if (yb && xa) atomicAdd(&dmap[row1], g * (wx0 * wy1));
  if (yb && xb) atomicAdd(&dmap[row1 + C], g * (wx1 * wy1));
}

static int gh_uv_check(const void* a, const void* b, const void* c, int P, int C, int Hm, int Wm) {
  if (P < 0 || C < 1 || Hm < 1 || Wm < 1) return GH_ERR_INVALID_ARG;
  if (P > 0 && (!a || !b || !c)) return GH_ERR_INVALID_ARG;
  return GH_OK;
}

extern "C" int gh_uv_sample_forward(const float* map, const float* uv, float* out, int P, int C, int Hm, int Wm, void* hip_stream) {
  int rc = gh_uv_check(map, uv, out, P, C, Hm, Wm);
  if (rc != GH_OK || P == 0) return rc;
  (void)hipGetLastError();
  const size_t n = (size_t)P * C;
  hipLaunchKernelGGL(gh_uv_sample_fwd_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0,
                     (hipStream_t)hip_stream, map, uv, out, P, C, Hm, Wm);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_uv_sample_backward(const float* uv, const float* dL_dout, float* dL_dmap, int P, int C, int Hm, int Wm,
                                     void* hip_stream) {
  int rc = gh_uv_check(uv, dL_dout, dL_dmap, P, C, Hm, Wm);
  if (rc != GH_OK || P == 0) return rc;
  (void)hipGetLastError();
  const size_t n = (size_t)P * C;
  hipLaunchKernelGGL(gh_uv_sample_bwd_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0,
                     (hipStream_t)hip_stream, uv, dL_dout, dL_dmap, P, C, Hm, Wm);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
