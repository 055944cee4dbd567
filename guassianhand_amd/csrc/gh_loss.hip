// gh_loss.hip — image-loss consumer of the rasteriser output: L = mean|img - gt| (the L1 term of
// utils.py:282-294) and its gradient dL/dimg = sign(img - gt)/n in ONE pass over the images, instead of torch's
// sub / abs / mean / sign / mul chain (five passes). HBM streaming: 2 reads + 1 write per element.
#include "gh_internal.h"

__global__ __launch_bounds__(GH_BLOCK) void gh_l1_loss_kernel(const float4* __restrict__ img, const float4* __restrict__ gt, size_t n4,
                                                               const float* __restrict__ img_tail, const float* __restrict__ gt_tail,
                                                               int n_tail, float grad_scale, float4* __restrict__ dimg,
                                                               float* __restrict__ dimg_tail, float* __restrict__ partials,
                                                               const GhCounters* __restrict__ guard) {
  __shared__ float s_w[GH_BLOCK / GH_WAVE];
  float acc = 0.0f;
  if (guard && (guard->overflow & GH_COUNTER_ERROR_MASK)) grad_scale = 0.0f;       // the image is invalid (instance overflow): no gradient leaves here
  auto sgn = [](float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); };      // torch.sign: sign(0) = 0
  for (size_t i = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * GH_BLOCK) {
    const float4 a = img[i], b = gt[i];
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
    dimg[i] = make_float4(grad_scale * sgn(d0), grad_scale * sgn(d1), grad_scale * sgn(d2), grad_scale * sgn(d3));
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) {
    const float d = img_tail[threadIdx.x] - gt_tail[threadIdx.x];
    acc += fabsf(d);
    dimg_tail[threadIdx.x] = grad_scale * sgn(d);
  }
  acc = gh_wave_sum_to63(acc);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

// fixed-order sum of the block partials (one block), scaled: loss[0] = scale * sum
// guard (optional): when the forward that produced the images overflowed its instance capacity the loss is NaN
__global__ __launch_bounds__(GH_BLOCK) void gh_partials_sum_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out,
                                                                    const GhCounters* __restrict__ guard) {
  __shared__ float s_w[GH_BLOCK / GH_WAVE];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += GH_BLOCK) s += partials[i];
  s = gh_wave_sum_to63(s);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (guard && (guard->overflow & GH_COUNTER_ERROR_MASK)) ? __uint_as_float(0x7FC00000u) : scale * (((s_w[0] + s_w[1]) + s_w[2]) + s_w[3]);
}

// The fused image loss's partials (GhLayout.loss_partials: 4 per tile, 21.5 k at 8 views of 512x334): one workgroup of 1024 threads,
// 16-byte loads, every thread a fixed stride of the array, then the fixed-order block sum.
__global__ __launch_bounds__(1024) void gh_partials_sum_wide_kernel(const float4* __restrict__ partials, size_t n4, float scale,
                                                                      float* __restrict__ out, bool scale_behind) {
  __shared__ float s_w[16];
  float a = 0.0f, b = 0.0f, c = 0.0f, d = 0.0f;
  for (size_t i = threadIdx.x; i < n4; i += 1024) { const float4 v = partials[i]; a += v.x; b += v.y; c += v.z; d += v.w; }
  float s = gh_wave_sum_to63((a + b) + (c + d));
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += s_w[k];
    out[0] = (scale_behind ? scale * ((const float*)partials)[n4 * 4] : scale) * t;
  }
}

void gh_launch_partials_sum(const float* partials, size_t n, float scale, float* out, hipStream_t s, bool scale_behind) {
  hipLaunchKernelGGL(gh_partials_sum_wide_kernel, dim3(1), dim3(1024), 0, s, (const float4*)partials, n / 4, scale, out, scale_behind);
}

extern "C" int gh_l1_loss(const float* image, const float* target, size_t n, float* loss_out, float* dL_dimage, float* partials,
                          int n_partials, const GhCounters* guard, void* hip_stream) {
  if (n == 0 || n_partials < 1 || !image || !target || !loss_out || !dL_dimage || !partials) return GH_ERR_INVALID_ARG;
  if ((((uintptr_t)image | (uintptr_t)target | (uintptr_t)dL_dimage) & 15) != 0) return GH_ERR_INVALID_ARG;   // float4 access
  (void)hipGetLastError();
  const size_t n4 = n / 4;
  const int tail = (int)(n - n4 * 4);
  const float inv = (float)(1.0 / (double)n);
  hipStream_t s = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(gh_l1_loss_kernel, dim3((unsigned)n_partials), dim3(GH_BLOCK), 0, s, (const float4*)image, (const float4*)target, n4,
                     image + n4 * 4, target + n4 * 4, tail, inv, (float4*)dL_dimage, dL_dimage + n4 * 4, partials, guard);
  hipLaunchKernelGGL(gh_partials_sum_kernel, dim3(1), dim3(GH_BLOCK), 0, s, partials, n_partials, inv, loss_out, guard);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

// ---- one-shot fit loss ---------------------------------------------------------------------------------------------
// Image part of the reference loss (utils.py:180-252, :282-294 with infer_one_shot.py:497, :507-510) for a stack of
// views, straight on the rasteriser's own layouts:
//   L = scale * sum_v [ lambda_l1 * mean_{c,y,x} |bb * rgb - gt| + lambda_m * mean_{y,x} (clip(alpha, -0.001, 1) - m)^2 ]
// image (NV,3,H,W) and alpha (NV,H,W) as the rasteriser writes them, gt_rgb (NV,H,W,3) channel-last and gt_mask
// (NV,H,W) as the reference holds them, bbox (NV,H,W) optional (pixels with bbox == 0 have their colour zeroed).
// One thread per pixel: reads 4 + 4 floats, writes the 4 gradients; block partials -> gh_partials_sum_kernel.
__global__ __launch_bounds__(GH_BLOCK) void gh_fit_loss_kernel(const float* __restrict__ image, const float* __restrict__ alpha,
                                                                const float* __restrict__ gt_rgb, const float* __restrict__ gt_mask,
                                                                const float* __restrict__ bbox, int NV, int HW, float k_l1, float k_m,
                                                                float* __restrict__ dimage, float* __restrict__ dalpha,
                                                                float* __restrict__ partials, const GhCounters* __restrict__ guard,
                                                                float rHW) {
  __shared__ float s_w[GH_BLOCK / GH_WAVE];
  float acc = 0.0f;
  const size_t npix = (size_t)NV * HW;
  const float live = (guard && (guard->overflow & GH_COUNTER_ERROR_MASK)) ? 0.0f : 1.0f;   // instance overflow: invalid images, no gradient leaves here
  for (size_t i = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x; i < npix; i += (size_t)gridDim.x * GH_BLOCK) {
    const size_t v = rHW > 0.0f ? (size_t)gh_div_small((uint32_t)i, (uint32_t)HW, rHW) : i / HW, p = i - v * HW;   // (no 64-bit division per pixel)
    const bool in_box = bbox ? bbox[i] != 0.0f : true;
    const float* im = image + v * 3 * HW + p;
    float* di = dimage + v * 3 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = (in_box ? im[(size_t)c * HW] : 0.0f) - gt_rgb[i * 3 + c];
      acc += k_l1 * fabsf(d);
      di[(size_t)c * HW] = in_box ? live * k_l1 * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) : 0.0f;
    }
    const float a = alpha[i];
    const float ac = fminf(fmaxf(a, -0.001f), 1.0f);
    const float e = ac - gt_mask[i];
    acc += k_m * e * e;
    dalpha[i] = (a >= -0.001f && a <= 1.0f) ? live * 2.0f * k_m * e : 0.0f;      // torch.clamp passes the gradient on [min, max]
  }
  acc = gh_wave_sum_to63(acc);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

extern "C" int gh_fit_loss(const float* image, const float* alpha, const float* gt_rgb, const float* gt_mask, const float* bbox,
                           int n_views, int H, int W, float lambda_l1, float lambda_mask, float scale, float* loss_out,
                           float* dL_dimage, float* dL_dalpha, float* partials, int n_partials, const GhCounters* guard,
                           void* hip_stream) {
  if (n_views < 1 || H < 1 || W < 1 || n_partials < 1) return GH_ERR_INVALID_ARG;
  if (!image || !alpha || !gt_rgb || !gt_mask || !loss_out || !dL_dimage || !dL_dalpha || !partials) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const int HW = H * W;
  hipStream_t s = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(gh_fit_loss_kernel, dim3((unsigned)n_partials), dim3(GH_BLOCK), 0, s, image, alpha, gt_rgb, gt_mask, bbox,
                     n_views, HW, scale * lambda_l1 / (3.0f * (float)HW), scale * lambda_mask / (float)HW, dL_dimage, dL_dalpha, partials, guard,
                     (long long)n_views * HW < (1ll << 24) ? 1.0f / (float)HW : 0.0f);
  hipLaunchKernelGGL(gh_partials_sum_kernel, dim3(1), dim3(GH_BLOCK), 0, s, partials, n_partials, 1.0f, loss_out, guard);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
