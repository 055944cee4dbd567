// gh_loss.hip — image-loss consumer of the rasteriser output: L = mean|img - gt| (the L1 term of
// utils.py:282-294) and its gradient dL/dimg = sign(img - gt)/n in ONE pass over the images, instead of torch's
// sub / abs / mean / sign / mul chain (five passes). HBM streaming: 2 reads + 1 write per element.
#include "gh_internal.h"

__global__ __launch_bounds__(GH_BLOCK) void gh_l1_loss_kernel(const float4* __restrict__ img, const float4* __restrict__ gt, size_t n4,
                                                               const float* __restrict__ img_tail, const float* __restrict__ gt_tail,
                                                               int n_tail, float grad_scale, float4* __restrict__ dimg,
                                                               float* __restrict__ dimg_tail, float* __restrict__ partials) {
  __shared__ float s_w[GH_BLOCK / GH_WAVE];
  float acc = 0.0f;
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
__global__ __launch_bounds__(GH_BLOCK) void gh_partials_sum_kernel(const float* __restrict__ partials, int n, float scale, float* __restrict__ out) {
  __shared__ float s_w[GH_BLOCK / GH_WAVE];
  float s = 0.0f;
  for (int i = threadIdx.x; i < n; i += GH_BLOCK) s += partials[i];
  s = gh_wave_sum_to63(s);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = scale * (((s_w[0] + s_w[1]) + s_w[2]) + s_w[3]);
}

extern "C" int gh_l1_loss(const float* image, const float* target, size_t n, float* loss_out, float* dL_dimage, float* partials,
                          int n_partials, void* hip_stream) {
  if (n == 0 || n_partials < 1 || !image || !target || !loss_out || !dL_dimage || !partials) return GH_ERR_INVALID_ARG;
  if ((((uintptr_t)image | (uintptr_t)target | (uintptr_t)dL_dimage) & 15) != 0) return GH_ERR_INVALID_ARG;   // float4 access
  (void)hipGetLastError();
  const size_t n4 = n / 4;
  const int tail = (int)(n - n4 * 4);
  const float inv = (float)(1.0 / (double)n);
  hipStream_t s = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(gh_l1_loss_kernel, dim3((unsigned)n_partials), dim3(GH_BLOCK), 0, s, (const float4*)image, (const float4*)target, n4,
                     image + n4 * 4, target + n4 * 4, tail, inv, (float4*)dL_dimage, dL_dimage + n4 * 4, partials);
  hipLaunchKernelGGL(gh_partials_sum_kernel, dim3(1), dim3(GH_BLOCK), 0, s, partials, n_partials, inv, loss_out);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
