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

// ---- active-texel form ---------------------------------------------------------------------------------------------
// The Gaussians' UV coordinates are constant during a one-shot fit, so only the <= 4P texels under their bilinear
// footprints ever receive an image gradient; every other texel of the zero-initialised maps has gradient
// 100*sign(0)/n = 0 and 2*0/n = 0 from the regularisers (infer_one_shot.py:514-518) and Adam leaves it at exactly 0.
// The fit loop therefore keeps only the active texels, compacted as (U, C) rows; slot[i][0..3] are the rows of the
// nw, ne, sw, se corners of Gaussian i (-1 = outside the map) and w[i][0..3] their bilinear weights.
__global__ __launch_bounds__(GH_BLOCK) void gh_uv_gather_fwd_kernel(const float* __restrict__ tex, const int32_t* __restrict__ slot,
                                                                     const float* __restrict__ w, float* __restrict__ out, int P, int C) {
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)P * C) return;
  const int i = (int)(idx / C), c = (int)(idx - (size_t)i * C);
  const int4 s4 = ((const int4*)slot)[i];
  const float4 w4 = ((const float4*)w)[i];
  float acc = 0.0f;                                               // same corner order as gh_uv_sample_fwd_kernel
  if (s4.x >= 0) acc += tex[(size_t)s4.x * C + c] * w4.x;
  if (s4.y >= 0) acc += tex[(size_t)s4.y * C + c] * w4.y;
  if (s4.z >= 0) acc += tex[(size_t)s4.z * C + c] * w4.z;
  if (s4.w >= 0) acc += tex[(size_t)s4.w * C + c] * w4.w;
  out[idx] = acc;
}

__global__ __launch_bounds__(GH_BLOCK) void gh_uv_gather_bwd_kernel(const int32_t* __restrict__ slot, const float* __restrict__ w,
                                                                     const float* __restrict__ dout, float* __restrict__ dtex, int P, int C) {
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)P * C) return;
  const int i = (int)(idx / C), c = (int)(idx - (size_t)i * C);
  const int4 s4 = ((const int4*)slot)[i];
  const float4 w4 = ((const float4*)w)[i];
  const float g = dout[idx];
  if (s4.x >= 0) atomicAdd(&dtex[(size_t)s4.x * C + c], g * w4.x);
  if (s4.y >= 0) atomicAdd(&dtex[(size_t)s4.y * C + c], g * w4.y);
  if (s4.z >= 0) atomicAdd(&dtex[(size_t)s4.z * C + c], g * w4.z);
  if (s4.w >= 0) atomicAdd(&dtex[(size_t)s4.w * C + c], g * w4.w);
}

// Deterministic form of the two scatter-adds above: the (Gaussian, corner) pairs that touch a texel are listed per texel
// once (CSR: row_ptr (U+1), pairs (nnz) = 4 * gaussian + corner, ascending within a texel), and the backward becomes a
// GATHER — one lane per (texel, channel) adds its contributions in list order. No atomics: bitwise reproducible, like every
// other kernel on the fit path. Lanes = channels, so the C floats of a contributing Gaussian's gradient row are read as
// one contiguous segment.
__global__ __launch_bounds__(GH_BLOCK) void gh_uv_scatter_sorted_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ pairs,
                                                                         const float* __restrict__ w, const float* __restrict__ dout,
                                                                         float* __restrict__ dtex, int U, int C) {
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)U * C) return;
  const int u = (int)(idx / C), c = (int)(idx - (size_t)u * C);
  const int j0 = row_ptr[u], j1 = row_ptr[u + 1];
  float acc = 0.0f;
  for (int j = j0; j < j1; ++j) {
    const int e = pairs[j];
    acc += dout[(size_t)(e >> 2) * C + c] * w[e];
  }
  dtex[idx] += acc;
}

// The fit looks TWO maps up at the same UVs every step (colour bias, opacity bias: renderer_one_shot.py:489-492): one launch
// for both, lane = channel of the concatenated (Ca + Cb) row — the same per-lane arithmetic as the single-map kernels, so the
// values are bit-identical to two separate launches.
__global__ __launch_bounds__(GH_BLOCK) void gh_uv_gather_fwd2_kernel(const float* __restrict__ tex_a, int Ca, const float* __restrict__ tex_b,
                                                                      int Cb, const int32_t* __restrict__ slot, const float* __restrict__ w,
                                                                      float* __restrict__ out_a, float* __restrict__ out_b, int P) {
  const int Ct = Ca + Cb;
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)P * Ct) return;
  const int i = (int)(idx / Ct), ct = (int)(idx - (size_t)i * Ct);
  const bool second = ct >= Ca;
  const float* tex = second ? tex_b : tex_a;
  const int C = second ? Cb : Ca, c = second ? ct - Ca : ct;
  const int4 s4 = ((const int4*)slot)[i];
  const float4 w4 = ((const float4*)w)[i];
  float acc = 0.0f;
  if (s4.x >= 0) acc += tex[(size_t)s4.x * C + c] * w4.x;
  if (s4.y >= 0) acc += tex[(size_t)s4.y * C + c] * w4.y;
  if (s4.z >= 0) acc += tex[(size_t)s4.z * C + c] * w4.z;
  if (s4.w >= 0) acc += tex[(size_t)s4.w * C + c] * w4.w;
  (second ? out_b : out_a)[(size_t)i * C + c] = acc;
}

__global__ __launch_bounds__(GH_BLOCK) void gh_uv_scatter_sorted2_kernel(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ pairs,
                                                                          const float* __restrict__ w, const float* __restrict__ dout_a, int Ca,
                                                                          float* __restrict__ dtex_a, const float* __restrict__ dout_b, int Cb,
                                                                          float* __restrict__ dtex_b, int U) {
  const int Ct = Ca + Cb;
  const size_t idx = (size_t)blockIdx.x * GH_BLOCK + threadIdx.x;
  if (idx >= (size_t)U * Ct) return;
  const int u = (int)(idx / Ct), ct = (int)(idx - (size_t)u * Ct);
  const bool second = ct >= Ca;
  const float* dout = second ? dout_b : dout_a;
  const int C = second ? Cb : Ca, c = second ? ct - Ca : ct;
  const int j0 = row_ptr[u], j1 = row_ptr[u + 1];
  float acc = 0.0f;
  for (int j = j0; j < j1; ++j) {
    const int e = pairs[j];
    acc += dout[(size_t)(e >> 2) * C + c] * w[e];
  }
  (second ? dtex_b : dtex_a)[(size_t)u * C + c] += acc;
}

// One pass over a parameter array: regulariser value (sum|p|, sum p^2 of the PRE-update values, block partials),
// regulariser gradient (l1*sign(p) + l2*2p) added to the accumulated image gradient, Adam update (torch.optim.Adam
// semantics, no amsgrad / weight decay: infer_one_shot.py:345), and the gradient buffer is cleared for the next step.
// beta^t for an integer step count by repeated squaring in double (a dozen multiplies; the library pow() in double is ~10 us of
// dependent instructions and every block waited for it: 13 us per launch whatever the tensor's size). Agrees with pow() to
// double rounding, i.e. the float factors derived from it are the same.
__device__ __forceinline__ double gh_powi(double b, int t) {
  double r = 1.0;
  for (unsigned e = (unsigned)t; e; e >>= 1) { if (e & 1u) r *= b; b *= b; }
  return r;
}

struct GhAdamOne {
  float* p; float* g; float* m; float* v; size_t n; float l1, l2; float* partials; int32_t* step_state; int nblk;
};

// The body of one block (index `b` of `nblk`) of the update of ONE tensor. Main loop over float4s (the four arrays stream at
// 16 bytes per lane; the scalar form ran at 1.7 TB/s), tail elements by block 0; also when an array is not 16-byte aligned.
__device__ __forceinline__ void gh_adam_block(const GhAdamOne& T, int b, int host_step, float lr, float beta1, float beta2, float eps,
                                              const GhCounters* __restrict__ guard) {
  __shared__ float s_a[GH_BLOCK / GH_WAVE], s_b[GH_BLOCK / GH_WAVE];
  __shared__ float s_bc[2];
  // Device-side guard: a step whose render overflowed its instance capacity must not touch the parameters. The step count
  // of the bias correction then lives on the device too: step_state[0] = steps applied so far, read by every block when it
  // starts; the block that FINISHES last (a ticket in step_state[1]) writes the new count — every other block has read the old
  // one by then. No host value changes from launch to launch, so a captured step replays correctly.
  const bool skip = guard && (guard->overflow & GH_COUNTER_ERROR_MASK) != 0u;
  int t = host_step;
  if (T.step_state) t = __hip_atomic_load(&T.step_state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  // bias corrections in double, as torch.optim.Adam's Python arithmetic — by ONE thread per block
  if (threadIdx.x == 0) {
    const double bc1 = 1.0 - gh_powi((double)beta1, t), bc2 = 1.0 - gh_powi((double)beta2, t);
    s_bc[0] = (float)((double)lr / bc1); s_bc[1] = (float)(1.0 / sqrt(bc2));
  }
  __syncthreads();
  const float lr_over_bc1 = s_bc[0], inv_sqrt_bc2 = s_bc[1];
  float sa = 0.0f, sb = 0.0f;
  auto one = [&](float pv, float gin, float& mv, float& vv) -> float {       // returns the new parameter value
    sa += fabsf(pv); sb += pv * pv;
    const float sgn = pv > 0.0f ? 1.0f : (pv < 0.0f ? -1.0f : 0.0f);
    const float gv = gin + T.l1 * sgn + T.l2 * 2.0f * pv;
    mv = beta1 * mv + (1.0f - beta1) * gv;
    vv = beta2 * vv + (1.0f - beta2) * gv * gv;
    return pv - lr_over_bc1 * (mv / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  };
  const bool wide = ((((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0);
  const size_t n4 = wide ? T.n / 4 : 0;
  float4* p4 = (float4*)T.p; float4* g4 = (float4*)T.g; float4* m4 = (float4*)T.m; float4* v4 = (float4*)T.v;
  for (size_t i = (size_t)b * GH_BLOCK + threadIdx.x; i < n4; i += (size_t)T.nblk * GH_BLOCK) {
    float4 pv = p4[i];
    if (skip) { sa += (fabsf(pv.x) + fabsf(pv.y)) + (fabsf(pv.z) + fabsf(pv.w)); sb += (pv.x * pv.x + pv.y * pv.y) + (pv.z * pv.z + pv.w * pv.w);
                g4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); continue; }
    const float4 gv = g4[i];
    float4 mv = m4[i], vv = v4[i];
    pv.x = one(pv.x, gv.x, mv.x, vv.x); pv.y = one(pv.y, gv.y, mv.y, vv.y);
    pv.z = one(pv.z, gv.z, mv.z, vv.z); pv.w = one(pv.w, gv.w, mv.w, vv.w);
    p4[i] = pv; m4[i] = mv; v4[i] = vv; g4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  // tail (n % 4 elements), or everything when the arrays are not 16-byte aligned: scalar, spread over the blocks
  for (size_t idx = n4 * 4 + (size_t)b * GH_BLOCK + threadIdx.x; idx < T.n; idx += (size_t)T.nblk * GH_BLOCK) {
    const float pv = T.p[idx];
    if (skip) { sa += fabsf(pv); sb += pv * pv; T.g[idx] = 0.0f; continue; }
    float mv = T.m[idx], vv = T.v[idx];
    T.p[idx] = one(pv, T.g[idx], mv, vv);
    T.m[idx] = mv; T.v[idx] = vv; T.g[idx] = 0.0f;
  }
  sa = gh_wave_sum_to63(sa); sb = gh_wave_sum_to63(sb);
  if ((threadIdx.x & 63) == 63) { s_a[threadIdx.x >> 6] = sa; s_b[threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0 && T.partials) {
    T.partials[2 * b] = ((s_a[0] + s_a[1]) + s_a[2]) + s_a[3];
    T.partials[2 * b + 1] = ((s_b[0] + s_b[1]) + s_b[2]) + s_b[3];
  }
  if (threadIdx.x == 0 && T.step_state) {
    // Relaxed agent-scope atomics only (an agent-scope release / acquire writes back / invalidates the whole L2: with 1,024
    // blocks that was most of this kernel's 25 us). Nothing but the two words themselves is handed over: every block has
    // consumed its own read of step_state[0] (its value feeds the update above) before it takes a ticket, and the new count
    // is read by the NEXT launch.
    const int ticket = __hip_atomic_fetch_add(&T.step_state[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == T.nblk - 1) {                          // the last block out: nobody is left to read the old count
      __hip_atomic_store(&T.step_state[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&T.step_state[0], skip ? t - 1 : t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(GH_BLOCK) void gh_adam_reg_kernel(GhAdamOne T, int host_step, float lr, float beta1, float beta2, float eps,
                                                                const GhCounters* __restrict__ guard) {
  gh_adam_block(T, (int)blockIdx.x, host_step, lr, beta1, beta2, eps, guard);
}

// Up to four tensors in one launch (blockIdx.y = tensor; blocks beyond a tensor's own count exit): the fit steps color_w,
// color_b and opacity_b together.
struct GhAdamFour { GhAdamOne t[4]; };
__global__ __launch_bounds__(GH_BLOCK) void gh_adam_reg_group_kernel(GhAdamFour G, int host_step, float lr, float beta1, float beta2, float eps,
                                                                      const GhCounters* __restrict__ guard) {
  const GhAdamOne& T = G.t[blockIdx.y];
  if ((int)blockIdx.x >= T.nblk) return;
  gh_adam_block(T, (int)blockIdx.x, host_step, lr, beta1, beta2, eps, guard);
}

// Loss assembly of the fit step (infer_one_shot.py:514-519): out[0] = base[0] + ka * sum_i pa[i * 2 + ca] + kb * sum_i pb[i * 2 + cb],
// out[1] = the regulariser part alone — the two sums over gh_adam_reg_step's block partials in fixed order, one block. Replaces
// two torch reductions and five scalar elementwise kernels per step.
__global__ __launch_bounds__(GH_BLOCK) void gh_reg_total_kernel(const float* __restrict__ pa, int na, int ca, float ka,
                                                                 const float* __restrict__ pb, int nb, int cb, float kb,
                                                                 const float* __restrict__ base, float* __restrict__ out) {
  __shared__ float s_a[GH_BLOCK / GH_WAVE], s_b[GH_BLOCK / GH_WAVE];
  float a = 0.0f, b = 0.0f;
  for (int i = threadIdx.x; i < na; i += GH_BLOCK) a += pa[2 * i + ca];
  for (int i = threadIdx.x; i < nb; i += GH_BLOCK) b += pb[2 * i + cb];
  a = gh_wave_sum_to63(a); b = gh_wave_sum_to63(b);
  if ((threadIdx.x & 63) == 63) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float reg = ka * (((s_a[0] + s_a[1]) + s_a[2]) + s_a[3]) + kb * (((s_b[0] + s_b[1]) + s_b[2]) + s_b[3]);
    out[1] = reg;
    out[0] = (base ? base[0] : 0.0f) + reg;
  }
}

extern "C" int gh_reg_total(const float* partials_a, int n_a, int col_a, float k_a, const float* partials_b, int n_b, int col_b,
                            float k_b, const float* base, float* out2, void* hip_stream) {
  if (n_a < 0 || n_b < 0 || (col_a | col_b) < 0 || col_a > 1 || col_b > 1 || !out2) return GH_ERR_INVALID_ARG;
  if ((n_a > 0 && !partials_a) || (n_b > 0 && !partials_b)) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gh_reg_total_kernel, dim3(1), dim3(GH_BLOCK), 0, (hipStream_t)hip_stream, partials_a, n_a, col_a, k_a,
                     partials_b, n_b, col_b, k_b, base, out2);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_uv_gather_forward(const float* texels, const int32_t* slot, const float* w, float* out, int P, int C, void* hip_stream) {
  if (P < 0 || C < 1) return GH_ERR_INVALID_ARG;
  if (P == 0) return GH_OK;
  if (!texels || !slot || !w || !out) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const size_t n = (size_t)P * C;
  hipLaunchKernelGGL(gh_uv_gather_fwd_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0,
                     (hipStream_t)hip_stream, texels, slot, w, out, P, C);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_uv_gather_backward(const int32_t* slot, const float* w, const float* dL_dout, float* dL_dtexels, int P, int C,
                                     void* hip_stream) {
  if (P < 0 || C < 1) return GH_ERR_INVALID_ARG;
  if (P == 0) return GH_OK;
  if (!slot || !w || !dL_dout || !dL_dtexels) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const size_t n = (size_t)P * C;
  hipLaunchKernelGGL(gh_uv_gather_bwd_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0,
                     (hipStream_t)hip_stream, slot, w, dL_dout, dL_dtexels, P, C);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_uv_scatter_sorted(const int32_t* row_ptr, const int32_t* pairs, const float* w, const float* dL_dout,
                                    float* dL_dtexels, int U, int C, void* hip_stream) {
  if (U < 0 || C < 1) return GH_ERR_INVALID_ARG;
  if (U == 0) return GH_OK;
  if (!row_ptr || !pairs || !w || !dL_dout || !dL_dtexels) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const size_t n = (size_t)U * C;
  hipLaunchKernelGGL(gh_uv_scatter_sorted_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0,
                     (hipStream_t)hip_stream, row_ptr, pairs, w, dL_dout, dL_dtexels, U, C);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_adam_reg_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, int step, float lr,
                                float beta1, float beta2, float eps, float reg_l1, float reg_l2, float* partials, int n_partials,
                                const GhCounters* guard, int32_t* step_state, void* hip_stream) {
  if (step < 1 || n_partials < 1) return GH_ERR_INVALID_ARG;
  if (n == 0) return GH_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !partials) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const GhAdamOne T{param, grad, exp_avg, exp_avg_sq, n, reg_l1, reg_l2, partials, step_state, n_partials};
  hipLaunchKernelGGL(gh_adam_reg_kernel, dim3((unsigned)n_partials), dim3(GH_BLOCK), 0, (hipStream_t)hip_stream, T, step, lr, beta1, beta2,
                     eps, guard);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_adam_reg_step_group(const GhAdamTensor* tensors, int n_tensors, int step, float lr, float beta1, float beta2, float eps,
                                      const GhCounters* guard, void* hip_stream) {
  if (step < 1 || n_tensors < 1 || n_tensors > 4 || !tensors) return GH_ERR_INVALID_ARG;
  GhAdamFour G;
  int max_blk = 0;
  for (int k = 0; k < 4; ++k) {
    if (k >= n_tensors) { G.t[k] = GhAdamOne{nullptr, nullptr, nullptr, nullptr, 0, 0.0f, 0.0f, nullptr, nullptr, 0}; continue; }
    const GhAdamTensor& a = tensors[k];
    if (a.n_partials < 1 || (a.n > 0 && (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq || !a.partials))) return GH_ERR_INVALID_ARG;
    G.t[k] = GhAdamOne{a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.n, a.reg_l1, a.reg_l2, a.partials, a.step_state, a.n == 0 ? 0 : a.n_partials};
    if (G.t[k].nblk > max_blk) max_blk = G.t[k].nblk;
  }
  if (max_blk == 0) return GH_OK;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gh_adam_reg_group_kernel, dim3((unsigned)max_blk, (unsigned)n_tensors), dim3(GH_BLOCK), 0, (hipStream_t)hip_stream, G, step,
                     lr, beta1, beta2, eps, guard);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_uv_gather_forward2(const float* texels_a, int Ca, const float* texels_b, int Cb, const int32_t* slot, const float* w,
                                     float* out_a, float* out_b, int P, void* hip_stream) {
  if (P < 0 || Ca < 1 || Cb < 1) return GH_ERR_INVALID_ARG;
  if (P == 0) return GH_OK;
  if (!texels_a || !texels_b || !slot || !w || !out_a || !out_b) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const size_t n = (size_t)P * (Ca + Cb);
  hipLaunchKernelGGL(gh_uv_gather_fwd2_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0, (hipStream_t)hip_stream,
                     texels_a, Ca, texels_b, Cb, slot, w, out_a, out_b, P);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}

extern "C" int gh_uv_scatter_sorted2(const int32_t* row_ptr, const int32_t* pairs, const float* w, const float* dL_dout_a, int Ca,
                                     float* dL_dtexels_a, const float* dL_dout_b, int Cb, float* dL_dtexels_b, int U, void* hip_stream) {
  if (U < 0 || Ca < 1 || Cb < 1) return GH_ERR_INVALID_ARG;
  if (U == 0) return GH_OK;
  if (!row_ptr || !pairs || !w || !dL_dout_a || !dL_dtexels_a || !dL_dout_b || !dL_dtexels_b) return GH_ERR_INVALID_ARG;
  (void)hipGetLastError();
  const size_t n = (size_t)U * (Ca + Cb);
  hipLaunchKernelGGL(gh_uv_scatter_sorted2_kernel, dim3((unsigned)((n + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK), 0, (hipStream_t)hip_stream,
                     row_ptr, pairs, w, dL_dout_a, Ca, dL_dtexels_a, dL_dout_b, Cb, dL_dtexels_b, U);
  return hipGetLastError() == hipSuccess ? GH_OK : GH_ERR_LAUNCH;
}
