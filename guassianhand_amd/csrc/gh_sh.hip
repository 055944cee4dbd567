// gh_sh.hip — spherical-harmonics colour stage (backward A.5-5/7) incl. the SH form of the
// attribute blend (tgs/models/renderer_one_shot.py:330-334, with its double multiply when color_b is given).
//
// Forward: one lane per (view, Gaussian), Gaussian-major, the fma chain over the coefficients IN ORDER in the lane
// (bit-identical to the sequential sum of the oracle); 192-byte coefficient rows as twelve 16-byte loads, or blended into LDS
// by the block for the pose batch. Backward: gh_sh_colour_bwd2_kernel (per (view, Gaussian), then per (Gaussian, coefficient));
// the first-generation kernel — 16 lanes per Gaussian = one DPP row, lane = coefficient, per-coefficient accumulators in
// registers across the view loop — serves calls with more than 256 views.
#include "gh_internal.h"

__device__ __forceinline__ float gh_row_sum(float v) {          // sum over the 16 lanes of the row, in every lane
  v += gh_dpp<0xB1>(v);           // quad_perm [1,0,3,2]
  v += gh_dpp<0x4E>(v);           // quad_perm [2,3,0,1]
  v += gh_dpp<0x141>(v);          // row_half_mirror
  v += gh_dpp<0x140>(v);          // row_mirror
  return v;
}

__device__ __forceinline__ float gh_pick16(const float* a, int k) {   // a[k] without a runtime-indexed register array
  float r = a[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) {
    r = (k == j) ? a[j] : r;
    asm("" : "+v"(r));          // keep the v_cndmask chain: without it the compiler spills the table to scratch and indexes it
  }
  return r;
}

// Blended coefficients of row i, e[3 k + channel] for k < nb (others 0): gh_blended_sh element by element; `wide` (M == 16 and
// 16-byte aligned arrays): twelve 16-byte loads per array.
__device__ __forceinline__ void gh_blended_row(const GhInputs& in, uint32_t flags, int M, int i, int nb, int wide, float* e) {
#pragma unroll
  for (int q = 0; q < 48; ++q) e[q] = 0.0f;
  if (wide) {
    const float4* sh4 = (const float4*)(in.shs + (size_t)i * 48);
    const float4* b4 = in.blend_color_b ? (const float4*)(in.blend_color_b + (size_t)i * 48) : nullptr;
    const float4* w4 = in.blend_color_w ? (const float4*)(in.blend_color_w + ((flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? (size_t)i * 48 : 0))
                                        : nullptr;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      if (4 * j < 3 * nb) {                        // nb is uniform over the launch (degree / M)
        const float4 sv = sh4[j];
        float x[4] = {sv.x, sv.y, sv.z, sv.w};
        if (w4) {
          const float4 wv = w4[j];
          const float w[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = x[q] * w[q];
          if (b4) {
            const float4 bv = b4[j];
            const float b[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { x[q] = x[q] * w[q]; x[q] = x[q] + b[q]; }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (4 * j + q < 3 * nb) e[4 * j + q] = x[q];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < nb) { e[3 * k] = gh_blended_sh(in, flags, M, i, k, 0); e[3 * k + 1] = gh_blended_sh(in, flags, M, i, k, 1); e[3 * k + 2] = gh_blended_sh(in, flags, M, i, k, 2); }
  }
}

// ------------------------------------------------------------------------------------------------
// The 256 rows [row0, row0 + 256) of blended coefficients (48 per row, M == 16) into s_rows[256][49], by the whole block with
// coalesced loads; wide: 16-byte pieces, 12 per thread, all loads in flight at once. Rows at or beyond n_rows are zero.
__device__ __forceinline__ void gh_stage_rows(const GhInputs& in, uint32_t flags, size_t row0, size_t n_rows, int wide,
                                              float* __restrict__ s_rows) {
    const size_t e0 = row0 * 48;
    const size_t e_end = n_rows * 48;
    const bool wpg = (flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
    if (wide) {                                    // 16-byte pieces: 12 per row, 12 per thread, all loads in flight at once
      const float4* sh4 = (const float4*)in.shs + e0 / 4;
      const float4* b4 = in.blend_color_b ? (const float4*)in.blend_color_b + e0 / 4 : nullptr;
      const float4* w4 = in.blend_color_w ? (const float4*)in.blend_color_w + (wpg ? e0 / 4 : 0) : nullptr;
      const int n4 = (int)((e_end - e0 < (size_t)GH_BLOCK * 48 ? e_end - e0 : (size_t)GH_BLOCK * 48) / 4);
      float4 sv[12], wv[12], bv[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int q4 = threadIdx.x + GH_BLOCK * j;
        const int c4 = q4 % 12;
        sv[j] = q4 < n4 ? sh4[q4] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (w4) wv[j] = wpg ? (q4 < n4 ? w4[q4] : sv[j]) : w4[c4];
        if (b4) bv[j] = q4 < n4 ? b4[q4] : sv[j];
      }
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int q4 = threadIdx.x + GH_BLOCK * j;
        const int row = q4 / 12, c4 = q4 - row * 12;
        float x[4] = {sv[j].x, sv[j].y, sv[j].z, sv[j].w};
        if (w4) {
          const float w[4] = {wv[j].x, wv[j].y, wv[j].z, wv[j].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] = x[q] * w[q];
          if (b4) {
            const float b[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { x[q] = x[q] * w[q]; x[q] = x[q] + b[q]; }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) s_rows[row * 49 + c4 * 4 + q] = x[q];
      }
    } else
#pragma unroll 8
    for (int idx = threadIdx.x; idx < GH_BLOCK * 48; idx += GH_BLOCK) {
      const size_t ge = e0 + (size_t)idx;
      const int row = idx / 48, el = idx - row * 48;
      float x = 0.0f;
      if (ge < e_end) {
        x = in.shs[ge];
        if (in.blend_color_w) {
          const float w = in.blend_color_w[wpg ? ge : (size_t)el];
          x = x * w;
          if (in.blend_color_b) { x = x * w; x = x + in.blend_color_b[ge]; }
        }
      }
      s_rows[row * 49 + el] = x;
    }
}

template <bool staged>                             // staged: pose batch (the 36 x 16 bytes in flight cost 144 registers: own kernel)
__global__ __launch_bounds__(GH_BLOCK) void gh_sh_colour_fwd_kernel(GhInputs in, int P, int NV, int N, int sh_degree, int M,
                                                                     uint32_t flags, float4* __restrict__ sh_rgb, float rdiv,
                                                                     int wide) {
  // One lane per (view, Gaussian): direction, basis, then the fma chain over the coefficients IN ORDER (bit-identical to the
  // oracle's sequential sum). Round 1 spread a (view, Gaussian) over 16 lanes (lane = coefficient, coalesced 192-byte rows, the
  // chain through row broadcasts): every lane evaluated the whole basis, 500 instructions per FOUR rows. Here a lane reads
  // its Gaussian's coefficients itself; with the Gaussian-major order below the views of a Gaussian sit in adjacent lanes,
  // so a wave's load touches 64 / n_views rows, each fetched once (pose batch: 64 rows per load, still 10x fewer instructions).
  extern __shared__ float s_rows[];                // staged: [256][49] blended coefficients of the block's 256 rows
  const int t = blockIdx.x * GH_BLOCK + threadIdx.x;
  if (staged) {
    // Pose batch: every lane has a row of its own, so per-lane row loads would touch 64 different 192-byte rows per
    // instruction. The block's 256 rows are contiguous: blend them with coalesced loads into LDS (row stride 49 words:
    // conflict-free when lane l then reads row l).
    gh_stage_rows(in, flags, (size_t)blockIdx.x * GH_BLOCK, (size_t)N, wide, s_rows);
    __syncthreads();
  }
  if (t >= N) return;
  int v, i, n;
  if (flags & GH_FLAG_PER_VIEW_GAUSSIANS) { v = rdiv > 0.0f ? (int)gh_div_small((uint32_t)t, (uint32_t)P, rdiv) : t / P; i = t; n = t; }
  else { i = rdiv > 0.0f ? (int)gh_div_small((uint32_t)t, (uint32_t)NV, rdiv) : t / NV; v = t - i * NV; n = v * P + i; }
  const float* cam = in.cams + (size_t)v * GH_CAM_FLOATS;
  float mx = in.means3D[3 * i], my = in.means3D[3 * i + 1], mz = in.means3D[3 * i + 2];
  if (in.blend_xyz_b) { mx = mx + in.blend_xyz_b[0]; my = my + in.blend_xyz_b[1]; mz = mz + in.blend_xyz_b[2]; }
  float dx = mx - cam[32], dy = my - cam[33], dz = mz - cam[34];
  const float len = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
  dx = dx / len; dy = dy / len; dz = dz / len;
  float Bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) Bv[j] = 0.0f;
  int nb = gh_sh_basis(sh_degree, dx, dy, dz, Bv);
  if (nb > M) nb = M;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;          // acc = fmaf(B_k, sh_k, acc) for k = 0 .. nb-1, in order
  if (staged) {
    const float* e = s_rows + threadIdx.x * 49;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < nb) { a0 = fmaf(Bv[k], e[3 * k], a0); a1 = fmaf(Bv[k], e[3 * k + 1], a1); a2 = fmaf(Bv[k], e[3 * k + 2], a2); }
    }
  } else if (wide) {                               // M == 16 and 16-byte aligned arrays (checked by the launcher)
    // 48 coefficients = 192 contiguous, 16-byte aligned bytes: twelve 16-byte loads per array; element e = 3 k + channel
    const float4* sh4 = (const float4*)(in.shs + (size_t)i * 48);
    const float4* b4 = in.blend_color_b ? (const float4*)(in.blend_color_b + (size_t)i * 48) : nullptr;
    const float4* w4 = in.blend_color_w ? (const float4*)(in.blend_color_w + ((flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) ? (size_t)i * 48 : 0))
                                        : nullptr;
    float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      if (4 * j < 3 * nb) {                        // nb is uniform over the launch (degree / M)
        const float4 sv = sh4[j];
        float e[4] = {sv.x, sv.y, sv.z, sv.w};
        if (w4) {                                  // the blend of gh_blended_sh, element by element (same operations)
          const float4 wv = w4[j];
          const float w[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = e[q] * w[q];
          if (b4) {
            const float4 bv = b4[j];
            const float b[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { e[q] = e[q] * w[q]; e[q] = e[q] + b[q]; }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int el = 4 * j + q;                // compile-time after unrolling
          if (el < 3 * nb) acc[el % 3] = fmaf(Bv[el / 3], e[q], acc[el % 3]);
        }
      }
    }
    a0 = acc[0]; a1 = acc[1]; a2 = acc[2];
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < nb) {
        a0 = fmaf(Bv[k], gh_blended_sh(in, flags, M, i, k, 0), a0);
        a1 = fmaf(Bv[k], gh_blended_sh(in, flags, M, i, k, 1), a1);
        a2 = fmaf(Bv[k], gh_blended_sh(in, flags, M, i, k, 2), a2);
      }
    }
  }
  a0 = a0 + 0.5f; a1 = a1 + 0.5f; a2 = a2 + 0.5f;
  unsigned cl = 0;
  if (a0 < 0.0f) { cl |= 1u; a0 = 0.0f; }
  if (a1 < 0.0f) { cl |= 2u; a1 = 0.0f; }
  if (a2 < 0.0f) { cl |= 4u; a2 = 0.0f; }
  sh_rgb[n] = make_float4(a0, a1, a2, __uint_as_float(cl));
}

// ------------------------------------------------------------------------------------------------
// d(basis_k)/d(x,y,z)
__device__ __forceinline__ void gh_sh_basis_grad(int deg, float x, float y, float z, float* dBx, float* dBy, float* dBz) {
#pragma unroll
  for (int k = 0; k < 16; ++k) dBx[k] = dBy[k] = dBz[k] = 0.0f;
  (void)deg;                    // all bands are evaluated; the caller masks coefficients k >= (deg+1)^2
  dBy[1] = -GH_SH_C1; dBz[2] = GH_SH_C1; dBx[3] = -GH_SH_C1;
  const float xx = x * x, yy = y * y, zz = z * z;
  dBx[4] = GH_SH_C2_0 * y;  dBy[4] = GH_SH_C2_0 * x;
  dBy[5] = GH_SH_C2_1 * z;  dBz[5] = GH_SH_C2_1 * y;
  dBx[6] = GH_SH_C2_2 * -2.0f * x; dBy[6] = GH_SH_C2_2 * -2.0f * y; dBz[6] = GH_SH_C2_2 * 4.0f * z;
  dBx[7] = GH_SH_C2_3 * z;  dBz[7] = GH_SH_C2_3 * x;
  dBx[8] = GH_SH_C2_4 * 2.0f * x; dBy[8] = GH_SH_C2_4 * -2.0f * y;
  dBx[9]  = GH_SH_C3_0 * 6.0f * x * y;  dBy[9] = GH_SH_C3_0 * (3.0f * xx - 3.0f * yy);
  dBx[10] = GH_SH_C3_1 * y * z; dBy[10] = GH_SH_C3_1 * x * z; dBz[10] = GH_SH_C3_1 * x * y;
  dBx[11] = GH_SH_C3_2 * -2.0f * x * y; dBy[11] = GH_SH_C3_2 * (4.0f * zz - xx - 3.0f * yy); dBz[11] = GH_SH_C3_2 * 8.0f * y * z;
  dBx[12] = GH_SH_C3_3 * -6.0f * x * z; dBy[12] = GH_SH_C3_3 * -6.0f * y * z; dBz[12] = GH_SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
  dBx[13] = GH_SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); dBy[13] = GH_SH_C3_4 * -2.0f * x * y; dBz[13] = GH_SH_C3_4 * 8.0f * x * z;
  dBx[14] = GH_SH_C3_5 * 2.0f * x * z; dBy[14] = GH_SH_C3_5 * -2.0f * y * z; dBz[14] = GH_SH_C3_5 * (xx - yy);
  dBx[15] = GH_SH_C3_6 * (3.0f * xx - 3.0f * yy); dBy[15] = GH_SH_C3_6 * -6.0f * x * y;
}

// One row (16 lanes) per Gaussian, lane = coefficient; loops the views with register accumulators.
__global__ __launch_bounds__(GH_BLOCK) void gh_sh_colour_bwd_kernel(
    GhInputs in, GhGrads gr, int P, int NV, int sh_degree, int M, uint32_t flags, const uint32_t* __restrict__ tiles_touched,
    const float4* __restrict__ sh_rgb, const float4* __restrict__ gsum, float4* __restrict__ dmean_sh,
    float* __restrict__ scratch) {
  __shared__ float s_cw[GH_BLOCK / 16][48];
  const int t = blockIdx.x * GH_BLOCK + threadIdx.x;
  const int i0 = t >> 4, k = t & 15;
  const bool per_view = (flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;      // pose batch: NV*P rows, row i belongs to view i / P
  const bool live = i0 < (per_view ? NV * P : P);
  const int i = live ? i0 : 0;
  const bool wpg = (flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
  const bool has_w = in.blend_color_w != nullptr, has_b = in.blend_color_b != nullptr;
  const bool coef = live && k < M;
  float raw[3] = {0, 0, 0}, wv[3] = {1, 1, 1}, shv[3] = {0, 0, 0};
  if (coef) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      raw[ch] = in.shs[((size_t)i * M + k) * 3 + ch];
      if (has_w) wv[ch] = in.blend_color_w[(wpg ? (size_t)i * 48 : 0) + k * 3 + ch];
      shv[ch] = gh_blended_sh(in, flags, M, i, k, ch);
    }
  }
  float mx = in.means3D[3 * i], my = in.means3D[3 * i + 1], mz = in.means3D[3 * i + 2];
  if (in.blend_xyz_b) { mx = mx + in.blend_xyz_b[0]; my = my + in.blend_xyz_b[1]; mz = mz + in.blend_xyz_b[2]; }
  float dsh[3] = {0, 0, 0}, dcb[3] = {0, 0, 0}, dcw[3] = {0, 0, 0};
  const int v_lo = per_view ? i / P : 0, v_hi = per_view ? v_lo + 1 : NV;
  for (int v = v_lo; v < v_hi; ++v) {
    const size_t n = per_view ? (size_t)i : (size_t)v * P + i;
    const bool vis = live && tiles_touched[n] != 0;                  // row-uniform
    float g[3] = {0, 0, 0};
    if (vis) {
      const float4 r1 = gsum[n * 3 + 1]; const float4 r2 = gsum[n * 3 + 2];
      const unsigned cl = __float_as_uint(sh_rgb[n].w);
      g[0] = (cl & 1u) ? 0.0f : r1.z; g[1] = (cl & 2u) ? 0.0f : r1.w; g[2] = (cl & 4u) ? 0.0f : r2.x;
    }
    const float* cam = in.cams + (size_t)v * GH_CAM_FLOATS;
    const float dx = mx - cam[32], dy = my - cam[33], dz = mz - cam[34];
    const float len = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    const float ux = dx / len, uy = dy / len, uz = dz / len;
    float Bv[16], dBx[16], dBy[16], dBz[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) Bv[j] = 0.0f;
    int nb = gh_sh_basis(sh_degree, ux, uy, uz, Bv);
    if (nb > M) nb = M;
    gh_sh_basis_grad(sh_degree, ux, uy, uz, dBx, dBy, dBz);
    const bool act = vis && k < nb;
    const float bk = act ? gh_pick16(Bv, k) : 0.0f;
    const float sg = shv[0] * g[0] + shv[1] * g[1] + shv[2] * g[2];      // sum_ch sh'_k[ch] * dL/drgb[ch]
    float dd0 = act ? gh_pick16(dBx, k) * sg : 0.0f, dd1 = act ? gh_pick16(dBy, k) * sg : 0.0f, dd2 = act ? gh_pick16(dBz, k) * sg : 0.0f;
    dd0 = gh_row_sum(dd0); dd1 = gh_row_sum(dd1); dd2 = gh_row_sum(dd2);
    if (live && k == 0) {
      const float dot = ux * dd0 + uy * dd1 + uz * dd2;                  // backward of d / |d|
      dmean_sh[n] = vis ? make_float4((dd0 - ux * dot) / len, (dd1 - uy * dot) / len, (dd2 - uz * dot) / len, 0.0f)
                        : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float gk = bk * g[ch];                                       // dL/d(blended coefficient)
      dcb[ch] += gk;
      if (has_w) {
        dsh[ch] += has_b ? gk * wv[ch] * wv[ch] : gk * wv[ch];
        dcw[ch] += has_b ? gk * 2.0f * raw[ch] * wv[ch] : gk * raw[ch];
      } else dsh[ch] += gk;
    }
  }
  if (coef) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (gr.dL_dshs) gr.dL_dshs[((size_t)i * M + k) * 3 + ch] = dsh[ch];
      if (has_b && gr.dL_dblend_color_b) gr.dL_dblend_color_b[(size_t)i * 48 + k * 3 + ch] = dcb[ch];
      if (has_w && wpg && gr.dL_dblend_color_w) gr.dL_dblend_color_w[(size_t)i * 48 + k * 3 + ch] = dcw[ch];
    }
  }
  if (has_w && !wpg && gr.dL_dblend_color_w) {         // global (48,) weights: fixed-order block partials
    const int row = threadIdx.x >> 4;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) s_cw[row][k * 3 + ch] = coef ? dcw[ch] : 0.0f;
    __syncthreads();
    if (threadIdx.x < 64) {
      float s = 0.0f;
      if (threadIdx.x < 48) for (int r = 0; r < GH_BLOCK / 16; ++r) s += s_cw[r][threadIdx.x];
      scratch[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
    }
  }
}

// Backward, two phases in one workgroup (n_views <= 256; the kernel above serves larger view counts):
//   1. one lane per (view, Gaussian) of the block's G = 256 / n_views Gaussians (Gaussian-major; pose batch: G = 256 rows with one
//      view each): masked dL/drgb, direction, basis and its gradient ONCE per (view, Gaussian) — the 16-lane form evaluates them
//      in every one of its 16 lanes —, the position gradient through the view direction (dmean_sh), and (basis, dL/drgb) left
//      in LDS;
//   2. one lane per (Gaussian, coefficient): the sums over the views of basis_k * dL/drgb in view order (registers), then the
//      blend's chain rule and coalesced stores of dL/dshs, dL/dcolor_b, dL/dcolor_w, as before.
template <bool STAGED>                                  // STAGED (pose batch, M == 16): the block's 256 rows blended into LDS first
__global__ __launch_bounds__(GH_BLOCK) void gh_sh_colour_bwd2_kernel(
    GhInputs in, GhGrads gr, int P, int NV, int sh_degree, int M, uint32_t flags, int G, int wide, float rdiv,
    const uint32_t* __restrict__ tiles_touched, const float4* __restrict__ sh_rgb, const float4* __restrict__ gsum,
    float4* __restrict__ dmean_sh, float* __restrict__ scratch) {
  extern __shared__ float s_rows[];
  // basis of pair p (0 where the pair contributes nothing; padded rows) and masked dL/drgb of pair p. STAGED: they take the place
  // of the staged rows once phase 1 has read those (the 50 KB of rows + 20 KB of these allowed two workgroups per CU, 50 KB three)
  __shared__ float s_B_own[STAGED ? 1 : GH_BLOCK][17];
  __shared__ float s_g_own[STAGED ? 1 : GH_BLOCK][3];
  float (*s_B)[17] = STAGED ? (float (*)[17])s_rows : s_B_own;
  float (*s_g)[3] = STAGED ? (float (*)[3])(s_rows + GH_BLOCK * 17) : s_g_own;
  __shared__ float s_cw[GH_BLOCK / 16][48];
  const bool per_view = (flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;
  const int nv = per_view ? 1 : NV;                     // views per row of the attribute arrays
  const int rows = per_view ? NV * P : P;
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * G;                      // first row of this block
  if (STAGED) {                                         // G == 256: one row per lane (see the forward)
    gh_stage_rows(in, flags, (size_t)row0, (size_t)rows, wide, s_rows);
    __syncthreads();
  }
  // ---- phase 1 ----
  {
    const int il = rdiv > 0.0f ? (int)gh_div_small((uint32_t)tid, (uint32_t)nv, rdiv) : tid / nv, vv = tid - il * nv;
    const int i = row0 + il;
    const bool live = il < G && i < rows;
    float Bv[16], g[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 16; ++j) Bv[j] = 0.0f;
    bool vis = false;
    if (live) {
      const int v = per_view ? i / P : vv;
      const size_t n = per_view ? (size_t)i : (size_t)v * P + i;
      vis = tiles_touched[n] != 0;
      if (vis) {
        const float4 r1 = gsum[n * 3 + 1]; const float4 r2 = gsum[n * 3 + 2];
        const unsigned cl = __float_as_uint(sh_rgb[n].w);
        g[0] = (cl & 1u) ? 0.0f : r1.z; g[1] = (cl & 2u) ? 0.0f : r1.w; g[2] = (cl & 4u) ? 0.0f : r2.x;
        float mx = in.means3D[3 * i], my = in.means3D[3 * i + 1], mz = in.means3D[3 * i + 2];
        if (in.blend_xyz_b) { mx = mx + in.blend_xyz_b[0]; my = my + in.blend_xyz_b[1]; mz = mz + in.blend_xyz_b[2]; }
        const float* cam = in.cams + (size_t)v * GH_CAM_FLOATS;
        const float dx = mx - cam[32], dy = my - cam[33], dz = mz - cam[34];
        const float len = sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
        const float ux = dx / len, uy = dy / len, uz = dz / len;
        int nb = gh_sh_basis(sh_degree, ux, uy, uz, Bv);
        if (nb > M) nb = M;
        float sg[16];                                     // sum_ch sh'_k[ch] * dL/drgb[ch]: the 48 coefficients die here,
        {                                                 // before the 48 basis-gradient values come alive
          float e[48];
          if (STAGED) {
#pragma unroll
            for (int q = 0; q < 48; ++q) e[q] = q < 3 * nb ? s_rows[tid * 49 + q] : 0.0f;
          } else gh_blended_row(in, flags, M, i, nb, wide, e);
#pragma unroll
          for (int k = 0; k < 16; ++k) sg[k] = k < nb ? e[3 * k] * g[0] + e[3 * k + 1] * g[1] + e[3 * k + 2] * g[2] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(sg[k]));
        float dBx[16], dBy[16], dBz[16];
        gh_sh_basis_grad(sh_degree, ux, uy, uz, dBx, dBy, dBz);
        float dd0 = 0.0f, dd1 = 0.0f, dd2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k < nb) { dd0 += dBx[k] * sg[k]; dd1 += dBy[k] * sg[k]; dd2 += dBz[k] * sg[k]; }
          else Bv[k] = 0.0f;
        }
        const float dot = ux * dd0 + uy * dd1 + uz * dd2;                                    // backward of d / |d|
        dmean_sh[n] = make_float4((dd0 - ux * dot) / len, (dd1 - uy * dot) / len, (dd2 - uz * dot) / len, 0.0f);
      } else {
        dmean_sh[n] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
    if (STAGED) __syncthreads();                          // every lane has taken its coefficients out of s_rows
#pragma unroll
    for (int k = 0; k < 16; ++k) s_B[tid][k] = vis ? Bv[k] : 0.0f;
    s_g[tid][0] = g[0]; s_g[tid][1] = g[1]; s_g[tid][2] = g[2];
  }
  __syncthreads();
  // ---- phase 2 ----
  const int k = tid & 15, grp = tid >> 4;
  const bool wpg = (flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
  const bool has_w = in.blend_color_w != nullptr, has_b = in.blend_color_b != nullptr;
  float cw[3] = {0.0f, 0.0f, 0.0f};                                         // this lane's part of the global (48,) weight gradient
  for (int il = grp; il < G; il += GH_BLOCK / 16) {
    const int i = row0 + il;
    if (i >= rows || k >= M) continue;
    float raw[3], wv[3] = {1.0f, 1.0f, 1.0f};
    {                                                                      // (k, channel) triples are 12 contiguous bytes
      const GhF3 r3 = *(const GhF3*)(in.shs + ((size_t)i * M + k) * 3);
      raw[0] = r3.x; raw[1] = r3.y; raw[2] = r3.z;
      if (has_w) {
        const GhF3 w3 = *(const GhF3*)(in.blend_color_w + (wpg ? (size_t)i * 48 : 0) + k * 3);
        wv[0] = w3.x; wv[1] = w3.y; wv[2] = w3.z;
      }
    }
    float dsh[3] = {0.0f, 0.0f, 0.0f}, dcb[3] = {0.0f, 0.0f, 0.0f}, dcw[3] = {0.0f, 0.0f, 0.0f};
    for (int vv = 0; vv < nv; ++vv) {                                        // views in order, as the sequential sum
      const int p = il * nv + vv;
      const float bk = s_B[p][k];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float gk = bk * s_g[p][ch];                                    // dL/d(blended coefficient)
        dcb[ch] += gk;
        if (has_w) {
          dsh[ch] += has_b ? gk * wv[ch] * wv[ch] : gk * wv[ch];
          dcw[ch] += has_b ? gk * 2.0f * raw[ch] * wv[ch] : gk * raw[ch];
        } else dsh[ch] += gk;
      }
    }
    if (gr.dL_dshs) *(GhF3*)(gr.dL_dshs + ((size_t)i * M + k) * 3) = GhF3{dsh[0], dsh[1], dsh[2]};
    if (has_b && gr.dL_dblend_color_b) *(GhF3*)(gr.dL_dblend_color_b + (size_t)i * 48 + k * 3) = GhF3{dcb[0], dcb[1], dcb[2]};
    if (has_w && wpg && gr.dL_dblend_color_w) *(GhF3*)(gr.dL_dblend_color_w + (size_t)i * 48 + k * 3) = GhF3{dcw[0], dcw[1], dcw[2]};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) cw[ch] += dcw[ch];
  }
  if (has_w && !wpg && gr.dL_dblend_color_w) {         // global (48,) weights: fixed-order block partials
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) s_cw[grp][k * 3 + ch] = cw[ch];
    __syncthreads();
    if (tid < 64) {
      float sum = 0.0f;
      if (tid < 48) for (int r = 0; r < GH_BLOCK / 16; ++r) sum += s_cw[r][tid];
      scratch[(size_t)tid * gridDim.x + blockIdx.x] = sum;
    }
  }
}

void gh_launch_sh_colour_fwd(const GhDims* d, const GhGrid& g, const GhInputs* in, char* ws, const GhLayout& L, hipStream_t s) {
  if (g.N == 0 || !in->shs) return;
  const size_t threads = (size_t)g.N;
  const int staged = ((d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) && d->M == 16) ? 1 : 0;       // rows of 48 coefficients, one per lane
  auto kern = staged ? gh_sh_colour_fwd_kernel<true> : gh_sh_colour_fwd_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)((threads + GH_BLOCK - 1) / GH_BLOCK)), dim3(GH_BLOCK),
                     staged ? GH_BLOCK * 49 * sizeof(float) : 0, s, *in, g.P,
                     g.NV, g.N, d->sh_degree, d->M, d->flags, (float4*)(ws + L.sh_rgb),
                     g.N < (1 << 24) && g.P > 0 ? 1.0f / (float)((d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) ? g.P : g.NV) : 0.0f,
                     (d->M == 16 && (((uintptr_t)in->shs | (uintptr_t)in->blend_color_w | (uintptr_t)in->blend_color_b) & 15) == 0) ? 1 : 0);
}

// returns the number of scratch blocks written (0 when the global colour-weight reduction is not needed)
int gh_launch_sh_colour_bwd(const GhDims* d, const GhGrid& g, const GhInputs* in, const GhGrads* gr, const char* wg, char* ws,
                            const GhLayout& L, hipStream_t s) {
  if (g.P == 0 || !in->shs) return 0;
  const bool per_view_rows = (d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) != 0;
  const bool wpg2 = (d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
  if (per_view_rows || g.NV <= GH_BLOCK) {               // two-phase kernel: one lane per (view, Gaussian), then per coefficient
    const int nv = per_view_rows ? 1 : g.NV, G = GH_BLOCK / nv;
    const int rows = per_view_rows ? g.N : g.P;
    const int nblk2 = (rows + G - 1) / G;
    const int wide = (d->M == 16 && (((uintptr_t)in->shs | (uintptr_t)in->blend_color_w | (uintptr_t)in->blend_color_b) & 15) == 0) ? 1 : 0;
    // (staging the block's coefficient rows in LDS as the forward does: no gain with shared Gaussians, and 1.67 -> 2.27 ms for
    // 32 poses x 98,562 rows, where the 50 KB more LDS per block halve the occupancy)
    const bool staged = per_view_rows && d->M == 16;
    auto kern2 = staged ? gh_sh_colour_bwd2_kernel<true> : gh_sh_colour_bwd2_kernel<false>;
    hipLaunchKernelGGL(kern2, dim3(nblk2), dim3(GH_BLOCK), staged ? GH_BLOCK * 49 * sizeof(float) : 0, s, *in, *gr,
                       g.P, g.NV, d->sh_degree, d->M, d->flags, G, wide, 1.0f / (float)nv, (const uint32_t*)(wg + L.tiles_touched), (const float4*)(ws + L.sh_rgb),
                       (const float4*)(ws + L.grad_sums), (float4*)(ws + L.dmean_sh), (float*)(ws + L.sh_scratch));
    return (in->blend_color_w && !wpg2 && gr->dL_dblend_color_w) ? nblk2 : 0;
  }
  const size_t threads = (size_t)((d->flags & GH_FLAG_PER_VIEW_GAUSSIANS) ? g.N : g.P) * 16;
  const int nblk = (int)((threads + GH_BLOCK - 1) / GH_BLOCK);
  hipLaunchKernelGGL(gh_sh_colour_bwd_kernel, dim3(nblk), dim3(GH_BLOCK), 0, s, *in, *gr, g.P, g.NV, d->sh_degree, d->M, d->flags,
                     (const uint32_t*)(wg + L.tiles_touched), (const float4*)(ws + L.sh_rgb), (const float4*)(ws + L.grad_sums),
                     (float4*)(ws + L.dmean_sh), (float*)(ws + L.sh_scratch));
  const bool wpg = (d->flags & GH_FLAG_BLEND_W_PER_GAUSSIAN) != 0;
  return (in->blend_color_w && !wpg && gr->dL_dblend_color_w) ? nblk : 0;
}
