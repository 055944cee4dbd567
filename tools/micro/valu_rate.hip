// Relative issue cost of the VALU instruction kinds the render kernels use (gfx950): 8 independent chains per lane,
// 2048x256 threads; reports giga-instructions/s per kind (wave64 instructions x 64 lanes).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ void k(float* out, float a, float b, int iters) {
  float v[8]; int m[8];
  for (int i = 0; i < 8; ++i) { v[i] = a + i + threadIdx.x; m[i] = threadIdx.x + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
      if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
      if (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      if (KIND == 6) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b));
      if (KIND == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(m[i]) : "v"(m[(i + 1) & 7]));
      if (KIND == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(b) : "vcc");
      if (KIND == 9) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 10) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m[i]));
      if (KIND == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 12) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(m[i]) : "v"(m[(i + 1) & 7]));
      if (KIND == 13) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(m[i]) : "v"(v[i]));
      if (KIND == 14) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v[i]) : "v"(m[i]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + m[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = KIND == 14 ? 2000 : 10000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, d, 1.0f, 0.999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double winst = 2048.0 * 4 * 8 * iters;       // wave instructions
  printf("%-16s %8.3f ms  %7.2f G wave-instr/s  -> %.2f cycles per wave instruction per SIMD at 2.4 GHz\n", name, ms, winst / ms / 1e6,
         1024.0 * 2.4e9 / (winst / (ms * 1e-3)));
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  run<0>("v_fma_f32", d); run<1>("v_mul_f32", d); run<2>("v_add_f32", d); run<3>("v_cndmask_b32", d); run<4>("v_min_f32", d);
  run<5>("v_mov_b32_dpp", d); run<6>("v_add_f32_dpp", d); run<7>("v_and_b32", d); run<8>("v_cmp_gt_f32", d); run<9>("v_rndne_f32", d);
  run<10>("v_ldexp_f32", d); run<11>("v_rcp_f32", d); run<12>("v_bfe_u32", d); run<13>("v_cvt_i32_f32", d); run<14>("ds_bpermute_b32", d);
  return 0;
}
