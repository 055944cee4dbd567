// Issue cost of packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against the plain forms, inline asm so that the
// compiler cannot re-pack anything: 8 independent chains per lane, 2048 x 256 threads (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float* out, float a, float b, int iters) {
  float s[8]; v2f x[8];
  for (int i = 0; i < 8; ++i) { s[i] = a + i + threadIdx.x; x[i] = v2f{a + i, a - i}; }
  v2f vb = {b, b}, vc = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b), "v"(a));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(vb), "v"(vc));
      if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(vb));
      if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(vb));
      if (KIND == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));
    }
  }
  float r = 0; for (int i = 0; i < 8; ++i) r += s[i] + x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> void run(const char* name, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, d, 1.0f, 0.999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  const double winst = 2048.0 * 4 * 8 * iters;
  printf("%-14s %8.3f ms -> %.2f cycles per wave instruction per SIMD at 2.4 GHz\n", name, ms, 1024.0 * 2.4e9 / (winst / (ms * 1e-3)));
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_pk_mul_f32", d); run<3>("v_pk_add_f32", d); run<4>("v_mul_f32", d);
  return 0;
}
