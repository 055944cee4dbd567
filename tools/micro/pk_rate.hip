// Issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950: N independent FMA chains per lane, wave-saturated.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void k(float* out, float a, float b, int iters) {
  v2f x0 = {a, a + 1}, x1 = {a + 2, a + 3}, x2 = {a + 4, a + 5}, x3 = {a + 6, a + 7};
  float s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3, s4 = a + 4, s5 = a + 5, s6 = a + 6, s7 = a + 7;
  const v2f vb = {b, b}, vc = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
    if (PK) {
      x0 = __builtin_elementwise_fma(x0, vb, vc); x1 = __builtin_elementwise_fma(x1, vb, vc);
      x2 = __builtin_elementwise_fma(x2, vb, vc); x3 = __builtin_elementwise_fma(x3, vb, vc);
    } else {
      s0 = fmaf(s0, b, 0.5f); s1 = fmaf(s1, b, 0.25f); s2 = fmaf(s2, b, 0.5f); s3 = fmaf(s3, b, 0.25f);
      s4 = fmaf(s4, b, 0.5f); s5 = fmaf(s5, b, 0.25f); s6 = fmaf(s6, b, 0.5f); s7 = fmaf(s7, b, 0.25f);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = PK ? x0.x + x0.y + x1.x + x1.y + x2.x + x2.y + x3.x + x3.y : s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int pk = 0; pk < 2; ++pk) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<true>, dim3(2048), dim3(256), 0, 0, d, 1.0f, 0.999f, iters);
      else hipLaunchKernelGGL(k<false>, dim3(2048), dim3(256), 0, 0, d, 1.0f, 0.999f, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fma = 2048.0 * 256 * 8 * iters;
      if (rep) printf("%s: %.3f ms, %.1f TFLOP/s (8 FMAs per lane per iteration as %d instructions)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, 2 * fma / ms / 1e9, pk ? 4 : 8);
    }
  }
  return 0;
}
