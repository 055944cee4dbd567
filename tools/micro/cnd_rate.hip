// v_cndmask_b32 issue cost on gfx950 in the forms the compiler emits (condition in VCC vs in an SGPR pair, constant vs
// per-iteration condition), against v_and_b32 / v_fma_f32 as yardsticks.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float* out, const float* in, float a, float b, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + 64 * i];
  const bool c0 = in[threadIdx.x] > 0.5f;                 // loop-invariant lane-varying condition -> SGPR pair
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) v[i] = c0 ? v[i] * b : v[i];                                    // mul + cndmask(sgpr pair)
      if (KIND == 1) v[i] = v[i] * b;                                                // mul only
      if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "s"(__ballot(c0)));
      if (KIND == 3) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(b));
      if (KIND == 4) { const bool c = v[i] > a; v[i] = c ? b : v[i]; }             // cmp + cndmask per element
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* d, float* in, int per) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 10000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, d, in, 1.0f, 0.999f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  const double winst = 2048.0 * 4 * 8 * iters * per;
  printf("%-34s %8.3f ms  %.2f cycles per wave instruction (x%d instr per element) at 2.4 GHz\n", name, ms, 1024.0 * 2.4e9 / (winst / (ms * 1e-3)), per);
}
int main() {
  float *d, *in; hipMalloc(&d, 256 * 2048 * 4); hipMalloc(&in, 4096 * 4); hipMemset(in, 0, 4096 * 4);
  run<1>("v_mul_f32", d, in, 1); run<0>("v_mul + v_cndmask(sgpr cond)", d, in, 2); run<2>("v_cndmask_e64 (sgpr pair)", d, in, 1);
  run<3>("v_cndmask_e32 (vcc)", d, in, 1); run<4>("v_cmp + v_cndmask per element", d, in, 2);
  return 0;
}
