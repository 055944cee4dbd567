// LDS access costs on gfx950 relevant to fetching a list entry's record for the four depth slots of a wave:
// ds_bpermute_b32 (crossbar, 4 B per lane) vs ds_read_b32 / ds_read_b128 from LDS memory with only 4 distinct addresses
// per wave (lanes of the same slot read the same record: broadcast).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float* out, int iters) {
  __shared__ float4 s_rec[256];
  s_rec[threadIdx.x] = make_float4(threadIdx.x, 1, 2, 3);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float acc = 0; int addr = (lane & 3) * 16 + (threadIdx.x >> 6) * 1024;      // 4 distinct records per wave
  int baddr = ((lane * 7) & 63) << 2;
  float v = (float)lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) { asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(baddr)); }
      if (KIND == 1) { float x; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr)); acc += x; }
      if (KIND == 2) { float4 x; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr)); acc += x.x + x.w; }
      if (KIND == 3) { float4 x; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(lane * 16)); acc += x.x + x.w; }   // all distinct
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v;
}
template <int KIND> void run(const char* name, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000; float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(2048), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  const double winst = 2048.0 * 4 * 8 * iters;
  printf("%-44s %8.3f ms  %.1f cycles per wave instruction per SIMD share at 2.4 GHz\n", name, ms, 1024.0 * 2.4e9 / (winst / (ms * 1e-3)));
}
int main() {
  float* d; hipMalloc(&d, 256 * 2048 * 4);
  run<0>("ds_bpermute_b32", d); run<1>("ds_read_b32, 4 addresses per wave", d); run<2>("ds_read_b128, 4 addresses per wave", d);
  run<3>("ds_read_b128, 64 addresses per wave", d);
  return 0;
}
