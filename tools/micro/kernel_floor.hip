// What does a kernel boundary cost inside a replayed HIP graph on MI355X, by kernel shape? (round 5: the one-view step is ~20 small
// kernels; rocprofv3 shows each of them at >= 4.4 us with no gaps — is that the dispatch floor or their own dependent-latency chains?)
// A graph of 6 kinds x 8 launches, replayed; run under `rocprofv3 --kernel-trace --stats` and read the per-kernel averages:
//   k_empty_1x64        one wave, nothing
//   k_empty_3080x256    a full grid of empty workgroups
//   k_store_3080x256    every thread stores one word (dirty lines at the kernel's end)
//   k_chain1 / 2 / 4    one workgroup, 1 / 2 / 4 DEPENDENT global loads (pointer chase through L2-resident data), then a store
// usage: hipcc --offload-arch=gfx950 -O2 -o /tmp/kernel_floor tools/micro/kernel_floor.hip && rocprofv3 --kernel-trace --stats -- /tmp/kernel_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty_1x64() {}
__global__ void k_empty_3080x256() {}
__global__ void k_store_3080x256(unsigned* o) { o[blockIdx.x * 256 + threadIdx.x] = threadIdx.x; }
template <int N> __global__ void k_chain(const unsigned* __restrict__ p, unsigned* o) {
  unsigned i = threadIdx.x;
#pragma unroll
  for (int k = 0; k < N; ++k) i = p[i];
  o[threadIdx.x] = i;
}
int main() {
  unsigned *p, *o;
  CHECK(hipMalloc(&p, 1 << 20)); CHECK(hipMalloc(&o, 3080 * 256 * 4));
  unsigned h[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) h[i] = (i * 97 + 13) & ((1 << 18) - 1);
  CHECK(hipMemcpy(p, h, sizeof h, hipMemcpyHostToDevice));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int r = 0; r < 8; ++r) {
    hipLaunchKernelGGL(k_empty_1x64, dim3(1), dim3(64), 0, s);
    hipLaunchKernelGGL(k_empty_3080x256, dim3(3080), dim3(256), 0, s);
    hipLaunchKernelGGL(k_store_3080x256, dim3(3080), dim3(256), 0, s, o);
    hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(256), 0, s, p, o);
    hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(256), 0, s, p, o);
    hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(256), 0, s, p, o);
  }
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 60; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipEventRecord(a, s));
  for (int i = 0; i < 100; ++i) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipEventRecord(b, s)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  std::printf("48 kernels per replay: %.2f us per replay = %.2f us per kernel (wall)\n", ms * 10.0f, ms * 10.0f / 48.0f);
  return 0;
}
