// Host cost of issuing the 17 small kernels of a one-view forward: 17 hipLaunchKernelGGL calls against ONE hipGraphLaunch of the same
// sequence captured once (and against a graph whose kernel-node parameters are rewritten before every launch) — the question
// behind VERDICT r3 item 6: of the drop-in's 131 us of raster_forward per view, 80 us are HIP launches inside the library.
// Kernels: a few microseconds of dependent work each (so the queue never runs dry of host-side slack in the direct case either).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void small(unsigned* p, int n, unsigned add) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = p[i] * 3u + add;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int NK = 17, N = 1 << 16, ITERS = 2000;
  unsigned* buf[2];
  CHECK(hipMalloc(&buf[0], N * 4)); CHECK(hipMalloc(&buf[1], N * 4));
  CHECK(hipMemset(buf[0], 0, N * 4)); CHECK(hipMemset(buf[1], 0, N * 4));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  auto direct = [&](unsigned* p) { for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(small, dim3(N / 256), dim3(256), 0, s, p, N, (unsigned)k); };
  for (int i = 0; i < 50; ++i) direct(buf[0]);
  CHECK(hipStreamSynchronize(s));
  // 1. direct launches
  double t0 = now_us();
  for (int i = 0; i < ITERS; ++i) direct(buf[i & 1]);
  double t_enq = now_us() - t0;
  CHECK(hipStreamSynchronize(s));
  double t_all = now_us() - t0;
  printf("direct: %d launches: host %.1f us per sequence, wall %.1f us per sequence\n", NK, t_enq / ITERS, t_all / ITERS);
  // 2. one graph per buffer, captured once
  hipGraphExec_t ge[2];
  hipGraph_t gr[2];
  double t_cap = now_us();
  for (int b = 0; b < 2; ++b) {
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    direct(buf[b]);
    CHECK(hipStreamEndCapture(s, &gr[b]));
    CHECK(hipGraphInstantiate(&ge[b], gr[b], nullptr, nullptr, 0));
  }
  printf("capture + instantiate: %.1f us per graph\n", (now_us() - t_cap) / 2);
  for (int i = 0; i < 50; ++i) CHECK(hipGraphLaunch(ge[i & 1], s));
  CHECK(hipStreamSynchronize(s));
  t0 = now_us();
  for (int i = 0; i < ITERS; ++i) CHECK(hipGraphLaunch(ge[i & 1], s));
  t_enq = now_us() - t0;
  CHECK(hipStreamSynchronize(s));
  t_all = now_us() - t0;
  printf("graph (cached per pointer set): host %.1f us per sequence, wall %.1f us per sequence\n", t_enq / ITERS, t_all / ITERS);
  // 3. one graph, every kernel node's parameters rewritten before each launch
  size_t nn = 0;
  CHECK(hipGraphGetNodes(gr[0], nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CHECK(hipGraphGetNodes(gr[0], nodes.data(), &nn));
  std::vector<hipKernelNodeParams> kp(nn);
  for (size_t k = 0; k < nn; ++k) CHECK(hipGraphKernelNodeGetParams(nodes[k], &kp[k]));
  t0 = now_us();
  int nN = N;
  for (int i = 0; i < ITERS; ++i) {
    unsigned* p = buf[i & 1];
    for (size_t k = 0; k < nn; ++k) {
      unsigned add = (unsigned)k;
      void* args[3] = {&p, &nN, &add};
      hipKernelNodeParams q = kp[k];
      q.kernelParams = args;
      CHECK(hipGraphExecKernelNodeSetParams(ge[0], nodes[k], &q));
    }
    CHECK(hipGraphLaunch(ge[0], s));
  }
  t_enq = now_us() - t0;
  CHECK(hipStreamSynchronize(s));
  t_all = now_us() - t0;
  printf("graph (all %zu nodes' parameters rewritten per launch): host %.1f us per sequence, wall %.1f us per sequence\n", nn, t_enq / ITERS, t_all / ITERS);
  // 4. capture + instantiate + launch + destroy every time (no cache at all)
  t0 = now_us();
  for (int i = 0; i < 200; ++i) {
    hipGraph_t g; hipGraphExec_t ex;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    direct(buf[i & 1]);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(ex, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipGraphExecDestroy(ex)); CHECK(hipGraphDestroy(g));
  }
  printf("capture + instantiate + launch + sync + destroy every time: %.1f us per sequence\n", (now_us() - t0) / 200);
  return 0;
}
