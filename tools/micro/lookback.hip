// Cost of a decoupled look-back over per-(block, digit) status words on MI355X — the piece a one-launch radix pass would add to
// the scatter kernel in exchange for the histogram + scan kernels. Every block takes a ticket (dynamic block
// id), "counts" its keys (a fixed pseudo-random count per (block, digit), after a delay loop that stands for loading and ranking
// 2048 keys), publishes flag|count as ONE 32-bit word per digit (relaxed agent-scope atomic store: no fence needed, flag and value
// travel together), then looks back: thread = digit, WIN predecessors' words in flight per round trip, until a word flagged
// PREFIX is met; publishes its inclusive prefix; writes the exclusive prefix out. Spins are bounded (a block that gives up sets
// an error flag and leaves), so the kernel always drains.
//   shapes: the tile partition (one segment, 1455 or 728 blocks, 128 digits) and the per-view depth sort (8 segments x 49 blocks,
//   256 digits). Reference: the kernels it would replace take 8.8 + 4.8 us (tile level) / 5.2 us (depth level) per pass.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define ST_AGG 0x40000000u
#define ST_PRE 0x80000000u
#define VAL_MASK 0x3FFFFFFFu

__host__ __device__ inline unsigned count_of(unsigned seg, unsigned b, unsigned d) {
  unsigned x = (seg * 7919u + b) * 2654435761u ^ (d * 40503u);
  x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
  return x & 63u;
}

template <int WIN>
__global__ __launch_bounds__(256) void lookback(unsigned* __restrict__ status, unsigned* __restrict__ ticket, unsigned* __restrict__ excl_out,
                                                unsigned* __restrict__ err, int nblk, int ndig, int delay) {
  __shared__ unsigned s_b;
  const unsigned seg = blockIdx.y;
  if (threadIdx.x == 0) s_b = __hip_atomic_fetch_add(&ticket[seg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned b = s_b;
  // stand-in for loading + ranking the block's keys
  float acc = (float)threadIdx.x;
  for (int i = 0; i < delay; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
  const int d = threadIdx.x;
  if (d >= ndig) { if (acc == 12345.678f) err[1] = 1; return; }
  unsigned* st = status + (size_t)seg * nblk * ndig;
  const unsigned agg = count_of(seg, b, d);
  __hip_atomic_store(&st[(size_t)b * ndig + d], (b == 0 ? ST_PRE : ST_AGG) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned excl = 0;
  if (b > 0) {
    int p = (int)b - 1;
    int spins = 0;
    bool done = false;
    while (!done) {
      unsigned w[WIN];
#pragma unroll
      for (int j = 0; j < WIN; ++j) w[j] = p - j >= 0 ? __hip_atomic_load(&st[(size_t)(p - j) * ndig + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_PRE;
#pragma unroll
      for (int j = 0; j < WIN; ++j) {
        if (done) break;
        if ((w[j] >> 30) == 0u) break;                     // not published yet: re-read from here
        excl += w[j] & VAL_MASK;
        --p;
        if (w[j] & ST_PRE) done = true;
      }
      if (++spins > (1 << 16)) { err[0] = 1; break; }     // bounded: never hang the box
    }
    __hip_atomic_store(&st[(size_t)b * ndig + d], ST_PRE | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  excl_out[((size_t)seg * nblk + b) * ndig + d] = excl + (acc == 12345.678f ? 1u : 0u);
}

__global__ void clear(unsigned* status, size_t n, unsigned* ticket, int segs) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) status[i] = 0u;
  if (i < (size_t)segs) ticket[i] = 0u;
}

template <int WIN>
static int run(const char* name, int segs, int nblk, int ndig, int delay) {
  const size_t n = (size_t)segs * nblk * ndig;
  unsigned *status, *ticket, *out, *err;
  CHECK(hipMalloc(&status, n * 4)); CHECK(hipMalloc(&ticket, 64 * 4)); CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&err, 8));
  CHECK(hipMemset(err, 0, 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f, best_clear = 1e9f;
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL(clear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, status, n, ticket, segs);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(lookback<WIN>, dim3(nblk, segs), dim3(256), 0, 0, status, ticket, out, err, nblk, ndig, delay);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    (void)best_clear;
  }
  // the same kernel without any look-back work to subtract: delay only (ndig = 0 -> every thread leaves after the delay loop)
  float base = 1e9f;
  for (int rep = 0; rep < 20; ++rep) {
    hipLaunchKernelGGL(clear, dim3(1), dim3(256), 0, 0, status, (size_t)0, ticket, segs);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(lookback<WIN>, dim3(nblk, segs), dim3(256), 0, 0, status, ticket, out, err, nblk, 0, delay);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < base) base = ms;
  }
  std::vector<unsigned> h(n); unsigned herr[2];
  CHECK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int s = 0; s < segs; ++s) for (int d = 0; d < ndig; ++d) {
    unsigned run_ = 0;
    for (int b = 0; b < nblk; ++b) { if (h[((size_t)s * nblk + b) * ndig + d] != run_) ++bad; run_ += count_of(s, b, d); }
  }
  printf("%-34s WIN %2d delay %5d: kernel %7.2f us, same grid without the look-back %7.2f us -> look-back adds %6.2f us; %zu wrong prefixes, gave-up flag %u\n",
         name, WIN, delay, best * 1e3f, base * 1e3f, (best - base) * 1e3f, bad, herr[0]);
  hipFree(status); hipFree(ticket); hipFree(out); hipFree(err);
  return 0;
}

int main() {
  for (int delay : {0, 2000}) {       // 2000 dependent fmas ~ 4 us: the time a scatter block spends before its counts exist
    if (run<4>("tile partition, 1455 blocks x 128", 1, 1455, 128, delay)) return 1;
    if (run<8>("tile partition, 1455 blocks x 128", 1, 1455, 128, delay)) return 1;
    if (run<16>("tile partition, 1455 blocks x 128", 1, 1455, 128, delay)) return 1;
    if (run<8>("tile partition, 728 blocks x 128", 1, 728, 128, delay)) return 1;
    if (run<16>("tile partition, 728 blocks x 128", 1, 728, 128, delay)) return 1;
    if (run<8>("depth sort, 8 x 49 blocks x 256", 8, 49, 256, delay)) return 1;
    if (run<16>("depth sort, 8 x 49 blocks x 256", 8, 49, 256, delay)) return 1;
    if (run<8>("depth sort 1 view, 97 blocks x 256", 1, 97, 256, delay)) return 1;
    if (run<8>("tile partition 1 view, 182 x 1024", 1, 182, 256, delay)) return 1;
  }
  return 0;
}
