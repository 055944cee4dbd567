// Cost of a software grid barrier on MI355X (all blocks co-resident: <= 2 per CU), with the agent-scope release / acquire a
// producer -> consumer phase boundary inside ONE kernel needs across the 8 XCDs' L2s — against the ~4.5-5 us a kernel boundary costs
// inside a replayed graph (profiles/r4: every small kernel of the 1-view step takes 4.4-5 us whatever it does).
// Each phase: every block writes 4 KB (its own slice), barrier, then reads the slice of block (b + 97) % G written in the phase
// before and checks it. Spins are bounded; a block that gives up sets a flag and the kernel still drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nblk, unsigned* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / nblk + 1u) * nblk;
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { *err = 1u; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// fence = 2: the same barrier without the release / acquire fences (what the atomics alone cost; reads may be stale)
__device__ __forceinline__ void grid_barrier_nofence(unsigned* bar, unsigned nblk, unsigned* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (old / nblk + 1u) * nblk;
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { *err = 1u; break; } }
  }
  __syncthreads();
}
// fence = 3: two levels — the blocks of a group of 16 meet on the group's own counter (16 atomics per address, the groups in
// parallel), the last arrival of a group adds ONE to the global counter; everybody spins on the global generation word
__device__ __forceinline__ void grid_barrier_tree(unsigned* bar, unsigned nblk, unsigned* err, unsigned gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned grp = blockIdx.x >> 4, ngrp = (nblk + 15u) >> 4;
    const unsigned gsize = (grp == ngrp - 1u) ? nblk - grp * 16u : 16u;
    const unsigned old = __hip_atomic_fetch_add(&bar[16 + grp * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old + 1u) == gsize * (gen + 1u)) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = ngrp * (gen + 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { *err = 1u; break; } }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void phases(unsigned* data, unsigned* bar, unsigned* err, unsigned* bad, int n_phase, int fence) {
  const unsigned G = gridDim.x, b = blockIdx.x;
  for (int p = 0; p < n_phase; ++p) {
    unsigned* mine = data + ((size_t)(p & 1) * G + b) * 1024;
    for (int i = threadIdx.x; i < 1024; i += 256) mine[i] = (unsigned)(p * 131071 + b * 1024 + i);
    if (fence == 1) grid_barrier(bar, G, err); else if (fence == 2) grid_barrier_nofence(bar, G, err);
    else if (fence == 3) grid_barrier_tree(bar, G, err, (unsigned)p); else __syncthreads();
    const unsigned o = (b + 97u) % G;
    const unsigned* theirs = data + ((size_t)(p & 1) * G + o) * 1024;
    unsigned wrong = 0;
    for (int i = threadIdx.x; i < 1024; i += 256) wrong += theirs[i] != (unsigned)(p * 131071 + o * 1024 + i);
    if (wrong && fence) atomicAdd(bad, wrong);
    if (fence == 3) grid_barrier_tree(bar + 4096, G, err, (unsigned)p);     // (the slice is rewritten two phases later: keep readers ahead of writers)
  }
}

int main() {
  unsigned *data, *bar, *err, *bad;
  CHECK(hipMalloc(&data, (size_t)2 * 1024 * 1024 * 4)); CHECK(hipMalloc(&bar, 65536)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int G : {97, 256, 512}) {
    for (int fence : {1, 2, 3, 0}) {
      float t[2];
      for (int k = 0; k < 2; ++k) {
        const int n_phase = k ? 33 : 1;
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
          CHECK(hipMemset(bar, 0, 65536)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipDeviceSynchronize());
          CHECK(hipEventRecord(e0));
          hipLaunchKernelGGL(phases, dim3(G), dim3(256), 0, 0, data, bar, err, bad, n_phase, fence);
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        t[k] = best;
      }
      unsigned herr, hbad; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      printf("%3d blocks, %s: 1 phase %6.2f us, 33 phases %7.2f us -> %5.2f us per phase boundary; stale reads %u, gave-up flag %u\n", G,
             fence == 1 ? "grid barrier + release/acquire" : fence == 2 ? "grid barrier, atomics only     " : fence == 3 ? "2 x tree barrier + rel/acq     " : "no barrier (block-local only)  ", t[0] * 1e3f, t[1] * 1e3f, (t[1] - t[0]) * 1e3f / 32.0f, hbad, herr);
    }
  }
  return 0;
}
