// Cost of the "last block finishes the reduction" pattern on MI355X: every block writes a 256-byte partial row (plus 64 KB of
// other stores, like a real producer), then takes a ticket; the last block sums all rows. Variants:
//   0: two kernels (producer, then a one-block reducer)                      — the baseline
//   1: one kernel, release on the ticket (agent scope) + acquire in the last block — the formally correct form
//   2: one kernel, partial rows stored / loaded as relaxed agent-scope atomics, workgroup-scope fence before a relaxed ticket
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float block_part(const float* __restrict__ in, float* __restrict__ out, int per_block) {
  float acc = 0.0f;
  const size_t base = (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) { const float v = in[base + i]; out[base + i] = v * 2.0f; acc += v; }
  return acc;
}

__global__ __launch_bounds__(256) void producer(const float* in, float* out, int per_block, float* rows) {
  const float acc = block_part(in, out, per_block);
  if (threadIdx.x < 64) rows[(size_t)blockIdx.x * 64 + threadIdx.x] = acc + (float)threadIdx.x;
}
// fixed-order sum of the rows by ONE block: thread t takes slot t & 63 of rows t >> 6, (t >> 6) + 4, ..; four partials per slot meet in LDS
template <bool ATOMIC>
__device__ __forceinline__ void reduce_rows(const float* rows, int nblk, float* result) {
  __shared__ float s_p[4][64];
  float s = 0.0f;
  for (int b = threadIdx.x >> 6; b < nblk; b += 4) {
    const float* p = &rows[(size_t)b * 64 + (threadIdx.x & 63)];
    s += ATOMIC ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
  }
  s_p[threadIdx.x >> 6][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64) result[threadIdx.x] = ((s_p[0][threadIdx.x] + s_p[1][threadIdx.x]) + s_p[2][threadIdx.x]) + s_p[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void reducer(const float* rows, int nblk, float* result) { reduce_rows<false>(rows, nblk, result); }
template <int MODE>
__global__ __launch_bounds__(256) void fused(const float* in, float* out, int per_block, float* rows, unsigned* ticket, float* result) {
  __shared__ unsigned s_last;
  const float acc = block_part(in, out, per_block);
  if (threadIdx.x < 64) {
    const float v = acc + (float)threadIdx.x;
    if (MODE == 1) rows[(size_t)blockIdx.x * 64 + threadIdx.x] = v;
    else __hip_atomic_store(&rows[(size_t)blockIdx.x * 64 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // s_waitcnt vmcnt(0): the write-through stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = MODE == 1 ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == gridDim.x - 1 ? 1u : 0u;
    if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (MODE == 1) reduce_rows<false>(rows, (int)gridDim.x, result); else reduce_rows<true>(rows, (int)gridDim.x, result);
}

int main() {
  const int nblk = 3080, per_block = 4096;      // 50 MB in, 50 MB out: a 25-us producer
  float *in, *out, *rows, *result; unsigned* ticket;
  CHECK(hipMalloc(&in, (size_t)nblk * per_block * 4)); CHECK(hipMalloc(&out, (size_t)nblk * per_block * 4));
  CHECK(hipMalloc(&rows, (size_t)nblk * 64 * 4)); CHECK(hipMalloc(&result, 64 * 4)); CHECK(hipMalloc(&ticket, 4));
  std::vector<float> h((size_t)nblk * per_block);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) * (1.0f / 1024.0f);
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(ticket, 0, 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ref[64], got[64];
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(result, 0, 256)); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int it = 0; it < 50; ++it) {
        if (mode == 0) { hipLaunchKernelGGL(producer, dim3(nblk), dim3(256), 0, 0, in, out, per_block, rows); hipLaunchKernelGGL(reducer, dim3(1), dim3(256), 0, 0, rows, nblk, result); }
        else if (mode == 1) hipLaunchKernelGGL(fused<1>, dim3(nblk), dim3(256), 0, 0, in, out, per_block, rows, ticket, result);
        else hipLaunchKernelGGL(fused<2>, dim3(nblk), dim3(256), 0, 0, in, out, per_block, rows, ticket, result);
      }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(got, result, 256, hipMemcpyDeviceToHost));
      if (mode == 0) for (int i = 0; i < 64; ++i) ref[i] = got[i];
      int bad = 0; for (int i = 0; i < 64; ++i) bad += got[i] != ref[i];
      printf("mode %d rep %d: %.2f us per iteration, %d of 64 sums differ from the two-kernel result\n", mode, rep, ms * 1e3f / 50, bad);
    }
  }
  return 0;
}
