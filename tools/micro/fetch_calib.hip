// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of THIS pipeline. MI355X_MICROARCH.md: "FETCH_SIZE reports exactly
// 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known byte count
// in your own access pattern before trusting an absolute." profiles/*_pmc_traffic.json doubles FETCH_SIZE for every kernel; the chain-rule
// kernel reads 36-byte sub-records with three 12-byte loads per lane at scattered slots, the render kernels 16-byte and 8-byte records.
// Every kernel below reads a KNOWN number of bytes exactly once from a buffer larger than the Infinity Cache (1 GiB, untouched before):
//   k_x4_stream    16 B per lane, coalesced                       (the guide's reference case: expect FETCH_SIZE = bytes / 2)
//   k_x2_stream     8 B per lane, coalesced                       (inst_r2)
//   k_x1_stream     4 B per lane, coalesced                       (keys, vals, flags)
//   k_x3_dense     3 x 12 B per lane, lanes 36 B apart            (sub-records of consecutive slots: the t-ordered record sum)
//   k_x3_slots     3 x 12 B per lane, lanes at pseudo-random 144-B slots, one 36-B sub-record each   (the chain rule before round 5)
//   k_line_gather  64-B line per lane (4 x 16 B), pseudo-random lines                                  (geometry-line gather of emit / ranges)
// usage (GPU box): bash tools/micro/fetch_calib.sh <out dir>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct F3 { float x, y, z; };
__device__ __forceinline__ uint32_t mix(uint32_t i) { i ^= i >> 16; i *= 0x7feb352du; i ^= i >> 15; i *= 0x846ca68bu; i ^= i >> 16; return i; }
__global__ void k_x4_stream(const float4* p, float* o, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { float4 v = p[i]; if (v.x == 1234.5f) o[0] = v.y; } }
__global__ void k_x2_stream(const float2* p, float* o, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { float2 v = p[i]; if (v.x == 1234.5f) o[0] = v.y; } }
__global__ void k_x1_stream(const float* p, float* o, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) { float v = p[i]; if (v == 1234.5f) o[0] = v; } }
__global__ void k_x3_dense(const float* p, float* o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const F3* r = (const F3*)(p + i * 9); F3 a = r[0], b = r[1], c = r[2]; if (a.x + b.y + c.z == 1234.5f) o[0] = a.y; }
}
__global__ void k_x3_slots(const float* p, float* o, size_t n, uint32_t n_slots) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const size_t slot = mix((uint32_t)i) % n_slots; const F3* r = (const F3*)(p + (slot * 4 + (i & 3)) * 9);
               F3 a = r[0], b = r[1], c = r[2]; if (a.x + b.y + c.z == 1234.5f) o[0] = a.y; }
}
__global__ void k_line_gather(const float4* p, float* o, size_t n, uint32_t n_lines) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float4* r = p + (size_t)(mix((uint32_t)i) % n_lines) * 4; float4 a = r[0], b = r[1], c = r[2], d = r[3];
               if (a.x + b.y + c.z + d.w == 1234.5f) o[0] = a.y; }
}
int main() {
  const size_t BYTES = (size_t)1 << 30;
  char* buf; float* o;
  CHECK(hipMalloc(&buf, BYTES)); CHECK(hipMalloc(&o, 64)); CHECK(hipMemset(buf, 0, BYTES));
  CHECK(hipDeviceSynchronize());
  const size_t MB256 = (size_t)256 << 20;         // bytes read by every streaming kernel: each from its own quarter of the buffer
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
  size_t n;
  n = MB256 / 16; hipLaunchKernelGGL(k_x4_stream, blocks(n), dim3(256), 0, 0, (const float4*)buf, o, n);
  n = MB256 / 8;  hipLaunchKernelGGL(k_x2_stream, blocks(n), dim3(256), 0, 0, (const float2*)(buf + MB256), o, n);
  n = MB256 / 4;  hipLaunchKernelGGL(k_x1_stream, blocks(n), dim3(256), 0, 0, (const float*)(buf + 2 * MB256), o, n);
  n = MB256 / 36; hipLaunchKernelGGL(k_x3_dense, blocks(n), dim3(256), 0, 0, (const float*)(buf + 3 * MB256), o, n);
  CHECK(hipDeviceSynchronize());
  // scattered: 2 M sub-records of 36 B = 72 MB useful, out of 1 GiB / 144 B = 7.4 M slots; 1 M lines of 64 B = 64 MB out of 16 M lines
  n = (size_t)2 << 20; hipLaunchKernelGGL(k_x3_slots, blocks(n), dim3(256), 0, 0, (const float*)buf, o, n, (uint32_t)(BYTES / 144));
  CHECK(hipDeviceSynchronize());
  n = (size_t)1 << 20; hipLaunchKernelGGL(k_line_gather, blocks(n), dim3(256), 0, 0, (const float4*)buf, o, n, (uint32_t)(BYTES / 64));
  CHECK(hipDeviceSynchronize());
  std::printf("known bytes: k_x4_stream / k_x2_stream / k_x1_stream 268435456 each; k_x3_dense %zu; k_x3_slots %zu useful (2097152 sub-records of 36 B at random 144-B slots);"
              " k_line_gather %zu useful (1048576 random 64-B lines)\n", (MB256 / 36) * 36, ((size_t)2 << 20) * 36, ((size_t)1 << 20) * 64);
  return 0;
}
