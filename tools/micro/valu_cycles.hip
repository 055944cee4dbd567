// Issue cost of the vector instructions the render kernels are made of, in SHADER CYCLES (s_memtime), on gfx950.
//
// tools/micro/valu_rate.hip converted wall time at an assumed 2.4 GHz and so could not pin the VALU
// ceiling. Here every wave brackets its own instruction stream with s_memtime (tick = shader cycle,
// MI355X_MICROARCH.md 'Per-instruction cycle constants'), and W = 1 / 2 / 4 / 8 waves are made resident per SIMD
// (workgroups of 4 W waves, one or two per CU), so the number reported is
//     cycles per wave64 instruction per SIMD = (cycles one wave needed) / (W x instructions per wave)
// for a stream of 8 independent dependency chains per lane. The guide's figure for plain FP32 is 2 cycles per SIMD
// (SIMD-32) and 4 for one wave alone. Whether an s_memtime tick IS a shader cycle under load is checked too: the W = 8
// launch is also timed with HIP events, which gives the tick rate (ticks per wall ns = GHz of the counter) and the absolute
// issue rate in wave-instructions per ns per SIMD — the number a kernel's (instructions / time / SIMDs) is compared with.
//
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_cycles tools/micro/valu_cycles.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define NCHAIN 8

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float a, float b, int iters) {
  float v[NCHAIN]; int m[NCHAIN]; float w[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i) { v[i] = a + i + threadIdx.x; m[i] = threadIdx.x + i; w[i] = b + i; }
  __shared__ float4 s_x[64];
  if (threadIdx.x < 64) s_x[threadIdx.x] = make_float4(a, b, a, b);
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "v"(a));
      if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b), "s"(0x5555555555555555ull));
      if (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      if (KIND == 6) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b));
      if (KIND == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(m[i]) : "v"(m[(i + 1) & 7]));
      if (KIND == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(b) : "vcc");
      if (KIND == 9) asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 10) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m[i]));
      if (KIND == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (KIND == 13) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(m[i]) : "v"(v[i]));
      if (KIND == 14) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v[i]) : "v"(m[i]));
      if (KIND == 15) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[i & ~1]) : "v"(*(double*)&w[0]), "v"(*(double*)&w[2]));
      if (KIND == 16) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&v[i & ~1]) : "v"(*(double*)&w[0]));
      if (KIND == 17) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      if (KIND == 18) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      if (KIND == 19) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[i]));
      if (KIND == 20) asm volatile("v_mul_f32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(b));
      if (KIND == 21) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 22) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(b));
      if (KIND == 23) { float4 t; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((m[i] & 63) * 16)); v[i] += t.x; }
      if (KIND == 24) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(m[i]) : "v"(m[(i + 1) & 7]));
      if (KIND == 25) asm volatile("s_nop 0");
      if (KIND == 26) asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(m[i]) : "s"(t0));
    }
  }
  asm volatile("s_nop 4\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float s = 0; for (int i = 0; i < NCHAIN; ++i) s += v[i] + m[i];
  if (s == 12345.678f) out[0] = 0;                        // keeps the chains alive
  if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, unsigned long long* d, int packed = 1) {
  const int iters = (KIND == 14 || KIND == 23) ? 4000 : 40000;      // W = 8: tens of ms per launch, so that the clocks settle
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-28s", name);
  for (int W : {1, 2, 4, 8}) {
    const int waves_per_block = W >= 4 ? 16 : 4 * W;          // W waves on each of the CU's 4 SIMDs (two blocks per CU for W = 8)
    const int blocks = 256 * (W == 8 ? 2 : 1);
    const int nw = blocks * waves_per_block;
    (void)hipMemset(d, 0, nw * 8);
    float ms = 0.0f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, d, 1.0f, 0.999f, iters);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[nw / 2];
    const double per = med / ((double)W * NCHAIN * iters) * (KIND == 15 || KIND == 16 ? 2.0 : 1.0);   // packed: 4 instructions per 8 chains
    printf("  W=%d %6.2f", W, per);
    if (W == 8) {
      const double n_inst = (double)NCHAIN * iters / (KIND == 15 || KIND == 16 ? 2.0 : 1.0);      // per wave
      printf("  | W=8: %.3f wave-instr/ns/SIMD, counter %.2f ticks/ns (kernel %.1f ms)", 8.0 * n_inst / (ms * 1e6), med / (ms * 1e6), ms);
    }
  }
  printf("\n");
}

int main() {
  printf("columns: s_memtime ticks per wave-instruction per SIMD with W waves resident per SIMD\n");
  unsigned long long* d; (void)hipMalloc(&d, 8 * 8192 * 8);
  run<0>("v_fma_f32", d); run<1>("v_mul_f32", d); run<2>("v_add_f32", d); run<21>("v_sub_f32", d); run<4>("v_min_f32", d); run<22>("v_max_f32", d);
  run<3>("v_cndmask_b32 (sgpr mask)", d); run<7>("v_and_b32", d); run<24>("v_bfe_u32", d); run<8>("v_cmp_gt_f32", d);
  run<9>("v_rndne_f32", d); run<10>("v_ldexp_f32", d); run<13>("v_cvt_i32_f32", d); run<11>("v_rcp_f32", d); run<12>("v_exp_f32", d);
  run<5>("v_mov_b32_dpp quad_perm", d); run<6>("v_add_f32_dpp quad_perm", d); run<20>("v_mul_f32_dpp quad bcast", d);
  run<17>("v_add_f32_dpp row_shr:1", d); run<18>("s_nop 1 + v_add_dpp row_shr", d); run<19>("v_add_f32_dpp row_bcast:15", d);
  run<15>("v_pk_fma_f32", d); run<16>("v_pk_mul_f32", d);
  run<14>("ds_bpermute_b32 + wait", d); run<23>("ds_read_b128 + wait", d); run<25>("s_nop 0", d); run<26>("s_ff1_i32_b64", d);
  return 0;
}
