// What one wave that has its SIMD to itself pays per instruction on gfx950 (the one-view forward IS such a wave: DESIGN App. R6-7).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/lwb tools/micro/lone_wave.hip && /tmp/lwb
// Every kernel runs ITER times a straight-line block of N instructions of one kind and reports cycles per instruction for
// 1 wave on the chip, and for 2 / 4 / 8 waves on ONE SIMD (a 512-thread workgroup puts two waves on each SIMD of its CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2000
#define STR2(x) #x
#define STR(x) STR2(x)
#define REPT(n, body) ".rept " STR(n) "\n\t" body "\n\t.endr\n\t"

template <int KIND>
__global__ void k(uint64_t* out, float seed) {
  float a = seed + threadIdx.x, b = seed * 2.0f, c = seed * 3.0f, d = seed * 5.0f, e = 1.0001f, f = 0.9999f;
  int si = (int)blockIdx.x + 1;
  uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {        // (not unrolled: a loop body of ~100 instructions, like the forward's trip)
    if (KIND == 0)        // 96 dependent 4-byte VALU (v_mul_f32 e32)
      asm volatile(REPT(96, "v_mul_f32 %0, %4, %0") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    else if (KIND == 1)   // 96 VALU, four independent chains interleaved
      asm volatile(REPT(24, "v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    else if (KIND == 2)   // 96 dependent 8-byte VALU (v_fma_f32)
      asm volatile(REPT(96, "v_fma_f32 %0, %4, %0, %5") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    else if (KIND == 3)   // 96 independent-ish 8-byte VALU (four chains)
      asm volatile(REPT(24, "v_fma_f32 %0, %4, %0, %5\n\tv_fma_f32 %1, %4, %1, %5\n\tv_fma_f32 %2, %4, %2, %5\n\tv_fma_f32 %3, %4, %3, %5") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
    else if (KIND == 4)   // 96 dependent DPP adds, the hazard covered by the chain through the non-DPP operand (as gh_quad_accumulate)
      asm volatile("s_nop 1\n\t" REPT(96, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    else if (KIND == 5)   // 96 DPP adds in three interleaved chains
      asm volatile("s_nop 1\n\t" REPT(32, "v_add_f32_dpp %0, %3, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %3, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %2, %3, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    else if (KIND == 6)   // 96 dependent SALU
      asm volatile(REPT(96, "s_add_u32 %0, %0, 3") : "+s"(si) : : "scc");
    else if (KIND == 7)   // 48 x (SALU, dependent VALU): the two kinds alternate
      asm volatile(REPT(48, "s_add_u32 %1, %1, 3\n\tv_mul_f32 %0, %2, %0") : "+v"(a), "+s"(si) : "v"(e) : "scc");
    else if (KIND == 8)   // VALU writes an SGPR pair (v_cmp), SALU reads it, VALU reads the SALU result: the ballot -> mask -> select pattern
      asm volatile(REPT(32, "v_cmp_gt_f32 vcc, %0, %2\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %0, %2, vcc") : "+v"(a), "+s"(si) : "v"(e) : "vcc", "scc");      // (s_and writes SCC: undeclared, it ate the loop's compare)
    else if (KIND == 9)   // ds_bpermute round trip, dependent
      asm volatile(REPT(16, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)") : "+v"(a) : "v"((int)(threadIdx.x * 4)));
    else if (KIND == 10)  // dependent v_ldexp / v_rndne / v_cvt (the software exponential's odd ones)
      asm volatile(REPT(32, "v_rndne_f32 %0, %0\n\tv_ldexp_f32 %0, %0, %1\n\tv_mul_f32 %0, %2, %0") : "+v"(a) : "v"(0), "v"(e));
  }
  uint64_t t1 = __builtin_readcyclecounter();
  if (a + b + c + d == 12345.678f || si == 0x7fffffff) out[63] = 1;          // keep the chains alive
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, int n_per_iter, uint64_t* d) {
  for (int threads : {64, 512, 1024}) {          // 1 wave; 2 waves per SIMD; 4 waves per SIMD
    uint64_t h[16] = {0};
    (void)hipMemset(d, 0, 64 * 8);
    k<KIND><<<1, threads>>>(d, 1.0f);
    k<KIND><<<1, threads>>>(d, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    uint64_t mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-58s waves/SIMD %d: %6.2f cycles per instruction of one wave (%.2f per SIMD issue)\n", name, threads == 64 ? 1 : threads / 256,
           (double)mx / ITER / n_per_iter, (double)mx / ITER / n_per_iter / (threads == 64 ? 1 : threads / 256));
  }
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  uint64_t* d; (void)hipMalloc(&d, 64 * 8);
  run<0>("v_mul_f32 e32, dependent chain", 96, d);
  run<1>("v_mul_f32 e32, four chains interleaved", 96, d);
  run<2>("v_fma_f32 (8 bytes), dependent chain", 96, d);
  run<3>("v_fma_f32 (8 bytes), four chains interleaved", 96, d);
  run<4>("v_add_f32_dpp, dependent chain", 96, d);
  run<5>("v_add_f32_dpp, three chains interleaved", 96, d);
  run<6>("s_add_u32, dependent chain", 96, d);
  run<7>("s_add_u32 / v_mul_f32 alternating", 96, d);
  run<8>("v_cmp -> s_and vcc -> v_cndmask", 96, d);
  run<9>("ds_bpermute_b32 + wait, dependent", 16, d);
  run<10>("v_rndne, v_ldexp, v_mul dependent", 96, d);
  return 0;
}
