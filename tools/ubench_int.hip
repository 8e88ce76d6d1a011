// Integer-ALU issue-rate microbenchmark for gfx950: how many wave-cycles do the instructions the
// Goldilocks/Poseidon kernels are built from actually cost?  (The CDNA4 guide lists fp/MFMA rates only.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_int.hip -o gpurun_out/ubench_int && ./gpurun_out/ubench_int
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 4096
#define UNROLL 16

template <int OP>
__global__ void k(uint64_t* out, uint32_t seed) {
  uint32_t a[UNROLL], b = seed | 1, c = seed * 3 + 7;
  uint64_t w[UNROLL];
  for (int i = 0; i < UNROLL; ++i) {
    a[i] = threadIdx.x * 17 + i + seed;
    w[i] = ((uint64_t)a[i] << 32) | (a[i] * 5);
  }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (OP == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 6) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(b) : "v"(c), "v"(seed) : "vcc");
      if (OP == 7) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));
      if (OP == 8) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(w[i]));
      if (OP == 9) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 10) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 11) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
      if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 13) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
      if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 16) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 18) asm volatile("v_alignbit_b32 %0, %0, %1, 10" : "+v"(a[i]) : "v"(b));
    }
  }
  uint64_t s = b;
  for (int i = 0; i < UNROLL; ++i) s += a[i] + w[i];
  if (s == 0x1234567) out[0] = s;
}

template <int OP>
void run(const char* name, int insts_per_iter) {
  uint64_t* d;
  hipMalloc(&d, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * 8, threads = 256;  // 8 waves / SIMD
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 3u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 5u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double wave_insts = (double)blocks * (threads / 64) * ITERS * UNROLL * insts_per_iter;
  // per SIMD: 1024 SIMDs; assume 2.4 GHz nominal
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-28s %8.3f ms  %6.2f cycles/wave-inst/SIMD (at 2.4 GHz)  %.2f T lane-ops/s\n", name, ms, cyc / (wave_insts / 1024.0),
         wave_insts * 64 / (ms * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  run<0>("v_mad_u64_u32 (acc)", 1);
  run<11>("v_mad_u64_u32 (c=0)", 1);
  run<1>("v_mul_lo_u32", 1);
  run<2>("v_mul_hi_u32", 1);
  run<3>("v_mad_u32_u24", 1);
  run<9>("v_mul_u32_u24", 1);
  run<10>("v_mul_hi_u32_u24", 1);
  run<4>("v_lshl_add_u32", 1);
  run<14>("v_lshl_or_b32", 1);
  run<15>("v_and_or_b32", 1);
  run<16>("v_dot4_u32_u8", 1);
  run<17>("v_perm_b32", 1);
  run<18>("v_alignbit_b32", 1);
  run<5>("v_add_u32", 1);
  run<6>("v_add_co+v_addc_co (pair)", 2);
  run<13>("v_sub_co_u32", 1);
  run<12>("v_cndmask_b32", 1);
  run<7>("v_lshlrev_b64", 1);
  run<8>("v_fma_f64", 1);
  return 0;
}
