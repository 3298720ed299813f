// VALU issue-rate microbenchmark (dev tool): cycles per wave-instruction with 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float sv) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  int lanesel = (int)sv;  // uniform
  asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
  asm volatile("s_mov_b64 vcc, 0x3333" ::: "vcc");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "s"(sv));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 3) { int s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(a[i])); asm volatile("" :: "s"(s)); }
        if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
        if (KIND == 5) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %0" :: "v"(a[i]) : "s20", "s21");
        if (KIND == 6) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
        if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
        if (KIND == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[20:21]" : "+v"(a[i]));
        if (KIND == 9) asm volatile("v_min_f32 %0, %0, %0" : "+v"(a[i]));
        if (KIND == 10) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (KIND == 11) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i+1)&7]));
        if (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
        if (KIND == 13) asm volatile("v_cmp_le_f32_e32 vcc, %0, %0" :: "v"(a[i]) : "vcc");
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int KIND> void run(const char* name, float* d, int blocks_per_cu) {
  int iters = 2000;
  int grid = 256 * blocks_per_cu;
  k<KIND><<<grid, 256>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<KIND><<<grid, 256>>>(d, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  double winstr_per_simd = (double)iters * REP * (grid * 4.0) / 1024.0;  // wave-instr per SIMD
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns/instr/SIMD, in-kernel %.2f cyc/instr/SIMD (memtime clk)\n", name, blocks_per_cu, ms,
         ms * 1e6 / winstr_per_simd, cyc / ((double)iters * REP * blocks_per_cu));
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4 + 1024);
  for (int b : {2, 8}) {
    run<0>("v_fma_f32 vvv", d, b); run<1>("v_fma_f32 sgpr", d, b); run<2>("v_exp_f32", d, b); run<3>("v_readlane_b32", d, b);
    run<4>("v_cndmask vcc", d, b); run<5>("v_cmp_e64 ->sgpr", d, b); run<6>("v_add_f32", d, b); run<7>("v_mul_f32", d, b);
    run<8>("v_cndmask e64 sgprpair", d, b); run<9>("v_min_f32", d, b); run<10>("v_add_f32_dpp", d, b); run<11>("v_cndmask vcc 2src", d, b);
    run<12>("v_mov_b32", d, b); run<13>("v_cmp_e32 ->vcc", d, b);
  }
}
