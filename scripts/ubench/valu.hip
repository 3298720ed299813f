// VALU / LDS issue-rate microbenchmark (dev tool): ns and cycles per wave-instruction at 1, 2 and 8 waves per SIMD.
// Output of round 4: profiles/r04_valu_issue.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float sv) {
  float a[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) { p[i].x = threadIdx.x * 0.001f + i; p[i].y = threadIdx.x * 0.002f - i; }
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 2654435761u + i;
  __shared__ float4 lds[1024];
  if (KIND >= 30 && KIND < 40) { for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3); __syncthreads(); }
  unsigned laddr = (unsigned)(size_t)&lds[(threadIdx.x * 37 + 11) & 1023];  // per-lane scattered 16-byte slots (conflict-free)
  unsigned lbase = (unsigned)(size_t)&lds[0];                                // one address for the whole wave
  unsigned lrand = (unsigned)(size_t)&lds[(threadIdx.x * 2654435761u >> 20) & 255];  // pseudo-random slots of a 256-record batch
  unsigned lrow = (unsigned)(size_t)&lds[((threadIdx.x >> 4) * 2654435761u >> 20) & 255];  // one slot per 16-lane row
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
        // round 4: packed-f32 VALU on independent 64-bit register pairs, and the integer / LDS ops of the compositing trips
        if (KIND == 14) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
        if (KIND == 15) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
        if (KIND == 16) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
        if (KIND == 17) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (KIND == 18) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 19) asm volatile("v_ffbl_b32 %0, %0" : "+v"(u[i]));
        if (KIND == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 21) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 22) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));
        if (KIND == 23) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 24) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 25) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 26) asm volatile("v_mul_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (KIND == 27) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (KIND == 28) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 29) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(p[i]));
        if (KIND == 30) { float4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(laddr)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; } }
        if (KIND == 31) { float q; asm volatile("ds_read_b32 %0, %1" : "=v"(q) : "v"(laddr)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q; } }
        if (KIND == 32) { f2 q; asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(laddr)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; } }
        if (KIND == 33) { float4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(laddr & ~1023u)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; } }
        // second batch of round 4: which operand forms keep an op in the 2-cycle class
        if (KIND == 40) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
        if (KIND == 41) asm volatile("v_fma_f32 %0, %0, %0, 1.0" : "+v"(a[i]));
        if (KIND == 42) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %0" : "+v"(a[i]));
        if (KIND == 43) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(a[i]));
        if (KIND == 44) asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0" : "+v"(a[i]));
        if (KIND == 45) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 46) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 47) asm volatile("v_med3_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 48) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 49) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 50) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
        if (KIND == 51) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 52) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 53) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 54) asm volatile("v_add_co_u32 %0, s[20:21], %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "s20", "s21");
        if (KIND == 55) asm volatile("v_or3_b32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 56) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 57) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 58) { asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" :: "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc"); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
        if (KIND == 59) { asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" :: "v"(a[i]), "v"(a[(i + 1) & 7]) : "s20", "s21"); asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
        if (KIND == 60) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
        if (KIND == 61) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 62) asm volatile("v_mbcnt_lo_u32_b32 %0, s20, %0" : "+v"(u[i]));
        if (KIND == 63) asm volatile("v_exp_f32_e64 %0, -%0" : "+v"(a[i]));
        if (KIND == 64) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sv));
        if (KIND == 65) asm volatile("v_mul_f32_e64 %0, -%0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 66) asm volatile("v_fma_f32 %0, -%0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 67) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f7fbe77" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 68) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 69) { int s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(a[i])); asm volatile("" :: "s"(s)); }
        if (KIND == 70) asm volatile("v_bfi_b32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 71) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));
        if (KIND == 72) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 3) & 7]));
        if (KIND == 73) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 74) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" :: "v"(a[i]), "s"(sv) : "s20", "s21");
        if (KIND == 75) asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x80" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 76) asm volatile("v_subrev_co_u32_e64 %0, s[20:21], 1, %0" : "+v"(u[i]) :: "s20", "s21");
        // third batch: shifts and a few more integer forms
        if (KIND == 77) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 78) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(u[i]));
        if (KIND == 79) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 80) asm volatile("v_not_b32 %0, %0" : "+v"(u[i]));
        if (KIND == 81) asm volatile("v_and_b32 %0, 15, %0" : "+v"(u[i]));
        if (KIND == 82) asm volatile("v_add_u32 %0, 16, %0" : "+v"(u[i]));
        if (KIND == 83) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 84) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 85) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 86) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 87) asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
        if (KIND == 88) asm volatile("v_fma_f32 %0, %0, %0, %0 clamp" : "+v"(a[i]));
        if (KIND == 89) asm volatile("v_mul_f32_e64 %0, %0, %0 mul:2" : "+v"(a[i]));
        if (KIND == 90) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 91) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        if (KIND == 92) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(u[(i + 1) & 7]));
        // fourth batch (round 4): 64-bit compare / min / max for sort keys held as register pairs
        if (KIND == 93) asm volatile("v_min_f64 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (KIND == 94) asm volatile("v_max_f64 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (KIND == 95) asm volatile("v_cmp_gt_u64_e64 s[20:21], %0, %1" :: "v"(p[i]), "v"(p[(i + 1) & 7]) : "s20", "s21");
        if (KIND == 96) asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1" :: "v"(p[i]), "v"(p[(i + 1) & 7]) : "s20", "s21");
        if (KIND == 97) asm volatile("v_mov_b64 %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        if (KIND == 98) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(u[i]) : "v"(laddr & 252u));
        if (KIND == 99) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" :: "v"(u[i]), "v"(u[(i + 1) & 7]) : "s20", "s21");
        if (KIND == 34) { float4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lbase)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; } }
        if (KIND == 35) { float q; asm volatile("ds_read_u8 %0, %1" : "=v"(q) : "v"(laddr)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q; } }
        if (KIND == 36) { float4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lrand)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; } }
        if (KIND == 37) { float4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lrow)); if (i == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a[0] += q.x; } }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int KIND> void run(const char* name, float* d, int blocks_per_cu) {
  int iters = 2000;
  int grid = 256 * blocks_per_cu;
  k<KIND><<<grid, 256>>>(d, 10, 1.0f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0); k<KIND><<<grid, 256>>>(d, iters, 1.0f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float cyc; (void)hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  double winstr_per_simd = (double)iters * REP * (grid * 4.0) / 1024.0;  // wave-instr per SIMD
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns/instr/SIMD, in-kernel %.2f cyc/instr/SIMD (memtime clk)\n", name, blocks_per_cu, ms,
         ms * 1e6 / winstr_per_simd, cyc / ((double)iters * REP * blocks_per_cu));
}
int main(int argc, char** argv) {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 + 1024);
  for (int b : {2, 8}) {
    run<93>("v_min_f64", d, b); run<94>("v_max_f64", d, b); run<95>("v_cmp_gt_u64 ->sgpr", d, b); run<96>("v_cmp_lt_f64 ->sgpr", d, b);
    run<97>("v_mov_b64", d, b); run<98>("ds_bpermute_b32 + wait", d, b); run<99>("v_cmp_gt_u32 ->sgpr", d, b);
    run<0>("v_fma_f32 vvv", d, b); run<8>("v_cndmask e64 sgprpair", d, b); run<28>("v_mov_b32_dpp quad_perm", d, b);
    if (argc > 1) continue;  // "valu.bin new": the fourth batch and three anchors only
    run<0>("v_fma_f32 vvv", d, b); run<1>("v_fma_f32 sgpr", d, b); run<2>("v_exp_f32", d, b); run<3>("v_readlane_b32", d, b);
    run<4>("v_cndmask vcc", d, b); run<5>("v_cmp_e64 ->sgpr", d, b); run<6>("v_add_f32", d, b); run<7>("v_mul_f32", d, b);
    run<8>("v_cndmask e64 sgprpair", d, b); run<9>("v_min_f32", d, b); run<10>("v_add_f32_dpp", d, b); run<11>("v_cndmask vcc 2src", d, b);
    run<12>("v_mov_b32", d, b); run<13>("v_cmp_e32 ->vcc", d, b);
    run<14>("v_pk_fma_f32", d, b); run<15>("v_pk_mul_f32", d, b); run<16>("v_pk_add_f32", d, b); run<17>("v_pk_fma_f32 2 pairs", d, b);
    run<27>("v_pk_mul_f32 op_sel_hi", d, b); run<18>("v_fmac_f32_e32", d, b); run<19>("v_ffbl_b32", d, b); run<20>("v_and_b32", d, b);
    run<21>("v_lshl_add_u32", d, b); run<22>("v_bfe_u32", d, b); run<23>("v_mad_u32_u24", d, b); run<24>("v_max3_f32", d, b);
    run<25>("v_rcp_f32", d, b); run<26>("v_mul_f32_dpp row_mirror", d, b); run<28>("v_mov_b32_dpp quad_perm", d, b);
    run<29>("v_lshlrev_b64", d, b); run<30>("ds_read_b128 per-lane slots", d, b); run<31>("ds_read_b32 per-lane", d, b);
    run<32>("ds_read_b64 per-lane", d, b); run<33>("ds_read_b128 broadcast", d, b);
    run<34>("ds_read_b128 one address", d, b); run<35>("ds_read_u8 per-lane", d, b); run<36>("ds_read_b128 random slots", d, b);
    run<37>("ds_read_b128 slot per row", d, b);
    run<40>("v_fma_f32 3 distinct", d, b); run<41>("v_fma_f32 inline 1.0", d, b); run<42>("v_fmamk_f32 literal", d, b);
    run<43>("v_sub_f32 1.0-x", d, b); run<44>("v_mul_f32 literal", d, b); run<45>("v_sub_f32", d, b); run<46>("v_max_f32", d, b);
    run<47>("v_med3_f32", d, b); run<48>("v_or_b32", d, b); run<49>("v_xor_b32", d, b); run<50>("v_lshlrev_b32 imm", d, b);
    run<51>("v_lshrrev_b32 vgpr", d, b); run<52>("v_add_u32", d, b); run<53>("v_sub_u32", d, b); run<54>("v_add_co_u32 ->sgpr", d, b);
    run<55>("v_or3_b32", d, b); run<56>("v_add3_u32", d, b); run<57>("v_cndmask e64 vcc", d, b); run<58>("v_cmp->vcc + cndmask vcc", d, b);
    run<59>("v_cmp->sgpr + cndmask sgpr", d, b); run<60>("v_cvt_f32_i32", d, b); run<61>("v_floor_f32", d, b); run<62>("v_mbcnt_lo", d, b);
    run<63>("v_exp_f32 neg", d, b); run<64>("v_mul_f32 sgpr", d, b); run<65>("v_mul_f32_e64 neg", d, b); run<66>("v_fma_f32 neg", d, b);
    run<67>("v_fmaak_f32", d, b); run<68>("v_lshlrev_b32_sdwa", d, b); run<69>("v_readfirstlane", d, b); run<70>("v_bfi_b32", d, b);
    run<71>("v_mul_f32 inline 0.5", d, b); run<72>("v_add_f32 2 distinct", d, b); run<73>("v_mul_lo_u32", d, b); run<74>("v_cmp_e64 sgpr operand", d, b);
    run<75>("v_bitop3_b32", d, b); run<76>("v_subrev_co_u32", d, b);
    run<77>("v_lshlrev_b32 vgpr", d, b); run<78>("v_lshrrev_b32 imm", d, b); run<79>("v_ashrrev_i32 vgpr", d, b); run<80>("v_not_b32", d, b);
    run<81>("v_and_b32 inline int", d, b); run<82>("v_add_u32 inline int", d, b); run<83>("v_mul_u32_u24", d, b); run<84>("v_add_lshl_u32", d, b);
    run<85>("v_and_or_b32", d, b); run<86>("v_cvt_u32_f32", d, b); run<87>("v_add_f32_dpp bank_mask", d, b); run<88>("v_fma_f32 clamp", d, b);
    run<89>("v_mul_f32 omod", d, b); run<90>("v_xad_u32", d, b); run<91>("v_min_u32", d, b); run<92>("v_ldexp_f32", d, b);
  }
}
