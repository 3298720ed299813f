// Compositing backward, 16-lane groups ("G16"): the vjp of gsplat.rasterize_to_pixels (IDX:14279)
// for the fused pipeline.  Replaces the quadrant walk of fused.hip (k_mraster_bwd) on the non-deterministic path.
//
// Why.  k_mraster_bwd lets all 64 lanes of a wave (an 8x8 pixel quadrant) evaluate one splat per trip; a sigma ~ 1 px
// splat reaches ~11 of those 64 pixels, 27 % of the trips find none (profiles/r02_pmc_sq_counters.txt), and the
// per-splat pixel sums cost a 64-lane reduction (DPP reduce-scatter, later 2 f32 MFMAs per splat = 35 % of the kernel).
// Here a wave still owns a quadrant, but its four 16-lane DPP rows each own one 4x4 pixel BLOCK and walk that block's
// OWN list: four different splats are in flight per trip, a splat occupies ~5 of 16 lanes instead of ~11 of 64, and
// the per-(block, entry) sums are a reduction over one DPP row only (22 full-rate VALU ops for 8 sums, no LDS, no
// matrix core).
//
// One wave = one workgroup = one QUADRANT of a tile (k_qraster_bwd below; the first builds of this round had one
// 256-thread workgroup per tile: DESIGN.md section 4 has the history and the measurements).  Per batch:
//   1. the wave scans the tile's list back to front, 64 entries per step (id + the forward's hit word), keeps the
//      entries whose hit nibble for this quadrant is non-zero -- without the forward's masks: whose alpha >= 1/255 disc
//      reaches one of the quadrant's four blocks -- and gathers only their records into LDS;
//   2. four ballots per step append the entry to the lists of the rows (blocks) that composited it;
//   3. row g walks its list: trip k evaluates entry rlist[g][k] on the row's 16 pixels (the reference loop's
//      recurrences: T /= (1 - alpha), v_alpha = T c.v - buf/(1 - alpha)), forms the 6 monomial sums
//      sum_p w {1, lx, ly, lx^2, lx ly, ly^2} (w = vis * v_alpha; lx, ly = pixel - tile centre) and the colour sums
//      sum_p (alpha T) v_c, reduce-scatters them over the row (lane 2j ends with sum j) and STORES them in the pair's
//      own slot pair[row][k] -- every (block, entry) pair is visited exactly once: no read-modify-write, no LDS atomic,
//      nothing to clear;
//   4. lane L adds up the pairs of slots L and L + 64 in row order, turns moments into the gradient row
//      [v_xy | v_conic | v_opacity | v_colour] (the splat's own centre, conic and opacity), and the rows leave as
//      packed 64-byte global atomics, one per (quadrant, entry) with a composited pixel.
// A row list longer than LCAP entries in one batch is walked in rounds of LCAP trips.
#include <stdlib.h>
#include <string.h>

#include "gsloc_common.h"

namespace gsl {

__device__ __forceinline__ float g16_sel(unsigned long long m, float t, float f) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}

// lane-constant row masks (bit = lane; p = lane & 15)
#define G16_M_LO8 0x00FF00FF00FF00FFull  // p < 8
#define G16_M_B2 0x0F0F0F0F0F0F0F0Full   // bit 2 of p clear
#define G16_M_B1 0x3333333333333333ull   // bit 1 of p clear
#define G16_M_B0 0x5555555555555555ull   // bit 0 of p clear

constexpr int G16_NG = GSL_NG;       // pixel groups (walk lists) per quadrant: 4 (4x4 blocks) or 8 (4x2 half blocks)
constexpr int G16_GL = 64 / G16_NG;  // lanes per group
static_assert(G16_NG == 4 || G16_NG == 8, "pixel groups are DPP rows or half rows");

// Reduce-scatter of 8 values over a 16-lane DPP row: returns, in lane p, the row total of value (p >> 1)
// (even and odd lane of a pair hold the same total).  4 + 2 + 1 exchange steps with halving payload + 1 plain add.
//
// Round 4: the first two steps select by BANK (lanes 0-7 | 8-15, then bit 2 of p: banks {0,2} | {1,3}), and a DPP
// instruction takes a bank mask for its destination write.  So instead of two v_cndmask (keep / send) + one v_add_f32_dpp
// per output, the first add writes  x + x[partner]  of one value in every lane and a second add with the complementary
// bank mask overwrites the other banks with the same sum of the other value: 2 instead of 3 four-cycle VALU per output, 12
// instead of 18 for the two steps (every v_cndmask, DPP and SGPR-operand op issues in 4 cycles on gfx950, plain
// fma / add / mul in 2: profiles/r04_valu_issue.txt).  a + b and b + a are the same float: the sums are bit-identical to
// the select form's.  The instructions are inline asm (the compiler folds a v_mov_dpp into an add only with a full bank
// mask), so the "VALU write -> DPP read of the same VGPR needs 2 wait states" hazard is handled here: s_nop 1 in front,
// and every later read is at least three instructions behind its write.
__device__ __forceinline__ float row_scatter8(const float (&v)[8]) {
  float n0, n1, n2, n3, m0, m1;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n\t"   // p <-> 15 - p; lanes 0-7 end with values 0-3,
      "v_add_f32_dpp %1, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %9, %9 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %10, %10 row_mirror row_mask:0xf bank_mask:0xc\n\t"  // ... lanes 8-15 with values 4-7
      "v_add_f32_dpp %1, %11, %11 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %12, %12 row_mirror row_mask:0xf bank_mask:0xc\n\t"
      // (value 7 is padding -- NS = 8 only for the 7 sums of the depth-only body -- so lanes 8-15 of n3 may keep the sums
      // of value 3: they end up in the slot of value 7, which nobody reads)
      "v_add_f32_dpp %4, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"  // p <-> 7 - p inside each half row
      "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"  // banks 1, 3 (bit 2 of p set)
      "v_add_f32_dpp %5, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(m0), "=&v"(m1)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]));
  float keep = g16_sel(G16_M_B1, m0, m1), send = g16_sel(G16_M_B1, m1, m0);
  float r = keep + dpp_get<0x4E>(send);  // quad_perm [2,3,0,1]
  r += dpp_get<0xB1>(r);                 // quad_perm [1,0,3,2]
  return r;
}

// 16 values: lane p ends with the row total of value p.
__device__ __forceinline__ float row_scatter16(const float (&v)[16]) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float keep = g16_sel(G16_M_LO8, v[i], v[i + 8]), send = g16_sel(G16_M_LO8, v[i + 8], v[i]);
    a[i] = keep + dpp_get<0x140>(send);
  }
  float n[4], m[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float keep = g16_sel(G16_M_B2, a[i], a[i + 4]), send = g16_sel(G16_M_B2, a[i + 4], a[i]);
    n[i] = keep + dpp_get<0x141>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float keep = g16_sel(G16_M_B1, n[i], n[i + 2]), send = g16_sel(G16_M_B1, n[i + 2], n[i]);
    m[i] = keep + dpp_get<0x4E>(send);
  }
  float keep = g16_sel(G16_M_B0, m[0], m[1]), send = g16_sel(G16_M_B0, m[1], m[0]);
  return keep + dpp_get<0xB1>(send);
}

// Half-row forms (GSL_NG = 8: a group is the 8 lanes of half a DPP row, two quads).  One bank-masked mirror step inside
// the half row leaves quad 0 with the pair sums of values 0-3 and quad 1 with those of values 4-7 (value 7: padding); two
// butterfly steps inside the quads then give EVERY lane of quad q the group totals of values 4q .. 4q+3, which its first
// lane stores as one float4.  7 + 8 = 15 DPP adds for a group of 8 lanes against 15 instructions for 16 lanes above: per
// trip the same, but the half blocks skip what the 4x4 blocks only half use.
__device__ __forceinline__ float4 half_scatter8(const float (&v)[8]) {
  float n0, n1, n2, n3;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"  // p <-> 7 - p inside each half row
      "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"   // banks 1, 3: the second quad of a group
      "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"  // (the compiler's DPP reads of n0 .. n3 follow; it does not see VALU writes inside an asm block)
      : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]));
  n0 += dpp_get<0x4E>(n0); n1 += dpp_get<0x4E>(n1); n2 += dpp_get<0x4E>(n2); n3 += dpp_get<0x4E>(n3);
  n0 += dpp_get<0xB1>(n0); n1 += dpp_get<0xB1>(n1); n2 += dpp_get<0xB1>(n2); n3 += dpp_get<0xB1>(n3);
  return make_float4(n0, n1, n2, n3);
}

// 16 values over a half row: every lane of quad q ends with the group totals of values 8q .. 8q+7 (lo: the first four).
__device__ __forceinline__ void half_scatter16(const float (&v)[16], float4& lo, float4& hi) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float keep = g16_sel(G16_M_B2, v[i], v[i + 8]), send = g16_sel(G16_M_B2, v[i + 8], v[i]);
    a[i] = keep + dpp_get<0x141>(send);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] += dpp_get<0x4E>(a[i]);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] += dpp_get<0xB1>(a[i]);
  lo = make_float4(a[0], a[1], a[2], a[3]);
  hi = make_float4(a[4], a[5], a[6], a[7]);
}

#ifdef GSL_G16_STATS  // dev build only: trip statistics of the walk (scripts/g16_stats.py)
__device__ unsigned long long g16_stats[8];
#define G16_STAT(i, n) do { unsigned long long n__ = (unsigned long long)(n); if (lane == 0) atomicAdd(&g16_stats[i], n__); } while (0)
#else
#define G16_STAT(i, n) do { } while (0)
#endif

#ifndef GSL_QB
// Staged entries per batch of one quadrant.  Round 3 (the wave scanned the tile's whole list and kept what its quadrant
// touched): 64: 273 us at R, 95: 257, 111: 269, 127: 264.  With the forward's hit lists every scanned entry is staged, a
// scan step is 64 entries and "room for a whole step" made every batch 64 entries whatever the capacity above that -- the
// slots 64..94 only cost LDS (8.4 -> 7.1 KB per wave, 19 -> 23 workgroups per CU): R 182 -> 179 us, X 437 -> 411.  (Filling
// 95-entry batches with a second, partial step was measured too: 198 us -- the gather and the flush run half empty on
// slots 64..94.)  Without hit lists (a caller that passes none) a batch is what one step of 64 list entries yields.
#define GSL_QB 64
#endif
#define GSL_QBS (GSL_QB + 1)  // + the sentinel slot
#ifndef GSL_Q_LCAP
// Trips per list per round (the pair slots of a round: 4 lists x LCAP x 8 floats).  24: 3 KB, what the 64 packed gradient
// rows of the flush need anyway (12-float pitch); the wave's LDS is then 6.1 KB and the registers (78) set the occupancy
// at 6 waves per SIMD.  32 / 40 / 48 measured at R: 178 / 179 / 189 us, 24: 177 (X: 402 / 432 / 458, 24: 392).
#define GSL_Q_LCAP 24
#endif

template <int D, int CG>
struct QStage {
  static constexpr int NV = 6 + CG;
  static constexpr int NS = (NV <= 8) ? 8 : 16;
  static constexpr int A = 6 + D;
  // trips per group per round: NG groups x LCAP x NS floats = 4 KiB
  static constexpr int LCAP = ((NS == 8) ? GSL_Q_LCAP : GSL_Q_LCAP / 2) * 4 / G16_NG;
  float4 s0[GSL_QBS];
  float4 s1[GSL_QBS];  // .w = the entry's absolute list index (bits), not r_cull: the walk tests it against last_ids
  float4 s2[(D >= 3 && CG == D) ? GSL_QBS : 1];
  int32_t id[GSL_QBS];
  // per slot, word w: position in the lists of groups 4w .. 4w+3 (7 bits each; 127 = not in that list)
  uint32_t posn[G16_NG / 4][GSL_QBS];
  alignas(16) float pair[G16_NG * LCAP * NS];  // [group][trip of the round][value]; reused for 64 packed gradient rows
  uint8_t rlist[G16_NG][GSL_QBS + 4];          // per group: slots in walk order, padded with the sentinel slot
  uint8_t nzlist[64];
};

template <int D, int CG>
__device__ __forceinline__ void qraster_bwd_body(
    QStage<D, CG>& sb, const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2,
    const uint4* __restrict__ Qh, const int32_t* __restrict__ flatten_ids, float* __restrict__ vacc, long long rs,
    long long re, int lane, int quad, float px, float py, float tx0, float ty0, int bin_final, float T_init,
    float Bp_init, const float (&vc)[D], const uint32_t* __restrict__ qhits, int n_qhits) {
  constexpr bool RGB = D >= 3;
  constexpr bool DEPTH = (D == 1) || (D == 4);
  constexpr int NV = QStage<D, CG>::NV;
  constexpr int NS = QStage<D, CG>::NS;
  constexpr int A = QStage<D, CG>::A;
  constexpr int LCAP = QStage<D, CG>::LCAP;
  constexpr int NG = G16_NG, GL = G16_GL;
  constexpr int PP = 12;  // pitch of a packed gradient row in LDS (floats): A = 6 + D <= 10
  static_assert(A <= PP && NG * LCAP * NS >= 64 * PP, "64 packed gradient rows reuse the pair slots");
  static_assert(GSL_QB < 127 && GSL_QB >= 64, "7-bit list positions with 127 reserved; a chunk appends up to 64 entries");
  const int grp = lane / GL, p = lane & (GL - 1);
  const float tcx = tx0 + 8.f, tcy = ty0 + 8.f;
  const float lx = px - tcx, ly = py - tcy;
  const float lxx = lx * lx, lxy = lx * ly, lyy = ly * ly;
  float T = T_init;
  float Bp = Bp_init;
  // which lane stores which totals of a trip.  Rows (NG = 4): lane p holds the total of value p >> 1 (NS = 8) or p (16)
  // and stores one float.  Half rows (NG = 8): every lane of quad q holds the totals of the q-th half of the values and
  // the quad's first lane stores them as float4s.
  const int myslot = (NG == 8) ? (p >> 2) * (NS / 2) : ((NS == 8) ? (p >> 1) : p);
  const bool writer = (NG == 8) ? ((p & 3) == 0) : ((NS == 8) ? ((p & 1) == 0) : true);
  float* const mypair = &sb.pair[grp * LCAP * NS + myslot];
  uint8_t* const mylist = sb.rlist[grp];
  const unsigned long long lt = (1ull << lane) - 1ull;
  // group rectangles of this quadrant (geometric fallback when the forward left no hit masks)
  const float qx0 = tx0 + 8.f * (float)(quad & 1), qy0 = ty0 + 8.f * (float)(quad >> 1);
  int grp_final[NG];  // per group: last list index any of its pixels composited (fallback test only)
  {
    int rf = bin_final;
    rf = max(rf, __shfl_xor(rf, 1, 64));
    rf = max(rf, __shfl_xor(rf, 2, 64));
    rf = max(rf, __shfl_xor(rf, 4, 64));
    if (GL == 16) rf = max(rf, __shfl_xor(rf, 8, 64));
#pragma unroll
    for (int g = 0; g < NG; ++g) grp_final[g] = __builtin_amdgcn_readlane(rf, GL * g);
  }
  if (lane == 0) {  // the sentinel record: opacity 0 fails alpha >= 1/255 on every pixel
    sb.s0[GSL_QB] = make_float4(0.f, 0.f, 0.f, 0.f);
    sb.s1[GSL_QB] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RGB && CG == D) sb.s2[GSL_QB] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // What is scanned, back to front, 64 per step: this quadrant's hit list from the forward (qhits: n_qhits entries of
  // group bits << GSL_HIT_SHIFT | list index, every one of them relevant), or -- without it -- the tile's list itself,
  // with the group tests done here.
  // (list positions fit an int: tile_offsets is int32.  readfirstlane: the scan, batch and trip loops below are
  // wave-uniform, and the compiler has to know it to keep their control on the scalar unit)
  int pos = __builtin_amdgcn_readfirstlane(qhits ? n_qhits : (int)re);
  const int pos_end = __builtin_amdgcn_readfirstlane(qhits ? 0 : (int)rs);
  while (pos > pos_end) {
    // ---- stage: scan chunks of 64 until the batch is (nearly) full
    int staged = 0;
    int cnt[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) cnt[g] = 0;
    __syncthreads();  // (one wave: orders the previous batch's LDS reads before these writes)
    while (pos > pos_end && staged <= GSL_QB - 64) {
      long long idx = (long long)(pos - 1 - lane);
      bool in = idx >= (long long)pos_end;
      unsigned nib = 0;
      int gid = 0;
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
      if (in) {
        if (qhits) {
          unsigned e = qhits[idx];
          nib = e >> GSL_HIT_SHIFT;
          idx = (long long)(e & GSL_HIT_INDEX_MASK);
          gid = flatten_ids[idx];
          load_record(Q0, Q1, Q2, Qh, gid, RGB && CG == D, r0, r1, r2);
        } else {
          gid = flatten_ids[idx];
          load_record(Q0, Q1, Q2, Qh, gid, RGB && CG == D, r0, r1, r2);
          if (r1.w >= 0.f) {
            float rr = r1.w * r1.w;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
              // group g: block b = g (NG = 4) or g >> 1 with its upper / lower two pixel rows (NG = 8)
              const int b = (NG == 8) ? (g >> 1) : g;
              const float cy = (NG == 8) ? 2.f * (float)(g & 1) + 1.f : 2.f, hy = (NG == 8) ? 0.5f : 1.5f;
              float ex = fmaxf(fabsf(r0.x - (qx0 + 4.f * (float)(b & 1) + 2.f)) - 1.5f, 0.f);
              float ey = fmaxf(fabsf(r0.y - (qy0 + 4.f * (float)(b >> 1) + cy)) - hy, 0.f);
              bool h = (ex * ex + ey * ey <= rr) && ((int)idx <= grp_final[g]);
              nib |= (h ? 1u : 0u) << g;
            }
          }
        }
      }
      unsigned long long R = __ballot(nib != 0);
      int slot = staged + __popcll(R & lt);
      // positions of this entry in the group lists, 7 bits each; 127 = "not in that list" (no round ever reaches it: a
      // list holds at most GSL_QB entries), so the gather needs no hit bit next to the position
      unsigned pack[NG / 4];
#pragma unroll
      for (int w = 0; w < NG / 4; ++w) pack[w] = 0x0FFFFFFFu;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        unsigned long long Bg = __ballot(nib & (1u << g));
        if (__builtin_amdgcn_inverse_ballot_w64(Bg)) {  // (exec = the ballot: not a second compare of the same bit)
          int q = cnt[g] + __popcll(Bg & lt);
          sb.rlist[g][q] = (uint8_t)slot;
          pack[g >> 2] ^= (127u ^ (unsigned)q) << (7 * (g & 3));
        }
        cnt[g] += __popcll(Bg);
      }
      if (__builtin_amdgcn_inverse_ballot_w64(R)) {
        sb.id[slot] = gid;
        sb.s0[slot] = r0;
        // the conic as the FORWARD stages it (times log2 e, diagonal halved: raster_px.hip praster_walk), so that a trip
        // evaluates alpha with the forward's very operations and both sides take the same alpha >= 1/255 decision for
        // every (pixel, entry) pair; the flush scales back
        sb.s1[slot] = make_float4(r1.x * (0.5f * GSL_LOG2E), r1.y * GSL_LOG2E, r1.z * (0.5f * GSL_LOG2E),
                                  __int_as_float((int)idx));
        if (RGB && CG == D) sb.s2[slot] = r2;
#pragma unroll
        for (int w = 0; w < NG / 4; ++w) sb.posn[w][slot] = pack[w];
      }
      staged += __popcll(R);
      pos -= 64;
    }
    if (staged == 0) continue;
    int kmax = 0, cnt_tot = 0, cnt_my = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      kmax = max(kmax, cnt[g]);
      cnt_tot += cnt[g];
      cnt_my = (grp == g) ? cnt[g] : cnt_my;
    }
    G16_STAT(6, kmax);
    G16_STAT(7, cnt_tot);
    (void)cnt_tot;
    for (int k = cnt_my + p; k < kmax + 3; k += GL) mylist[k] = (uint8_t)GSL_QB;  // sentinel padding (+3: read-ahead)
    __syncthreads();
    constexpr int NU = (GSL_QB + 63) / 64;  // staged slots per lane (1 since a batch holds 64 entries)
    float mo[NU][NV];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int q = 0; q < NV; ++q) mo[u][q] = 0.f;
    // One trip: the gradient sums of list entry (slot t, records c0 / c1) over this lane's pixel group, r (and r2)
    // = what this lane stores of them.
    auto trip = [&](int t, const float4& c0, const float4& c1, float4& r, float4& r2) {
      float dx = c0.x - px, dy = c0.y - py;
      float sigma = fmaf(c1.y * dx, dy, fmaf(c1.x * dx, dx, c1.z * dy * dy));  // log2(e) sigma, the forward's expression
      float vis = __builtin_amdgcn_exp2f(-sigma);
      float opv = c0.w * vis;
      float alpha = fminf(GSL_ALPHA_MAX, opv);
      unsigned long long validm = __ballot(__float_as_int(c1.w) <= bin_final) & __ballot(sigma >= 0.f) &
                                  __ballot(alpha >= GSL_ALPHA_MIN);
      G16_STAT(0, 1);
      r = make_float4(0.f, 0.f, 0.f, 0.f);
      r2 = r;
      if (validm) {
        G16_STAT(1, 1);
#ifdef GSL_G16_STATS
        {
          int groups_hit = 0;
          for (int g = 0; g < NG; ++g) groups_hit += ((validm >> (GL * g)) & ((1ull << GL) - 1ull)) != 0;
          G16_STAT(3, groups_hit);
        }
#endif
        G16_STAT(5, __popcll(validm));
        unsigned long long capm = __ballot(opv <= GSL_ALPHA_MAX);
        float am = g16_sel(validm, alpha, 0.f);
        float ra = __builtin_amdgcn_rcpf(1.f - am);
        T *= ra;
        float fac = am * T;
        float cdot;
        if (CG == D) {
          cdot = 0.f;
          if (RGB) {
            float4 q2 = sb.s2[t];
            cdot = q2.x * vc[0] + q2.y * vc[1] + q2.z * vc[2];
          }
          if (DEPTH) cdot += c0.z * vc[D - 1];
        } else {
          cdot = c0.z * vc[D - 1];
        }
        float v_alpha = T * cdot - ra * Bp;
        Bp += fac * cdot;
        float w = g16_sel(validm & capm, vis, 0.f) * v_alpha;
        float val[NS];
        val[0] = w; val[1] = w * lx; val[2] = w * ly; val[3] = w * lxx; val[4] = w * lxy; val[5] = w * lyy;
        if (CG == D) {
#pragma unroll
          for (int ch = 0; ch < D; ++ch) val[6 + ch] = fac * vc[ch];
        } else {
          val[6] = fac * vc[D - 1];
        }
#pragma unroll
        for (int q = NV; q < NS; ++q) val[q] = 0.f;
        static_assert(NS != 8 || NV == 7, "the 8-value scatters treat value 7 as padding");
        if (NG == 8) {
          if (NS == 8) r = half_scatter8(reinterpret_cast<const float(&)[8]>(val));
          else half_scatter16(reinterpret_cast<const float(&)[16]>(val), r, r2);
        } else {
          if (NS == 8) r.x = row_scatter8(reinterpret_cast<const float(&)[8]>(val));
          else r.x = row_scatter16(reinterpret_cast<const float(&)[16]>(val));
        }
      }
    };
    auto put = [&](float* dst, const float4& r, const float4& r2) {
      if (NG == 8) {
        *reinterpret_cast<float4*>(dst) = r;
        if (NS == 16) *reinterpret_cast<float4*>(dst + 4) = r2;
      } else {
        *dst = r.x;
      }
    };
    for (int k0 = 0; k0 < kmax; k0 += LCAP) {
      const int k1 = min(k0 + LCAP, kmax);
      // Two trips per turn of the loop, records in two register sets that swap roles (round 4: a one-trip body rotated
      // its look-ahead through seven v_mov per trip -- 12 % of its VALU slots).  The list index runs two entries and the
      // records one entry ahead of the trip; a trip's result is stored at the start of the next one (the LDS queue is in
      // order: stores before the next loads).  An odd count makes the second trip of the last turn an extra one: it
      // meets sentinel slots (lists are padded to kmax + 3), finds no valid pixel and stores zeros in a pair slot of its
      // own (the trip count of a round is even whenever the round is full) that nobody gathers.
      int t0 = mylist[k0], t1 = mylist[k0 + 1];
      float4 qa0 = sb.s0[t0], qa1 = sb.s1[t0];
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), ra2 = ra, rb = ra, rb2 = ra;
      float* dst = mypair;
      for (int k = k0; k < k1; k += 2) {
        if (k > k0 && writer) put(dst - NS, rb, rb2);
        const int t2 = mylist[k + 2];
        const float4 qb0 = sb.s0[t1], qb1 = sb.s1[t1];
        trip(t0, qa0, qa1, ra, ra2);
        if (writer) put(dst, ra, ra2);
        const int t3 = mylist[k + 3];
        qa0 = sb.s0[t2];
        qa1 = sb.s1[t2];
        trip(t1, qb0, qb1, rb, rb2);
        t0 = t2;
        t1 = t3;
        dst += 2 * NS;
      }
      if (writer) put(dst - NS, rb, rb2);
      __syncthreads();
      // gather: lane L adds up the pairs of slots L and L + 64 that lie in this round, in group order
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int slot = lane + 64 * u;
        if (slot < staged) {
#pragma unroll
          for (int w = 0; w < NG / 4; ++w) {
            unsigned pk = sb.posn[w][slot];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int g = 4 * w + g4;
              const unsigned q = ((pk >> (7 * g4)) & 127u) - (unsigned)k0;
              if (q < (unsigned)LCAP) {
                const float4* src = reinterpret_cast<const float4*>(&sb.pair[(g * LCAP + q) * NS]);
                float4 a = src[0], c = src[1];
                mo[u][0] += a.x; mo[u][1] += a.y; mo[u][2] += a.z; mo[u][3] += a.w;
                mo[u][4] += c.x; mo[u][5] += c.y; mo[u][6] += c.z;
                if (NV > 7) mo[u][7 < NV ? 7 : 0] += c.w;
                if (NS == 16) {
                  float4 e = src[2];
                  if (NV > 8) mo[u][8 < NV ? 8 : 0] += e.x;
                  if (NV > 9) mo[u][9 < NV ? 9 : 0] += e.y;
                }
              }
            }
          }
        }
      }
      __syncthreads();
    }
    // ---- moments -> gradient rows, 64 slots at a time (the packed rows reuse the pair slots)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int slot = lane + 64 * u;
      bool nz = false;
      float row[A];
#pragma unroll
      for (int k = 0; k < A; ++k) row[k] = 0.f;
      if (slot < staged) {
        {  // any sum non-zero?  (one compare on the OR of the bit patterns, sign bit shifted out: -0.0 is zero)
          unsigned bits = 0u;
#pragma unroll
          for (int k = 0; k < NV; ++k) bits |= __float_as_uint(mo[u][k]);
          nz = (bits << 1) != 0u;
        }
        if (nz) {
          float4 r0 = sb.s0[slot], r1 = sb.s1[slot];
          float X = r0.x - tcx, Y = r0.y - tcy, S = mo[u][0];
          float Sx = X * S - mo[u][1], Sy = Y * S - mo[u][2];
          float Sxx = X * (X * S - 2.f * mo[u][1]) + mo[u][3];
          float Sxy = X * (Y * S - mo[u][2]) - Y * mo[u][1] + mo[u][4];
          float Syy = Y * (Y * S - 2.f * mo[u][2]) + mo[u][5];
          float no = -r0.w;
          r1.x *= 2.f / GSL_LOG2E; r1.y *= 1.f / GSL_LOG2E; r1.z *= 2.f / GSL_LOG2E;  // (staged times log2 e: see the gather)
          row[0] = no * (r1.x * Sx + r1.y * Sy);
          row[1] = no * (r1.y * Sx + r1.z * Sy);
          row[2] = 0.5f * no * Sxx;
          row[3] = no * Sxy;
          row[4] = 0.5f * no * Syy;
          row[5] = S;
          if (CG == D) {
#pragma unroll
            for (int ch = 0; ch < D; ++ch) row[6 + ch] = mo[u][6 + ch];
          } else {
            row[6 + D - 1] = mo[u][6];
          }
        }
      }
      float* packed = sb.pair;
      if (nz) {
#pragma unroll
        for (int k = 0; k < A; ++k) packed[lane * PP + k] = row[k];
      }
      unsigned long long mask = __ballot(nz);
      int cntz = __popcll(mask);
      if (nz) sb.nzlist[__popcll(mask & lt)] = (uint8_t)lane;
      __syncthreads();
      // A lanes per row, 64 / A rows per atomic instruction (6 rows of 10 floats in "RGB+ED", 9 of 7 in "ED"; a row's
      // lanes stay contiguous: one 64-byte line per row)
      constexpr int RPI = 64 / A;
      const int arow = lane / A, f = lane - arow * A;
      for (int i0 = 0; i0 < cntz; i0 += RPI) {
        int gi = i0 + arow;
        if (gi < cntz && arow < RPI) {
          int sl = sb.nzlist[gi];
          // (byte offset in 32 bits -- Gaussian ids are below 2^26, include/gsloc_hip.h -- so the atomic takes the scalar
          // base + a 32-bit lane offset instead of three instructions of 64-bit address arithmetic)
          const unsigned off = ((unsigned)sb.id[sl + 64 * u] << 6) | ((unsigned)f << 2);
          atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(vacc) + off), packed[sl * PP + f]);
        }
      }
      __syncthreads();
    }
  }
}

// One work item: a (tile, quadrant) pair, or LONG a (segment slot, quadrant) pair.  b = the item's index of n_items.
template <int D, bool ED, int CG, bool LONG>
__device__ __forceinline__ void qraster_bwd_item(
    QStage<D, CG>& sb, int b, int n_items, const float4* __restrict__ Q0, const float4* __restrict__ Q1,
    const float4* __restrict__ Q2, int W, int H, int tile_w, int ty0, const int32_t* __restrict__ tile_offsets,
    const int32_t* __restrict__ flatten_ids, long long capacity, const float* __restrict__ render,
    const float* __restrict__ alphas, const int32_t* __restrict__ last_ids, const float* __restrict__ v_render,
    const float* __restrict__ v_alphas, float* __restrict__ vacc, int row0, int row1, const uint4* __restrict__ Qh,
    const uint32_t* __restrict__ isect_hits, int32_t* __restrict__ isect_hit_counts, int long_min, LongWs lw,
    int32_t* __restrict__ rgb_flag, int32_t* __restrict__ clear_counts, int32_t* __restrict__ clear_state) {
  // (a forward that sorted its own bins could not clear the tile counters, raster_px.hip k_praster_fwd SORT: the first
  // launch of the backward does, one lane per tile)
  if (!LONG && (CG == 1 || D == 3) && clear_counts && (b & 3) == 0 && threadIdx.x == 0) {
    clear_counts[ty0 * tile_w + (b >> 2)] = 0;  // (any one-to-one map of items to tiles does for clearing)
    if (b == 0 && clear_state) *clear_state = 0;
  }
  // (LONG: items are (tile, segment) slots; otherwise the (tile, quadrant) pairs are dealt to the XCDs in contiguous
  // spans: gsloc_common.h)
  const int wi = LONG ? b : xcd_span_item(b, n_items);
  const int quad = wi & 3, item = wi >> 2;
  int tile, sgm = 0, gseg = 0;
  if (LONG) {
    gseg = item;
    if (gseg >= lw.n_seg[0]) return;
    tile = lw.seg_tile[gseg];
    sgm = lw.seg_idx[gseg];
    if (tile < 0) return;  // (segments that did not fit the workspace: flagged by k_long_map)
  } else {
    tile = ty0 * tile_w + item;
  }
  int tyi = tile / tile_w, txi = tile - tyi * tile_w;
  int lane = threadIdx.x, grp = lane >> 4, p = lane & 15;
  int j = txi * 16 + (quad & 1) * 8 + (grp & 1) * 4 + (p & 3);
  int i = tyi * 16 + (quad >> 1) * 8 + (grp >> 1) * 4 + (p >> 2);
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W) && (i >= row0) && (i < row1);

  long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
  if (re > capacity) re = capacity;
  if (rs >= re) return;
  if (!LONG && long_min > 0 && re - rs > long_min) return;
  if (LONG) {
    rs += (long long)sgm * GSL_SEG;
    re = min(rs + (long long)GSL_SEG, re);
  }
  size_t pid = inside ? ((size_t)i * W + j) : 0;
  int bin_final = inside ? last_ids[pid] : -1;
  // nothing behind the last entry a pixel of this quadrant composited: start there (and leave if that is before rs)
  int quad_final = bin_final;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) quad_final = max(quad_final, __shfl_xor(quad_final, o, 64));
  if ((long long)quad_final + 1 < re) re = (long long)quad_final + 1;
  if (rs >= re) return;
  float vc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) vc[k] = inside ? v_render[pid * D + k] : 0.f;
  if (D == 4) {
    bool rgb_grad = (vc[0] != 0.f) || (vc[1] != 0.f) || (vc[2] != 0.f);
    bool any_rgb = __ballot(rgb_grad) != 0ull;
    if ((CG == 1) == any_rgb) {  // the other kernel's quadrant
      if (CG == 1 && rgb_flag && lane == 0) atomicOr(rgb_flag, 1);
      return;
    }
  }
  float Aimg = inside ? alphas[pid] : 0.f;
  float T_final = 1.f - Aimg;
  float va = inside ? v_alphas[pid] : 0.f;
  if (ED && inside) {
    float dn = render[pid * D + (D - 1)];
    float vd = vc[D - 1];
    if (Aimg >= 1e-10f) va += -vd * dn / Aimg;
    vc[D - 1] = vd / fmaxf(Aimg, 1e-10f);
  }
  float T_init = T_final, Bp_init = -T_final * va;
  if (LONG) {
    int nseg = lw.seg_cnt[gseg];
    long long seg_end = min(tile_offsets[tile] + (long long)(sgm + 1) * GSL_SEG, (long long)tile_offsets[tile + 1]);
    int pslot = quad * 64 + lane;  // the forward's thread index of this pixel
    if (inside && (long long)bin_final >= seg_end) {
      T_init = fabsf(lw.Tend[(size_t)gseg * 256 + pslot]);
      // colour partials of the later segments up to the one the pixel stopped in, in segment order; eight segments'
      // records are requested together (a fringe pixel of a pile composites through tens of segments)
      const float4* C4 = reinterpret_cast<const float4*>(lw.C);
      bool stop = false;
      for (int s0 = sgm + 1; s0 < nseg && !stop; s0 += 8) {
        float t8[8];
        float4 c8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          size_t slot = (size_t)(gseg - sgm + min(s0 + u, nseg - 1)) * 256 + pslot;
          t8[u] = lw.Tend[slot];
          c8[u] = C4[slot];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (stop || s0 + u >= nseg) continue;
          if (t8[u] == 2.f) { stop = true; continue; }
          float cc[4] = {c8[u].x, c8[u].y, c8[u].z, c8[u].w};
          float dot = 0.f;
#pragma unroll
          for (int k = 0; k < D; ++k) dot += vc[k] * cc[k];
          Bp_init += dot;
          // as k_long_combine: a pixel that stopped INSIDE this segment composited nothing behind it, whatever the later
          // segments' own restart products say (their T0 may round to the other side of 1e-4)
          if (t8[u] < 0.f) stop = true;
        }
      }
    }
  }
  // this quadrant's hit list: at 4 x (start of the list or segment) + quadrant x its length
  const uint32_t* qh = nullptr;
  int nqh = 0;
  if (isect_hits) {
    long long ls = tile_offsets[tile], le = tile_offsets[tile + 1];
    if (le > capacity) le = capacity;
    if (LONG) {
      ls += (long long)sgm * GSL_SEG;
      le = min(ls + (long long)GSL_SEG, le);
      nqh = lw.seg_qcnt[gseg * 4 + quad];
    } else {
      nqh = isect_hit_counts[tile * 4 + quad];
    }
    qh = isect_hits + 4 * ls + (long long)quad * (le - ls);
    if (nqh == 0) return;
  }
  qraster_bwd_body<D, CG>(sb, Q0, Q1, Q2, Qh, flatten_ids, vacc, rs, re, lane, quad, px, py, (float)(txi * 16),
                          (float)(tyi * 16), bin_final, T_init, Bp_init, vc, qh, nqh);
}

// One wave per workgroup.  The grid is one workgroup per item for every launch but the full-colour second one of
// "RGB+ED" (CG = D = 4), which gets a capped grid whose workgroups walk the items with a stride: that launch finds its
// flag clear on every iteration of GsplatLoc's loss, and 12 900 workgroups that load one word and leave cost the graph
// about 2.5 us more than 2 048 do (the stage of both backward launches at R: 184 -> 181.6 us).
template <int D, bool ED, int CG, bool LONG>
__global__ __launch_bounds__(64) void k_qraster_bwd(
    const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2, int W, int H,
    int tile_w, int ty0, const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    long long capacity, const float* __restrict__ render, const float* __restrict__ alphas,
    const int32_t* __restrict__ last_ids, const float* __restrict__ v_render, const float* __restrict__ v_alphas,
    float* __restrict__ vacc, int row0, int row1, const uint4* __restrict__ Qh,
    const uint32_t* __restrict__ isect_hits, int32_t* __restrict__ isect_hit_counts, int long_min, LongWs lw,
    int32_t* __restrict__ rgb_flag, int32_t* __restrict__ clear_counts, int32_t* __restrict__ clear_state, int n_items) {
  __shared__ QStage<D, CG> sb;
  // rgb_flag (may be NULL; D = 4 only): raised by the depth-only kernel when it leaves a quadrant to the full-colour
  // kernel, cleared by the compositing forward.  The full-colour launch returns on a clear flag after one load per
  // workgroup instead of reading its 64 pixels' upstream gradient to find out that it has nothing to do (GsplatLoc's
  // loss: always).  A stale raised flag (a second backward after the same forward) only costs that reading.
  if (D == 4 && CG == D && rgb_flag && *rgb_flag == 0) return;
  for (int b = (int)blockIdx.x; b < n_items; b += (int)gridDim.x) {
    qraster_bwd_item<D, ED, CG, LONG>(sb, b, n_items, Q0, Q1, Q2, W, H, tile_w, ty0, tile_offsets, flatten_ids, capacity,
                                      render, alphas, last_ids, v_render, v_alphas, vacc, row0, row1, Qh, isect_hits,
                                      isect_hit_counts, long_min, lw, rgb_flag, clear_counts, clear_state);
    __syncthreads();  // (the next item stages into the same LDS)
  }
}

}  // namespace gsl

#ifdef GSL_G16_STATS
extern "C" int gsl_g16_stats(unsigned long long* host_out, int reset) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gsl::g16_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gsl::g16_stats), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#endif

extern "C" int32_t* gsl_fused_bin_state(void* ws, int n_tiles);  // fused.hip: the state word inside ws

// Launch of the G16 backward (called by gsl_fused_raster_bwd in fused.hip for the non-deterministic path).
extern "C" int gsl_g16_raster_bwd_launch(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                         int height, int tile_w, int ty0, int ty1, const int32_t* tile_offsets,
                                         const int32_t* flatten_ids, int64_t capacity, const float* render,
                                         const float* alphas, const int32_t* last_ids, const float* v_render,
                                         const float* v_alphas, float* vacc, int row0, int row1, const void* Qh,
                                         const uint32_t* isect_hits, const int32_t* isect_hit_counts, int long_min,
                                         void* long_ws, int max_seg, int rgb_flag_index, void* clear_ws,
                                         void* stream) {
  // long_ws == NULL: the tiles of the strip (those longer than long_min, if > 0, are skipped);
  // long_ws != NULL: only the (tile, segment) pairs the forward's long-list pass listed there
  hipStream_t st = (hipStream_t)stream;
  int nblk = long_ws ? max_seg : (ty1 - ty0) * tile_w;
  gsl::LongWs lw = gsl::long_ws_views(long_ws ? long_ws : (void*)0, long_ws ? max_seg : 0);
  const bool lng = long_ws != nullptr;
  // (the flag sits behind the 4 n_tiles hit-list lengths; the long-list launches, which have no lengths array, go without)
  int32_t* rgb_flag = isect_hit_counts ? const_cast<int32_t*>(isect_hit_counts) + rgb_flag_index : nullptr;
  int32_t* clear_counts = (int32_t*)clear_ws;
  int32_t* clear_state = clear_ws ? gsl_fused_bin_state(clear_ws, rgb_flag_index / 4) : nullptr;  // (4 n_tiles)
#define CALL_Q(DD, EE, CC)                                                                                   \
  do {                                                                                                       \
    const int n_items = 4 * nblk;                                                                            \
    const int grid = ((DD) == 4 && (CC) == 4 && rgb_flag && n_items > 2048) ? 2048 : n_items;                \
    if (lng)                                                                                                 \
      hipLaunchKernelGGL((gsl::k_qraster_bwd<DD, EE, CC, true>), dim3(grid), dim3(64), 0, st, (const float4*)Q0, \
                         (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,     \
                         flatten_ids, (long long)capacity, render, alphas, last_ids, v_render, v_alphas, vacc, \
                         row0, row1, (const uint4*)Qh, isect_hits, const_cast<int32_t*>(isect_hit_counts), long_min, lw, rgb_flag, clear_counts, clear_state, n_items);                   \
    else                                                                                                     \
      hipLaunchKernelGGL((gsl::k_qraster_bwd<DD, EE, CC, false>), dim3(grid), dim3(64), 0, st, (const float4*)Q0, \
                         (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,     \
                         flatten_ids, (long long)capacity, render, alphas, last_ids, v_render, v_alphas, vacc, \
                         row0, row1, (const uint4*)Qh, isect_hits, const_cast<int32_t*>(isect_hit_counts), long_min, lw, rgb_flag, clear_counts, clear_state, n_items);                   \
  } while (0)
  if (channels == 1) { if (ed) CALL_Q(1, true, 1); else CALL_Q(1, false, 1); }
  else if (channels == 3) { CALL_Q(3, false, 3); }
  else if (channels == 4) {
    if (ed) { CALL_Q(4, true, 1); CALL_Q(4, true, 4); }
    else { CALL_Q(4, false, 1); CALL_Q(4, false, 4); }
  }
  else return GSL_ERR_BAD_ARG;
#undef CALL_Q
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
