// Compositing backward, 16-lane groups ("G16"): the vjp of gsplat.rasterize_to_pixels (IDX:14279)
// for the fused pipeline.  Replaces the quadrant walk of fused.hip (k_mraster_bwd) on the non-deterministic path.
//
// Why.  k_mraster_bwd lets all 64 lanes of a wave (an 8x8 pixel quadrant) evaluate one splat per trip; a sigma ~ 1 px
// splat reaches ~11 of those 64 pixels, 27 % of the trips find none (profiles/r02_pmc_sq_counters.txt), and the
// per-splat pixel sums cost a 64-lane reduction (DPP reduce-scatter, later 2 f32 MFMAs per splat = 35 % of the kernel).
// Here a wave still owns a quadrant, but its four 16-lane DPP rows each own one 4x4 pixel BLOCK and walk that block's
// OWN list: four different splats are in flight per trip, a splat occupies ~5 of 16 lanes instead of ~11 of 64, and
// the per-(block, splat) sums are a reduction over one DPP row only (22 full-rate VALU ops for 8 sums, no LDS, no
// matrix core).
//
// Per workgroup (one 16x16 tile), per batch of GSL_GB list entries (back to front):
//   1. stage the records in LDS (gathered into registers during the previous batch's walk);
//   2. every staging thread tests its splat's alpha >= 1/255 disc against the 16 blocks' pixel centres (exact
//      circle/rectangle distance, and nothing behind the block's last composited entry) -> 16 ballots -> order-
//      preserving per-block lists blist[16][.] (u8 slots);
//   3. wave = quadrant, row g = block: trip k evaluates entry blist[b][k] on the row's 16 pixels (same recurrences as
//      the reference loop: T /= (1 - alpha), v_alpha = T c.v - buf/(1 - alpha)), forms the 6 monomial sums
//      sum_p w {1, lx, ly, lx^2, lx ly, ly^2} (w = vis * v_alpha; lx, ly = pixel - tile centre) and the colour sums
//      sum_p (alpha T) v_c, reduce-scatters them over the row (lane 2j ends with sum j) and STORES them in the pair's
//      own slot pair[block][k] -- every (block, entry) pair is visited exactly once, so there is no read-modify-write,
//      no LDS atomic and nothing to clear (first build: per-wave moment rows with ds_add_f32 whenever two rows of a
//      wave held the same entry in one trip -- 29 % of the trips -- 536 us against the 418 us it had to beat);
//   4. the thread that staged an entry knows its position in each block list: it adds up the entry's pairs in block
//      order (fixed order: the tile's contribution is deterministic), turns moments into the gradient row
//      [v_xy | v_conic | v_opacity | v_colour] (the splat's own centre, conic and opacity), and the rows leave as
//      packed 64-byte global atomics exactly as before.
// A block list longer than LCAP entries in one batch (large splats) is walked in rounds of LCAP trips.
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

// Reduce-scatter of 8 values over a 16-lane DPP row: returns, in lane p, the row total of value (p >> 1)
// (even and odd lane of a pair hold the same total).  4 + 2 + 1 exchange steps with halving payload + 1 plain add.
__device__ __forceinline__ float row_scatter8(const float (&v)[8]) {
  float n[4], m[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float keep = g16_sel(G16_M_LO8, v[i], v[i + 4]), send = g16_sel(G16_M_LO8, v[i + 4], v[i]);
    n[i] = keep + dpp_get<0x140>(send);  // row_mirror: p <-> 15 - p
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float keep = g16_sel(G16_M_B2, n[i], n[i + 2]), send = g16_sel(G16_M_B2, n[i + 2], n[i]);
    m[i] = keep + dpp_get<0x141>(send);  // row_half_mirror: p <-> 7 - p inside each half row
  }
  float keep = g16_sel(G16_M_B1, m[0], m[1]), send = g16_sel(G16_M_B1, m[1], m[0]);
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

#define GSL_GB 255   // list entries per batch
#define GSL_GBS 256  // staged record slots: the batch + one SENTINEL record (opacity 0: fails alpha >= 1/255 on every
                     // pixel) that pads the block lists, so that a trip needs no "is this row still active" logic

#ifdef GSL_G16_STATS  // dev build only: trip statistics of the walk (scripts/g16_stats.py)
__device__ unsigned long long g16_stats[8];
#define G16_STAT(i, n) do { unsigned long long n__ = (unsigned long long)(n); if (lane == 0) atomicAdd(&g16_stats[i], n__); } while (0)
#else
#define G16_STAT(i, n) do { } while (0)
#endif

template <int D, int CG>
struct GStage {
  static constexpr int NV = 6 + CG;                 // sums per (block, entry): 6 monomial + CG colour
  static constexpr int NS = (NV <= 8) ? 8 : 16;     // slots of the row reduce-scatter
  static constexpr int A = 6 + D;                   // gradient row [v_xy 2 | v_conic 3 | v_opacity 1 | v_colour D]
  static constexpr int LCAP = (NS == 8) ? 48 : 24;  // trips per block per round (pair slots)
  float4 s0[GSL_GBS];
  float4 s1[GSL_GBS];
  float4 s2[(D >= 3 && CG == D) ? GSL_GBS : 1];
  int32_t id[GSL_GBS];
  alignas(16) float pair[16 * LCAP * NS];  // [block][trip of the round][slot]; reused for the packed gradient rows
  uint8_t blist[16][GSL_GBS];              // per block: batch slots in walk order, padded with the sentinel slot
  int bcnt[4][16];                         // [staging wave][block]
  int btot[16];                            // per block: entries of this batch
  uint16_t list[4][64];                    // flush: non-zero slots per wave
};

template <int D, int CG>
__device__ __forceinline__ void graster_bwd_body(
    GStage<D, CG>& sb, const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2,
    const uint4* __restrict__ Qh, const int32_t* __restrict__ flatten_ids, float* __restrict__ vacc, long long rs,
    long long re, int tid, float px, float py, float tx0, float ty0, bool inside, int bin_final, float T_init,
    float Bp_init, const float (&vc)[D], const int* __restrict__ bfinal, const uint16_t* __restrict__ isect_hits) {
  constexpr bool RGB = D >= 3;
  constexpr bool DEPTH = (D == 1) || (D == 4);
  constexpr int NV = GStage<D, CG>::NV;
  constexpr int NS = GStage<D, CG>::NS;
  constexpr int A = GStage<D, CG>::A;
  constexpr int LCAP = GStage<D, CG>::LCAP;
  static_assert(16 * LCAP * NS >= GSL_GBS * 16, "the packed gradient rows reuse the pair slots");
  const int lane = tid & 63, wv = tid >> 6, grp = lane >> 4, p = lane & 15;
  const int blk = 4 * wv + grp;  // block id = 4 * quadrant + row (the bit numbering of the forward's hit masks)
  const float tcx = tx0 + 8.f, tcy = ty0 + 8.f;
  const float lx = px - tcx, ly = py - tcy;
  const float lxx = lx * lx, lxy = lx * ly, lyy = ly * ly;
  // T_init: transmittance after the last entry of [rs, re) this pixel composited; Bp_init: -T_final v_alpha + the
  // colour partials (dotted with v_colour) of everything this pixel composited BEHIND re (0 entries for a whole list)
  float T = T_init;
  float Bp = Bp_init;
  (void)inside;  // a pixel outside the image / pixel-row window has bin_final = -1: no entry passes its age test
  // which slot of the reduce-scatter this lane ends up with, and whether it stores it
  const int myslot = (NS == 8) ? (p >> 1) : p;
  const bool writer = (NS == 8) ? ((p & 1) == 0) : true;
  float* const mypair = &sb.pair[blk * LCAP * NS + myslot];
  uint8_t* const mylist = sb.blist[blk];
  if (tid == 0) {  // the sentinel record (never overwritten: batches stage slots 0 .. GSL_GB - 1)
    sb.s0[GSL_GB] = make_float4(0.f, 0.f, 0.f, 0.f);
    sb.s1[GSL_GB] = make_float4(0.f, 0.f, 0.f, -1.f);
    if (RGB && CG == D) sb.s2[GSL_GB] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  const int nb = (int)((re - rs + GSL_GB - 1) / GSL_GB);
  // records of batch b + 1 are gathered into registers while batch b is walked
  int pg = 0;
  unsigned ph = 0;
  float4 pr0 = make_float4(0.f, 0.f, 0.f, 0.f), pr1 = make_float4(0.f, 0.f, 0.f, -1.f), pr2 = pr0;
  auto gather = [&](int b) {
    long long bend = re - 1 - (long long)b * GSL_GB;
    int bsize = (int)min((long long)GSL_GB, bend + 1 - rs);
    if (tid < bsize) {
      pg = flatten_ids[bend - tid];
      if (isect_hits) ph = isect_hits[bend - tid];
      load_record(Q0, Q1, Q2, Qh, pg, RGB && CG == D, pr0, pr1, pr2);
    }
  };
  gather(0);

  for (int b = 0; b < nb; ++b) {
    const long long bend = re - 1 - (long long)b * GSL_GB;  // slot t <-> absolute list index bend - t (back to front)
    const int bsize = (int)min((long long)GSL_GB, bend + 1 - rs);
    __syncthreads();  // the previous batch's flush has read pair / id / s0 / s1 / list
    const bool staged = tid < bsize;
    const float4 r0 = pr0, r1 = pr1;
    if (staged) {
      sb.id[tid] = pg;
      sb.s0[tid] = pr0;
      sb.s1[tid] = pr1;
      if (RGB && CG == D) sb.s2[tid] = pr2;
    }
    // which of the 16 blocks walk the staged entry?  Exactly those that composited it on some pixel in the forward
    // (isect_hits); without the forward's masks: alpha >= 1/255 disc (radius r1.w) against the rectangle of the
    // block's pixel centres, and nothing behind the block's last composited entry.  Block q = 4 * quadrant + row sits
    // at (2 (quadrant & 1) + (row & 1), 2 (quadrant >> 1) + (row >> 1)) of the 4x4 blocks of the tile.
    unsigned hits = 0;
    if (isect_hits) {
      hits = staged ? ph : 0u;
    } else if (staged && r1.w >= 0.f) {
      const float rr = r1.w * r1.w;
      const int age = (int)(bend - tid);
      float ddx[4], ddy[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float ex = fmaxf(fabsf(r0.x - (tx0 + 4.f * (float)q + 2.f)) - 1.5f, 0.f);
        float ey = fmaxf(fabsf(r0.y - (ty0 + 4.f * (float)q + 2.f)) - 1.5f, 0.f);
        ddx[q] = ex * ex;
        ddy[q] = rr - ey * ey;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int bx = 2 * ((q >> 2) & 1) + (q & 1), by = 2 * (q >> 3) + ((q >> 1) & 1);
        bool h = (ddx[bx] <= ddy[by]) && (age <= bfinal[q]);
        hits |= (h ? 1u : 0u) << q;
      }
    }
    unsigned long long B[16];
    int mycnt = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      B[q] = __ballot((hits >> q) & 1u);
      if (lane == q) mycnt = __popcll(B[q]);
    }
    if (lane < 16) sb.bcnt[wv][lane] = mycnt;
    __syncthreads();
    if (b + 1 < nb) gather(b + 1);
    // order-preserving positions of this thread's entry in the lists of the blocks it reaches (4 x 4 packed bytes)
    unsigned posn[4] = {0u, 0u, 0u, 0u};
    {
      const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        int base = 0;
        if (wv > 0) base += sb.bcnt[0][q];
        if (wv > 1) base += sb.bcnt[1][q];
        if (wv > 2) base += sb.bcnt[2][q];
        int pos = base + __popcll(B[q] & lt);
        if ((hits >> q) & 1u) {
          sb.blist[q][pos] = (uint8_t)tid;
          posn[q >> 2] |= (unsigned)pos << (8 * (q & 3));
        }
      }
      if (tid < 16) sb.btot[tid] = sb.bcnt[0][tid] + sb.bcnt[1][tid] + sb.bcnt[2][tid] + sb.bcnt[3][tid];
    }
    __syncthreads();
    // ---- walk: row g of wave wv walks the list of block blk, LCAP trips per round
    const int cnt = sb.btot[blk];
    int kwave = cnt;
    kwave = max(kwave, __shfl_xor(kwave, 16, 64));
    kwave = max(kwave, __shfl_xor(kwave, 32, 64));
    kwave = __builtin_amdgcn_readfirstlane(kwave);
    // pad this row's list with the sentinel up to the wave's trip count (+2: the walk reads two trips ahead); only this
    // wave reads these lists, and a wave's LDS operations complete in order
    for (int k = cnt + p; k < min(kwave + 2, GSL_GBS); k += 16) mylist[k] = (uint8_t)GSL_GB;
    const int lim = (int)bend - bin_final;  // entry t was composited by this pixel iff bend - t <= bin_final
    int kall = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) kall = max(kall, sb.btot[q]);
    kall = __builtin_amdgcn_readfirstlane(kall);
#ifdef GSL_G16_STATS
    if (tid == 0) {
      int kw[4];
      for (int w = 0; w < 4; ++w) {
        kw[w] = 0;
        for (int g2 = 0; g2 < 4; ++g2) kw[w] = max(kw[w], sb.btot[4 * w + g2]);
      }
      atomicAdd(&g16_stats[6], (unsigned long long)max(max(kw[0], kw[1]), max(kw[2], kw[3])));
      int tot = 0;
      for (int q = 0; q < 16; ++q) tot += sb.btot[q];
      atomicAdd(&g16_stats[7], (unsigned long long)tot);
    }
#endif
    float mo[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) mo[q] = 0.f;
    for (int k0 = 0; k0 < kall; k0 += LCAP) {
      const int k1 = min(k0 + LCAP, kwave);
      // two-deep software pipeline: list index of trip k + 2, records of trip k + 1
      int t_cur = mylist[min(k0, GSL_GBS - 1)];
      int t_nxt = mylist[min(k0 + 1, GSL_GBS - 1)];
      float4 q0 = sb.s0[t_cur], q1 = sb.s1[t_cur];
      // the store of trip k is issued at the top of trip k + 1, BEFORE that trip's loads: LDS operations complete in
      // order, so a store issued after the loads would be waited for together with them at the next loop head
      float r_prev = 0.f;
      bool st_prev = false;
      float* dst_prev = mypair;
      for (int k = k0; k < k1; ++k) {
        const int t = t_cur;
        const float4 c0 = q0, c1 = q1;
        if (st_prev) *dst_prev = r_prev;
        t_cur = t_nxt;
        t_nxt = mylist[min(k + 2, GSL_GBS - 1)];
        q0 = sb.s0[t_cur];
        q1 = sb.s1[t_cur];
        float dx = c0.x - px, dy = c0.y - py;
        float gx = c1.x * dx + c1.y * dy;
        float gy = c1.y * dx + c1.z * dy;
        float sigma = 0.5f * (dx * gx + dy * gy);
        float vis = __expf(-sigma);
        float opv = c0.w * vis;
        float alpha = fminf(GSL_ALPHA_MAX, opv);
        unsigned long long validm = __ballot(t >= lim) & __ballot(sigma >= 0.f) & __ballot(alpha >= GSL_ALPHA_MIN);
        G16_STAT(0, 1);
        G16_STAT(2, __popcll(__ballot(k < cnt)) >> 4);
        float r = 0.f;
        if (validm) {  // some pixel of the quadrant composited one of the (up to four) entries of this trip
          G16_STAT(1, 1);
          G16_STAT(3, ((validm & 0xFFFFull) != 0) + ((validm & 0xFFFF0000ull) != 0) + ((validm & 0xFFFF00000000ull) != 0) + ((validm >> 48) != 0));
          G16_STAT(5, __popcll(validm));
          unsigned long long capm = __ballot(opv <= GSL_ALPHA_MAX);
          float am = g16_sel(validm, alpha, 0.f);  // other lanes: alpha = 0 => ra = 1, fac = 0: state unchanged
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
          float w = g16_sel(validm & capm, vis, 0.f) * v_alpha;  // alpha clamped at 0.999 => no geometric gradient
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
          if (NS == 8) r = row_scatter8(reinterpret_cast<const float(&)[8]>(val));
          else r = row_scatter16(reinterpret_cast<const float(&)[16]>(val));
        }
        r_prev = r;  // the pair's own slot: written exactly once
        st_prev = writer;
        dst_prev = &mypair[(k - k0) * NS];
      }
      if (st_prev) *dst_prev = r_prev;
      __syncthreads();
      // gather: the staging thread of an entry adds up its pairs of this round, in block order
      if (staged) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          int pos = (int)((posn[q >> 2] >> (8 * (q & 3))) & 0xFFu) - k0;
          if (((hits >> q) & 1u) && pos >= 0 && pos < LCAP) {
            const float4* src = reinterpret_cast<const float4*>(&sb.pair[(q * LCAP + pos) * NS]);
            float4 a = src[0], c = src[1];
            mo[0] += a.x; mo[1] += a.y; mo[2] += a.z; mo[3] += a.w;
            mo[4] += c.x; mo[5] += c.y; mo[6] += c.z;
            if (NV > 7) mo[7 < NV ? 7 : 0] += c.w;
            if (NS == 16) {
              float4 e = src[2];
              if (NV > 8) mo[8 < NV ? 8 : 0] += e.x;
              if (NV > 9) mo[9 < NV ? 9 : 0] += e.y;
            }
          }
        }
      }
      __syncthreads();  // pair slots are free again (next round, or the packed rows below)
    }
    // ---- moments -> gradient row (the staging thread of each entry)
    {
      bool nz = false;
      float row[A];
#pragma unroll
      for (int k = 0; k < A; ++k) row[k] = 0.f;
      if (staged) {
#pragma unroll
        for (int k = 0; k < NV; ++k) nz = nz || (mo[k] != 0.f);
        if (nz) {
          float X = r0.x - tcx, Y = r0.y - tcy, S = mo[0];
          float Sx = X * S - mo[1], Sy = Y * S - mo[2];
          float Sxx = X * (X * S - 2.f * mo[1]) + mo[3];
          float Sxy = X * (Y * S - mo[2]) - Y * mo[1] + mo[4];
          float Syy = Y * (Y * S - 2.f * mo[2]) + mo[5];
          float no = -r0.w;  // v_sigma = -opacity * w
          row[0] = no * (r1.x * Sx + r1.y * Sy);
          row[1] = no * (r1.y * Sx + r1.z * Sy);
          row[2] = 0.5f * no * Sxx;
          row[3] = no * Sxy;
          row[4] = 0.5f * no * Syy;
          row[5] = S;
          if (CG == D) {
#pragma unroll
            for (int ch = 0; ch < D; ++ch) row[6 + ch] = mo[6 + ch];
          } else {
            row[6 + D - 1] = mo[6];
          }
        }
      }
      float* packed = sb.pair;  // GSL_GB rows of 16 floats
      if (nz) {
#pragma unroll
        for (int k = 0; k < A; ++k) packed[tid * 16 + k] = row[k];
      }
      // pack non-zero slots so that 16 consecutive lanes add one Gaussian's 64-byte row
      unsigned long long mask = __ballot(nz);
      int cntz = __popcll(mask);
      if (nz) sb.list[wv][__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)tid;
      __syncthreads();
      int f = lane & 15;
      for (int i0 = 0; i0 < cntz; i0 += 4) {
        int gi = i0 + (lane >> 4);
        if (gi < cntz && f < A) {
          int sl = sb.list[wv][gi];
          size_t g = (size_t)sb.id[sl];
          atomicAdd(&vacc[g * 16 + f], packed[sl * 16 + f]);
        }
      }
    }
  }
}

// CG = 1 (only for D = 4): the tiles whose upstream gradient lives in the depth channel alone (GsplatLoc's loss) -- one
// colour sum instead of four, 8 slots per pair, <= 128 VGPRs; CG = D: every other tile.  For D = 4 both kernels are
// launched and each returns at once on the other's tiles (a workgroup that only reads its tile's v_render costs
// ~1 ns of a CU): one kernel holding both bodies is limited by the larger one's registers and LDS on every tile.
template <int D, bool ED, int CG, bool LONG>
__global__ __launch_bounds__(256) void k_graster_bwd(
    const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2, int W, int H,
    int tile_w, int ty0, const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    long long capacity, const float* __restrict__ render, const float* __restrict__ alphas,
    const int32_t* __restrict__ last_ids, const float* __restrict__ v_render, const float* __restrict__ v_alphas,
    float* __restrict__ vacc, int row0, int row1, const uint4* __restrict__ Qh,
    const uint16_t* __restrict__ isect_hits, int long_min, LongWs lw) {
  __shared__ GStage<D, CG> sb;
  __shared__ int s_final[4];
  __shared__ int s_bfinal[16];  // per 4x4 block: last list index any of its pixels composited
  // LONG = false: one workgroup per tile of the strip (tiles with lists longer than long_min are skipped when
  // long_min > 0); LONG = true: one workgroup per (tile, segment) pair of the long tiles (lw, raster_px.hip)
  int tile, sgm = 0, gseg = 0;
  if (LONG) {
    gseg = blockIdx.x;
    if (gseg >= lw.n_seg[0]) return;
    tile = lw.seg_tile[gseg];
    sgm = lw.seg_idx[gseg];
  } else {
    tile = ty0 * tile_w + GSL_TILE_OF_BLOCK();
  }
  int tyi = tile / tile_w, txi = tile - tyi * tile_w;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, grp = lane >> 4, p = lane & 15;
  // wave = 8x8 quadrant, DPP row = 4x4 block, lane p of a row = pixel (p & 3, p >> 2) of the block
  int j = txi * 16 + (wv & 1) * 8 + (grp & 1) * 4 + (p & 3);
  int i = tyi * 16 + (wv >> 1) * 8 + (grp >> 1) * 4 + (p >> 2);
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
  float vc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) vc[k] = inside ? v_render[pid * D + k] : 0.f;
  if (D == 4) {
    bool rgb_grad = (vc[0] != 0.f) || (vc[1] != 0.f) || (vc[2] != 0.f);
    int any_rgb = __syncthreads_or(rgb_grad);
    if ((CG == 1) == (any_rgb != 0)) return;  // the other kernel's tile
  }
  float Aimg = inside ? alphas[pid] : 0.f;
  float T_final = 1.f - Aimg;
  int bin_final = inside ? last_ids[pid] : -1;
  float va = inside ? v_alphas[pid] : 0.f;
  if (ED && inside) {
    float dn = render[pid * D + (D - 1)];
    float vd = vc[D - 1];
    if (Aimg >= 1e-10f) va += -vd * dn / Aimg;
    vc[D - 1] = vd / fmaxf(Aimg, 1e-10f);
  }
  // last composited list index per block (row), per wave, per tile
  int row_final = bin_final;
  row_final = max(row_final, __shfl_xor(row_final, 1, 64));
  row_final = max(row_final, __shfl_xor(row_final, 2, 64));
  row_final = max(row_final, __shfl_xor(row_final, 4, 64));
  row_final = max(row_final, __shfl_xor(row_final, 8, 64));
  int blk = 4 * wv + grp;
  if (p == 0) s_bfinal[blk] = row_final;
  int wave_final = max(row_final, __shfl_xor(row_final, 16, 64));
  wave_final = max(wave_final, __shfl_xor(wave_final, 32, 64));
  if (lane == 0) s_final[wv] = wave_final;
  __syncthreads();
  int block_final = max(max(s_final[0], s_final[1]), max(s_final[2], s_final[3]));
  // nothing behind block_final was composited by any pixel of the tile: start there
  if ((long long)block_final + 1 < re) re = max((long long)block_final + 1, rs);
  if (rs >= re) return;
  float T_init = T_final, Bp_init = -T_final * va;
  if (LONG) {
    // the pixel went on compositing behind this segment: its T at the segment's end, and the colour partials of the
    // later segments (the forward left both); otherwise its last entry lies in (or before) this segment: T_final
    int nseg = lw.seg_cnt[gseg];
    if (inside && (long long)bin_final >= re) {
      T_init = fabsf(lw.Tend[(size_t)gseg * 256 + tid]);
      for (int s2 = sgm + 1; s2 < nseg; ++s2) {
        size_t slot = (size_t)(gseg - sgm + s2) * 256 + tid;
        if (lw.Tend[slot] == 2.f) break;  // dead on arrival from there on
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) dot += vc[k] * lw.C[slot * 4 + k];
        Bp_init += dot;
      }
    }
  }
  graster_bwd_body<D, CG>(sb, Q0, Q1, Q2, Qh, flatten_ids, vacc, rs, re, tid, px, py, (float)(txi * 16), (float)(tyi * 16),
                          inside, bin_final, T_init, Bp_init, vc, s_bfinal, isect_hits);
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

// Launch of the G16 backward (called by gsl_fused_raster_bwd in fused.hip for the non-deterministic path).
extern "C" int gsl_g16_raster_bwd_launch(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                         int height, int tile_w, int ty0, int ty1, const int32_t* tile_offsets,
                                         const int32_t* flatten_ids, int64_t capacity, const float* render,
                                         const float* alphas, const int32_t* last_ids, const float* v_render,
                                         const float* v_alphas, float* vacc, int row0, int row1, const void* Qh,
                                         const uint16_t* isect_hits, int long_min, void* long_ws, int max_seg,
                                         void* stream) {
  // long_ws == NULL: the tiles of the strip (those longer than long_min, if > 0, are skipped);
  // long_ws != NULL: only the (tile, segment) pairs the forward's long-list pass listed there
  hipStream_t st = (hipStream_t)stream;
  int nblk = long_ws ? max_seg : (ty1 - ty0) * tile_w;
  gsl::LongWs lw = gsl::long_ws_views(long_ws ? long_ws : (void*)0, long_ws ? max_seg : 0);
  const bool lng = long_ws != nullptr;
#define CALL_G(DD, EE, CC)                                                                                   \
  do {                                                                                                       \
    if (lng)                                                                                                 \
      hipLaunchKernelGGL((gsl::k_graster_bwd<DD, EE, CC, true>), dim3(nblk), dim3(256), 0, st, (const float4*)Q0, \
                         (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,     \
                         flatten_ids, (long long)capacity, render, alphas, last_ids, v_render, v_alphas, vacc, \
                         row0, row1, (const uint4*)Qh, isect_hits, long_min, lw);                            \
    else                                                                                                     \
      hipLaunchKernelGGL((gsl::k_graster_bwd<DD, EE, CC, false>), dim3(nblk), dim3(256), 0, st, (const float4*)Q0, \
                         (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,     \
                         flatten_ids, (long long)capacity, render, alphas, last_ids, v_render, v_alphas, vacc, \
                         row0, row1, (const uint4*)Qh, isect_hits, long_min, lw);                            \
  } while (0)
  if (channels == 1) { if (ed) CALL_G(1, true, 1); else CALL_G(1, false, 1); }
  else if (channels == 3) { CALL_G(3, false, 3); }
  else if (channels == 4) {
    if (ed) { CALL_G(4, true, 1); CALL_G(4, true, 4); }
    else { CALL_G(4, false, 1); CALL_G(4, false, 4); }
  }
  else return GSL_ERR_BAD_ARG;
#undef CALL_G
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
