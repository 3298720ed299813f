// Per-pixel-mask compositing (forward and backward) for the fused pipeline.
//
// GsplatLoc's splats are tiny: the reference's kNN scales collapse to the 0.3 px^2 blur (alpha >= 1/255 only
// within ~1.8 px of the centre, ~10 pixels), and even sigma = 1 px splats cover ~45 of a tile's 256 pixels.
// Walking a tile's depth-sorted list with all 64 lanes of a wave per splat (raster.hip, and the first fused
// kernels) keeps 8-35 % of the lanes busy.  Here every lane walks ITS OWN pixel's candidates:
//
//   1. a batch of 256 records is staged in LDS once per workgroup (16x16 tile, wave = 8x8 quadrant);
//   2. the staging threads test their record against the four quadrants; order-preserving compaction
//      (ballot + mbcnt + a 4x4 count table in LDS) gives each wave the ~35 % of the batch it can see;
//   3. per chunk of 64 surviving candidates, lane e turns candidate e's bounding box into a column range
//      and a row range of the quadrant; 8 + 8 ballots transpose that into "which candidates cover column
//      x / row y"; a lane's 64-bit candidate mask is colmask[x] & rowmask[y];
//   4. each lane pops the set bits of its own mask in list order (front to back, or back to front in the
//      backward pass), fetching the record from LDS with a per-lane address.  No cross-lane operation is
//      needed in the loop, so the lanes diverge freely; the wave runs max-over-lanes iterations per chunk
//      (~3 for the reference's splats instead of ~22 wave-wide trips).
//   5. backward: every lane adds its contribution to the splat's LDS accumulator row with ds_add_f32
//      (lanes work on different splats, so there is nothing to reduce across the wave); rows are flushed
//      per batch as packed 64-byte global atomics exactly as in fused.hip.
//
// Skipped candidates are exactly those the reference loop would `continue` over (alpha < 1/255), so the
// per-pixel sequence of composited splats -- and therefore every output -- is unchanged.
#include "gsloc_common.h"
#include "loss_dev.h"
#include "sort_dev.h"

namespace gsl {

template <int D>
struct PStage {
  float4 s0[256];
  float4 s1[256];
  float4 s2[(D >= 3) ? 256 : 1];
  uint16_t qlist[4][256];  // per-quadrant candidate slots, in list order
  int qcnt[4][4];          // [staging wave][quadrant]
};

// Order-preserving compaction of the staged batch into per-quadrant candidate lists.
// Call with the record of slot `tid` (valid_rec = slot < bsize).  Two barriers inside.
// Returns the number of candidates of quadrant `wv`.
// SHIFT = 4: the lists hold the records' BYTE offsets (slot x 16) -- what a trip of praster_walk adds to the array base,
// sparing it a 4-cycle shift per candidate; 0: slot numbers (k_tiny_bwd compares them with list positions).
template <int SHIFT = 0, typename Stage>
__device__ __forceinline__ int compact_quadrants(Stage& sb, int tid, bool valid_rec, float x, float y, float r,
                                                 float tile_x0, float tile_y0) {
  int lane = tid & 63, wv = tid >> 6;
  unsigned long long B[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float cx = tile_x0 + 4.f + 8.f * (float)(q & 1), cy = tile_y0 + 4.f + 8.f * (float)(q >> 1);
    // (one ballot per compare, combined as scalar masks: a ballot of a compound predicate is compiled as
    // v_cndmask 0/1 + v_cmp_ne on top of the compares -- two more four-cycle VALU each)
    B[q] = __ballot(valid_rec) & __ballot(fabsf(x - cx) <= r + 3.5f) & __ballot(fabsf(y - cy) <= r + 3.5f);
  }
  if (lane < 4) {
    unsigned long long b = lane == 0 ? B[0] : (lane == 1 ? B[1] : (lane == 2 ? B[2] : B[3]));
    sb.qcnt[wv][lane] = __popcll(b);
  }
  __syncthreads();
  unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if ((B[q] >> lane) & 1ull) {
      int base = 0;
      for (int w = 0; w < wv; ++w) base += sb.qcnt[w][q];
      sb.qlist[q][base + __popcll(B[q] & lt)] = (uint16_t)(tid << SHIFT);
    }
  }
  __syncthreads();
  return sb.qcnt[0][wv] + sb.qcnt[1][wv] + sb.qcnt[2][wv] + sb.qcnt[3][wv];
}

// Per-lane 64-bit candidate mask of one chunk: bit k set <=> candidate k's box covers this lane's pixel.
// lo/hi are the candidate's inclusive column (row) ranges inside the quadrant (empty if hi < lo).
// The 8+8 ballots are wave-uniform and are handed to the lanes of column/row v with two exec-masked moves each (no
// v_cndmask, no VCC).  exec is saved, narrowed to (current exec & lanes) for the two moves and restored, so the
// helper is correct under any control flow the compiler leaves around it.
__device__ __forceinline__ void masked_mov2(unsigned& dlo, unsigned& dhi, unsigned long long value,
                                            unsigned long long lanes) {
  unsigned vlo = (unsigned)value, vhi = (unsigned)(value >> 32);
  unsigned long long saved;
  asm volatile("s_mov_b64 %2, exec\n\ts_and_b64 exec, %2, %5\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\ts_mov_b64 exec, %2"
               : "+v"(dlo), "+v"(dhi), "=&s"(saved)
               : "s"(vlo), "s"(vhi), "s"(lanes)
               : "scc");
}
// BLOCKROWS = false: lane = 8 y + x of the quadrant (k_tiny_bwd); true: each 16-lane DPP row is one 4x4 pixel block,
// lane = 16 g + p with x = 4 (g & 1) + (p & 3), y = 4 (g >> 1) + (p >> 2) (k_praster_fwd: the compositing backward of
// raster_g16.hip uses the same mapping, and a per-block OR of "who composited what" is then a row reduction).
template <bool BLOCKROWS = false>
__device__ __forceinline__ void pixel_masks(int lox, int hix, int loy, int hiy, int lane, unsigned& mlo, unsigned& mhi) {
  unsigned clo = 0, chi = 0, rlo = 0, rhi = 0;
  (void)lane;
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    unsigned long long mc = __ballot(lox <= v) & __ballot(v <= hix);  // (not a ballot of the conjunction: see above)
    unsigned long long mr = __ballot(loy <= v) & __ballot(v <= hiy);
    if (BLOCKROWS) {
      masked_mov2(clo, chi, mc, (0x0000111100001111ull << (v & 3)) << (16 * (v >> 2)));  // lanes of pixel column v
      masked_mov2(rlo, rhi, mr, (0x00000000000F000Full << (4 * (v & 3))) << (32 * (v >> 2)));  // lanes of pixel row v
    } else {
      masked_mov2(clo, chi, mc, 0x0101010101010101ull << v);  // lanes of pixel column v
      masked_mov2(rlo, rhi, mr, 0xFFull << (8 * v));          // lanes of pixel row v
    }
  }
  mlo = clo & rlo;
  mhi = chi & rhi;
}

// OR over the lanes of a pixel group of the backward: the 16 lanes of a DPP row (GSL_NG = 4) or the 8 of a half row
// (GSL_NG = 8); every lane of the group gets the result.
__device__ __forceinline__ unsigned group_or(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
#if GSL_NG == 4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
#endif
  return v;
}

// t in the lanes of the scalar mask m, f elsewhere (v_cndmask_b32_e64 on an SGPR pair; no VCC round trip).
__device__ __forceinline__ unsigned sel_u32(unsigned long long m, unsigned t, unsigned f) {
  unsigned r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}
__device__ __forceinline__ float sel_f32(unsigned long long m, float t, float f) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}

// Index of the lowest set bit; 0xFFFFFFFF for 0 (v_ffbl_b32 as it is: __builtin_ctz(0) is undefined and __ffs costs two more
// four-cycle instructions for the zero case, which the trips below do not need).
__device__ __forceinline__ int ffbl_raw(unsigned v) {
  int r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

// record at byte offset `off` (slot x 16) of a staged array
__device__ __forceinline__ float4 rec_at(const float4* arr, unsigned off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(arr) + off);
}

__device__ __forceinline__ void box_range(float centre_rel, float r, int& lo, int& hi) {
  // pixel centres of the quadrant sit at 0..7 in these coordinates
  float l = ceilf(centre_rel - r), h = floorf(centre_rel + r);
  lo = (int)fmaxf(l, 0.f);
  hi = (int)fminf(h, 7.f);
}

// The walk of one tile list (or of one SEGMENT [rs, re) of a long list) by the 256 pixels of the tile.
// MODE 0: composite (the reference loop): pix += feat * alpha * T, T *= 1 - alpha, stop before T would drop to 1e-4.
// MODE 1: transmittance product only -- T *= 1 - alpha over every entry with alpha >= 1/255, no stop, no colours:
//         pass A of the long-list split below.
// On entry: T, done (pixels outside the window, or dead on arrival in a later segment), pix = 0, cur_idx.
template <int D, int MODE>
__device__ __forceinline__ void praster_walk(
    PStage<D>& sb, const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2,
    const uint4* __restrict__ Qh, const int32_t* __restrict__ flatten_ids, long long rs, long long re, int tid, float px,
    float py, int qx, int qy, int txi, int tyi, bool& done, float& T, float (&pix)[D], int& cur_idx,
    uint32_t* __restrict__ isect_hits, int& n_hits) {
  constexpr bool RGB = D >= 3;
  constexpr bool DEPTH = (D == 1) || (D == 4);
  const int lane = tid & 63, wv = tid >> 6;
  int nb = (int)((re - rs + 255) / 256);
  if (MODE == 1) isect_hits = nullptr;
  // isect_hits (may be NULL): this quadrant's HIT LIST -- the entries of [rs, re) that at least one of its pixels
  // composited, in list order, each as (bits of its GSL_NG pixel groups that did) << GSL_HIT_SHIFT | absolute list index, appended
  // chunk by chunk at  isect_hits[4 rs + quadrant (re - rs) + n_hits ...]: the compositing backward walks exactly those
  // (block, entry) pairs and never scans an entry its quadrant did not touch
  uint32_t* const qout = isect_hits ? isect_hits + 4 * rs + (long long)wv * (re - rs) : nullptr;
  n_hits = 0;
  // The pixels' predicates live as SCALAR lane masks from here on (round 4): each compare is one ballot, their
  // conjunctions, negations and the running "done" set are s_and / s_andn2 / s_or on SGPR pairs, and a lane takes a
  // value under a mask with one v_cndmask_b32_e64.  Left to the compiler, per-lane bools cost a second compare for
  // every negation (v_cmp_nge next to v_cmp_ge) and a v_cndmask 0/1 + v_cmp_ne for every bool that crosses a loop.
  unsigned long long DONE = __ballot(done);
  for (int b = 0; b < nb; ++b) {
    if (__syncthreads_and(DONE == ~0ull)) break;
    long long bstart = rs + (long long)b * 256;
    int bsize = (int)min((long long)256, re - bstart);
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 0.f, 0.f, -1.f);
    if (tid < bsize) {
      int g = flatten_ids[bstart + tid];
      float4 r2 = make_float4(0.f, 0.f, 0.f, 0.f);
      load_record(Q0, Q1, Q2, Qh, g, RGB && MODE == 0, r0, r1, r2);
      // staged for the walk: the conic times log2(e) (and its diagonal halved), so that a trip gets
      // log2(e) sigma = a' dx^2 + c' dy^2 + b' dx dy in six operations and alpha = opacity exp2(-that) without the scaling
      // multiply of expf.  Laid out for the trip's LDS reads (round 4; the LDS array was 47 % busy, profiles/r04_*):
      //   s0 = (x, y, a', b')   s1 = (c', opacity, r, g)   s2 = (b, depth, r_cull, -)      [colour modes]
      //   s0 = (x, y, a', b')   s1 = (c', opacity, depth, r_cull)                          [depth only]
      // two 16-byte reads and one 8-byte read per candidate (10 LDS cycles) instead of 8 + 4 + 12 + 16 bytes (16 cycles: a
      // 12-byte read costs 8)
      const float ca = r1.x * (0.5f * GSL_LOG2E), cb = r1.y * GSL_LOG2E, cc = r1.z * (0.5f * GSL_LOG2E);
      sb.s0[tid] = make_float4(r0.x, r0.y, ca, cb);
      if (RGB) {
        sb.s1[tid] = make_float4(cc, r0.w, r2.x, r2.y);
        sb.s2[tid] = make_float4(r2.z, r0.z, r1.w, 0.f);
      } else {
        sb.s1[tid] = make_float4(cc, r0.w, r0.z, r1.w);
      }
    } else {
      // every slot holds finite numbers: a lane without a candidate reads SOME slot in the straight-line trip and
      // multiplies what it finds by an exact zero (stale LDS bits can be NaN)
      sb.s0[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (RGB) {
        sb.s1[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        sb.s2[tid] = make_float4(0.f, 0.f, -1.f, 0.f);
      } else {
        sb.s1[tid] = make_float4(0.f, 0.f, 0.f, -1.f);
      }
    }
    // (read back from LDS: wave-uniform, but only readfirstlane tells the compiler so -- the chunk loop's control then
    // stays on the scalar unit)
    const int n = __builtin_amdgcn_readfirstlane(
        compact_quadrants<4>(sb, tid, tid < bsize, r0.x, r0.y, r1.w, (float)(txi * 16), (float)(tyi * 16)));
    for (int c = 0; c < n; c += 64) {
      if (DONE == ~0ull) break;
      int e = c + lane;
      int lox = 1, hix = 0, loy = 1, hiy = 0;
      if (e < n) {
        const unsigned off = sb.qlist[wv][e];  // (byte offset of the record: slot x 16)
        float4 a0 = rec_at(sb.s0, off);
        float r = RGB ? rec_at(sb.s2, off).z : rec_at(sb.s1, off).w;
        box_range(a0.x - ((float)qx + 0.5f), r, lox, hix);
        box_range(a0.y - ((float)qy + 0.5f), r, loy, hiy);
      }
      unsigned mlo, mhi;
      pixel_masks<true>(lox, hix, loy, hiy, lane, mlo, mhi);
      mlo = sel_u32(DONE, 0u, mlo);
      mhi = sel_u32(DONE, 0u, mhi);
      unsigned cm[2] = {0u, 0u};  // candidates of this chunk this pixel composited
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        unsigned m = half ? mhi : mlo;
        if (half) m = sel_u32(DONE, 0u, m);  // (a pixel that stopped in the first half; inside the loop only m is cleared)
        const uint16_t* const ql = &sb.qlist[wv][c + half * 32];
        // Straight-line trips under a wave-uniform loop: a lane that has run out of candidates goes through the
        // arithmetic with alpha 0.  No divergent region, so none of the copies the structurizer makes of the nine values
        // that live across the loop (they were a quarter of the VALU instructions of a trip written with branches:
        // 110 -> 80 per trip, forward 175 -> 166 us at R).  Two candidates per trip: both records are requested and
        // both alphas evaluated before the transmittance updates are applied in list order (halves the dependent LDS
        // round trips).
        // Round 4 (profiles/r04_valu_issue.txt: v_cndmask / v_cmp / v_ffbl / carry-out ops issue in 4 cycles, and / xor /
        // sub / fma / mul in 2): the candidates are popped as one-bit MASKS (m & -m: two 2-cycle ops), which also are what
        // the "composited" mask collects; the index of a lane WITHOUT a candidate is left as v_ffbl gives it (-1: the
        // list slot in front of the chunk's, some valid slot of the batch whose alpha is multiplied away) instead of
        // being selected into range; the last composited index is taken from the composited mask once per chunk, not
        // selected per trip.  72 -> 62 VALU per trip, ten of the twelve removed ones of the 4-cycle kind.
        unsigned long long ACT = __ballot(m != 0);  // lanes with a candidate left (one compare per trip: it also is
        if (ACT) do {                                // the loop's condition)
          const unsigned b0 = m & (0u - m);
          m ^= b0;
          const unsigned b1 = m & (0u - m);
          m ^= b1;
          const unsigned long long TWO = __ballot(b1 != 0u);
          const unsigned t0 = ql[ffbl_raw(b0)] & 0xFF0u;  // byte offsets (masked: a lane without a candidate reads
          const unsigned t1 = ql[ffbl_raw(b1)] & 0xFF0u;  // whatever sits in front of the chunk's list)
          float4 p0 = rec_at(sb.s0, t0), p1 = rec_at(sb.s1, t0);  // (x, y, a', b'), (c', opacity, r | depth, g | r_cull)
          float4 u0 = rec_at(sb.s0, t1), u1 = rec_at(sb.s1, t1);
          float dx0 = p0.x - px, dy0 = p0.y - py, dx1 = u0.x - px, dy1 = u0.y - py;
          float sg0 = fmaf(p0.w * dx0, dy0, fmaf(p0.z * dx0, dx0, p1.x * dy0 * dy0));  // log2(e) sigma (see the staging)
          float sg1 = fmaf(u0.w * dx1, dy1, fmaf(u0.z * dx1, dx1, u1.x * dy1 * dy1));
          float al0 = fminf(GSL_ALPHA_MAX, p1.y * __builtin_amdgcn_exp2f(-sg0));
          float al1 = fminf(GSL_ALPHA_MAX, u1.y * __builtin_amdgcn_exp2f(-sg1));
          const unsigned long long OK0 = ACT & __ballot(sg0 >= 0.f) & __ballot(al0 >= GSL_ALPHA_MIN);
          const unsigned long long OK1 = TWO & __ballot(sg1 >= 0.f) & __ballot(al1 >= GSL_ALPHA_MIN);
          if (MODE == 1) {
            T *= 1.f - sel_f32(OK0, al0, 0.f);
            T *= 1.f - sel_f32(OK1, al1, 0.f);
            ACT = __ballot(m != 0);
            continue;
          }
          // T > 1e-4 on entry, so a skipped candidate (alpha 0) leaves T as it is and cannot stop the pixel.
          // ONE select per candidate: the alpha that takes effect is al if the candidate passes its tests and does not
          // stop the pixel, else 0 -- and with alpha 0 the plain products give vis = 0 and T unchanged, bit for bit what
          // selects on vis and T would (two 2-cycle ops for two 4-cycle selects per candidate).
          const unsigned long long S0R = __ballot(T * (1.f - al0) <= GSL_T_STOP);  // candidate 0 would stop the pixel
          const unsigned long long STOP0 = OK0 & S0R, EFF0 = OK0 & ~S0R;
          const float a0 = sel_f32(EFF0, al0, 0.f);
          const float vis0 = a0 * T;
          const float T1 = T * (1.f - a0);
          const unsigned long long LIVE1 = OK1 & ~STOP0;
          const unsigned long long S1R = __ballot(T1 * (1.f - al1) <= GSL_T_STOP);
          const unsigned long long STOP1 = LIVE1 & S1R, EFF1 = LIVE1 & ~S1R;
          const float a1 = sel_f32(EFF1, al1, 0.f);
          const float vis1 = a1 * T1;
          T = T1 * (1.f - a1);
          if (RGB) {
            const float2 q20 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(sb.s2) + t0);  // (b, depth)
            const float2 q21 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(sb.s2) + t1);
            pix[0] += p1.z * vis0; pix[1] += p1.w * vis0; pix[2] += q20.x * vis0;
            if (DEPTH) pix[D - 1] += q20.y * vis0;
            pix[0] += u1.z * vis1; pix[1] += u1.w * vis1; pix[2] += q21.x * vis1;
            if (DEPTH) pix[D - 1] += q21.y * vis1;
          } else if (DEPTH) {
            pix[D - 1] += p1.z * vis0;
            pix[D - 1] += u1.z * vis1;
          }
          // composited <=> alpha >= 1/255 and the pixel did not stop on this entry (alpha T > 0 then: T > 1e-4): the
          // masks the alphas were selected under
          cm[half] |= sel_u32(EFF0, b0, 0u) | sel_u32(EFF1, b1, 0u);
          const unsigned long long STOPPED = STOP0 | STOP1;
          DONE |= STOPPED;
          m = sel_u32(STOPPED, 0u, m);
          ACT = __ballot(m != 0);
        } while (ACT);
      }
      {  // the last entry this pixel composited in the chunk = the highest bit of its composited mask (list order)
        const bool any = (cm[0] | cm[1]) != 0u;
        const int pos = cm[1] ? 63 - __clz((int)cm[1]) : 31 - __clz((int)(cm[0] | 1u));
        const int tl = (sb.qlist[wv][c + pos] & 0xFF0u) >> 4;
        cur_idx = any ? (int)bstart + tl : cur_idx;
      }
      if (isect_hits) {
        // per pixel group (GSL_NG = 4: DPP row = 4x4 block; 8: half row = 4x2 half block): OR of the pixels'
        // composited-candidate masks; lane e then owns candidate c + e and collects its bit from the groups
        unsigned rlo = group_or(cm[0]), rhi = group_or(cm[1]);
        // group g's OR is wave-uniform: as a scalar lane mask it IS "which lanes' candidates group g composited", so a
        // lane picks its bit up with one select per group (no 64-bit shifts), and the union is a scalar OR, not a ballot
        unsigned nib = 0;
        unsigned long long Rm = 0ull;
#pragma unroll
        for (int g = 0; g < GSL_NG; ++g) {
          unsigned glo = (unsigned)__builtin_amdgcn_readlane((int)rlo, (64 / GSL_NG) * g);
          unsigned ghi = (unsigned)__builtin_amdgcn_readlane((int)rhi, (64 / GSL_NG) * g);
          unsigned long long gm = ((unsigned long long)ghi << 32) | glo;
          nib |= sel_u32(gm, 1u << g, 0u);
          Rm |= gm;
        }
        // (a set bit implies e < n)
        if (nib) qout[n_hits + __popcll(Rm & ((1ull << lane) - 1ull))] = (nib << GSL_HIT_SHIFT) | (unsigned)((int)bstart + (sb.qlist[wv][e] >> 4));
        n_hits += __popcll(Rm);
      }
    }
  }
  done = ((DONE >> lane) & 1ull) != 0ull;
}

// lane -> pixel of the compositing kernels: wave = 8x8 quadrant, DPP row g = 4x4 block g of the quadrant, lane p of
// the row = pixel (p & 3, p >> 2) of the block
struct TilePixel {
  int tyi, txi, qx, qy, i, j;
};
__device__ __forceinline__ TilePixel tile_pixel(int tile, int tile_w, int tid) {
  TilePixel t;
  int lane = tid & 63, wv = tid >> 6, grp = lane >> 4, pp = lane & 15;
  t.tyi = tile / tile_w;
  t.txi = tile - t.tyi * tile_w;
  t.qx = t.txi * 16 + (wv & 1) * 8;
  t.qy = t.tyi * 16 + (wv >> 1) * 8;
  t.j = t.qx + 4 * (grp & 1) + (pp & 3);
  t.i = t.qy + 4 * (grp >> 1) + (pp >> 2);
  return t;
}

// SORT = 2 / 3 (binned projection, whole frame, every list <= 1024 / 2048 keys, no long lists): the kernel turns its
// tile's bin into the sorted list ITSELF before compositing it -- what k_tile_sort_wg does in a launch of its own: adds up
// the sizes of the tiles before its own (offsets, total, overflow flags), sorts (four waves sort quarters in registers,
// two merge-path passes in LDS, sort_dev.h) and writes tile_offsets / flatten_ids for the backward.  The tracker's
// iteration loses a launch (10 of 85 us at 102 k Gaussians).  The tile counters cannot be cleared here any more -- other
// workgroups are still adding them up -- so the compositing BACKWARD clears them (clear_counts of k_tiny_bwd /
// k_qraster_bwd); a forward nobody back-propagates leaves them to the host (RenderContext zeroes them).
struct FwdSort {
  uint64_t* bins;
  const int32_t* counts;
  int32_t* tile_offsets;  // out
  int32_t* flatten_ids;   // out
  int32_t* n_isects;      // out (total)
  int32_t* flags;         // flags[1], flags[2]: a tile outgrew its bin
  int bin_cap;
  const int32_t* storage_of;  // tile-order placement (gsloc_common.h, list_id); may be NULL
};
template <int SORT> struct FwdListPtr { typedef const int32_t* __restrict__ type; };
template <> struct FwdListPtr<2> { typedef const int32_t* type; };  // (the kernel writes what these point to)
template <> struct FwdListPtr<3> { typedef const int32_t* type; };

// long_min > 0: tiles whose list is longer than long_min entries are left to the long-list kernels below.
template <int D, bool ED, int SORT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SORT == 3 ? 4 : 6))) void k_praster_fwd(
    const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2, int W, int H,
    int tile_w, int ty0, typename FwdListPtr<SORT>::type tile_offsets, typename FwdListPtr<SORT>::type flatten_ids,
    long long capacity, float* __restrict__ render, float* __restrict__ alphas, int32_t* __restrict__ last_ids,
    int row0, int row1, const uint4* __restrict__ Qh, int32_t* __restrict__ clear_counts,
    int32_t* __restrict__ clear_state, uint32_t* __restrict__ isect_hits, int32_t* __restrict__ isect_hit_counts,
    int long_min, int n_tiles_total, FwdSort fs) {
  constexpr size_t SORT_BYTES = SORT ? (size_t)8 * 64 * (1 << SORT) * sizeof(uint64_t) : 0;  // two buffers of 4 runs
  constexpr size_t LDS_BYTES = sizeof(PStage<D>) > SORT_BYTES ? sizeof(PStage<D>) : SORT_BYTES;
  __shared__ __align__(16) unsigned char smem[LDS_BYTES];  // the sort's merge buffers, then the compositing stage
  PStage<D>& sb = *reinterpret_cast<PStage<D>*>(smem);
  __shared__ int s_scan[5];
  int tile = ty0 * tile_w + GSL_TILE_OF_BLOCK();
  long long sorted_rs = 0, sorted_re = 0;
  if constexpr (SORT != 0) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int acc = prefix_count_share(fs.counts, tile, fs.bin_cap, tid);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) s_scan[wv] = acc;
    if (tid == 0) {
      int c_own = fs.counts[tile];
      s_scan[4] = min(c_own, fs.bin_cap);
      if (c_own > fs.bin_cap && fs.flags) { fs.flags[1] = 1; atomicMax(&fs.flags[2], c_own); }
    }
    __syncthreads();
    long long s0 = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3], e0 = s0 + s_scan[4];
    if (tid == 0) {
      fs.tile_offsets[tile] = (int32_t)s0;
      if (tile == n_tiles_total - 1) {
        fs.tile_offsets[tile + 1] = (int32_t)e0;
        if (fs.n_isects) fs.n_isects[0] = (int32_t)e0;
      }
    }
    if (e0 > capacity) e0 = capacity;  // (the total tells the host; what fits is sorted and composited)
    int n = (int)max(e0 - s0, (long long)0);
    if (n > 256) {
      wg_sort_tile<SORT>(fs.bins + (size_t)tile * (size_t)fs.bin_cap, n, s0, tile, tid, reinterpret_cast<uint64_t*>(smem),
                         nullptr, fs.flatten_ids, nullptr, 0, fs.storage_of);
    } else if (n > 0 && wv == 0) {  // a short list: one wave's registers, no merge passes
      const uint64_t* src = fs.bins + (size_t)tile * (size_t)fs.bin_cap;
      uint64_t k[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) k[r] = (lane * 4 + r < n) ? src[lane * 4 + r] : GSL_SORT_PAD;
      wave_sort_regs<2>(k, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lane * 4 + r < n) fs.flatten_ids[s0 + lane * 4 + r] = list_id(fs.storage_of, k[r]);
    }
    __syncthreads();  // the list is in flatten_ids (this workgroup's own stores: visible after the barrier's fence)
    sorted_rs = s0;
    sorted_re = s0 + n;
  }
  // binned projection: the tile-size counter of this tile has been consumed by the sort kernel; clear it for the next
  // projection (no clearing launch; sizes of tiles outside the strip are never raised)
  if (clear_counts && threadIdx.x == 0) {
    clear_counts[tile] = 0;
    if (blockIdx.x == 0 && clear_state) *clear_state = 0;  // counters consumed and cleared: the next projection may bin
  }
  // "some quadrant needs the full-colour backward" flag behind the hit-list lengths (raster_g16.hip): cleared here
  if (isect_hit_counts && blockIdx.x == 0 && threadIdx.x == 0) isect_hit_counts[4 * n_tiles_total] = 0;
  int tid = threadIdx.x;
  TilePixel tp = tile_pixel(tile, tile_w, tid);
  int i = tp.i, j = tp.j;
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W) && (i >= row0) && (i < row1);
  bool done = !inside;

  long long rs, re;
  if constexpr (SORT != 0) {
    rs = sorted_rs;
    re = sorted_re;
  } else {
    rs = tile_offsets[tile];
    re = tile_offsets[tile + 1];
    if (re > capacity) re = capacity;
    if (rs > re) rs = re;
    if (long_min > 0 && re - rs > long_min) return;
  }

  float T = 1.f;
  int cur_idx = 0;
  float pix[D];
#pragma unroll
  for (int k = 0; k < D; ++k) pix[k] = 0.f;
  int n_hits = 0;
  praster_walk<D, 0>(sb, Q0, Q1, Q2, Qh, flatten_ids, rs, re, tid, px, py, tp.qx, tp.qy, tp.txi, tp.tyi, done, T, pix,
                     cur_idx, isect_hits, n_hits);
  if (isect_hits && (tid & 63) == 0) isect_hit_counts[tile * 4 + (tid >> 6)] = n_hits;
  if (inside) {
    size_t pid = (size_t)i * W + j;
    float A = 1.f - T;
    alphas[pid] = A;
    if (ED) pix[D - 1] = pix[D - 1] / fmaxf(A, 1e-10f);
#pragma unroll
    for (int k = 0; k < D; ++k) render[pid * D + k] = pix[k];
    last_ids[pid] = cur_idx;
  }
}

// ---------------------------------------------------------------------------------------------------
// Long tile lists split over workgroups (segment transmittances).
// A pile of splats in one tile -- the invalid pixels of a TUM depth frame all sit at the previous camera's origin
// (/root/reference/src/data/Image.py:29-35, my_gsplat/geometry.py:138-161 do not filter zero depths) and land on one
// spot when the camera has moved backwards -- used to be composited by ONE workgroup: 23 k entries, 3.8 + 4.9 ms forward
// + backward (round 2, scripts/pile_bench.py).  A list longer than long_min entries is now cut into segments of
// GSL_SEG entries, one workgroup each:
//   map  : k_long_map lists the (tile, segment) pairs of this frame (device side; the grid is a fixed upper bound);
//   pass A (k_long_fwd<.., 0>): per pixel, the segment's transmittance product  P_s = prod (1 - alpha)  over its entries
//          with alpha >= 1/255 (no stop rule, no colours);
//   pass B (k_long_fwd<.., 1>): the reference loop on the segment, started from  T0 = P_0 ... P_{s-1}  -- a pixel
//          whose T0 is already <= 1e-4 stopped in an earlier segment (the product only shrinks) and does nothing -- with
//          the stop rule; leaves the segment's colour partial, its last composited index and T (negative if the pixel
//          stopped inside);
//   combine (k_long_combine): adds up a tile's partials in segment order, takes T and the last index of the stopping
//          (or last) segment, normalises the expected depth and writes the image.
// The per-pixel sequence of composited splats is the reference's; only the association of the float32 transmittance
// product changes ((P_0 P_1) x ... instead of one running product), a last-bit effect.  The backward
// (raster_g16.hip, k_long_bwd) restarts each segment from the stored T and the colour partials of the later segments.
// ---------------------------------------------------------------------------------------------------
template <int D, int PASS>
__global__ __launch_bounds__(256) void k_long_fwd(
    const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2, int W, int H,
    int tile_w, const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids, long long capacity,
    int row0, int row1, const uint4* __restrict__ Qh, uint32_t* __restrict__ isect_hits, LongWs w) {
  __shared__ PStage<D> sb;
  int g = blockIdx.x;
  if (g >= w.n_seg[0]) return;
  int tile = w.seg_tile[g], sgm = w.seg_idx[g];
  if (tile < 0) return;  // a tile whose segments did not fit the workspace (flagged by k_long_map)
  int tid = threadIdx.x;
  TilePixel tp = tile_pixel(tile, tile_w, tid);
  int i = tp.i, j = tp.j;
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W) && (i >= row0) && (i < row1);
  long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
  if (re > capacity) re = capacity;
  long long ss = rs + (long long)sgm * GSL_SEG, se = min(ss + (long long)GSL_SEG, re);
  float pix[D];
#pragma unroll
  for (int k = 0; k < D; ++k) pix[k] = 0.f;
  size_t slot = (size_t)g * 256 + tid;
  if (PASS == 0) {
    float T = 1.f;
    int cur = 0;
    bool done = !inside;
    int nh = 0;
    praster_walk<D, 1>(sb, Q0, Q1, Q2, Qh, flatten_ids, ss, se, tid, px, py, tp.qx, tp.qy, tp.txi, tp.tyi, done, T, pix,
                       cur, nullptr, nh);
    w.P[slot] = T;
  } else {
    // T0 = P_0 ... P_{sgm-1}, multiplied in segment order; the loads are issued eight at a time (one after the other
    // they were 25 us of this pass on a 90-segment list)
    float T = 1.f;
    {
      const float* Pp = w.P + (size_t)(g - sgm) * 256 + tid;
      int s = 0;
      for (; s + 8 <= sgm; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Pp[(size_t)(s + u) * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) T *= v[u];
      }
      for (; s < sgm; ++s) T *= Pp[(size_t)s * 256];
    }
    bool dead = !inside || T <= GSL_T_STOP;  // stopped in an earlier segment
    bool done = dead;
    int cur = -1;
    int nh = 0;  // the segment's hit lists sit at 4 ss + quadrant (se - ss), their lengths in the workspace
    praster_walk<D, 0>(sb, Q0, Q1, Q2, Qh, flatten_ids, ss, se, tid, px, py, tp.qx, tp.qy, tp.txi, tp.tyi, done, T, pix,
                       cur, isect_hits, nh);
    if ((tid & 63) == 0) w.seg_qcnt[g * 4 + (tid >> 6)] = isect_hits ? nh : 0;
    // negative: the pixel stopped inside this segment (T is the value before the stop); 2: dead on arrival (a
    // transmittance is never 2), its T, index and partial are not this segment's to report
    w.Tend[slot] = dead ? 2.f : ((done && !dead) ? -T : T);
    w.last[slot] = cur;
#pragma unroll
    for (int k = 0; k < 4; ++k) w.C[slot * 4 + k] = (k < D) ? pix[k < D ? k : 0] : 0.f;
  }
}

// 1024 threads: pixel = tid & 255, and the tile's segments are dealt to the four 256-thread groups in contiguous
// quarters, walked in segment order (eight segments' records requested together), then joined in quarter order through
// LDS -- a pixel that never stops reads every segment, and one thread doing that alone was 55 us on a 90-segment list.
template <int D, bool ED>
__global__ __launch_bounds__(1024) void k_long_combine(int W, int H, int tile_w, float* __restrict__ render,
                                                       float* __restrict__ alphas, int32_t* __restrict__ last_ids, int row0,
                                                       int row1, LongWs w) {
  __shared__ float4 s_pix[3][256];
  __shared__ float s_T[3][256];
  __shared__ int s_last[3][256];
  __shared__ int s_state[3][256];  // bit 0: visited a segment, bit 1: stopped
  int g = blockIdx.x;
  if (g >= w.n_seg[0] || w.seg_idx[g] != 0) return;
  int tile = w.seg_tile[g], nseg = w.seg_cnt[g];
  if (tile < 0) return;
  int tid = threadIdx.x & 255, quarter = threadIdx.x >> 8;
  TilePixel tp = tile_pixel(tile, tile_w, tid);
  int i = tp.i, j = tp.j;
  bool inside = (i < H) && (j < W) && (i >= row0) && (i < row1);
  float pix[4] = {0.f, 0.f, 0.f, 0.f};
  float T = 1.f;
  int last = -1;
  bool stop = false, visited = false;
  const float4* C4 = reinterpret_cast<const float4*>(w.C);
  int per = (nseg + 3) >> 2, sb = quarter * per, se = min(sb + per, nseg);
  for (int s0 = sb; s0 < se && !stop && inside; s0 += 8) {
    float t8[8];
    int l8[8];
    float4 c8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      size_t slot = (size_t)(g + min(s0 + u, se - 1)) * 256 + tid;
      t8[u] = w.Tend[slot];
      l8[u] = w.last[slot];
      c8[u] = C4[slot];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (stop || s0 + u >= se) continue;
      float t = t8[u];
      if (t == 2.f) { stop = true; continue; }  // dead on arrival: the pixel stopped at the very end of the previous segment
      pix[0] += c8[u].x; pix[1] += c8[u].y; pix[2] += c8[u].z; pix[3] += c8[u].w;
      if (l8[u] >= 0) last = l8[u];
      T = fabsf(t);
      visited = true;
      if (t < 0.f) stop = true;  // the pixel stopped inside this segment
    }
  }
  if (quarter > 0) {
    s_pix[quarter - 1][tid] = make_float4(pix[0], pix[1], pix[2], pix[3]);
    s_T[quarter - 1][tid] = T;
    s_last[quarter - 1][tid] = last;
    s_state[quarter - 1][tid] = (visited ? 1 : 0) | (stop ? 2 : 0);
  }
  __syncthreads();
  if (quarter != 0 || !inside) return;
  for (int q = 0; q < 3 && !stop; ++q) {  // what the later quarters found counts only while no earlier one stopped
    float4 c = s_pix[q][tid];
    int st = s_state[q][tid];
    pix[0] += c.x; pix[1] += c.y; pix[2] += c.z; pix[3] += c.w;
    if (s_last[q][tid] >= 0) last = s_last[q][tid];
    if (st & 1) T = s_T[q][tid];
    stop = (st & 2) != 0;
  }
  if (last < 0) last = 0;
  size_t pid = (size_t)i * W + j;
  float A = 1.f - T;
  alphas[pid] = A;
  if (ED) pix[D - 1] = pix[D - 1] / fmaxf(A, 1e-10f);
#pragma unroll
  for (int k = 0; k < D; ++k) render[pid * D + k] = pix[k];
  last_ids[pid] = last;
}

// ---------------------------------------------------------------------------------------------------
// "Tiny splat" backward (every r_cull < 2 px: the alpha >= 1/255 disc covers at most 4x4 pixel centres --
// the situation of GsplatLoc's as-coded kNN scales, where every splat is the 0.3 px^2 blur).
// Pass 1 (this kernel, lane = pixel): the back-to-front walk of the px scheme; per composited
// (pixel, splat) it stores  (w, fac) = (vis * v_alpha [0 when alpha is clamped], alpha * T)  into the
// splat's own 4x4 slab  trec[g][row - r0][col - c0]  (plain 8-byte stores, no atomics, no reduction), and
// every pixel leaves its (expected-depth-chained) upstream gradient in vcT[H,W,D].
// Pass 2 runs inside the projection backward (k_fproject_bwd, tiny_slab_load + tiny_slab_fold in gsloc_common.h): four lanes per
// Gaussian read its slab, rebuild dx, dy from the Gaussian's own record and sum the 16 slots into its gradient row,
// which never leaves LDS.
// ---------------------------------------------------------------------------------------------------

template <int D>
struct TStage {
  float4 s0[256];
  float4 s1[256];
  float4 s2[(D >= 3) ? 256 : 1];
  uint16_t qlist[4][256];
  int qcnt[4][4];
  int32_t id[256];
};

// LOSS (the tracker's iteration, whole frame, no normal term): the kernel computes its tile's share of the depth + edge
// loss and the upstream gradient itself (loss_dev.h; the tile IS the loss kernel's 16x16 block) -- the separate loss
// launch, 9 of the 90 us of an iteration at 102 k Gaussians, disappears.  v_render is still written (the long-list
// backward and the tests read it) and partial[tile] as the loss kernel would.
struct TinyLoss {
  const float* gt;    // target depth [H,W]
  float* partial;     // [tiles][2]
  float* v_render;    // [H,W,D], channel D-1 written
  float depth_w, edge_w, inv_P;
};
template <bool LOSS>
struct TinyLossLds {
  LossLds L;
  float grad[16][16];
};
template <>
struct TinyLossLds<false> {};

template <int D, bool ED, bool LOSS>
__global__ __launch_bounds__(256) void k_tiny_bwd(
    const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2, int W, int H,
    int tile_w, int ty0, const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    long long capacity, const float* __restrict__ render, const float* __restrict__ alphas,
    const int32_t* __restrict__ last_ids, const float* __restrict__ v_render, const float* __restrict__ v_alphas,
    float2* __restrict__ trec, float* __restrict__ vcT, int row0, int row1, int32_t* __restrict__ flags, int long_min,
    TinyLoss tl, int32_t* __restrict__ clear_counts, int32_t* __restrict__ clear_state) {
  constexpr bool RGB = D >= 3;
  constexpr bool DEPTH = (D == 1) || (D == 4);
  __shared__ TStage<D> sb;
  __shared__ int s_final[4];
  __shared__ TinyLossLds<LOSS> sl;
  int tile = ty0 * tile_w + GSL_TILE_OF_BLOCK();
  // (the forward sorted its own bins and could not clear the tile counters: see k_praster_fwd, SORT)
  if (clear_counts && threadIdx.x == 0) {
    clear_counts[tile] = 0;
    if (blockIdx.x == 0 && clear_state) *clear_state = 0;
  }
  int tyi = tile / tile_w, txi = tile - tyi * tile_w;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int qx = txi * 16 + (wv & 1) * 8, qy = tyi * 16 + (wv >> 1) * 8;
  int j = qx + (lane & 7), i = qy + (lane >> 3);
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W) && (i >= row0) && (i < row1);
  size_t pid = inside ? ((size_t)i * W + j) : 0;
  if constexpr (LOSS) {
    float g, l1, le;
    bool has_pixel;
    loss_block(sl.L, render, D, tl.gt, W, H, 0, H, H, tl.depth_w, tl.edge_w, tl.inv_P, txi * 16, tyi * 16, tid, g,
               has_pixel, l1, le);
    sl.grad[tid >> 4][tid & 15] = g;
    if (has_pixel) tl.v_render[((size_t)(tyi * 16 + (tid >> 4)) * W + (txi * 16 + (tid & 15))) * D + (D - 1)] = g;
    float s1 = wave_sum(l1), s2 = wave_sum(le);
    if (lane == 0) { sl.L.red[wv][0] = s1; sl.L.red[wv][1] = s2; }
    __syncthreads();
    if (tid < 2)
      tl.partial[2 * (size_t)tile + tid] = sl.L.red[0][tid] + sl.L.red[1][tid] + sl.L.red[2][tid] + sl.L.red[3][tid];
  }
  float Aimg = inside ? alphas[pid] : 0.f;
  float T_final = 1.f - Aimg;
  int bin_final = inside ? last_ids[pid] : -1;
  float vc[D];
  float va = inside ? v_alphas[pid] : 0.f;
  if constexpr (LOSS) {
#pragma unroll
    for (int k = 0; k < D; ++k) vc[k] = 0.f;  // (the tracker's loss has no colour term)
    if (inside) vc[D - 1] = sl.grad[(wv >> 1) * 8 + (lane >> 3)][(wv & 1) * 8 + (lane & 7)];
  } else {
#pragma unroll
    for (int k = 0; k < D; ++k) vc[k] = inside ? v_render[pid * D + k] : 0.f;
  }
  if (ED && inside) {
    float dn = render[pid * D + (D - 1)];
    float vd = vc[D - 1];
    if (Aimg >= 1e-10f) va += -vd * dn / Aimg;
    vc[D - 1] = vd / fmaxf(Aimg, 1e-10f);
  }
  if (inside) {
#pragma unroll
    for (int k = 0; k < D; ++k) vcT[pid * D + k] = vc[k];
  }
  long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
  if (re > capacity) re = capacity;
  if (rs >= re) return;
  if (long_min > 0 && re - rs > long_min) return;  // long list: gsl_long_raster_bwd adds this tile's rows to vacc
  int wave_final = bin_final;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wave_final = max(wave_final, __shfl_xor(wave_final, o, 64));
  if (lane == 0) s_final[wv] = wave_final;
  __syncthreads();
  int block_final = max(max(s_final[0], s_final[1]), max(s_final[2], s_final[3]));
  if ((long long)block_final + 1 < re) re = max((long long)block_final + 1, rs);
  if (rs >= re) return;
  int nb = (int)((re - rs + 255) / 256);
  float T = T_final;
  float Bp = -T_final * va;

  for (int b = 0; b < nb; ++b) {
    long long bend = re - 1 - (long long)b * 256;
    int bsize = (int)min((long long)256, bend + 1 - rs);
    __syncthreads();
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = make_float4(0.f, 0.f, 0.f, -1.f);
    if (tid < bsize) {
      int g = flatten_ids[bend - tid];
      sb.id[tid] = g;
      r0 = GSL_Q(Q0, g);
      r1 = GSL_Q(Q1, g);
      sb.s0[tid] = r0;
      // the conic as the FORWARD stages it (times log2 e, diagonal halved: praster_walk), so that a trip evaluates alpha
      // with the forward's very operations and both sides take the same alpha >= 1/255 decision for every pair
      sb.s1[tid] = make_float4(r1.x * (0.5f * GSL_LOG2E), r1.y * GSL_LOG2E, r1.z * (0.5f * GSL_LOG2E), r1.w);
      if (RGB) sb.s2[tid] = GSL_Q(Q2, g);
    } else {  // (finite numbers in every slot: see praster_walk)
      sb.id[tid] = 0;
      sb.s0[tid] = r0;
      sb.s1[tid] = r1;
      if (RGB) sb.s2[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int n = compact_quadrants(sb, tid, tid < bsize, r0.x, r0.y, r1.w, (float)(txi * 16), (float)(tyi * 16));
    int t_first = (int)max((long long)0, bend - (long long)wave_final);
    int t_lane = inside ? (int)max((long long)0, bend - (long long)bin_final) : 1 << 30;
    for (int c = 0; c < n; c += 64) {
      int e = c + lane;
      int lox = 1, hix = 0, loy = 1, hiy = 0;
      if (e < n) {
        int t = sb.qlist[wv][e];
        if (t >= t_first) {
          float4 a0 = sb.s0[t];
          float r = sb.s1[t].w;
          box_range(a0.x - ((float)qx + 0.5f), r, lox, hix);
          box_range(a0.y - ((float)qy + 0.5f), r, loy, hiy);
        }
      }
      unsigned mlo, mhi;
      pixel_masks(lox, hix, loy, hiy, lane, mlo, mhi);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        unsigned m = half ? mhi : mlo;
        // straight-line trips under a wave-uniform loop, two candidates per trip, candidates popped as one-bit masks,
        // predicates as scalar lane masks (as in praster_walk): a lane without a candidate (it reads whatever slot number
        // sits in front of the chunk's list: a valid slot, finite numbers), or whose candidate fails the alpha tests,
        // runs the arithmetic with alpha 0 (1 / (1 - alpha) = 1: T and the running sum stay as they are) and stores
        // nothing
        const uint16_t* const ql = &sb.qlist[wv][c + half * 32];
        unsigned long long ACT = __ballot(m != 0);
        if (ACT) do {
          const unsigned b0 = m & (0u - m);
          m ^= b0;
          const unsigned b1 = m & (0u - m);
          m ^= b1;
          const unsigned long long TWO = __ballot(b1 != 0u);
          const int t0 = ql[ffbl_raw(b0)] & 255;
          const int t1 = ql[ffbl_raw(b1)] & 255;
          float4 p0 = sb.s0[t0], p1 = sb.s1[t0];
          float4 u0 = sb.s0[t1], u1 = sb.s1[t1];
          float dx0 = p0.x - px, dy0 = p0.y - py, dx1 = u0.x - px, dy1 = u0.y - py;
          float sg0 = fmaf(p1.y * dx0, dy0, fmaf(p1.x * dx0, dx0, p1.z * dy0 * dy0));  // log2(e) sigma, the forward's
          float sg1 = fmaf(u1.y * dx1, dy1, fmaf(u1.x * dx1, dx1, u1.z * dy1 * dy1));  // expression
          float vis0 = __builtin_amdgcn_exp2f(-sg0), vis1 = __builtin_amdgcn_exp2f(-sg1);
          float opv0 = p0.w * vis0, opv1 = u0.w * vis1;
          float al0 = fminf(GSL_ALPHA_MAX, opv0), al1 = fminf(GSL_ALPHA_MAX, opv1);
          const unsigned long long OK0 =
              ACT & __ballot(t0 >= t_lane) & __ballot(sg0 >= 0.f) & __ballot(al0 >= GSL_ALPHA_MIN);
          const unsigned long long OK1 =
              TWO & __ballot(t1 >= t_lane) & __ballot(sg1 >= 0.f) & __ballot(al1 >= GSL_ALPHA_MIN);
          float cd0 = 0.f, cd1 = 0.f;
          if constexpr (RGB) {
            float4 q20 = sb.s2[t0], q21 = sb.s2[t1];
            cd0 = q20.x * vc[0] + q20.y * vc[1] + q20.z * vc[2];
            cd1 = q21.x * vc[0] + q21.y * vc[1] + q21.z * vc[2];
          }
          if (DEPTH) { cd0 += p0.z * vc[D - 1]; cd1 += u0.z * vc[D - 1]; }
          const float a0 = sel_f32(OK0, al0, 0.f), a1 = sel_f32(OK1, al1, 0.f);
          const float ra0 = __builtin_amdgcn_rcpf(1.f - a0), ra1 = __builtin_amdgcn_rcpf(1.f - a1);  // (alpha 0: exactly 1)
          const float T0 = T * ra0;
          const float fac0 = a0 * T0;
          const float va0 = T0 * cd0 - ra0 * Bp;
          const float Bp0 = Bp + fac0 * cd0;
          const float T1 = T0 * ra1;
          const float fac1 = a1 * T1;
          const float va1 = T1 * cd1 - ra1 * Bp0;
          T = T1;
          Bp = Bp0 + fac1 * cd1;
          if (__builtin_amdgcn_inverse_ballot_w64(OK0)) {  // (exec = the mask: no per-lane test)
            float w = (opv0 <= GSL_ALPHA_MAX) ? vis0 * va0 : 0.f;
            int cc = j - tiny_origin(p0.x, p1.w), rr = i - tiny_origin(p0.y, p1.w);
            if ((unsigned)cc < 4u && (unsigned)rr < 4u)
              trec[(size_t)sb.id[t0] * 16 + rr * 4 + cc] = make_float2(w, fac0);
            else if (flags)
              flags[0] = 1;  // sticky: the splat outgrew its 4x4 slab (r_cull >= 2 px); polled by the host
          }
          if (__builtin_amdgcn_inverse_ballot_w64(OK1)) {
            float w = (opv1 <= GSL_ALPHA_MAX) ? vis1 * va1 : 0.f;
            int cc = j - tiny_origin(u0.x, u1.w), rr = i - tiny_origin(u0.y, u1.w);
            if ((unsigned)cc < 4u && (unsigned)rr < 4u)
              trec[(size_t)sb.id[t1] * 16 + rr * 4 + cc] = make_float2(w, fac1);
            else if (flags)
              flags[0] = 1;
          }
          ACT = __ballot(m != 0);
        } while (ACT);
      }
    }
  }
}


}  // namespace gsl

extern "C" int32_t* gsl_fused_bin_state(void* ws, int n_tiles);  // fused.hip: the state word inside ws

extern "C" int gsl_tiny_raster_bwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                   int height, int tile_w, int tile_h, int ty0, int ty1, const int32_t* tile_offsets,
                                   const int32_t* flatten_ids, int64_t capacity, const float* render,
                                   const float* alphas, const int32_t* last_ids, const float* v_render,
                                   const float* v_alphas, float* trec, float* vcT, int row0, int row1,
                                   int32_t* flags, int long_min, const float* loss_depth_gt, float depth_lambda,
                                   float edge_lambda, float* loss_partials, void* clear_ws, void* stream) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0 || row0 < 0 || row0 > row1)
    return GSL_ERR_BAD_ARG;
  if (tile_w * 16 < width || tile_h * 16 < height) return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render || !alphas || !last_ids || !v_render || !v_alphas || !trec || !vcT)
    return GSL_ERR_BAD_ARG;
  if (ed && channels == 3) return GSL_ERR_BAD_ARG;
  // loss_depth_gt != NULL: the kernel computes the tracking loss of gsl_tracking_loss itself and WRITES v_render's
  // depth channel and loss_partials[tiles][2]; whole frame only (the loss block of a tile is the tile)
  const bool loss = loss_depth_gt != nullptr;
  if (loss && (!loss_partials || ty0 != 0 || ty1 != tile_h || row0 != 0 || row1 < height ||
               tile_w != (width + 15) / 16 || tile_h != (height + 15) / 16 || (channels != 1 && channels != 4)))
    return GSL_ERR_BAD_ARG;
  if (ty0 == ty1) return GSL_OK;
  if (capacity > 0 && (!Q0 || !Q1 || !flatten_ids || (channels >= 3 && !Q2))) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int nblk = (ty1 - ty0) * tile_w;
  gsl::TinyLoss tl{loss_depth_gt, loss_partials, const_cast<float*>(v_render), depth_lambda, edge_lambda,
                   1.0f / ((float)width * (float)height)};
#define CALL_TB(DD, EE, LL)                                                                                  \
  hipLaunchKernelGGL((gsl::k_tiny_bwd<DD, EE, LL>), dim3(nblk), dim3(256), 0, st, (const float4*)Q0,        \
                     (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,         \
                     flatten_ids, (long long)capacity, render, alphas, last_ids, v_render, v_alphas,         \
                     (float2*)trec, vcT, row0, row1, flags, long_min, tl, (int32_t*)clear_ws,                \
                     clear_ws ? gsl_fused_bin_state(clear_ws, tile_w * tile_h) : (int32_t*)nullptr)
#define CALL_TL(DD, EE) do { if (loss) CALL_TB(DD, EE, true); else CALL_TB(DD, EE, false); } while (0)
  if (channels == 1) { if (ed) CALL_TL(1, true); else CALL_TL(1, false); }
  else if (channels == 3) { CALL_TB(3, false, false); }
  else if (channels == 4) { if (ed) CALL_TL(4, true); else CALL_TL(4, false); }
  else return GSL_ERR_BAD_ARG;
#undef CALL_TL
#undef CALL_TB
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}


#define GSL_P_DISPATCH(D, ED, CALL)                                \
  if (D == 1) { if (ED) CALL(1, true); else CALL(1, false); }      \
  else if (D == 3) { CALL(3, false); }                             \
  else if (D == 4) { if (ED) CALL(4, true); else CALL(4, false); } \
  else return GSL_ERR_BAD_ARG;

// Compositing forward (k_praster_fwd).  Pixel rows outside [row0, row1) are not touched.
// sort_bins != NULL (binned projection, whole frame, bin_cap <= 2048, long_min == 0): the kernel also does gsl_fused_bin's
// work for its tile -- gsl_fused_bin is then NOT called; tile_offsets, flatten_ids and n_isects are outputs, flags as in
// gsl_fused_bin, and the tile counters in binned_ws stay set until a compositing backward given clear_ws clears them.
extern "C" int gsl_fused_raster_fwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                 int height, int tile_w, int tile_h, int ty0, int ty1, const int32_t* tile_offsets,
                                 const int32_t* flatten_ids, int64_t capacity, float* render, float* alphas,
                                 int32_t* last_ids, int row0, int row1, const void* Qh, void* binned_ws,
                                 uint32_t* isect_hits, int32_t* isect_hit_counts, int long_min, void* sort_bins,
                                 int bin_cap, int32_t* n_isects, int32_t* flags, const int32_t* storage_of,
                                 void* stream) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0 || row0 < 0 || row0 > row1)
    return GSL_ERR_BAD_ARG;
  if (tile_w * 16 < width || tile_h * 16 < height) return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render || !alphas || !last_ids) return GSL_ERR_BAD_ARG;
  if (capacity > 0 && !flatten_ids) return GSL_ERR_BAD_ARG;
  if (capacity > 0 && !Qh && (!Q0 || !Q1 || (channels >= 3 && !Q2))) return GSL_ERR_BAD_ARG;
  if (ed && channels == 3) return GSL_ERR_BAD_ARG;
  if (isect_hits && !isect_hit_counts) return GSL_ERR_BAD_ARG;
  if (isect_hits && capacity >= ((int64_t)1 << GSL_HIT_SHIFT)) return GSL_ERR_BAD_ARG;  // a hit entry keeps the list index below its group bits
  const bool sort = sort_bins != nullptr;
  if (sort && (!binned_ws || !n_isects || bin_cap <= 0 || bin_cap > 2048 || long_min != 0 || ty0 != 0 || ty1 != tile_h ||
               !flatten_ids))
    return GSL_ERR_BAD_ARG;
  if (ty0 == ty1) return GSL_OK;
  hipStream_t st = (hipStream_t)stream;
  int nblk = (ty1 - ty0) * tile_w;
  gsl::FwdSort fs{(uint64_t*)sort_bins, (const int32_t*)binned_ws, const_cast<int32_t*>(tile_offsets),
                  const_cast<int32_t*>(flatten_ids), n_isects, flags, bin_cap, storage_of};
  // (sorting variant: the counters are still being added up by other workgroups -- the backward clears them)
  int32_t* clear_counts = sort ? nullptr : (int32_t*)binned_ws;
  int32_t* clear_state = sort ? nullptr : gsl_fused_bin_state(binned_ws, tile_w * tile_h);
  const int lk = !sort ? 0 : (bin_cap <= 1024 ? 2 : 3);
#define CALL_PF3(DD, EE, SS)                                                                                  \
  hipLaunchKernelGGL((gsl::k_praster_fwd<DD, EE, SS>), dim3(nblk), dim3(256), 0, st, (const float4*)Q0,      \
                     (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,          \
                     flatten_ids, (long long)capacity, render, alphas, last_ids, row0, row1, (const uint4*)Qh,     \
                     clear_counts, clear_state, isect_hits, isect_hit_counts, long_min, tile_w * tile_h, fs)
#define CALL_PF(DD, EE)                                                                          \
  do {                                                                                           \
    if (lk == 0) CALL_PF3(DD, EE, 0); else if (lk == 2) CALL_PF3(DD, EE, 2); else CALL_PF3(DD, EE, 3); \
  } while (0)
  GSL_P_DISPATCH(channels, ed, CALL_PF)
#undef CALL_PF
#undef CALL_PF3
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// ---- long tile lists (see the comment above k_long_map) ----------------------------------------------------------------
extern "C" int gsl_long_segment(void) { return GSL_SEG; }
extern "C" int gsl_long_sort_segment(void) { return GSL_SORT_SEG; }

extern "C" size_t gsl_long_ws_bytes(int max_seg) {
  if (max_seg <= 0) return 0;
  size_t b = 16 + (size_t)(3 + 4) * max_seg * 4 + 256;
  return b + (size_t)max_seg * 256 * 4 * (1 + 1 + 1 + 4);
}

extern "C" int gsl_long_raster_fwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                   int height, int tile_w, int tile_h, int ty0, int ty1, const int32_t* tile_offsets,
                                   const int32_t* flatten_ids, int64_t capacity, float* render, float* alphas,
                                   int32_t* last_ids, int row0, int row1, const void* Qh, uint32_t* isect_hits,
                                   int long_min, void* long_ws, size_t long_ws_bytes, int max_seg, int map_ready,
                                   void* stream) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0 || row0 < 0 || row0 > row1 || long_min <= 0 || max_seg <= 0)
    return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render || !alphas || !last_ids || !long_ws) return GSL_ERR_BAD_ARG;
  if (long_ws_bytes < gsl_long_ws_bytes(max_seg)) return GSL_ERR_WORKSPACE;
  if (capacity > 0 && (!flatten_ids || (!Qh && (!Q0 || !Q1 || (channels >= 3 && !Q2))))) return GSL_ERR_BAD_ARG;
  if (ed && channels == 3) return GSL_ERR_BAD_ARG;
  if (ty0 == ty1 || capacity == 0) return GSL_OK;
  hipStream_t st = (hipStream_t)stream;
  gsl::LongWs w = gsl::long_ws_views(long_ws, max_seg);
  if (!map_ready) {  // (gsl_long_sort of the same frame, strip and long_min has listed the segments already)
    hipLaunchKernelGGL(gsl::k_long_map, dim3(1), dim3(1024), 0, st, tile_offsets, ty0 * tile_w, (ty1 - ty0) * tile_w,
                       (long long)capacity, long_min, max_seg, 0, w);
    GSL_CHECK_LAUNCH();
  }
#define CALL_LF(DD, EE)                                                                                          \
  do {                                                                                                           \
    hipLaunchKernelGGL((gsl::k_long_fwd<DD, 0>), dim3(max_seg), dim3(256), 0, st, (const float4*)Q0,             \
                       (const float4*)Q1, (const float4*)Q2, width, height, tile_w, tile_offsets, flatten_ids,   \
                       (long long)capacity, row0, row1, (const uint4*)Qh, isect_hits, w);                        \
    hipLaunchKernelGGL((gsl::k_long_fwd<DD, 1>), dim3(max_seg), dim3(256), 0, st, (const float4*)Q0,             \
                       (const float4*)Q1, (const float4*)Q2, width, height, tile_w, tile_offsets, flatten_ids,   \
                       (long long)capacity, row0, row1, (const uint4*)Qh, isect_hits, w);                        \
    hipLaunchKernelGGL((gsl::k_long_combine<DD, EE>), dim3(max_seg), dim3(1024), 0, st, width, height, tile_w,   \
                       render, alphas, last_ids, row0, row1, w);                                                 \
  } while (0)
  GSL_P_DISPATCH(channels, ed, CALL_LF)
#undef CALL_LF
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
