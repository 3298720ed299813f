// Device-side pieces of the per-tile sort shared by the binning kernels (binning.hip) and by the compositing forward
// that sorts its own tile's bin (raster_px.hip): the register bitonic network of one wave, the merge-path search, and
// the one-tile-per-workgroup sort built from them.
#pragma once
#include "gsloc_common.h"

namespace gsl {

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = (unsigned)__shfl_xor((int)lo, mask, 64);
  hi = (unsigned)__shfl_xor((int)hi, mask, 64);
  return ((uint64_t)hi << 32) | lo;
}
// 32-bit select on a 64-bit scalar mask (v_cndmask_b32_e64 with an SGPR pair).  Written with ?: the compiler sends every
// compare of the network through VCC: v_cmp_lt -> s_nop -> two v_cndmask (VCC) -> v_cmp_gt -> s_nop -> two v_cndmask per
// compare-exchange -- two compares where one decides both outputs, and a hazard bubble per compare (round 4 found ~640 s_nop
// and 2 x 480 v_cmp_u64 in one 1024-key sort).  With the outcome as a ballot every compare-exchange is ONE v_cmp into its
// own SGPR pair and four selects, and consecutive compares do not wait for each other.
__device__ __forceinline__ unsigned sort_sel(unsigned long long m, unsigned t, unsigned f) {
  unsigned r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}
// Compare-exchange of two keys held in one lane: a <- min, b <- max.
// A key is (depth bits << 32) | Gaussian index with the depth a POSITIVE FINITE float (every projection of this library
// culls z <= 0 and z > far_plane), so its 64 bits read as a double are a positive finite double, and positive doubles
// order like their bit patterns: v_min_f64 / v_max_f64 return the smaller / larger KEY, bit for bit (they return one of
// their operands; no NaN, no -0 here).  Two instructions per compare-exchange instead of one 64-bit compare + four
// selects; all of them issue at the same rate on gfx950 (scripts/ubench/valu.hip kinds 93-99: 1.9 ns), so the in-register
// stages of the network cost 2/5 of what they did.  The padding key of a short list is GSL_SORT_PAD = +infinity (hi word
// 0x7FF00000: above every finite float's bits); ~0 would be a NaN, which min AND max both drop.
#define GSL_SORT_PAD 0x7FF0000000000000ull
__device__ __forceinline__ void cswap(uint64_t& a, uint64_t& b) {
  const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
  double lo, hi;
  asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
  asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
  a = (uint64_t)__double_as_longlong(lo);
  b = (uint64_t)__double_as_longlong(hi);
}
// keep_min_mask: ballot of the lanes that keep the smaller key of (a, partner's o)
__device__ __forceinline__ uint64_t pick(uint64_t a, uint64_t o, unsigned long long keep_min_mask) {
  const unsigned long long take = ~(__ballot(o < a) ^ keep_min_mask);  // take the partner's key where (o < a) == keep_min
  return ((uint64_t)sort_sel(take, (unsigned)(o >> 32), (unsigned)(a >> 32)) << 32) |
         sort_sel(take, (unsigned)o, (unsigned)a);
}

// Lane exchanges of the network.  Round 4: the sort kernel's waves spent a quarter of their cycles waiting to ISSUE an LDS
// instruction (SQ_WAIT_INST_LDS 1.4e7 of 5.8e7 wave-cycles; the compositing kernels: 0.7 %): 21 of the 55 stages of a
// 1024-key sort exchange 16 keys x 2 dwords through ds_bpermute.  Every exchange inside a 16-lane row is a DPP pattern --
// the flips of spans 2, 4, 8, 16 are quad_perm [1,0,3,2] / [3,2,1,0], row_half_mirror, row_mirror; the half-cleaners at
// distance 1, 2, 8 are quad_perm [1,0,3,2] / [2,3,0,1] and row_ror:8, distance 4 two bank-masked row shifts -- and costs 4-cycle
// VALU moves instead of an LDS round trip; only distance 16 and the two widest flips stay on ds_bpermute (3 of the 21 stages).
struct XShfl {
  int mask;
  __device__ __forceinline__ uint64_t operator()(uint64_t v) const { return shfl_xor_u64(v, mask); }
};
template <int CTRL>
struct XDpp {
  __device__ __forceinline__ uint64_t operator()(uint64_t v) const {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((uint64_t)hi << 32) | lo;
  }
};
struct XDppXor4 {  // lane <-> lane ^ 4: banks 0, 2 of a row take lane + 4 (row_shl:4), banks 1, 3 lane - 4 (row_shr:4)
  __device__ __forceinline__ static unsigned one(unsigned v) {
    int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
  }
  __device__ __forceinline__ uint64_t operator()(uint64_t v) const {
    return ((uint64_t)one((unsigned)(v >> 32)) << 32) | one((unsigned)v);
  }
};
template <int KPT, typename X>
__device__ __forceinline__ void flip_stage(uint64_t (&k)[KPT], X xchg, unsigned long long keep_min) {
#pragma unroll
  for (int r = 0; r < KPT / 2; ++r) {
    uint64_t o_r = xchg(k[KPT - 1 - r]);
    uint64_t o_p = xchg(k[r]);
    k[r] = pick(k[r], o_r, keep_min);
    k[KPT - 1 - r] = pick(k[KPT - 1 - r], o_p, keep_min);
  }
}
template <int KPT, typename X>
__device__ __forceinline__ void clean_stage(uint64_t (&k)[KPT], X xchg, unsigned long long km) {
#pragma unroll
  for (int r = 0; r < KPT; ++r) k[r] = pick(k[r], xchg(k[r]), km);
}

template <int LK>
__device__ __forceinline__ void wave_sort_regs(uint64_t (&k)[1 << LK], int lane) {
  constexpr int KPT = 1 << LK;
  // stages whose blocks fit inside one lane's registers
#pragma unroll
  for (int lk = 1; lk <= LK; ++lk) {
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
      int p = r ^ ((1 << lk) - 1);
      if (r < p) cswap(k[r], k[p]);
    }
#pragma unroll
    for (int lj = lk - 2; lj >= 0; --lj)
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        int p = r ^ (1 << lj);
        if (r < p) cswap(k[r], k[p]);
      }
  }
  // stages that span 2^tb lanes: the flip pairs (lane, r) with (lane ^ (2^tb - 1), KPT - 1 - r), the half-cleaners
  // at lane distance 2^b pair equal registers, the rest is register-to-register again
#pragma unroll 1
  for (int tb = 1; tb <= 6; ++tb) {
    int mask = (1 << tb) - 1;
    const unsigned long long keep_min = __ballot(((lane >> (tb - 1)) & 1) == 0);
    switch (tb) {  // lane <-> lane ^ mask: the mirror of a span of 2^tb lanes
      case 1: flip_stage<KPT>(k, XDpp<0xB1>(), keep_min); break;   // quad_perm [1,0,3,2]
      case 2: flip_stage<KPT>(k, XDpp<0x1B>(), keep_min); break;   // quad_perm [3,2,1,0]
      case 3: flip_stage<KPT>(k, XDpp<0x141>(), keep_min); break;  // row_half_mirror
      case 4: flip_stage<KPT>(k, XDpp<0x140>(), keep_min); break;  // row_mirror
      default: flip_stage<KPT>(k, XShfl{mask}, keep_min); break;
    }
#pragma unroll 1
    for (int b = tb - 2; b >= 0; --b) {
      const unsigned long long km = __ballot(((lane >> b) & 1) == 0);
      switch (b) {  // lane <-> lane ^ 2^b
        case 0: clean_stage<KPT>(k, XDpp<0xB1>(), km); break;   // quad_perm [1,0,3,2]
        case 1: clean_stage<KPT>(k, XDpp<0x4E>(), km); break;   // quad_perm [2,3,0,1]
        case 2: clean_stage<KPT>(k, XDppXor4(), km); break;
        case 3: clean_stage<KPT>(k, XDpp<0x128>(), km); break;  // row_ror:8
        default: clean_stage<KPT>(k, XShfl{1 << b}, km); break;
      }
    }
#pragma unroll
    for (int lj = LK - 1; lj >= 0; --lj)
#pragma unroll
      for (int r = 0; r < KPT; ++r) {
        int p = r ^ (1 << lj);
        if (r < p) cswap(k[r], k[p]);
      }
  }
}

// Sum of min(counts[i], cap) over i < n by the 256 threads of a workgroup (this thread's share; the caller folds lanes and
// waves): the offset of a tile's list = the sizes of the tiles before it.  Eight loads in flight per thread: written as
// "for (i = tid; i < n; i += 256) acc += ..." the compiler waits for every load before it issues the next, and the last
// workgroups of a 3 225-tile frame paid 13 dependent L2 round trips -- that loop, not the sorting network, was the sort
// kernel's 47 us at R (round 4: the kernel WITHOUT the network took as long).
__device__ __forceinline__ int prefix_count_share(const int32_t* __restrict__ counts, int n, int cap, int tid) {
  int acc = 0;
  for (int base = 0; base < n; base += 256 * 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + tid;
      v[u] = (i < n) ? counts[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += min(v[u], cap);
  }
  return acc;
}

// smallest ia in [lo, hi] such that the first d merged elements take ia from A (keys are unique)
__device__ __forceinline__ int merge_diag(const uint64_t* __restrict__ A, int lenA, const uint64_t* __restrict__ B, int lenB,
                                          int d) {
  int lo = max(0, d - lenB), hi = min(d, lenA);
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (A[mid] <= B[d - 1 - mid]) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------------
// One tile per 256-thread workgroup (lists of a few hundred to 2048 keys).  One wave sorting a 1 460-key list in registers
// is 60 us of pure latency when a strip has fewer tiles than the chip has SIMDs (8 strips of workload X), and a bitonic
// network over the padded power of two does n log^2 n work.  Here every wave sorts a quarter of the list in registers
// (wave_sort_regs), the four runs go to LDS and two merge-path passes (every thread merges its 2^LK outputs) finish the
// job: a third of the compare-exchanges, a quarter of the critical path.  Same result: the keys are unique.
// ------------------------------------------------------------------------------------------------
template <int LK>
__device__ __forceinline__ void wg_sort_tile(const uint64_t* __restrict__ src, int n, long long s, int t, int tid,
                                             uint64_t* __restrict__ lds, uint64_t* __restrict__ keys_out,
                                             int32_t* __restrict__ flatten_ids, int64_t* __restrict__ isect_ids,
                                             int64_t cam_enc, const int32_t* __restrict__ storage_of = nullptr) {
  constexpr int KPT = 1 << LK, RUN = 64 * KPT;
  const int lane = tid & 63, wv = tid >> 6;
  uint64_t k[KPT];
  const int e0 = wv * RUN + lane * KPT;
#pragma unroll
  for (int r = 0; r < KPT; ++r) k[r] = (e0 + r < n) ? src[e0 + r] : GSL_SORT_PAD;
  wave_sort_regs<LK>(k, lane);
  uint64_t* bufA = lds;
  uint64_t* bufB = lds + 4 * RUN;
#pragma unroll
  for (int r = 0; r < KPT; ++r) bufA[e0 + r] = k[r];
  __syncthreads();
  {  // runs (0, 1) and (2, 3) -> two runs of 2 RUN in bufB
    const int p = tid >> 7, l = tid & 127;
    const uint64_t* A = bufA + p * 2 * RUN;
    const uint64_t* B = A + RUN;
    const int d0 = l * KPT;
    int ia = merge_diag(A, RUN, B, RUN, d0), ib = d0 - ia;
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      uint64_t v;
      if (ib >= RUN || (ia < RUN && A[ia] <= B[ib])) v = A[ia++];
      else v = B[ib++];
      bufB[p * 2 * RUN + d0 + q] = v;
    }
  }
  __syncthreads();
  {  // the two runs of 2 RUN -> the result
    const uint64_t* A = bufB;
    const uint64_t* B = bufB + 2 * RUN;
    const int d0 = tid * KPT;
    if (d0 < n) {
      int ia = merge_diag(A, 2 * RUN, B, 2 * RUN, d0), ib = d0 - ia;
#pragma unroll
      for (int q = 0; q < KPT; ++q) {
        uint64_t v;
        if (ib >= 2 * RUN || (ia < 2 * RUN && A[ia] <= B[ib])) v = A[ia++];
        else v = B[ib++];
        if (d0 + q < n) {
          flatten_ids[s + d0 + q] = list_id(storage_of, v);
          if (isect_ids) isect_ids[s + d0 + q] = cam_enc | ((int64_t)t << 32) | (int64_t)(v >> 32);
          if (keys_out) keys_out[s + d0 + q] = v;
        }
      }
    }
  }
}

}  // namespace gsl
