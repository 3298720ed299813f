// Tile binning: which 16x16 screen tiles does each projected Gaussian touch, and in what
// depth order does each tile see them.  Replaces gsplat.isect_tiles (two-pass count/emit +
// a device-wide 64-bit radix sort) and isect_offset_encode (IDX:14360, IDX:14369).
//
// MI355X design: instead of one global radix sort over (tile|depth) keys -- 6 passes x 24 B
// per intersection through HBM -- intersections are bucketed by tile with one counting pass
// (per-tile histogram -> exclusive scan -> scatter) and every tile's list is then sorted on
// (depth bits, Gaussian index) by one workgroup entirely inside LDS (160 KiB per CU).  The
// composite key makes the result independent of scatter order and identical to a stable
// global sort of the gsplat key.  Wave64 lanes that hit the same tile are merged with a
// ballot so hot tiles see one atomic per wave instead of 64.
#include <cstdlib>
#include "gsloc_common.h"
#include "sort_dev.h"

namespace gsl {

#define GSL_SORT_LDS_CAP 4096  // keys per LDS block of the long-list sort (32 KiB: the wave sorts are limited to four workgroups per CU by their registers anyway)

// Per-wave merged atomic add of 1 on ctr[key]: lanes holding the same key elect a leader.
// Returns the position (old value + rank among equal lanes) for `active` lanes.
// At most MERGE_ROUNDS leader rounds, the rest fall back to plain atomics (random screen order).
__device__ __forceinline__ int merged_atomic_inc(int32_t* __restrict__ ctr, int key, bool active) {
  int pos = 0;
  unsigned long long todo = __ballot(active);
  int lane = threadIdx.x & 63;
#pragma unroll 1
  for (int round = 0; round < 4 && todo; ++round) {
    int leader = __ffsll((long long)todo) - 1;
    int lkey = __shfl(key, leader, 64);
    unsigned long long same = __ballot(active && key == lkey) & todo;
    int cnt = __popcll(same);
    int base = 0;
    if (lane == leader) base = atomicAdd(&ctr[lkey], cnt);
    base = __shfl(base, leader, 64);
    if ((same >> lane) & 1ull) {
      pos = base + __popcll(same & ((1ull << lane) - 1ull));
      active = false;
    }
    todo &= ~same;
  }
  if (active) pos = atomicAdd(&ctr[key], 1);
  return pos;
}

// Pass 1: per-Gaussian tile count (strip-clipped) + per-tile histogram.
__global__ __launch_bounds__(256) void k_isect_count(const float* __restrict__ means2d,
                                                     const int32_t* __restrict__ radii, int N, int tile_size,
                                                     int tile_w, int tile_h, int ty0, int ty1,
                                                     int32_t* __restrict__ tiles_per_gauss,
                                                     int32_t* __restrict__ tile_counts) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int xmin = 0, ymin = 0, xmax = 0, ymax = 0;
  if (i < N) {
    int r = radii[i];
    if (r > 0) {
      tile_rect(means2d[2 * (size_t)i], means2d[2 * (size_t)i + 1], r, tile_size, tile_w, tile_h, xmin, ymin, xmax,
                ymax);
      ymin = max(ymin, ty0);
      ymax = min(ymax, ty1);
      if (ymax < ymin) ymax = ymin;
    }
    if (tiles_per_gauss) tiles_per_gauss[i] = (xmax - xmin) * (ymax - ymin);
  }
  // walk the rectangle; all lanes of the wave iterate together so the ballot merge sees them
  int w = xmax - xmin, n = w * (ymax - ymin);
  int nmax = n;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
  for (int k = 0; k < nmax; ++k) {
    bool act = k < n;
    int t = 0;
    if (act) t = (ymin + k / w) * tile_w + xmin + k % w;
    merged_atomic_inc(tile_counts, t, act);
  }
}

// ---- LDS-privatised variants (strip of <= GSL_HIST_LDS_TILES tiles) -------------------------
// Each 512-thread workgroup histograms its Gaussians' tiles in LDS (ds_add, no return) and
// touches global memory once per distinct tile: for raster-ordered splats (one per pixel of the
// previous depth frame) that is a few dozen atomics per workgroup instead of one per intersection.
#define GSL_HIST_LDS_TILES 8192
#define GSL_BIN_THREADS 512

__device__ __forceinline__ void strip_rect(const float* __restrict__ means2d, const int32_t* __restrict__ radii, int i,
                                           int N, int tile_size, int tile_w, int tile_h, int ty0, int ty1, int& xmin,
                                           int& ymin, int& xmax, int& ymax) {
  xmin = ymin = xmax = ymax = 0;
  if (i < N) {
    int r = radii[i];
    if (r > 0) {
      tile_rect(means2d[2 * (size_t)i], means2d[2 * (size_t)i + 1], r, tile_size, tile_w, tile_h, xmin, ymin, xmax,
                ymax);
      ymin = max(ymin, ty0);
      ymax = min(ymax, ty1);
      if (ymax < ymin) ymax = ymin;
    }
  }
}

__global__ __launch_bounds__(GSL_BIN_THREADS) void k_isect_count_lds(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, int N, int tile_size, int tile_w, int tile_h,
    int ty0, int ty1, int32_t* __restrict__ tiles_per_gauss, int32_t* __restrict__ tile_counts) {
  extern __shared__ int s_hist[];
  int nst = (ty1 - ty0) * tile_w, tbase = ty0 * tile_w;
  for (int k = threadIdx.x; k < nst; k += GSL_BIN_THREADS) s_hist[k] = 0;
  __syncthreads();
  int i = blockIdx.x * GSL_BIN_THREADS + threadIdx.x;
  int xmin, ymin, xmax, ymax;
  strip_rect(means2d, radii, i, N, tile_size, tile_w, tile_h, ty0, ty1, xmin, ymin, xmax, ymax);
  if (i < N && tiles_per_gauss) tiles_per_gauss[i] = (xmax - xmin) * (ymax - ymin);
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) atomicAdd(&s_hist[y * tile_w + x - tbase], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < nst; k += GSL_BIN_THREADS) {
    int c = s_hist[k];
    if (c) atomicAdd(&tile_counts[tbase + k], c);
  }
}

__global__ __launch_bounds__(GSL_BIN_THREADS) void k_isect_scatter_lds(
    const float* __restrict__ means2d, const int32_t* __restrict__ radii, const float* __restrict__ depths, int N,
    int tile_size, int tile_w, int tile_h, int ty0, int ty1, const int32_t* __restrict__ tile_offsets,
    int32_t* __restrict__ cursors, long long capacity, uint64_t* __restrict__ keys) {
  extern __shared__ int s_mem[];
  int nst = (ty1 - ty0) * tile_w, tbase = ty0 * tile_w;
  int* s_cnt = s_mem;
  int* s_base = s_mem + nst;
  for (int k = threadIdx.x; k < nst; k += GSL_BIN_THREADS) s_cnt[k] = 0;
  __syncthreads();
  int i = blockIdx.x * GSL_BIN_THREADS + threadIdx.x;
  int xmin, ymin, xmax, ymax;
  strip_rect(means2d, radii, i, N, tile_size, tile_w, tile_h, ty0, ty1, xmin, ymin, xmax, ymax);
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) atomicAdd(&s_cnt[y * tile_w + x - tbase], 1);
  __syncthreads();
  // one returning global atomic per distinct tile reserves this workgroup's span of the bucket
  for (int k = threadIdx.x; k < nst; k += GSL_BIN_THREADS) {
    int c = s_cnt[k];
    if (c) {
      s_base[k] = tile_offsets[tbase + k] + atomicAdd(&cursors[tbase + k], c);
      s_cnt[k] = 0;
    }
  }
  __syncthreads();
  if (xmax > xmin && ymax > ymin) {
    uint64_t key = ((uint64_t)__float_as_uint(depths[i]) << 32) | (uint32_t)i;
    for (int y = ymin; y < ymax; ++y)
      for (int x = xmin; x < xmax; ++x) {
        int lt = y * tile_w + x - tbase;
        long long pos = (long long)s_base[lt] + atomicAdd(&s_cnt[lt], 1);
        if (pos < capacity) keys[pos] = key;
      }
  }
}

// Exclusive scan of tile_counts[n] -> offsets[n+1]; total -> n_isects; zero the cursors.
__global__ __launch_bounds__(1024) void k_tile_scan(const int32_t* __restrict__ counts, int n,
                                                    int32_t* __restrict__ offsets, int32_t* __restrict__ n_isects,
                                                    int32_t* __restrict__ cursors) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int v = (i < n) ? counts[i] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    int carry = carry_s;
    if (i < n) {
      offsets[i] = carry + woff + x - v;
      cursors[i] = 0;
    }
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) {
    offsets[n] = carry_s;
    n_isects[0] = carry_s;
  }
}

// Pass 2: scatter (depth bits, Gaussian id) into the tile buckets.
__global__ __launch_bounds__(256) void k_isect_scatter(const float* __restrict__ means2d,
                                                       const int32_t* __restrict__ radii,
                                                       const float* __restrict__ depths, int N, int tile_size,
                                                       int tile_w, int tile_h, int ty0, int ty1,
                                                       const int32_t* __restrict__ tile_offsets,
                                                       int32_t* __restrict__ cursors, long long capacity,
                                                       uint64_t* __restrict__ keys) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int xmin = 0, ymin = 0, xmax = 0, ymax = 0;
  uint64_t key = 0;
  if (i < N) {
    int r = radii[i];
    if (r > 0) {
      tile_rect(means2d[2 * (size_t)i], means2d[2 * (size_t)i + 1], r, tile_size, tile_w, tile_h, xmin, ymin, xmax,
                ymax);
      ymin = max(ymin, ty0);
      ymax = min(ymax, ty1);
      if (ymax < ymin) ymax = ymin;
      key = ((uint64_t)__float_as_uint(depths[i]) << 32) | (uint32_t)i;
    }
  }
  int w = xmax - xmin, n = w * (ymax - ymin);
  int nmax = n;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
  for (int k = 0; k < nmax; ++k) {
    bool act = k < n;
    int t = 0;
    if (act) t = (ymin + k / w) * tile_w + xmin + k % w;
    int p = merged_atomic_inc(cursors, t, act);
    if (act) {
      long long pos = (long long)tile_offsets[t] + p;
      if (pos < capacity) keys[pos] = key;
    }
  }
}

// Ascending-only bitonic network on n (arbitrary) 64-bit keys; comparators whose upper index
// falls past n are skipped (equivalent to +inf padding).  One comparator sub-step in global memory: used by the
// long-list sort below for the sub-steps that cross 4096-key blocks.
__device__ __forceinline__ void bitonic_flip_step(uint64_t* a, int n, int half, int k, int tid, int nthreads) {
  int hk = k >> 1;
  for (int i = tid; i < half; i += nthreads) {
    int blk = i / hk, off = i - blk * hk;
    int lo = blk * k + off;
    int hi = blk * k + (k - 1 - off);
    if (hi < n) {
      uint64_t x = a[lo], y = a[hi];
      if (x > y) { a[lo] = y; a[hi] = x; }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void bitonic_half_step(uint64_t* a, int n, int half, int j, int tid, int nthreads) {
  for (int i = tid; i < half; i += nthreads) {
    int blk = i / j, off = i - blk * j;
    int lo = blk * 2 * j + off;
    int hi = lo + j;
    if (hi < n) {
      uint64_t x = a[lo], y = a[hi];
      if (x > y) { a[lo] = y; a[hi] = x; }
    }
  }
  __syncthreads();
}

// LDS version for a 256-thread workgroup.  Each of `nw` working waves owns a contiguous segment
// of S = P/nw keys; every sub-step whose comparator block fits inside a segment needs no
// workgroup barrier (a wave's LDS operations complete in order), which leaves 2-5 s_barriers
// per sort instead of log^2(P)/2.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* a, int n, int tid) {
  int lgP = 0;
  while ((1 << lgP) < n) ++lgP;
  int P = 1 << lgP;
  if (P < 2) return;
  int nw = P >= 512 ? 4 : (P >= 256 ? 2 : 1);
  int S = P / nw;           // keys per wave segment
  int pairs_w = S >> 1;     // comparators per wave per sub-step
  int wv = tid >> 6, lane = tid & 63;
  bool work = wv < nw;
  int pbase = wv * pairs_w;
  for (int lk = 1; lk <= lgP; ++lk) {  // stage k = 2^lk
    int k = 1 << lk, hk = k >> 1;
    if (work) {
      for (int q = lane; q < pairs_w; q += 64) {
        int i = pbase + q;
        int off = i & (hk - 1);
        int base = (i >> (lk - 1)) << lk;
        int lo = base + off;
        int hi = base + (k - 1 - off);
        if (hi < n) {
          uint64_t x = a[lo], y = a[hi];
          if (x > y) { a[lo] = y; a[hi] = x; }
        }
      }
    }
    if (k > S) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    for (int lj = lk - 2; lj >= 0; --lj) {  // distance j = 2^lj
      int j = 1 << lj;
      if (work) {
        for (int q = lane; q < pairs_w; q += 64) {
          int i = pbase + q;
          int lo = ((i >> lj) << (lj + 1)) + (i & (j - 1));
          int hi = lo + j;
          if (hi < n) {
            uint64_t x = a[lo], y = a[hi];
            if (x > y) { a[lo] = y; a[hi] = x; }
          }
        }
      }
      // a barrier is needed whenever this or the next sub-step crosses wave segments
      bool cross = (2 * j > S) || (j > 1 ? (j > S) : (2 * k > S));
      if (cross) __syncthreads();
      else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    }
  }
  __syncthreads();
}

// Lists longer than the LDS capacity (a pile of splats in one tile: e.g. the invalid pixels of a TUM depth frame,
// which all sit at the previous camera's origin).  Same network, run block-wise: every stage k <= CAP is the LDS sort
// of one aligned CAP-key block; of a stage k > CAP only the sub-steps at distance >= CAP touch global memory, the
// remaining log2(CAP) sub-steps stay inside aligned blocks and run in LDS.  For n = 24 k: 6 global sub-steps
// instead of 120.
__device__ __forceinline__ void bitonic_sort_long(uint64_t* a, int n, uint64_t* lds, int tid) {
  constexpr int CAP = GSL_SORT_LDS_CAP;
  int nblk = (n + CAP - 1) / CAP;
  for (int b = 0; b < nblk; ++b) {
    int nb = min(CAP, n - b * CAP);
    __syncthreads();
    for (int i = tid; i < nb; i += 256) lds[i] = a[b * CAP + i];
    __syncthreads();
    bitonic_sort_lds(lds, nb, tid);
    for (int i = tid; i < nb; i += 256) a[b * CAP + i] = lds[i];
  }
  __syncthreads();
  int P = CAP;
  while (P < n) P <<= 1;
  int half = P >> 1;
  for (int k = 2 * CAP; k <= P; k <<= 1) {
    bitonic_flip_step(a, n, half, k, tid, 256);
    for (int j = k >> 2; j >= CAP; j >>= 1) bitonic_half_step(a, n, half, j, tid, 256);
    for (int b = 0; b < nblk; ++b) {
      int nb = min(CAP, n - b * CAP);
      for (int i = tid; i < nb; i += 256) lds[i] = a[b * CAP + i];
      __syncthreads();
      for (int j = CAP >> 1; j >= 1; j >>= 1) bitonic_half_step(lds, nb, CAP >> 1, j, tid, 256);
      for (int i = tid; i < nb; i += 256) a[b * CAP + i] = lds[i];
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Per-tile sort, one WAVE per tile: the same ascending bitonic network, with the keys held in registers.
// Element e = lane * KPT + r lives in register r of lane `lane` (KPT = 4 .. 32 keys per lane, P = 64 KPT >= n, padded
// with +inf).  Comparator distances below KPT are register-to-register (no data movement at all: 34 of the 55
// sub-steps at P = 1024), the others exchange through the cross-lane network (ds_bpermute, no memory), and nothing
// needs a barrier or LDS.  Measured against the LDS version it replaces: see DESIGN.md.  Lists longer than 2048
// entries (a pile of splats in one tile) are sorted by the whole workgroup block-wise, bitonic_sort_long.
// ------------------------------------------------------------------------------------------------
// I/O (round 4): the kernel WITHOUT the network took as long as with it -- its 47 us at R were the loads and stores: lane L
// holding elements 16 L .. 16 L + 15 reads (and writes) 64 different cache lines per instruction.  The network does not care
// where an input key starts, so the keys are loaded lane-interleaved (element r * 64 + L: one 512-byte run per instruction);
// the sorted ids are transposed through the wave's own 8 KiB of LDS (rows rotated by the lane: no bank pile-up) and leave as
// contiguous 256-byte runs.  `ids_lds`: 2048 ints private to this wave.
template <int LK>
__device__ __forceinline__ void wave_sort_tile(const uint64_t* __restrict__ src, int n, long long s, int t, int lane,
                                               uint64_t* __restrict__ keys_out, int32_t* __restrict__ flatten_ids,
                                               int64_t* __restrict__ isect_ids, int64_t cam_enc,
                                               const int32_t* __restrict__ storage_of, int32_t* ids_lds) {
  constexpr int KPT = 1 << LK;
  uint64_t k[KPT];
  int e0 = lane * KPT;
#pragma unroll
  for (int r = 0; r < KPT; ++r) k[r] = (r * 64 + lane < n) ? src[r * 64 + lane] : GSL_SORT_PAD;
  wave_sort_regs<LK>(k, lane);
  if (LK <= 4 && !isect_ids && !keys_out) {  // (32 keys per lane: the ids' registers would cost the instance a wave per SIMD)
#pragma unroll
    for (int r = 0; r < KPT; ++r)
      ids_lds[e0 + ((r + lane) & (KPT - 1))] = (e0 + r < n) ? list_id(storage_of, k[r]) : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < KPT; ++r) {
      const int e = r * 64 + lane, row = e >> LK, col = e & (KPT - 1);
      if (e < n) flatten_ids[s + e] = ids_lds[row * KPT + ((col + row) & (KPT - 1))];
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < KPT; ++r)
    if (e0 + r < n) {
      uint64_t v = k[r];
      flatten_ids[s + e0 + r] = list_id(storage_of, v);
      if (isect_ids) isect_ids[s + e0 + r] = cam_enc | ((int64_t)t << 32) | (int64_t)(v >> 32);
      if (keys_out) keys_out[s + e0 + r] = v;
    }
}

#define GSL_SORT_WAVE_MAX 2048  // longest list one wave sorts in registers (32 keys per lane)

// Four tiles per 256-thread workgroup, one per wave; write flatten_ids (+ gsplat-style isect_ids, + the sorted keys
// when the deterministic backward wants them).  The unsorted keys of a tile are its span of `keys`, or its
// fixed-capacity bin (binned projection).
// counts != nullptr (binned projection, launched over ALL tiles): tile_offsets is an OUTPUT -- every workgroup adds up
// the sizes of the tiles before its own (a few coalesced loads per thread) and its waves write the offsets of their
// tiles, the last one also the total; a tile that outgrew its bin keeps bin_cap entries and raises flags[1] (its size
// goes to flags[2]).  The separate single-workgroup scan launch disappears; the counters are cleared later by the
// compositing forward (every workgroup may still be reading them here).
// MAXLK = 4 / 5: the longest list one wave sorts in registers is 1024 / 2048 keys; longer ones go to the workgroup's LDS
// sort below.  The 32-keys-per-lane network is what sets the kernel's register count (141 VGPR once every compare is a
// ballot: three waves per SIMD, one fewer than a frame of 3 225 tiles needs to be resident at once), so frames whose lists
// are expected to stay below 1024 keys run the instance without it (gsl_tile_sort_keys).
template <int MAXLK>
__global__ __launch_bounds__(256) void k_tile_sort(int32_t* __restrict__ tile_offsets, int tile_begin,
                                                   int n_strip_tiles, long long capacity,
                                                   uint64_t* __restrict__ keys, int32_t* __restrict__ flatten_ids,
                                                   int64_t* __restrict__ isect_ids, int64_t cam_enc,
                                                   int write_sorted_keys, uint64_t* __restrict__ bins, int bin_cap,
                                                   const int32_t* __restrict__ counts, int32_t* __restrict__ n_isects,
                                                   int32_t* __restrict__ flags, int long_min,
                                                   const int32_t* __restrict__ storage_of) {
  __shared__ uint64_t skeys[GSL_SORT_LDS_CAP];
  __shared__ int s_scan[8];
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (counts) {
    int first = tile_begin + blockIdx.x * 4;
    int acc = prefix_count_share(counts, first, bin_cap, tid);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    int t_own = first + wv;
    int c_own = (blockIdx.x * 4 + wv < n_strip_tiles) ? counts[t_own] : 0;
    if (lane == 0) {
      s_scan[wv] = acc;
      s_scan[4 + wv] = min(c_own, bin_cap);
      if (c_own > bin_cap && flags) { flags[1] = 1; atomicMax(&flags[2], c_own); }
    }
    __syncthreads();
    int base = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    for (int w = 0; w < wv; ++w) base += s_scan[4 + w];
    if (lane == 0 && blockIdx.x * 4 + wv < n_strip_tiles) {
      tile_offsets[t_own] = base;
      if (blockIdx.x * 4 + wv == n_strip_tiles - 1) {
        tile_offsets[t_own + 1] = base + s_scan[4 + wv];
        if (n_isects) n_isects[0] = base + s_scan[4 + wv];
      }
    }
  }
  // span of tile q of this workgroup in the packed arrays: from the scan above, or from the offsets given
  auto span = [&](int q, long long& s, long long& e) {
    int t = tile_begin + blockIdx.x * 4 + q;
    if (counts) {
      s = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
      for (int w = 0; w < q; ++w) s += s_scan[4 + w];
      e = s + s_scan[4 + q];
    } else {
      s = tile_offsets[t];
      e = tile_offsets[t + 1];
    }
  };
  int local = blockIdx.x * 4 + wv;
  if (local < n_strip_tiles) {
    int t = tile_begin + local;
    long long s, e;
    span(wv, s, e);
    if (e > capacity) e = capacity;
    int n = (int)max(e - s, (long long)0);
    if (bins && n > bin_cap) n = bin_cap;
    const uint64_t* src = bins ? bins + (size_t)t * (size_t)bin_cap : keys + s;
    // sorted keys go to the packed array; in place when that is also the source (every lane has read its keys
    // into registers before any lane writes)
    uint64_t* kout = write_sorted_keys ? keys : nullptr;
    int32_t* const ids_lds = reinterpret_cast<int32_t*>(skeys) + wv * 2048;  // (this wave's quarter of the LDS block)
    if (n > 0 && n <= (64 << MAXLK)) {
      if (n <= 256) wave_sort_tile<2>(src, n, s, t, lane, kout, flatten_ids, isect_ids, cam_enc, storage_of, ids_lds);
      else if (n <= 512) wave_sort_tile<3>(src, n, s, t, lane, kout, flatten_ids, isect_ids, cam_enc, storage_of, ids_lds);
      else if (MAXLK == 4 || n <= 1024) wave_sort_tile<4>(src, n, s, t, lane, kout, flatten_ids, isect_ids, cam_enc, storage_of, ids_lds);
      else wave_sort_tile<(MAXLK > 4 ? 5 : 4)>(src, n, s, t, lane, kout, flatten_ids, isect_ids, cam_enc, storage_of, ids_lds);
    }
  }
  // rare: lists too long for one wave, sorted in place by the whole workgroup, one after the other
  for (int q = 0; q < 4; ++q) {
    int lq = blockIdx.x * 4 + q;
    if (lq >= n_strip_tiles) break;
    int t = tile_begin + lq;
    long long s, e;
    span(q, s, e);
    if (e > capacity) e = capacity;
    int n = (int)max(e - s, (long long)0);
    if (bins && n > bin_cap) n = bin_cap;
    if (n <= (64 << MAXLK)) continue;
    if (long_min > 0 && bins && n > long_min) continue;  // sorted by several workgroups: gsl_long_sort
    uint64_t* src = bins ? bins + (size_t)t * (size_t)bin_cap : keys + s;
    __syncthreads();
    bitonic_sort_long(src, n, skeys, tid);
    for (int i = tid; i < n; i += 256) {
      uint64_t k = src[i];
      flatten_ids[s + i] = list_id(storage_of, k);
      if (isect_ids) isect_ids[s + i] = cam_enc | ((int64_t)t << 32) | (int64_t)(k >> 32);
      if (write_sorted_keys && bins) keys[s + i] = k;
    }
  }
}

// isect_tiles(sort=False): emit in Gaussian order at cum_tiles positions.
__global__ __launch_bounds__(256) void k_isect_emit(const float* __restrict__ means2d,
                                                    const int32_t* __restrict__ radii,
                                                    const float* __restrict__ depths,
                                                    const int64_t* __restrict__ cum_tiles, int N, int tile_size,
                                                    int tile_w, int tile_h, int64_t cam_enc, int id_offset,
                                                    int64_t* __restrict__ isect_ids,
                                                    int32_t* __restrict__ flatten_ids) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int r = radii[i];
  if (r <= 0) return;
  int xmin, ymin, xmax, ymax;
  tile_rect(means2d[2 * (size_t)i], means2d[2 * (size_t)i + 1], r, tile_size, tile_w, tile_h, xmin, ymin, xmax, ymax);
  int64_t cur = (i == 0) ? 0 : cum_tiles[i - 1];
  int64_t dbits = (int64_t)__float_as_uint(depths[i]);
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) {
      int64_t tile = (int64_t)y * tile_w + x;
      isect_ids[cur] = cam_enc | (tile << 32) | dbits;
      flatten_ids[cur] = id_offset + i;
      ++cur;
    }
}

// isect_offset_encode: offsets[q] = first index whose (cam,tile) >= q.
__global__ __launch_bounds__(256) void k_isect_offsets(const int64_t* __restrict__ isect_ids, long long n,
                                                       int n_cameras, int n_tiles, int tile_n_bits,
                                                       int32_t* __restrict__ offsets) {
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)n_cameras * n_tiles;
  if (n == 0) {
    if (idx < total) offsets[idx] = 0;
    return;
  }
  if (idx >= n) return;
  int64_t mask = ((int64_t)1 << tile_n_bits) - 1;
  int64_t id = isect_ids[idx];
  long long cur = (id >> (32 + tile_n_bits)) * n_tiles + ((id >> 32) & mask);
  if (idx == 0) {
    for (long long q = 0; q <= cur; ++q) offsets[q] = 0;
  }
  if (idx == n - 1) {
    for (long long q = cur + 1; q < total; ++q) offsets[q] = (int32_t)n;
  }
  if (idx > 0) {
    int64_t pid = isect_ids[idx - 1];
    long long prev = (pid >> (32 + tile_n_bits)) * n_tiles + ((pid >> 32) & mask);
    for (long long q = prev + 1; q <= cur; ++q) offsets[q] = (int32_t)idx;
  }
}


// ------------------------------------------------------------------------------------------------
// Sort of a LONG tile list by several workgroups (binned mode).  One workgroup sorted such a list block-wise in 1.6 ms
// (23 k keys: the pile of invalid TUM points) -- after the compositing of that list had been split
// over workgroups it was three quarters of the iteration.  Now: every GSL_SORT_SEG-key segment of the list is sorted in
// registers by one wave (k_long_sort_seg, into the packed key array), then `passes` merge passes double the run length,
// one wave per GSL_SORT_SEG outputs (merge path: the two diagonals of the chunk are located by binary search in the two runs,
// the <= GSL_SORT_SEG inputs staged in LDS, every lane merges its share of the outputs), ping-ponging between the packed key array and the
// tile's bin; the last pass writes flatten_ids.  Same result as any stable sort of the (depth bits, id) keys.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_long_sort_seg(const int32_t* __restrict__ tile_offsets, long long capacity,
                                                      const uint64_t* __restrict__ bins, int bin_cap,
                                                      uint64_t* __restrict__ keys, LongWs w) {
  int g = blockIdx.x;
  if (g >= w.n_seg[0]) return;
  int tile = w.seg_tile[g], sgm = w.seg_idx[g];
  if (tile < 0) return;  // a tile whose segments did not fit the workspace (flagged by k_long_map)
  // (the map lists compositing segments; the first of every GSL_SORT_SEG / GSL_SEG works as a sort segment)
  if (sgm & (GSL_SORT_SEG / GSL_SEG - 1)) return;
  sgm >>= GSL_SORT_SEG_LOG2 - GSL_SEG_LOG2;
  long long s = tile_offsets[tile], e = tile_offsets[tile + 1];
  if (e > capacity) e = capacity;
  int n = (int)min((long long)bin_cap, e - s);
  int lane = threadIdx.x;
  const uint64_t* src = bins + (size_t)tile * (size_t)bin_cap + (size_t)sgm * GSL_SORT_SEG;
  int m = min(GSL_SORT_SEG, n - sgm * GSL_SORT_SEG);
  constexpr int KPL = GSL_SORT_SEG / 64;  // keys per lane
  uint64_t k[KPL];
#pragma unroll
  for (int r = 0; r < KPL; ++r) k[r] = (lane * KPL + r < m) ? src[lane * KPL + r] : GSL_SORT_PAD;
  wave_sort_regs<GSL_SORT_SEG_LOG2 - 6>(k, lane);
  uint64_t* dst = keys + s + (size_t)sgm * GSL_SORT_SEG;
#pragma unroll
  for (int r = 0; r < KPL; ++r)
    if (lane * KPL + r < m) dst[lane * KPL + r] = k[r];
}

// merge_diag by the 64 lanes of a wave together (every lane calls it and gets the result): 64 probes per round instead
// of one, so a diagonal of a 16 k-key run costs 3 dependent global loads instead of 14 (the searches were most of a
// merge pass: 7 us each, eight passes per frame)
__device__ __forceinline__ int merge_diag_wave(const uint64_t* __restrict__ A, int lenA, const uint64_t* __restrict__ B,
                                               int lenB, int d, int lane) {
  int lo = max(0, d - lenB), hi = min(d, lenA);
  while (lo < hi) {  // (wave-uniform)
    int step = (hi - lo + 63) >> 6;
    int mid = lo + lane * step;
    bool pred = mid < hi && A[mid] <= B[d - 1 - mid];  // true exactly for the probes below the answer: a prefix of lanes
    int k = __popcll(__ballot(pred));
    int nhi = (lo + k * step < hi) ? lo + k * step : hi;  // probe k (if there is one) answered "not below"
    lo = k > 0 ? lo + (k - 1) * step + 1 : lo;
    hi = nhi;
  }
  return lo;
}

// pass p: runs of (GSL_SORT_SEG << p) keys -> runs of twice that.  src / dst: the packed key array and the bins, alternating.
__global__ __launch_bounds__(64) void k_long_merge(const int32_t* __restrict__ tile_offsets, long long capacity,
                                                   uint64_t* __restrict__ bins, int bin_cap, uint64_t* __restrict__ keys,
                                                   int pass, int last, int32_t* __restrict__ flatten_ids, LongWs w,
                                                   const int32_t* __restrict__ storage_of) {
  __shared__ uint64_t sk[GSL_SORT_SEG];
  __shared__ int s_split[4];
  int g = blockIdx.x;
  if (g >= w.n_seg[0]) return;
  int tile = w.seg_tile[g], sgm = w.seg_idx[g];
  if (tile < 0) return;  // a tile whose segments did not fit the workspace (flagged by k_long_map)
  // (the map lists compositing segments; the first of every GSL_SORT_SEG / GSL_SEG works as a sort segment)
  if (sgm & (GSL_SORT_SEG / GSL_SEG - 1)) return;
  sgm >>= GSL_SORT_SEG_LOG2 - GSL_SEG_LOG2;
  long long s = tile_offsets[tile], e = tile_offsets[tile + 1];
  if (e > capacity) e = capacity;
  int n = (int)min((long long)bin_cap, e - s);
  int lane = threadIdx.x;
  uint64_t* kbase = keys + s;
  uint64_t* bbase = bins + (size_t)tile * (size_t)bin_cap;
  const uint64_t* src = (pass & 1) ? bbase : kbase;
  uint64_t* dst = (pass & 1) ? kbase : bbase;
  int L = GSL_SORT_SEG << pass;
  int pair_start = (sgm * GSL_SORT_SEG) / (2 * L) * (2 * L);
  int o = sgm * GSL_SORT_SEG - pair_start;
  int lenA = max(0, min(L, n - pair_start)), lenB = max(0, min(L, n - pair_start - L));
  int out_len = min(GSL_SORT_SEG, lenA + lenB - o);
  const uint64_t* A = src + pair_start;
  const uint64_t* B = src + pair_start + L;
  {
    int ia_lo = merge_diag_wave(A, lenA, B, lenB, o, lane);
    int ia_hi = merge_diag_wave(A, lenA, B, lenB, o + out_len, lane);
    if (lane == 0) {
      s_split[0] = ia_lo;
      s_split[1] = o - ia_lo;
      s_split[2] = ia_hi;
      s_split[3] = o + out_len - ia_hi;
    }
  }
  __syncthreads();
  int ia0 = s_split[0], ib0 = s_split[1], na = s_split[2] - ia0, nb = s_split[3] - ib0;
  for (int q = lane; q < na; q += 64) sk[q] = A[ia0 + q];
  for (int q = lane; q < nb; q += 64) sk[na + q] = B[ib0 + q];
  __syncthreads();
  // every lane merges its GSL_SORT_SEG / 64 outputs from the staged pieces
  constexpr int KPL = GSL_SORT_SEG / 64;
  int d0 = min(lane * KPL, out_len), d1 = min(lane * KPL + KPL, out_len);
  int ia = merge_diag(sk, na, sk + na, nb, d0), ib = d0 - ia;
  for (int d = d0; d < d1; ++d) {
    uint64_t v;
    if (ib >= nb || (ia < na && sk[ia] <= sk[na + ib])) v = sk[ia++];
    else v = sk[na + ib++];
    dst[pair_start + o + d] = v;
    if (last) flatten_ids[s + pair_start + o + d] = list_id(storage_of, v);
  }
}


// Same contract as k_tile_sort (offsets from the counters in binned mode, overflow flags, long lists left to
// gsl_long_sort), one tile per workgroup.
__global__ __launch_bounds__(256) void k_tile_sort_wg(int32_t* __restrict__ tile_offsets, int tile_begin,
                                                      int n_strip_tiles, long long capacity,
                                                      uint64_t* __restrict__ keys, int32_t* __restrict__ flatten_ids,
                                                      int64_t* __restrict__ isect_ids, int64_t cam_enc,
                                                      int write_sorted_keys, uint64_t* __restrict__ bins, int bin_cap,
                                                      const int32_t* __restrict__ counts, int32_t* __restrict__ n_isects,
                                                      int32_t* __restrict__ flags, int long_min,
                                                      const int32_t* __restrict__ storage_of) {
  __shared__ uint64_t skeys[GSL_SORT_LDS_CAP];
  __shared__ int s_scan[5];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = tile_begin + blockIdx.x;
  long long s, e;
  if (counts) {
    int acc = prefix_count_share(counts, t, bin_cap, tid);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) s_scan[wv] = acc;
    if (tid == 0) {
      int c_own = counts[t];
      s_scan[4] = min(c_own, bin_cap);
      if (c_own > bin_cap && flags) { flags[1] = 1; atomicMax(&flags[2], c_own); }
    }
    __syncthreads();
    s = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    e = s + s_scan[4];
    if (tid == 0) {
      tile_offsets[t] = (int32_t)s;
      if ((int)blockIdx.x == n_strip_tiles - 1) {
        tile_offsets[t + 1] = (int32_t)e;
        if (n_isects) n_isects[0] = (int32_t)e;
      }
    }
  } else {
    s = tile_offsets[t];
    e = tile_offsets[t + 1];
  }
  if (e > capacity) e = capacity;
  int n = (int)max(e - s, (long long)0);
  if (bins && n > bin_cap) n = bin_cap;
  if (n == 0) return;
  uint64_t* src = bins ? bins + (size_t)t * (size_t)bin_cap : keys + s;
  uint64_t* kout = write_sorted_keys ? keys : nullptr;
  if (n <= 1024) {
    wg_sort_tile<2>(src, n, s, t, tid, skeys, kout, flatten_ids, isect_ids, cam_enc, storage_of);
  } else if (n <= 2048) {
    wg_sort_tile<3>(src, n, s, t, tid, skeys, kout, flatten_ids, isect_ids, cam_enc, storage_of);
  } else {
    if (long_min > 0 && bins && n > long_min) return;  // sorted by several workgroups: gsl_long_sort
    bitonic_sort_long(src, n, skeys, tid);
    for (int i = tid; i < n; i += 256) {
      uint64_t k = src[i];
      flatten_ids[s + i] = list_id(storage_of, k);
      if (isect_ids) isect_ids[s + i] = cam_enc | ((int64_t)t << 32) | (int64_t)(k >> 32);
      if (write_sorted_keys && bins) keys[s + i] = k;
    }
  }
}

}  // namespace gsl

extern "C" size_t gsl_isect_ws_bytes(int n_tiles) {
  // [tile_counts n_tiles][cursors n_tiles]
  return (size_t)2 * (size_t)(n_tiles > 0 ? n_tiles : 1) * sizeof(int32_t);
}

extern "C" int gsl_isect_count(const float* means2d, const int32_t* radii, int N, int tile_size, int tile_w,
                               int tile_h, int ty0, int ty1, int32_t* tiles_per_gauss, int32_t* tile_offsets,
                               int32_t* n_isects, void* ws, size_t ws_bytes, void* stream) {
  if (N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1)
    return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !n_isects || (N > 0 && (!means2d || !radii))) return GSL_ERR_BAD_ARG;
  int n_tiles = tile_w * tile_h;
  if (!ws || ws_bytes < gsl_isect_ws_bytes(n_tiles)) return GSL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int32_t* counts = (int32_t*)ws;
  int32_t* cursors = counts + n_tiles;
  if (gsl::zero_u32(counts, (size_t)n_tiles, st) != GSL_OK) return GSL_ERR_HIP;
  if (N > 0) {
    int nst = (ty1 - ty0) * tile_w;
    if (nst > 0 && nst <= GSL_HIST_LDS_TILES) {
      hipLaunchKernelGGL(gsl::k_isect_count_lds, dim3((N + GSL_BIN_THREADS - 1) / GSL_BIN_THREADS),
                         dim3(GSL_BIN_THREADS), (size_t)nst * sizeof(int), st, means2d, radii, N, tile_size, tile_w,
                         tile_h, ty0, ty1, tiles_per_gauss, counts);
    } else {
      hipLaunchKernelGGL(gsl::k_isect_count, dim3((N + 255) / 256), dim3(256), 0, st, means2d, radii, N, tile_size,
                         tile_w, tile_h, ty0, ty1, tiles_per_gauss, counts);
    }
    GSL_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(gsl::k_tile_scan, dim3(1), dim3(1024), 0, st, counts, n_tiles, tile_offsets, n_isects, cursors);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" int gsl_isect_fill(const float* means2d, const int32_t* radii, const float* depths, int N, int tile_size,
                              int tile_w, int tile_h, int ty0, int ty1, int cam_id, int tile_n_bits,
                              const int32_t* tile_offsets, int64_t capacity, uint64_t* sort_keys,
                              int32_t* flatten_ids, int64_t* isect_ids, void* ws, size_t ws_bytes, void* stream) {
  if (N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 || capacity < 0)
    return GSL_ERR_BAD_ARG;
  if (!tile_offsets) return GSL_ERR_BAD_ARG;
  if (N == 0 || capacity == 0 || ty0 == ty1) return GSL_OK;
  if (!means2d || !radii || !depths || !sort_keys || !flatten_ids) return GSL_ERR_BAD_ARG;
  int n_tiles = tile_w * tile_h;
  if (!ws || ws_bytes < gsl_isect_ws_bytes(n_tiles)) return GSL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int32_t* cursors = (int32_t*)ws + n_tiles;
  int nst = (ty1 - ty0) * tile_w;
  if (nst <= GSL_HIST_LDS_TILES) {
    hipLaunchKernelGGL(gsl::k_isect_scatter_lds, dim3((N + GSL_BIN_THREADS - 1) / GSL_BIN_THREADS),
                       dim3(GSL_BIN_THREADS), (size_t)2 * nst * sizeof(int), st, means2d, radii, depths, N, tile_size,
                       tile_w, tile_h, ty0, ty1, tile_offsets, cursors, (long long)capacity, sort_keys);
  } else {
    hipLaunchKernelGGL(gsl::k_isect_scatter, dim3((N + 255) / 256), dim3(256), 0, st, means2d, radii, depths, N,
                       tile_size, tile_w, tile_h, ty0, ty1, tile_offsets, cursors, (long long)capacity, sort_keys);
  }
  GSL_CHECK_LAUNCH();
  int64_t cam_enc = (int64_t)cam_id << (32 + tile_n_bits);
  return gsl_tile_sort(tile_offsets, ty0 * tile_w, (ty1 - ty0) * tile_w, capacity, sort_keys, flatten_ids, isect_ids,
                       cam_enc, stream);
}

extern "C" int gsl_tile_sort_keys(int32_t* tile_offsets, int tile_begin, int n_strip_tiles, int64_t capacity,
                                  uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, int64_t cam_enc,
                                  int write_sorted_keys, uint64_t* bins, int bin_cap, const int32_t* counts,
                                  int32_t* n_isects, int32_t* flags, int long_min, int occupied_tiles,
                                  const int32_t* storage_of, void* stream);

extern "C" int gsl_tile_sort(const int32_t* tile_offsets, int tile_begin, int n_strip_tiles, int64_t capacity,
                             uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, int64_t cam_enc,
                             void* stream) {
  if (!tile_offsets || tile_begin < 0 || n_strip_tiles < 0 || capacity < 0) return GSL_ERR_BAD_ARG;
  if (n_strip_tiles == 0 || capacity == 0) return GSL_OK;
  if (!sort_keys || !flatten_ids) return GSL_ERR_BAD_ARG;
  return gsl_tile_sort_keys(const_cast<int32_t*>(tile_offsets), tile_begin, n_strip_tiles, capacity, sort_keys, flatten_ids,
                            isect_ids, cam_enc, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, stream);
}

// gsl_tile_sort that can also leave the sorted (depth bits, id) keys in sort_keys and read the unsorted keys from
// fixed-capacity per-tile bins instead of sort_keys (internal: gsl_fused_bin)
extern "C" int gsl_tile_sort_keys(int32_t* tile_offsets, int tile_begin, int n_strip_tiles, int64_t capacity,
                                  uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, int64_t cam_enc,
                                  int write_sorted_keys, uint64_t* bins, int bin_cap, const int32_t* counts,
                                  int32_t* n_isects, int32_t* flags, int long_min, int occupied_tiles,
                                  const int32_t* storage_of, void* stream) {
  if (!tile_offsets || tile_begin < 0 || n_strip_tiles < 0 || capacity < 0) return GSL_ERR_BAD_ARG;
  if (counts && (!bins || tile_begin != 0)) return GSL_ERR_BAD_ARG;  // the scan runs over all tiles, bins only
  if (n_strip_tiles == 0 || (capacity == 0 && !counts)) return GSL_OK;
  if (capacity > 0 && (!sort_keys || !flatten_ids)) return GSL_ERR_BAD_ARG;
  // lists of several hundred keys: one tile per workgroup (waves sort quarters, merged in LDS); short lists: one per wave
  // (occupied_tiles: the tiles that can hold entries -- a strip's, when the launch runs over all tiles of the image)
  const int occ = occupied_tiles > 0 ? occupied_tiles : n_strip_tiles;
  const long long mean_list = capacity / (long long)occ;
  static const char* const force = getenv("GSL_DEV_TILE_SORT");  // dev / test switch: "wave" / "wg" (read once)
  // (a latency matter: with more tiles than the chip has room for wave sorts at once, one tile per wave keeps more
  // lists in flight and is as fast or faster -- X: 159 against 169 us; with a strip's few hundred tiles the workgroup
  // kernel's shorter critical path decides)
  const bool wg = force ? force[1] == 'g' : (mean_list > 320 && occ <= 2048);
  if (wg)
    hipLaunchKernelGGL(gsl::k_tile_sort_wg, dim3(n_strip_tiles), dim3(256), 0, (hipStream_t)stream, tile_offsets,
                       tile_begin, n_strip_tiles, (long long)capacity, sort_keys, flatten_ids, isect_ids, cam_enc,
                       write_sorted_keys, bins, bin_cap, counts, n_isects, flags, long_min, storage_of);
  else if (mean_list <= 1150 && !(force && force[0] == 'W'))
    // (capacity carries ~1.3 x head-room: a mean list of <= ~880 keys, whose longest lists stay below 1024 in a frame of
    // evenly spread splats; a tile that does exceed 1024 takes the workgroup's LDS sort -- slower, never wrong)
    hipLaunchKernelGGL((gsl::k_tile_sort<4>), dim3((n_strip_tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, tile_offsets,
                       tile_begin, n_strip_tiles, (long long)capacity, sort_keys, flatten_ids, isect_ids, cam_enc,
                       write_sorted_keys, bins, bin_cap, counts, n_isects, flags, long_min, storage_of);
  else
    hipLaunchKernelGGL((gsl::k_tile_sort<5>), dim3((n_strip_tiles + 3) / 4), dim3(256), 0, (hipStream_t)stream, tile_offsets,
                       tile_begin, n_strip_tiles, (long long)capacity, sort_keys, flatten_ids, isect_ids, cam_enc,
                       write_sorted_keys, bins, bin_cap, counts, n_isects, flags, long_min, storage_of);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" int gsl_isect_emit(const float* means2d, const int32_t* radii, const float* depths,
                              const int64_t* cum_tiles, int N, int tile_size, int tile_w, int tile_h, int cam_id,
                              int tile_n_bits, int id_offset, int64_t* isect_ids, int32_t* flatten_ids,
                              void* stream) {
  if (N < 0 || tile_size <= 0 || tile_w <= 0 || tile_h <= 0) return GSL_ERR_BAD_ARG;
  if (N == 0) return GSL_OK;
  if (!means2d || !radii || !depths || !cum_tiles || !isect_ids || !flatten_ids) return GSL_ERR_BAD_ARG;
  int64_t cam_enc = (int64_t)cam_id << (32 + tile_n_bits);
  hipLaunchKernelGGL(gsl::k_isect_emit, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, means2d, radii,
                     depths, cum_tiles, N, tile_size, tile_w, tile_h, cam_enc, id_offset, isect_ids, flatten_ids);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" int gsl_isect_offsets(const int64_t* isect_ids, int64_t n_isects, int n_cameras, int n_tiles,
                                 int tile_n_bits, int32_t* offsets, void* stream) {
  if (n_isects < 0 || n_cameras <= 0 || n_tiles <= 0 || !offsets) return GSL_ERR_BAD_ARG;
  if (n_isects > 0 && !isect_ids) return GSL_ERR_BAD_ARG;
  long long work = n_isects > 0 ? (long long)n_isects : (long long)n_cameras * n_tiles;
  hipLaunchKernelGGL(gsl::k_isect_offsets, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     isect_ids, (long long)n_isects, n_cameras, n_tiles, tile_n_bits, offsets);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// Multi-workgroup sort of the long tile lists (binned mode; see k_long_sort_seg).  Call after gsl_fused_bin(long_min).
extern "C" int gsl_long_sort(const int32_t* tile_offsets, int tile_w, int tile_h, int ty0, int ty1, int64_t capacity,
                             uint64_t* bins, int bin_cap, uint64_t* sort_keys, int32_t* flatten_ids, int long_min,
                             void* long_ws, size_t long_ws_bytes, int max_seg, int passes, const int32_t* storage_of,
                             void* stream) {
  if (tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 || capacity < 0 || long_min <= 0 ||
      max_seg <= 0 || passes < 0 || passes > 12 || bin_cap <= 0)
    return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !bins || !sort_keys || !flatten_ids || !long_ws) return GSL_ERR_BAD_ARG;
  if (long_ws_bytes < gsl_long_ws_bytes(max_seg)) return GSL_ERR_WORKSPACE;
  if (ty0 == ty1 || capacity == 0) return GSL_OK;
  hipStream_t st = (hipStream_t)stream;
  gsl::LongWs w = gsl::long_ws_views(long_ws, max_seg);
  hipLaunchKernelGGL(gsl::k_long_map, dim3(1), dim3(1024), 0, st, tile_offsets, ty0 * tile_w, (ty1 - ty0) * tile_w,
                     (long long)capacity, long_min, max_seg, GSL_SORT_SEG << passes, w);
  hipLaunchKernelGGL(gsl::k_long_sort_seg, dim3(max_seg), dim3(64), 0, st, tile_offsets, (long long)capacity, bins,
                     bin_cap, sort_keys, w);
  for (int p = 0; p < passes; ++p)
    hipLaunchKernelGGL(gsl::k_long_merge, dim3(max_seg), dim3(64), 0, st, tile_offsets, (long long)capacity, bins, bin_cap,
                       sort_keys, p, p == passes - 1 ? 1 : 0, flatten_ids, w, storage_of);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
