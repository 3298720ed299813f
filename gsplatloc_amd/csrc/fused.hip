// Fused single-camera render pipeline for GsplatLoc's pose-tracking loop (the hot path):
//   forward : project + SH colour + pack + tile histogram  -> scan -> scatter -> per-tile LDS sort
//             -> composite (expected-depth normalisation fused)
//   backward: composite vjp (packed 64-byte gradient rows)  -> projection/SH vjp + pose reduction
// Same arithmetic as the stage operators (project.hip / binning.hip / raster.hip / sh.hip), which
// restate gsplat.rasterization (IDX:14954) as called from /root/reference/src/my_gsplat/model.py:195-213;
// the difference is data layout and launch count.
//
// HBM layout (SoA of 16-byte records, one per Gaussian, written once by the projection kernel and
// gathered by the compositing kernels with one or two dwordx4 loads):
//   Q0 = (x, y, depth, opacity_eff)   Q1 = (conic_a, conic_b, conic_c, r_cull)   Q2 = (r, g, b, 0)
// r_cull is a conservative radius of the alpha >= 1/255 region, used by the per-quadrant ballot test.
// gsplat's meta tensors (means2d, depths, conics, opacities) are strided views of Q0/Q1 on the host.
#include <stdlib.h>
#include <string.h>

#include "project_dev.h"
#include "sh_dev.h"

namespace gsl {

#ifndef GSL_F_BIN_THREADS
#define GSL_F_BIN_THREADS 512
#endif
#define GSL_F_MAX_STRIP_TILES 8192

// Radius of the smallest disc around the centre that holds the whole alpha >= 1/255 ellipse {sigma <= tau}:
// sqrt(2 tau / lambda_min(conic)).  (Round 2 stored the half-extent of the ellipse's axis-aligned bounding box, which is
// smaller for a rotated anisotropic splat; every user treats the value as conservative -- the forward's pixel boxes,
// the quadrant tests, the 4x4 slabs of the tiny backward -- and the 16-lane-group backward tests the DISC against its
// 4x4 pixel blocks.  Identical for isotropic splats, GsplatLoc's only kind.)
__device__ __forceinline__ float cull_radius(float ca, float cb, float cc, float op) {
  float tau = __logf(255.f * op) * 1.01f + 0.01f;
  float det = ca * cc - cb * cb;
  if (!(tau > 0.f)) return -1.f;  // opacity < 1/255: can never reach the alpha threshold
  if (!(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) return 1e30f;  // degenerate conic: never cull
  float hd = 0.5f * (ca - cc);
  float root = sqrtf(hd * hd + cb * cb);
  float lmin = det / (0.5f * (ca + cc) + root);  // = mean - root, without the cancellation
  if (!(lmin > 0.f)) return 1e30f;
  return sqrtf(2.f * tau / lmin) * 1.0001f + 1e-3f;
}

// ------------------------------------------------------------------------------------------------
// Forward 1: projection + colour + pack + tile histogram.
// ------------------------------------------------------------------------------------------------
// BINNED: the kernel also reserves each intersection's slot in its tile's fixed-capacity bin (one returning atomic per
// distinct tile per workgroup, ranks inside the workgroup from LDS) and writes the (depth bits | id) key there: the
// separate scatter pass and its second read of the records disappear.  tile_counts ends up holding the tile sizes.
template <bool RGB, bool BINNED>
__global__ __launch_bounds__(GSL_F_BIN_THREADS) void k_fproject(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ colors, int sh_degree, int K_sh,
    const float* __restrict__ V, const float* __restrict__ Kmat, int N, int W, int H, float eps2d, float near_plane,
    float far_plane, float radius_clip, int antialiased, int tile_w, int tile_h, int ty0, int ty1,
    int32_t* __restrict__ radii, float4* __restrict__ Q0, float4* __restrict__ Q1, float4* __restrict__ Q2,
    float* __restrict__ comps, int32_t* __restrict__ tiles_per_gauss, int32_t* __restrict__ tile_counts,
    uint4* __restrict__ Qh, uint64_t* __restrict__ bins, int bin_cap, int32_t* __restrict__ bin_state,
    int32_t* __restrict__ flags, const int32_t* __restrict__ order_ids) {
  extern __shared__ int s_hist[];
  int nst = (ty1 - ty0) * tile_w, tbase = ty0 * tile_w;
  // Counter contract of the binned mode: the tile counters must be zero on entry -- the compositing forward of the
  // previous iteration clears them and marks the state word clean.  A projection that finds the state dirty (a forward
  // was skipped or failed between two projections) raises flags[3] instead of binning on top of stale sizes silently.
  if (BINNED && bin_state && nst > 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    if (atomicExch(bin_state, 1) != 0 && flags) flags[3] = 1;
  }
  // s_hist[nst], s_hist[nst + 1]: lowest / highest strip-tile index a Gaussian of this workgroup touches.  The passes over
  // the counters below cover that range only: with the Gaussians stored in tile order (context.py:_choose_placement) a
  // workgroup's 512 touch a band of two or three tile rows, not the frame's 3 225 tiles (the fixed 16-step loops were a
  // quarter of the kernel's VALU instructions).
  int* const s_rng = s_hist + nst;
  for (int k = threadIdx.x; k < nst; k += GSL_F_BIN_THREADS) s_hist[k] = 0;
  if (threadIdx.x == 0) { s_rng[0] = nst; s_rng[1] = -1; }
  __syncthreads();
  int i = blockIdx.x * GSL_F_BIN_THREADS + threadIdx.x;
  Cam cam = load_cam(V, Kmat);
  int xmin = 0, ymin = 0, xmax = 0, ymax = 0;
  uint64_t key = 0;
  if (i < N) {
    ProjMid p;
    float q[4], s[3];
    load_gaussian(means, quats, scales, i, cam, p, q, s);
    // EVERY global load of the thread is issued here, before the first value is used.  Left alone, the compiler sinks each
    // load into the branch that needs it -- mean, then (depth test) rotation and scale, then (visibility test) opacity,
    // then one coefficient triple per turn of the colour loop -- and a wave pays five to seven DEPENDENT memory round
    // trips for 92 bytes (round 4: a wave of this kernel lived 36 k cycles for 3.6 k cycles of arithmetic).  The empty
    // asm consumes all of them at once: one wait.
    float op_in = opacities[i];
    float shc[12];  // the colour itself (sh_degree < 0) or the first four coefficient triples (everything up to degree 1)
#pragma unroll
    for (int k = 0; k < 12; ++k) shc[k] = 0.f;
    if (RGB) {
      if (sh_degree < 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) shc[k] = colors[3 * (size_t)i + k];
      } else {
        const float* cf = colors + (size_t)i * K_sh * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) shc[k] = cf[k];
        if (sh_degree >= 1) {  // (one uniform branch for the three triples of degree 1: their loads leave together)
#pragma unroll
          for (int k = 3; k < 12; ++k) shc[k] = cf[k];
        }
      }
    }
    asm volatile("" : "+v"(p.mean[0]), "+v"(p.mean[1]), "+v"(p.mean[2]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]),
                 "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(op_in), "+v"(shc[0]), "+v"(shc[1]), "+v"(shc[2]), "+v"(shc[3]),
                 "+v"(shc[4]), "+v"(shc[5]), "+v"(shc[6]), "+v"(shc[7]), "+v"(shc[8]), "+v"(shc[9]), "+v"(shc[10]),
                 "+v"(shc[11]));
    int radius = 0;
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = make_float4(0.f, 0.f, 0.f, -1.f);
    float comp = 0.f;
    if (p.mc[2] >= near_plane && p.mc[2] <= far_plane) {
      p.covar = quat_scale_to_covar(q, s);
      p.covar_c = mul_bt(mul(cam.R, p.covar), cam.R);
      persp_mid(cam, W, H, p);
      float a, b, c;
      cov2d_from(p.J, p.covar_c, a, b, c);
      float det_orig = a * c - b * b;
      a += eps2d;
      c += eps2d;
      float det = a * c - b * b;
      if (det > 0.f) {
        float bb = 0.5f * (a + c);
        float v1 = bb + sqrtf(fmaxf(0.01f, bb * bb - det));
        float rad = ceilf(3.f * sqrtf(v1));
        float mx = cam.fx * p.mc[0] * p.rz + cam.cx;
        float my = cam.fy * p.mc[1] * p.rz + cam.cy;
        bool ok = rad > radius_clip;
        ok = ok && !(mx + rad <= 0.f || mx - rad >= (float)W || my + rad <= 0.f || my - rad >= (float)H);
        if (ok) {
          float inv = 1.f / det;
          radius = (int)rad;
          comp = sqrtf(fmaxf(0.f, det_orig / det));
          float op = op_in;
          if (antialiased) op *= comp;
          float ca = c * inv, cb = -b * inv, cc = a * inv;
          o0 = make_float4(mx, my, p.mc[2], op);
          o1 = make_float4(ca, cb, cc, cull_radius(ca, cb, cc, op));
        }
      }
    }
    radii[i] = radius;
    GSL_Q(Q0, i) = o0;
    GSL_Q(Q1, i) = o1;
    if (comps) comps[i] = comp;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (RGB) {
      if (sh_degree < 0) {
        c0 = shc[0]; c1 = shc[1]; c2 = shc[2];
      } else {
        if (radius > 0) {  // masks = radii > 0
          M3 Ri;
          float cp[3];
          cam_inverse(cam, Ri, cp);
          float x = p.mean[0] - cp[0], y = p.mean[1] - cp[1], z = p.mean[2] - cp[2];
          float inorm = rsqrtf(x * x + y * y + z * z);
          float Y[16];
          sh_basis(sh_degree, x * inorm, y * inorm, z * inorm, Y);
          int nK = (sh_degree + 1) * (sh_degree + 1);
          const float* cf = colors + (size_t)i * K_sh * 3;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < nK) { c0 += Y[k] * shc[3 * k]; c1 += Y[k] * shc[3 * k + 1]; c2 += Y[k] * shc[3 * k + 2]; }
          for (int k = 4; k < nK; ++k) {
            c0 += Y[k] * cf[3 * k]; c1 += Y[k] * cf[3 * k + 1]; c2 += Y[k] * cf[3 * k + 2];
          }
        }
        c0 = fmaxf(c0 + 0.5f, 0.f); c1 = fmaxf(c1 + 0.5f, 0.f); c2 = fmaxf(c2 + 0.5f, 0.f);
      }
      if (!Qh) GSL_Q(Q2, i) = make_float4(c0, c1, c2, 0.f);  // (fp16 staging: the compositing kernels read the colour from Qh)
    }
    if (Qh) store_half_record(Qh, (size_t)i, o0, o1, make_float4(c0, c1, c2, 0.f));
    if (radius > 0) {
      tile_rect(o0.x, o0.y, radius, 16, tile_w, tile_h, xmin, ymin, xmax, ymax);
      ymin = max(ymin, ty0);
      ymax = min(ymax, ty1);
      if (ymax < ymin) ymax = ymin;
      key = ((uint64_t)__float_as_uint(o0.z) << 32) | (uint32_t)(order_ids ? order_ids[i] : i);
    }
    if (tiles_per_gauss) tiles_per_gauss[i] = (xmax - xmin) * (ymax - ymin);
  }
  {
    const bool has = (ymin < ymax) && (xmin < xmax);
    int lo = has ? ymin * tile_w + xmin - tbase : nst, hi = has ? (ymax - 1) * tile_w + (xmax - 1) - tbase : -1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o, 64));
      hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && hi >= 0) { atomicMin(&s_rng[0], lo); atomicMax(&s_rng[1], hi); }
  }
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) atomicAdd(&s_hist[y * tile_w + x - tbase], 1);
  __syncthreads();
  const int k_lo = __builtin_amdgcn_readfirstlane(s_rng[0]), k_hi = __builtin_amdgcn_readfirstlane(s_rng[1]);
  if (!BINNED) {
    for (int k = k_lo + (int)threadIdx.x; k <= k_hi; k += GSL_F_BIN_THREADS) {
      int c = s_hist[k];
      if (c) atomicAdd(&tile_counts[tbase + k], c);
    }
    return;
  }
  // all of a thread's returning atomics are issued before the first result is consumed
  int res[GSL_F_MAX_STRIP_TILES / GSL_F_BIN_THREADS];
#pragma unroll
  for (int u = 0; u < GSL_F_MAX_STRIP_TILES / GSL_F_BIN_THREADS; ++u) {
    res[u] = 0;
    if (k_lo + u * GSL_F_BIN_THREADS <= k_hi) {  // (wave-uniform)
      int k = k_lo + (int)threadIdx.x + u * GSL_F_BIN_THREADS;
      int c = (k <= k_hi) ? s_hist[k] : 0;
      res[u] = c ? atomicAdd(&tile_counts[tbase + k], c) : 0;
    }
  }
#pragma unroll
  for (int u = 0; u < GSL_F_MAX_STRIP_TILES / GSL_F_BIN_THREADS; ++u) {
    if (k_lo + u * GSL_F_BIN_THREADS <= k_hi) {
      int k = k_lo + (int)threadIdx.x + u * GSL_F_BIN_THREADS;
      if (k <= k_hi && s_hist[k]) s_hist[k] = res[u];  // first slot of this workgroup's span; ranks count up from it
    }
  }
  __syncthreads();
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) {
      int lt = y * tile_w + x - tbase;
      int slot = atomicAdd(&s_hist[lt], 1);
      if (slot < bin_cap) bins[(size_t)(tbase + lt) * (size_t)bin_cap + slot] = key;
    }
}

// Exclusive scan of tile counts (single workgroup) -> offsets[n+1], total, zeroed cursors.  The counts are cleared
// after they are read (the next projection accumulates into them again).  bin_cap > 0: a tile keeps at most bin_cap
// entries (what its bin holds); a larger count raises flags[1] and leaves the largest count seen in flags[2].
__global__ __launch_bounds__(1024) void k_ftile_scan(int32_t* __restrict__ counts, int n,
                                                     int32_t* __restrict__ offsets, int32_t* __restrict__ n_isects,
                                                     int32_t* __restrict__ cursors, int bin_cap,
                                                     int32_t* __restrict__ flags) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + tid;
    int v = (i < n) ? counts[i] : 0;
    if (i < n) counts[i] = 0;
    if (bin_cap > 0 && v > bin_cap) {
      if (flags) { flags[1] = 1; atomicMax(&flags[2], v); }
      v = bin_cap;
    }
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

// Forward 2: scatter (depth bits | Gaussian id) keys into the tile buckets.
__global__ __launch_bounds__(GSL_F_BIN_THREADS) void k_fscatter(
    const float4* __restrict__ Q0, const int32_t* __restrict__ radii, int N, int tile_w, int tile_h, int ty0, int ty1,
    const int32_t* __restrict__ tile_offsets, int32_t* __restrict__ cursors, long long capacity,
    uint64_t* __restrict__ keys, const int32_t* __restrict__ order_ids) {
  extern __shared__ int s_mem[];
  int nst = (ty1 - ty0) * tile_w, tbase = ty0 * tile_w;
  int* s_cnt = s_mem;
  int* s_base = s_mem + nst;
  for (int k = threadIdx.x; k < nst; k += GSL_F_BIN_THREADS) s_cnt[k] = 0;
  __syncthreads();
  int i = blockIdx.x * GSL_F_BIN_THREADS + threadIdx.x;
  int xmin = 0, ymin = 0, xmax = 0, ymax = 0;
  uint64_t key = 0;
  if (i < N) {
    int r = radii[i];
    if (r > 0) {
      float4 q0 = GSL_Q(Q0, i);
      tile_rect(q0.x, q0.y, r, 16, tile_w, tile_h, xmin, ymin, xmax, ymax);
      ymin = max(ymin, ty0);
      ymax = min(ymax, ty1);
      if (ymax < ymin) ymax = ymin;
      key = ((uint64_t)__float_as_uint(q0.z) << 32) | (uint32_t)(order_ids ? order_ids[i] : i);
    }
  }
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) atomicAdd(&s_cnt[y * tile_w + x - tbase], 1);
  __syncthreads();
  // one returning global atomic per distinct tile reserves this workgroup's span of the bucket;
  // all of a thread's atomics are issued before the first result is consumed
  {
    int res[GSL_F_MAX_STRIP_TILES / GSL_F_BIN_THREADS];
#pragma unroll
    for (int u = 0; u < GSL_F_MAX_STRIP_TILES / GSL_F_BIN_THREADS; ++u) {
      int k = threadIdx.x + u * GSL_F_BIN_THREADS;
      int c = (k < nst) ? s_cnt[k] : 0;
      res[u] = c ? atomicAdd(&cursors[tbase + k], c) : 0;
    }
#pragma unroll
    for (int u = 0; u < GSL_F_MAX_STRIP_TILES / GSL_F_BIN_THREADS; ++u) {
      int k = threadIdx.x + u * GSL_F_BIN_THREADS;
      if (k < nst) {
        if (s_cnt[k]) s_base[k] = tile_offsets[tbase + k] + res[u];
        s_cnt[k] = 0;
      }
    }
  }
  __syncthreads();
  for (int y = ymin; y < ymax; ++y)
    for (int x = xmin; x < xmax; ++x) {
      int lt = y * tile_w + x - tbase;
      long long pos = (long long)s_base[lt] + atomicAdd(&s_cnt[lt], 1);
      if (pos < capacity) keys[pos] = key;
    }
}

// Lane select driven by a 64-bit scalar mask (v_cndmask_b32_e64 with an SGPR pair).  Measured on
// MI355X: the VOP2 form that reads VCC issues ~5x slower than this form (9.5 vs 1.8 ns per
// wave-instruction per SIMD), and hipcc picks the VCC form for plain ?: selects -- so the hot
// loops keep their predicates as scalar masks (__ballot) and select through this helper.
__device__ __forceinline__ float sel64(unsigned long long m, float t, float f) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}
__device__ __forceinline__ int sel64i(unsigned long long m, int t, int f) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}

// ------------------------------------------------------------------------------------------------
// Compositing backward.  256-thread workgroup per 16x16 tile, one wave64 per 8x8 quadrant; a batch of list entries
// is gathered once per workgroup into LDS, each wave ballots the quadrant test over 64 of them and walks the
// survivors back to front, broadcasting each record from LDS (wave-uniform address).  The per-splat pixel sums run
// on the matrix cores: hardware ablation of the previous build (profiles/r02_backward_ablation.txt) showed its
// atomics to be free and its 64-lane DPP reduce-scatter of 7-10 values per splat to cost a quarter of the kernel.
// Every one of those sums is linear in two per-pixel
// scalars of the (pixel, splat) pair,
//     w = vis * v_alpha (0 where alpha is clamped or the pixel did not composite the splat)    f = alpha * T,
// with per-pixel weights that do not depend on the splat once dx = X - px is expanded around the tile centre:
//     sum_p w * {1, lx, ly, lx^2, lx ly, ly^2}      (lx, ly = pixel centre - tile centre)
//     sum_p f * v_colour_k(p)
// i.e. [splats x pixels] . [pixels x 16] products.  The walk stores w and f of 8 splats as rows of a 16 x 64 tile in
// LDS (lane = pixel = column: conflict-free stores); 16 v_mfma_f32_16x16x4_f32 (exact f32 fma chains) then produce the
// 8 x (6 + CG) sums, which are added to the splat's moment row.  At the end of a batch one thread per splat turns its
// moments into the gradient row  [v_xy | v_conic | v_opacity | v_colour]  (X, Y, conic and opacity are the splat's
// own), and the rows leave as packed 64-byte global atomics exactly as before.  The matrix pipe runs beside the
// vector pipe, so the reduction costs the walk ~2 LDS stores per splat.
// ------------------------------------------------------------------------------------------------
typedef float gsl_f32x4 __attribute__((ext_vector_type(4)));

#define GSL_MB 192        // list entries staged per batch (three 64-entry chunks)
#define GSL_MPITCH 72     // floats per tile row: 16-byte reads of lane (k, i) at [i][16 m + 4 k] hit 16 distinct bank quads

template <int D, bool DET = false>
struct FStageM {
  static constexpr int A = 6 + D;   // moment / gradient row: [6 geometric][D colour]
  static constexpr int AP = A | 1;  // odd LDS pitch
  static constexpr int NW = DET ? 4 : 1;  // deterministic mode: one moment row per (wave, slot), summed in wave order
  float4 s0[GSL_MB];
  float4 s1[GSL_MB];
  float4 s2[(D >= 3) ? GSL_MB : 1];
  int32_t id[GSL_MB];
  float acc[NW * GSL_MB * AP];        // per-slot moments, converted in place to the gradient row at flush
  float tile[4][16 * GSL_MPITCH];     // per wave: rows 0-7 = w of the group's splats, rows 8-15 = f
  int32_t gslot[4][8];                // batch slot of each splat of the group
  uint16_t list[4][64];
};

template <int D, int CG, bool DET>
__device__ __forceinline__ void mraster_group_flush(FStageM<D, DET>& sb, int wv, int lane, int count,
                                                    const float (&bmat)[16]) {
  // D[i][j] = sum_k A[i][k] B[k][j]; lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; K-step kb
  // of lane-row k is pixel 16 (kb >> 2) + 4 k + (kb & 3); result row 4 (l >> 4) + r, column l & 15 in register r.
  constexpr int AP = FStageM<D, DET>::AP;
  int k = lane >> 4, ij = lane & 15;
  const float* row = &sb.tile[wv][ij * GSL_MPITCH + 4 * k];
  gsl_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // rows 0-7 (w) meet the monomial columns, rows 8-15 (f) the colour columns: one B per lane serves both because
  // the unwanted blocks of the product are simply not read back
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float4 a = *reinterpret_cast<const float4*>(row + 16 * m);
    float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int kb = 4 * m + u;
      if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bmat[kb], acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bmat[kb], acc0, 0, 0, 0);
    }
  }
  float d[4] = {acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]};
  // lane rows 0,1 hold the w rows (splat 4 k + r), columns < 6; lane rows 2,3 the f rows (splat 4 (k - 2) + r),
  // columns 6 .. 6 + CG - 1
  bool wrow = k < 2;
  int sbase = 4 * (k & 1);
  bool col_ok = wrow ? (ij < 6) : (ij >= 6 && ij < 6 + CG);
  int col = (CG == D || wrow) ? ij : (6 + D - 1);  // depth-only variant: its single colour column is the last one
  int4 gs = *reinterpret_cast<const int4*>(&sb.gslot[wv][sbase]);
  int sl[4] = {gs.x, gs.y, gs.z, gs.w};
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (col_ok && sbase + r < count && d[r] != 0.f) {
      // a wave visits a slot once per batch: in deterministic mode its row is a plain store into the wave's own copy
      if (DET) sb.acc[(wv * GSL_MB + sl[r]) * AP + col] = d[r];
      else atomicAdd(&sb.acc[sl[r] * AP + col], d[r]);
    }
}

template <int D, int CG, bool DET>
__device__ __forceinline__ void mraster_bwd_body(
    FStageM<D, DET>& sb, const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2,
    const uint4* __restrict__ Qh, const int32_t* __restrict__ flatten_ids, float* __restrict__ vacc, long long rs,
    long long re, int nb, int tid, float px, float py, float qcx, float qcy, float tcx, float tcy, bool inside,
    int bin_final, int wave_final, float T_final, const float (&vc)[D], float va) {
  constexpr bool RGB = D >= 3;
  constexpr bool DEPTH = (D == 1) || (D == 4);
  constexpr int A = FStageM<D, DET>::A;
  constexpr int AP = FStageM<D, DET>::AP;
  constexpr int NW = FStageM<D, DET>::NW;
  int lane = tid & 63, wv = tid >> 6;
  float T = T_final;
  float Bp = -T_final * va;
  unsigned long long insidem = __ballot(inside);
  // B operand of this lane: K-step kb of lane-row k is pixel (= walk lane) pl = 16 (kb >> 2) + 4 k + (kb & 3)
  float phi[16], bcol[16];
  {
    int k = lane >> 4, j = lane & 15;
    float wlx = __shfl(px - tcx, 0, 64), wly = __shfl(py - tcy, 0, 64);  // offset of the quadrant's first pixel
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      int pl = 16 * (kb >> 2) + 4 * k + (kb & 3);
      float lx = wlx + (float)(pl & 7), ly = wly + (float)(pl >> 3);
      float m = (j == 0) ? 1.f : (j == 1) ? lx : (j == 2) ? ly : (j == 3) ? lx * lx : (j == 4) ? lx * ly : (j == 5) ? ly * ly : 0.f;
      phi[kb] = m;
      bcol[kb] = 0.f;
    }
  }
  if (CG == D) {
    // colour columns: B[pixel][6 + ch] = upstream gradient of channel ch at that pixel
    // (exchanged once through this wave's tile, which is not in use yet)
    float* vcs = sb.tile[wv];
#pragma unroll
    for (int ch = 0; ch < D; ++ch) vcs[lane * D + ch] = vc[ch];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int k = lane >> 4, j = lane & 15;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      int pl = 16 * (kb >> 2) + 4 * k + (kb & 3);
      if (j >= 6 && j < 6 + D) bcol[kb] = vcs[pl * D + (j - 6)];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  } else {
    int j = lane & 15;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) bcol[kb] = (j == 6) ? 1.f : 0.f;  // f already carries v_depth of its pixel
  }
  float bmat[16];  // this lane's column of B: a monomial column (j < 6) or a colour column
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) bmat[kb] = ((lane & 15) < 6) ? phi[kb] : bcol[kb];
  float* wrow = &sb.tile[wv][lane];

  // The records of batch b + 1 are gathered into registers while batch b is walked (the gather is two dependent
  // global loads per thread; nothing else in the kernel can cover their latency at 3 workgroups per CU).
  int pg = 0;
  float4 pr0 = make_float4(0.f, 0.f, 0.f, 0.f), pr1 = pr0, pr2 = pr0;
  auto gather = [&](int b) {
    long long bend = re - 1 - (long long)b * GSL_MB;
    int bsize = (int)min((long long)GSL_MB, bend + 1 - rs);
    if (tid < bsize) {
      pg = flatten_ids[bend - tid];
      load_record(Q0, Q1, Q2, Qh, pg, RGB && CG == D, pr0, pr1, pr2);
    }
  };
  gather(0);

  for (int b = 0; b < nb; ++b) {
    long long bend = re - 1 - (long long)b * GSL_MB;  // slot t <-> absolute index bend - t (back to front)
    int bsize = (int)min((long long)GSL_MB, bend + 1 - rs);
    __syncthreads();
    if (tid < bsize) {
      sb.id[tid] = pg;
      sb.s0[tid] = pr0;
      sb.s1[tid] = pr1;
      if (RGB && CG == D) sb.s2[tid] = pr2;
    }
    if (tid < GSL_MB) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int k = 0; k < A; ++k) sb.acc[(w * GSL_MB + tid) * AP + k] = 0.f;
    }
    __syncthreads();
    if (b + 1 < nb) gather(b + 1);
    int t_first = (int)max((long long)0, bend - (long long)wave_final);
    int hh = 0;  // splats in the open group
    for (int c = (t_first / 64) * 64; c < bsize; c += 64) {
      int e = c + lane;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = make_float4(0.f, 0.f, 0.f, -1.f);
      if (e < bsize && e >= t_first) {
        a0 = sb.s0[e];
        a1 = sb.s1[e];
      }
      unsigned long long m = __ballot(fabsf(a0.x - qcx) <= a1.w + 3.5f) & __ballot(fabsf(a0.y - qcy) <= a1.w + 3.5f);
      if (!m) continue;
      int t = c + __ffsll((long long)m) - 1;
      m &= m - 1;
      float4 q0 = sb.s0[t], q1 = sb.s1[t];
      for (;;) {
        // the next survivor's records are requested before this one's arithmetic (wave-uniform LDS addresses)
        bool more = m != 0;
        int tn = t;
        float4 q0n = q0, q1n = q1;
        if (more) {
          tn = c + __ffsll((long long)m) - 1;
          m &= m - 1;
          q0n = sb.s0[tn];
          q1n = sb.s1[tn];
        }
        float dx = q0.x - px, dy = q0.y - py;
        float gx = q1.x * dx + q1.y * dy;
        float gy = q1.y * dx + q1.z * dy;
        float sigma = 0.5f * (dx * gx + dy * gy);
        float vis = __expf(-sigma);
        float opv = q0.w * vis;
        float alpha = fminf(GSL_ALPHA_MAX, opv);
        unsigned long long validm = insidem & __ballot((int)(bend - t) <= bin_final) & __ballot(sigma >= 0.f) &
                                    __ballot(alpha >= GSL_ALPHA_MIN);
        if (validm) {  // some pixel of this quadrant composited the splat
          unsigned long long capm = __ballot(opv <= GSL_ALPHA_MAX);
          float am = sel64(validm, alpha, 0.f);  // other lanes: alpha = 0 => ra = 1, fac = 0, nothing changes
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
            if (DEPTH) cdot += q0.z * vc[D - 1];
          } else {
            cdot = q0.z * vc[D - 1];
          }
          float v_alpha = T * cdot - ra * Bp;
          Bp += fac * cdot;
          float vism = sel64(validm & capm, vis, 0.f);  // alpha clamped at 0.999 => no geometric gradient
          wrow[hh * GSL_MPITCH] = vism * v_alpha;
          wrow[(8 + hh) * GSL_MPITCH] = (CG == D) ? fac : fac * vc[D - 1];
          if (lane == 0) sb.gslot[wv][hh] = t;
          if (++hh == 8) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mraster_group_flush<D, CG, DET>(sb, wv, lane, 8, bmat);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            hh = 0;
          }
        }
        if (!more) break;
        t = tn;
        q0 = q0n;
        q1 = q1n;
      }
    }
    if (hh) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      mraster_group_flush<D, CG, DET>(sb, wv, lane, hh, bmat);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // Moments -> gradient row (one thread per slot).
    {
      bool nz = false;
      float mo[A];
#pragma unroll
      for (int k = 0; k < A; ++k) mo[k] = 0.f;
      if (tid < bsize) {
#pragma unroll
        for (int k = 0; k < A; ++k) {
          float m = sb.acc[tid * AP + k];
#pragma unroll
          for (int w = 1; w < NW; ++w) m += sb.acc[(w * GSL_MB + tid) * AP + k];  // fixed wave order
          mo[k] = m;
          nz = nz || (m != 0.f);
        }
        if (nz) {
          float4 q0 = sb.s0[tid], q1 = sb.s1[tid];
          float X = q0.x - tcx, Y = q0.y - tcy, S = mo[0];
          float Sx = X * S - mo[1], Sy = Y * S - mo[2];
          float Sxx = X * (X * S - 2.f * mo[1]) + mo[3];
          float Sxy = X * (Y * S - mo[2]) - Y * mo[1] + mo[4];
          float Syy = Y * (Y * S - 2.f * mo[2]) + mo[5];
          float no = -q0.w;  // v_sigma = -opacity * w
          mo[0] = no * (q1.x * Sx + q1.y * Sy);
          mo[1] = no * (q1.y * Sx + q1.z * Sy);
          mo[2] = 0.5f * no * Sxx;
          mo[3] = no * Sxy;
          mo[4] = 0.5f * no * Syy;
          mo[5] = S;
          if (!DET) {  // the packed flush below reads the rows from LDS
#pragma unroll
            for (int k = 0; k < 6; ++k) sb.acc[tid * AP + k] = mo[k];
          }
        }
      }
      if (DET) {
        // deterministic mode: the (tile, splat) row goes to the intersection's own slot with plain stores (zero rows
        // included); the projection backward sums a Gaussian's rows in tile order
        if (tid < bsize) {
          float pad[12];
#pragma unroll
          for (int k = 0; k < 12; ++k) pad[k] = (k < A && nz) ? mo[k] : 0.f;
          float4* dst = reinterpret_cast<float4*>(vacc) + 4 * (size_t)(bend - tid);
          dst[0] = make_float4(pad[0], pad[1], pad[2], pad[3]);
          dst[1] = make_float4(pad[4], pad[5], pad[6], pad[7]);
          dst[2] = make_float4(pad[8], pad[9], pad[10], pad[11]);
        }
      } else {
        // pack non-zero slots so that 16 consecutive lanes add one Gaussian's 64-byte row
        unsigned long long mask = __ballot(nz);
        int cnt = __popcll(mask);
        if (nz) sb.list[wv][__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)tid;
        __syncthreads();
        int f = lane & 15;
        for (int i0 = 0; i0 < cnt; i0 += 4) {
          int gi = i0 + (lane >> 4);
          if (gi < cnt && f < A) {
            int sl = sb.list[wv][gi];
            size_t g = (size_t)sb.id[sl];
            atomicAdd(&vacc[g * 16 + f], sb.acc[sl * AP + f]);
          }
        }
      }
    }
  }
}

template <int D, bool ED, bool DET>
__global__ __launch_bounds__(256) void k_mraster_bwd(
    const float4* __restrict__ Q0, const float4* __restrict__ Q1, const float4* __restrict__ Q2, int W, int H,
    int tile_w, int ty0, const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids,
    long long capacity, const float* __restrict__ render, const float* __restrict__ alphas,
    const int32_t* __restrict__ last_ids, const float* __restrict__ v_render, const float* __restrict__ v_alphas,
    float* __restrict__ vacc, int row0, int row1, const uint4* __restrict__ Qh) {
  // DET: vacc is vrow[capacity][16], one row per intersection; otherwise vacc[N][16], one row per Gaussian (atomics)
  __shared__ FStageM<D, DET> sb;
  __shared__ int s_final[4];
  int tile = ty0 * tile_w + GSL_TILE_OF_BLOCK();
  int tyi = tile / tile_w, txi = tile - tyi * tile_w;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int qx = txi * 16 + (wv & 1) * 8, qy = tyi * 16 + (wv >> 1) * 8;
  int j = qx + (lane & 7), i = qy + (lane >> 3);
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W) && (i >= row0) && (i < row1);
  float qcx = (float)qx + 4.f, qcy = (float)qy + 4.f;
  float tcx = (float)(txi * 16) + 8.f, tcy = (float)(tyi * 16) + 8.f;

  long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
  if (re > capacity) re = capacity;
  if (rs >= re) return;

  size_t pid = inside ? ((size_t)i * W + j) : 0;
  float Aimg = inside ? alphas[pid] : 0.f;
  float T_final = 1.f - Aimg;
  int bin_final = inside ? last_ids[pid] : -1;
  float vc[D];
  float va = inside ? v_alphas[pid] : 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) vc[k] = inside ? v_render[pid * D + k] : 0.f;
  if (ED && inside) {
    float dn = render[pid * D + (D - 1)];
    float vd = vc[D - 1];
    if (Aimg >= 1e-10f) va += -vd * dn / Aimg;
    vc[D - 1] = vd / fmaxf(Aimg, 1e-10f);
  }
  int wave_final = bin_final;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wave_final = max(wave_final, __shfl_xor(wave_final, o, 64));
  if (lane == 0) s_final[wv] = wave_final;
  bool rgb_grad = false;
  if (D == 4) rgb_grad = (vc[0] != 0.f) || (vc[1] != 0.f) || (vc[2] != 0.f);
  int any_rgb = __syncthreads_or(rgb_grad);
  int block_final = max(max(s_final[0], s_final[1]), max(s_final[2], s_final[3]));
  // nothing behind block_final was composited by any pixel of the tile: start there
  long long re_all = re;
  if ((long long)block_final + 1 < re) re = max((long long)block_final + 1, rs);
  if (DET) {  // rows of the entries that are not walked are zero
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long q = 4 * re + tid; q < 4 * re_all; q += 256) reinterpret_cast<float4*>(vacc)[q] = z;
  }
  if (rs >= re) return;
  int nb = (int)((re - rs + GSL_MB - 1) / GSL_MB);
  if (D == 4 && !any_rgb)
    mraster_bwd_body<D, 1, DET>(sb, Q0, Q1, Q2, Qh, flatten_ids, vacc, rs, re, nb, tid, px, py, qcx, qcy, tcx, tcy, inside,
                           bin_final, wave_final, T_final, vc, va);
  else
    mraster_bwd_body<D, D, DET>(sb, Q0, Q1, Q2, Qh, flatten_ids, vacc, rs, re, nb, tid, px, py, qcx, qcy, tcx, tcy, inside,
                           bin_final, wave_final, T_final, vc, va);
}

// ------------------------------------------------------------------------------------------------
// Backward 2: per-Gaussian vjp of projection + colour, and the pose reduction.
// Reads (and clears) the 64-byte gradient rows.  partial rows: [v_R 9][v_t 3][v_campos 3].
// ------------------------------------------------------------------------------------------------
template <bool FULL, int D>
__global__ __launch_bounds__(256) void k_fproject_bwd(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ colors, int sh_degree, int K_sh,
    const float* __restrict__ V, const float* __restrict__ Kmat, int N, int W, int H, float eps2d, int antialiased,
    const int32_t* __restrict__ radii, const float4* __restrict__ Q1, const float* __restrict__ comps,
    float4* __restrict__ vacc, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales, float* __restrict__ v_opacities, float* __restrict__ v_colors,
    float* __restrict__ partials, const float4* __restrict__ vrow, const uint64_t* __restrict__ skeys,
    const int32_t* __restrict__ tile_offsets, const float4* __restrict__ Q0, int tile_w, int tile_h, int ty0, int ty1,
    long long capacity, float4* __restrict__ trec, const float* __restrict__ vcT, int32_t* __restrict__ vc_state) {
  constexpr bool RGB = D >= 3;
  int i = blockIdx.x * 256 + threadIdx.x;
  Cam cam = load_cam(V, Kmat);
  // vc_state (may be NULL): 1 = the caller's v_colors buffer is known to hold zeros only (it is the same buffer call after
  // call, and nobody has written a non-zero since); a Gaussian without a colour gradient -- every Gaussian, under
  // GsplatLoc's depth-only loss -- then skips its 48 bytes of zero stores.  A thread that writes a real gradient marks
  // the buffer dirty (2); k_freduce_viewmat, which runs after the whole grid, turns "nobody wrote a non-zero in a launch
  // that stored everything" into 1 again.
  const bool vc_zero = FULL && RGB && vc_state && *vc_state == 1;
  // tiny-splat backward, pass 2 fused in: four lanes per Gaussian fold its 4x4 slab of (w, alpha*T) records into the
  // gradient row, which stays in LDS for the thread that owns the Gaussian (no row round trip, no gather launch)
  __shared__ float4 srow[256][3];
  if (trec) {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {  // two (Gaussian, slab row) items per turn: their loads leave together
      TinySlabIn in[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int t = (2 * half + u) * 256 + threadIdx.x;
        in[u] = tiny_slab_load(radii, Q0, Q1, trec, blockIdx.x * 256 + (t >> 2), t & 3, N);
      }
      GSL_TINY_SLAB_PIN(in[0]);
      GSL_TINY_SLAB_PIN(in[1]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        int t = (2 * half + u) * 256 + threadIdx.x;
        int lg = t >> 2, r = t & 3, gid = blockIdx.x * 256 + lg;
        float v[6 + D];
        tiny_slab_fold<D>(in[u], W, H, trec, vcT, gid, r, v);
        float pad[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) pad[k] = (k < 6 + D) ? v[k] : 0.f;
        if (r == 0) srow[lg][0] = make_float4(pad[0], pad[1], pad[2], pad[3]);
        if (r == 1) srow[lg][1] = make_float4(pad[4], pad[5], pad[6], pad[7]);
        if (r == 2) srow[lg][2] = make_float4(pad[8], pad[9], pad[10], pad[11]);
      }
    }
    __syncthreads();
  }
  float acc15[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) acc15[k] = 0.f;
  float vmean[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
  float vop = 0.f;
  float vrgb[3] = {0.f, 0.f, 0.f};
  bool live = (i < N) && (radii[i] > 0);
  bool sh_live = false;
  if (live) {
    float4 r0, r1, r2;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vrow) {
      // deterministic mode: the Gaussian's rows were stored per intersection; find its entry in each tile list of
      // its rectangle (the lists are sorted by (depth bits, id): binary search) and add the rows in tile order
      r0 = r1 = r2 = z;
      float4 q0 = GSL_Q(Q0, i);
      int xmin, ymin, xmax, ymax;
      tile_rect(q0.x, q0.y, radii[i], 16, tile_w, tile_h, xmin, ymin, xmax, ymax);
      ymin = max(ymin, ty0);
      ymax = min(ymax, ty1);
      uint64_t want = ((uint64_t)__float_as_uint(q0.z) << 32) | (uint32_t)i;
      for (int y = ymin; y < ymax; ++y)
        for (int x = xmin; x < xmax; ++x) {
          long long lo = tile_offsets[y * tile_w + x], hi = tile_offsets[y * tile_w + x + 1];
          if (hi > capacity) hi = capacity;
          while (lo < hi) {
            long long mid = (lo + hi) >> 1;
            if (skeys[mid] < want) lo = mid + 1; else hi = mid;
          }
          if (lo < capacity && skeys[lo] == want) {
            float4 a = vrow[4 * lo], b = vrow[4 * lo + 1], c = vrow[4 * lo + 2];
            r0.x += a.x; r0.y += a.y; r0.z += a.z; r0.w += a.w;
            r1.x += b.x; r1.y += b.y; r1.z += b.z; r1.w += b.w;
            r2.x += c.x; r2.y += c.y; r2.z += c.z; r2.w += c.w;
          }
        }
    } else if (trec) {
      r0 = srow[threadIdx.x][0]; r1 = srow[threadIdx.x][1]; r2 = srow[threadIdx.x][2];
      if (vacc) {  // tiny-splat mode with long tile lists: those tiles' rows arrive through vacc (gsl_long_raster_bwd)
        float4 a = vacc[4 * (size_t)i], b = vacc[4 * (size_t)i + 1], c = vacc[4 * (size_t)i + 2];
        if (a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f || b.x != 0.f || b.y != 0.f || b.z != 0.f || b.w != 0.f ||
            c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f) {
          r0.x += a.x; r0.y += a.y; r0.z += a.z; r0.w += a.w;
          r1.x += b.x; r1.y += b.y; r1.z += b.z; r1.w += b.w;
          r2.x += c.x; r2.y += c.y; r2.z += c.z; r2.w += c.w;
          vacc[4 * (size_t)i] = z; vacc[4 * (size_t)i + 1] = z; vacc[4 * (size_t)i + 2] = z;
        }
      }
    } else {
      r0 = vacc[4 * (size_t)i]; r1 = vacc[4 * (size_t)i + 1]; r2 = vacc[4 * (size_t)i + 2];
      vacc[4 * (size_t)i] = z; vacc[4 * (size_t)i + 1] = z; vacc[4 * (size_t)i + 2] = z;
    }
    // row = [vx vy | va vb vc | vop | col0 col1 col2 col3 ...]
    float vm2x = r0.x, vm2y = r0.y, v_ca = r0.z, v_cb = r0.w, v_cc = r1.x, vop_eff = r1.y;
    float col[4] = {r1.z, r1.w, r2.x, r2.y};
    float vdepth = (D == 1) ? col[0] : ((D == 4) ? col[3] : 0.f);
    if (RGB) { vrgb[0] = col[0]; vrgb[1] = col[1]; vrgb[2] = col[2]; }
    float comp = 0.f, vcomp = 0.f;
    vop = vop_eff;
    if (antialiased) {
      comp = comps[i];
      vcomp = vop_eff * opacities[i];
      vop = vop_eff * comp;
    }
    ProjMid p;
    float q[4], s[3];
    load_gaussian(means, quats, scales, i, cam, p, q, s);
    p.covar = quat_scale_to_covar(q, s);
    p.covar_c = mul_bt(mul(cam.R, p.covar), cam.R);
    persp_mid(cam, W, H, p);
    float4 q1 = GSL_Q(Q1, i);
    project_vjp<FULL>(cam, eps2d, p, q, s, q1.x, q1.y, q1.z, vm2x, vm2y, vdepth, v_ca, v_cb, v_cc, antialiased != 0,
                      comp, vcomp, acc15, vmean, vq, vs);
    sh_live = RGB && (sh_degree >= 0) && (vrgb[0] != 0.f || vrgb[1] != 0.f || vrgb[2] != 0.f);
    if (sh_live) {
      // colour = max(SH(dir) + 0.5, 0), dir = mean - campos
      M3 Ri;
      float cp[3];
      cam_inverse(cam, Ri, cp);
      float rx = p.mean[0] - cp[0], ry = p.mean[1] - cp[1], rz = p.mean[2] - cp[2];
      float inorm = rsqrtf(rx * rx + ry * ry + rz * rz);
      float x = rx * inorm, y = ry * inorm, zz = rz * inorm;
      float Y[16];
      sh_basis(sh_degree, x, y, zz, Y);
      int nK = (sh_degree + 1) * (sh_degree + 1);
      const float* cf = colors + (size_t)i * K_sh * 3;
      float c0 = 0.5f, c1 = 0.5f, c2 = 0.5f;
      for (int k = 0; k < nK; ++k) {
        c0 += Y[k] * cf[3 * k]; c1 += Y[k] * cf[3 * k + 1]; c2 += Y[k] * cf[3 * k + 2];
      }
      if (!(c0 > 0.f)) vrgb[0] = 0.f;  // clamp_min(x, 0) passes the gradient where x > 0
      if (!(c1 > 0.f)) vrgb[1] = 0.f;
      if (!(c2 > 0.f)) vrgb[2] = 0.f;
      float sk[16];
      for (int k = 0; k < nK; ++k) {
        sk[k] = cf[3 * k] * vrgb[0] + cf[3 * k + 1] * vrgb[1] + cf[3 * k + 2] * vrgb[2];
        if (FULL) {
          if (k == 0 && vc_state) *vc_state = 2;
          v_colors[((size_t)i * K_sh + k) * 3] = Y[k] * vrgb[0];
          v_colors[((size_t)i * K_sh + k) * 3 + 1] = Y[k] * vrgb[1];
          v_colors[((size_t)i * K_sh + k) * 3 + 2] = Y[k] * vrgb[2];
        }
      }
      if (FULL)
        for (int k = nK * 3; k < K_sh * 3; ++k) v_colors[(size_t)i * K_sh * 3 + k] = 0.f;
      float g[3];
      sh_basis_grad(sh_degree, x, y, zz, sk, g);
      float dd = g[0] * x + g[1] * y + g[2] * zz;
      float gd[3] = {(g[0] - dd * x) * inorm, (g[1] - dd * y) * inorm, (g[2] - dd * zz) * inorm};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vmean[k] += gd[k];
        acc15[12 + k] = -gd[k];
      }
    }
  }
  if (FULL && i < N) {
#pragma unroll
    for (int k = 0; k < 3; ++k) v_means[3 * (size_t)i + k] = vmean[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quats[4 * (size_t)i + k] = vq[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) v_scales[3 * (size_t)i + k] = vs[k];
    v_opacities[i] = vop;
    if (RGB && !sh_live) {
      if (sh_degree < 0) {
        const bool nz = vrgb[0] != 0.f || vrgb[1] != 0.f || vrgb[2] != 0.f;
        if (nz && vc_state) *vc_state = 2;
        if (nz || !vc_zero) {
          v_colors[3 * (size_t)i] = vrgb[0]; v_colors[3 * (size_t)i + 1] = vrgb[1]; v_colors[3 * (size_t)i + 2] = vrgb[2];
        }
      } else if (!vc_zero) {
        // (12 coefficients = 48 bytes per Gaussian for SH degree 1: three 16-byte stores instead of twelve strided
        // 4-byte ones; any other band count keeps the loop)
        if (K_sh == 4) {
          float4* dst = reinterpret_cast<float4*>(v_colors + (size_t)i * 12);
          float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
          dst[0] = z4; dst[1] = z4; dst[2] = z4;
        } else {
          for (int k = 0; k < K_sh * 3; ++k) v_colors[(size_t)i * K_sh * 3 + k] = 0.f;
        }
      }
    }
  }
  if (partials != nullptr) {
    __shared__ float red[4][15];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      float sum = wave_sum(acc15[k]);
      if (lane == 0) red[wv][k] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 15)
      partials[(size_t)blockIdx.x * 16 + threadIdx.x] =
          red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

// Fixed-order sum of the partial rows, chain of the SH view direction through the camera
// position (campos = -R^-1 t), result into v_viewmat[16] (row 3 = 0: that row is constant).
// Stage 1 for many rows (N > 1 M: one workgroup summing 19 532 rows took 41 us at workload X): 64 workgroups each add
// up a contiguous span of rows (same thread layout as reduce_viewmat_rows: thread = (row mod 64, quarter), fixed order)
// into one row of `out`; k_freduce_viewmat then sums those 64.
__global__ __launch_bounds__(256) void k_freduce_rows(const float* __restrict__ partials, int nb, float* __restrict__ out) {
  __shared__ float red[4][16];
  int span = (nb + (int)gridDim.x - 1) / (int)gridDim.x;
  int r0 = blockIdx.x * span, r1 = min(nb, r0 + span);
  int q = threadIdx.x & 3;
  const float4* rows = reinterpret_cast<const float4*>(partials) + q;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = r0 + (threadIdx.x >> 2); b < r1; b += 64) {
    float4 x = rows[(size_t)b * 4];
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
  }
  float v4[4] = {a.x, a.y, a.z, a.w};
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = v4[c];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
    if (lane < 4) red[wv][4 * q + c] = x;
  }
  __syncthreads();
  if (threadIdx.x < 16)
    out[(size_t)blockIdx.x * 16 + threadIdx.x] =
        threadIdx.x < 15 ? (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]) : 0.f;
}

__global__ __launch_bounds__(1024) void k_freduce_viewmat(const float* __restrict__ partials, int nb,
                                                         const float* __restrict__ V, const float* __restrict__ Kmat,
                                                         float* __restrict__ v_viewmat, int32_t* __restrict__ vc_state) {
  __shared__ float red[16][15];
  __shared__ float tot[15];
  float v = reduce_viewmat_rows_wide(partials, nb, V, Kmat, red, tot);
  if (threadIdx.x < 16) v_viewmat[threadIdx.x] = v;
  // (see k_fproject_bwd) 2: a real colour gradient was written in the launch before this one -> unknown; otherwise every
  // Gaussian's slot holds zeros now
  if (vc_state && threadIdx.x == 0) *vc_state = (*vc_state == 2) ? 0 : 1;
}

}  // namespace gsl

// ---------------------------------------------------------------------------------------------- C ABI
// ws = [tile_counts n_tiles][cursors n_tiles][pad to 16 bytes][pose-gradient rows ceil(N/256) x 16 floats]
#define GSL_VM_STAGE_ROWS 64  // scratch rows behind the ceil(N / 256) pose-gradient rows: stage 1 of the row reduction
static inline size_t gsl_vm_rows_offset(int n_tiles) { return ((size_t)2 * (size_t)n_tiles * sizeof(int32_t) + 15) & ~(size_t)15; }

// state word of the binned mode's counter contract: the 16th float of the first pose-gradient row (rows use 15)
static inline int32_t* gsl_bin_state(void* ws, int n_tiles) { return (int32_t*)((char*)ws + gsl_vm_rows_offset(n_tiles)) + 15; }
extern "C" int32_t* gsl_fused_bin_state(void* ws, int n_tiles) { return (ws && n_tiles > 0) ? gsl_bin_state(ws, n_tiles) : nullptr; }

extern "C" size_t gsl_fused_ws_bytes(int N, int n_tiles) {
  // [tile_counts n_tiles][cursors n_tiles][partials ceil(N/256)*16 floats]
  size_t nb = ((size_t)(N > 0 ? N : 1) + 255) / 256;
  return gsl_vm_rows_offset(n_tiles > 0 ? n_tiles : 1) + (nb + GSL_VM_STAGE_ROWS) * 16 * sizeof(float);
}

// where gsl_fused_project_bwd leaves the pose-gradient rows inside ws: ceil(N / 256) rows of 16 floats (15 used)
extern "C" const float* gsl_fused_viewmat_rows(const void* ws, int n_tiles) {
  if (!ws || n_tiles <= 0) return nullptr;
  return (const float*)((const char*)ws + gsl_vm_rows_offset(n_tiles));
}

extern "C" int gsl_fused_project(const float* means, const float* quats, const float* scales, const float* opacities,
                                 const float* colors, int sh_degree, int K_sh, const float* viewmat, const float* K,
                                 int N, int width, int height, float eps2d, float near_plane, float far_plane,
                                 float radius_clip, int antialiased, int tile_w, int tile_h, int ty0, int ty1,
                                 int32_t* radii, float* Q0, float* Q1, float* Q2, float* compensations,
                                 int32_t* tiles_per_gauss, int32_t* tile_offsets, int32_t* n_isects, void* ws,
                                 size_t ws_bytes, void* Qh, void* bins, int bin_cap, int32_t* flags,
                                 const int32_t* order_ids, void* stream) {
  if (N < 0 || width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1)
    return GSL_ERR_BAD_ARG;
  if (N > GSL_MAX_GAUSSIANS) return GSL_ERR_BAD_ARG;  // (packed gradient rows are addressed by 32-bit byte offsets)
  if (tile_w * 16 < width || tile_h * 16 < height) return GSL_ERR_BAD_ARG;
  int n_tiles = tile_w * tile_h, nst = (ty1 - ty0) * tile_w;
  if (nst > GSL_F_MAX_STRIP_TILES) return GSL_ERR_BAD_ARG;
  if (!viewmat || !K || !tile_offsets || !n_isects) return GSL_ERR_BAD_ARG;
  if (N > 0 && (!means || !quats || !scales || !opacities || !radii || !Q0 || !Q1)) return GSL_ERR_BAD_ARG;
  if (Q2 && !colors) return GSL_ERR_BAD_ARG;
  if (Q2 && sh_degree >= 0 && (sh_degree > 3 || K_sh < (sh_degree + 1) * (sh_degree + 1))) return GSL_ERR_BAD_ARG;
  if (antialiased && !compensations) return GSL_ERR_BAD_ARG;
  if (!ws || ws_bytes < gsl_fused_ws_bytes(N, n_tiles)) return GSL_ERR_WORKSPACE;
  if (bins && bin_cap <= 0) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int32_t* counts = (int32_t*)ws;
  int32_t* cursors = counts + n_tiles;
  GSL_CLAMP_DEPTH_WINDOW(near_plane, far_plane);
  // binned mode relies on the scan leaving the counters cleared (ws zero-filled once by the caller): no clearing launch
  if (!bins && gsl::zero_u32(counts, (size_t)n_tiles, st) != GSL_OK) return GSL_ERR_HIP;
  if (N > 0) {
    dim3 grid((N + GSL_F_BIN_THREADS - 1) / GSL_F_BIN_THREADS), block(GSL_F_BIN_THREADS);
    size_t lds = (size_t)(nst + 2) * sizeof(int);  // counters + the touched range
#define CALL_P(RGBV, BINV)                                                                                            \
  hipLaunchKernelGGL((gsl::k_fproject<RGBV, BINV>), grid, block, lds, st, means, quats, scales, opacities, colors,    \
                     sh_degree, K_sh, viewmat, K, N, width, height, eps2d, near_plane, far_plane, radius_clip,        \
                     antialiased, tile_w, tile_h, ty0, ty1, radii, (float4*)Q0, (float4*)Q1, (float4*)Q2,             \
                     compensations, tiles_per_gauss, counts, (uint4*)Qh, (uint64_t*)bins, bin_cap,                   \
                     bins ? gsl_bin_state(ws, n_tiles) : (int32_t*)nullptr, flags, order_ids)
    if (Q2) { if (bins) CALL_P(true, true); else CALL_P(true, false); }
    else { if (bins) CALL_P(false, true); else CALL_P(false, false); }
#undef CALL_P
    GSL_CHECK_LAUNCH();
  }
  if (bins) return GSL_OK;  // binned mode: gsl_fused_bin's sort kernel adds up the tile sizes itself
  hipLaunchKernelGGL(gsl::k_ftile_scan, dim3(1), dim3(1024), 0, st, counts, n_tiles, tile_offsets, n_isects, cursors, 0,
                     flags);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// defined in binning.hip
extern "C" int gsl_tile_sort_keys(int32_t* tile_offsets, int tile_begin, int n_strip_tiles, int64_t capacity,
                                  uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, int64_t cam_enc,
                                  int write_sorted_keys, uint64_t* bins, int bin_cap, const int32_t* counts,
                                  int32_t* n_isects, int32_t* flags, int long_min, int occupied_tiles,
                                  const int32_t* storage_of, void* stream);

extern "C" int gsl_fused_bin(const float* Q0, const int32_t* radii, int N, int tile_w, int tile_h, int ty0, int ty1,
                             int tile_n_bits, int32_t* tile_offsets, int64_t capacity, uint64_t* sort_keys,
                             int32_t* flatten_ids, int64_t* isect_ids, void* ws, size_t ws_bytes,
                             int write_sorted_keys, void* bins, int bin_cap, int32_t* n_isects, int32_t* flags,
                             int long_min, const int32_t* order_ids, const int32_t* storage_of, void* stream) {
  if (N < 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 || capacity < 0)
    return GSL_ERR_BAD_ARG;
  int n_tiles = tile_w * tile_h, nst = (ty1 - ty0) * tile_w;
  if (nst > GSL_F_MAX_STRIP_TILES || !tile_offsets) return GSL_ERR_BAD_ARG;
  if (bins) {
    // gsl_fused_project already put every key into its tile's bin and left the tile sizes in the counters: the sort
    // kernel runs over ALL tiles, writes tile_offsets[n_tiles + 1] and n_isects itself (sizes outside the strip are 0)
    if (bin_cap <= 0 || !n_isects) return GSL_ERR_BAD_ARG;
    if (!ws || ws_bytes < gsl_fused_ws_bytes(N, n_tiles)) return GSL_ERR_WORKSPACE;
    if (capacity > 0 && (!sort_keys || !flatten_ids)) return GSL_ERR_BAD_ARG;
    return gsl_tile_sort_keys(tile_offsets, 0, n_tiles, capacity, sort_keys, flatten_ids, isect_ids, 0,
                              write_sorted_keys, (uint64_t*)bins, bin_cap, (const int32_t*)ws, n_isects, flags,
                              write_sorted_keys ? 0 : long_min, nst, storage_of, stream);
  }
  if (N == 0 || capacity == 0 || nst == 0) return GSL_OK;
  if (!Q0 || !radii || !sort_keys || !flatten_ids) return GSL_ERR_BAD_ARG;
  if (!ws || ws_bytes < gsl_fused_ws_bytes(N, n_tiles)) return GSL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int32_t* cursors = (int32_t*)ws + n_tiles;
  hipLaunchKernelGGL(gsl::k_fscatter, dim3((N + GSL_F_BIN_THREADS - 1) / GSL_F_BIN_THREADS), dim3(GSL_F_BIN_THREADS),
                     (size_t)2 * nst * sizeof(int), st, (const float4*)Q0, radii, N, tile_w, tile_h, ty0, ty1,
                     tile_offsets, cursors, (long long)capacity, sort_keys, order_ids);
  GSL_CHECK_LAUNCH();
  return gsl_tile_sort_keys(tile_offsets, ty0 * tile_w, nst, capacity, sort_keys, flatten_ids, isect_ids, 0,
                            write_sorted_keys, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, storage_of, stream);
}

// defined in raster_g16.hip
extern "C" int gsl_g16_raster_bwd_launch(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                         int height, int tile_w, int ty0, int ty1, const int32_t* tile_offsets,
                                         const int32_t* flatten_ids, int64_t capacity, const float* render,
                                         const float* alphas, const int32_t* last_ids, const float* v_render,
                                         const float* v_alphas, float* vacc, int row0, int row1, const void* Qh,
                                         const uint32_t* isect_hits, const int32_t* isect_hit_counts, int long_min,
                                         void* long_ws, int max_seg, int rgb_flag_index, void* clear_ws,
                                         void* stream);

#define GSL_F_DISPATCH(D, ED, CALL)                     \
  if (D == 1) { if (ED) CALL(1, true); else CALL(1, false); }   \
  else if (D == 3) { CALL(3, false); }                          \
  else if (D == 4) { if (ED) CALL(4, true); else CALL(4, false); } \
  else return GSL_ERR_BAD_ARG;

// Compositing backward (k_mraster_bwd): quadrant walk, per-splat pixel sums on the matrix cores.  Adds into vacc.
// Pixel rows outside [row0, row1) are not touched (strip rendering with a one-pixel halo).
extern "C" int gsl_fused_raster_bwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed,
                                   int width, int height, int tile_w, int tile_h, int ty0, int ty1,
                                   const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                                   const float* render, const float* alphas, const int32_t* last_ids,
                                   const float* v_render, const float* v_alphas, float* vacc, int row0, int row1,
                                   const void* Qh, float* vrow, const uint32_t* isect_hits,
                                   const int32_t* isect_hit_counts, int long_min, void* clear_ws, void* stream) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0 || row0 < 0 || row0 > row1)
    return GSL_ERR_BAD_ARG;
  if (tile_w * 16 < width || tile_h * 16 < height) return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render || !alphas || !last_ids || !v_render || !v_alphas) return GSL_ERR_BAD_ARG;
  if (ed && channels == 3) return GSL_ERR_BAD_ARG;
  if (capacity == 0 || ty0 == ty1 || row0 == row1) return GSL_OK;
  if (!flatten_ids || (!vacc && !vrow)) return GSL_ERR_BAD_ARG;
  if (isect_hits && !isect_hit_counts) return GSL_ERR_BAD_ARG;
  if (!Qh && (!Q0 || !Q1 || (channels >= 3 && !Q2))) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int nblk = (ty1 - ty0) * tile_w;
  if (vrow && clear_ws) return GSL_ERR_BAD_ARG;  // (the deterministic mode keeps the separate sort launch)
  if (vrow) {  // deterministic mode: one row per intersection, plain stores, no atomics anywhere
#define CALL_MD(DD, EE)                                                                                       \
  hipLaunchKernelGGL((gsl::k_mraster_bwd<DD, EE, true>), dim3(nblk), dim3(256), 0, st, (const float4*)Q0,    \
                     (const float4*)Q1, (const float4*)Q2, width, height, tile_w, ty0, tile_offsets,          \
                     flatten_ids, (long long)capacity, render, alphas, last_ids, v_render, v_alphas, vrow,    \
                     row0, row1, (const uint4*)Qh)
    GSL_F_DISPATCH(channels, ed, CALL_MD)
#undef CALL_MD
    GSL_CHECK_LAUNCH();
    return GSL_OK;
  }
  // non-deterministic path: 16-lane groups, one workgroup per quadrant (raster_g16.hip)
  return gsl_g16_raster_bwd_launch(Q0, Q1, Q2, channels, ed, width, height, tile_w, ty0, ty1, tile_offsets, flatten_ids,
                                   capacity, render, alphas, last_ids, v_render, v_alphas, vacc, row0, row1, Qh,
                                   isect_hits, isect_hit_counts, long_min, nullptr, 0, 4 * tile_w * tile_h, clear_ws,
                                   stream);
}

// Compositing backward of the long tile lists (the segments gsl_long_raster_fwd listed in long_ws): adds into vacc.
extern "C" int gsl_long_raster_bwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                                   int height, int tile_w, int tile_h, int ty0, int ty1, const int32_t* tile_offsets,
                                   const int32_t* flatten_ids, int64_t capacity, const float* render,
                                   const float* alphas, const int32_t* last_ids, const float* v_render,
                                   const float* v_alphas, float* vacc, int row0, int row1, const void* Qh,
                                   const uint32_t* isect_hits, int long_min, void* long_ws, int max_seg,
                                   void* stream) {
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0 || row0 < 0 || row0 > row1 || long_min <= 0 || max_seg <= 0 || !long_ws)
    return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render || !alphas || !last_ids || !v_render || !v_alphas || !vacc) return GSL_ERR_BAD_ARG;
  if (ed && channels == 3) return GSL_ERR_BAD_ARG;
  if (capacity == 0 || ty0 == ty1 || row0 == row1) return GSL_OK;
  if (!flatten_ids || (!Qh && (!Q0 || !Q1 || (channels >= 3 && !Q2)))) return GSL_ERR_BAD_ARG;
  return gsl_g16_raster_bwd_launch(Q0, Q1, Q2, channels, ed, width, height, tile_w, ty0, ty1, tile_offsets, flatten_ids,
                                   capacity, render, alphas, last_ids, v_render, v_alphas, vacc, row0, row1, Qh,
                                   isect_hits, nullptr, long_min, long_ws, max_seg, 0, nullptr, stream);
}

extern "C" int gsl_fused_project_bwd(const float* means, const float* quats, const float* scales,
                                     const float* opacities, const float* colors, int sh_degree, int K_sh,
                                     const float* viewmat, const float* K, int N, int width, int height,
                                     float eps2d, int antialiased, int channels, const int32_t* radii,
                                     const float* Q1, const float* compensations, float* vacc, float* v_means,
                                     float* v_quats, float* v_scales, float* v_opacities, float* v_colors,
                                     float* v_viewmat, void* ws, size_t ws_bytes, int n_tiles, const float* vrow,
                                     const uint64_t* sorted_keys, const int32_t* tile_offsets, const float* Q0,
                                     int tile_w, int tile_h, int ty0, int ty1, int64_t capacity, float* tiny_trec,
                                     const float* tiny_vcT, int reduce_viewmat, int32_t* v_colors_state, void* stream) {
  if (N < 0 || width <= 0 || height <= 0 || n_tiles <= 0) return GSL_ERR_BAD_ARG;
  if (reduce_viewmat && !v_viewmat) return GSL_ERR_BAD_ARG;
  if (channels != 1 && channels != 3 && channels != 4) return GSL_ERR_BAD_ARG;
  bool full = v_means != nullptr;
  if (full != (v_quats != nullptr) || full != (v_scales != nullptr) || full != (v_opacities != nullptr))
    return GSL_ERR_BAD_ARG;
  if (full && channels >= 3 && !v_colors) return GSL_ERR_BAD_ARG;
  if (antialiased && !compensations) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    if (reduce_viewmat && gsl::zero_u32(v_viewmat, 16, st) != GSL_OK) return GSL_ERR_HIP;
    return GSL_OK;
  }
  if (!means || !quats || !scales || !opacities || !viewmat || !K || !radii || !Q1) return GSL_ERR_BAD_ARG;
  if (!vrow && !vacc && !tiny_trec) return GSL_ERR_BAD_ARG;
  if (tiny_trec && (!tiny_vcT || !Q0 || vrow)) return GSL_ERR_BAD_ARG;
  if (vrow && (!sorted_keys || !tile_offsets || !Q0 || tile_w <= 0 || tile_h <= 0 || tile_w * tile_h != n_tiles ||
               ty0 < 0 || ty1 > tile_h || ty0 > ty1 || capacity < 0))
    return GSL_ERR_BAD_ARG;
  if (channels >= 3 && !colors) return GSL_ERR_BAD_ARG;
  if (!ws || ws_bytes < gsl_fused_ws_bytes(N, n_tiles)) return GSL_ERR_WORKSPACE;
  // one row of 15 sums per workgroup; reduce_viewmat = 0 leaves them for gsl_pose_step / gsl_pack_pose_reduce
  float* partials = (float*)((char*)ws + gsl_vm_rows_offset(n_tiles));
  int grid = (N + 255) / 256;
  int32_t* vcs = (full && channels >= 3) ? v_colors_state : nullptr;
#define CALL_PB(FF, DD)                                                                                          \
  hipLaunchKernelGGL((gsl::k_fproject_bwd<FF, DD>), dim3(grid), dim3(256), 0, st, means, quats, scales, opacities, \
                     colors, sh_degree, K_sh, viewmat, K, N, width, height, eps2d, antialiased, radii,             \
                     (const float4*)Q1, compensations, (float4*)vacc, v_means, v_quats, v_scales, v_opacities,    \
                     v_colors, partials, (const float4*)vrow, sorted_keys, tile_offsets, (const float4*)Q0, tile_w,   \
                     tile_h, ty0, ty1, (long long)capacity, (float4*)tiny_trec, tiny_vcT, vcs)
  if (full) {
    if (channels == 1) CALL_PB(true, 1); else if (channels == 3) CALL_PB(true, 3); else CALL_PB(true, 4);
  } else {
    if (channels == 1) CALL_PB(false, 1); else if (channels == 3) CALL_PB(false, 3); else CALL_PB(false, 4);
  }
#undef CALL_PB
  GSL_CHECK_LAUNCH();
  if (reduce_viewmat) {
    if (grid > 8192) {  // two stages (fixed order either way)
      float* stage = partials + (size_t)grid * 16;
      hipLaunchKernelGGL(gsl::k_freduce_rows, dim3(GSL_VM_STAGE_ROWS), dim3(256), 0, st, partials, grid, stage);
      hipLaunchKernelGGL(gsl::k_freduce_viewmat, dim3(1), dim3(1024), 0, st, stage, GSL_VM_STAGE_ROWS, viewmat, K, v_viewmat, vcs);
    } else {
      hipLaunchKernelGGL(gsl::k_freduce_viewmat, dim3(1), dim3(1024), 0, st, partials, grid, viewmat, K, v_viewmat, vcs);
    }
    GSL_CHECK_LAUNCH();
  }
  return GSL_OK;
}
