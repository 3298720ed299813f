// Shared device helpers for libgsloc_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsloc_hip.h"

#define GSL_WAVE 64
#define GSL_ALPHA_MAX 0.999f
#define GSL_ALPHA_MIN (1.0f / 255.0f)
#define GSL_T_STOP 1e-4f
#define GSL_LOG2E 1.4426950408889634f

#define GSL_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return GSL_ERR_HIP;               \
  } while (0)

namespace gsl {

// misc.hip: zero n dwords with a kernel of the library (never hipMemsetAsync: see the note there)
int zero_u32(void* p, size_t n_dwords, hipStream_t st);

struct M3 {  // row-major 3x3
  float m[9];
  __device__ __forceinline__ float& operator()(int r, int c) { return m[r * 3 + c]; }
  __device__ __forceinline__ float operator()(int r, int c) const { return m[r * 3 + c]; }
};

__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
  M3 o;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
  return o;
}
__device__ __forceinline__ M3 mul_bt(const M3& a, const M3& b) {  // a * b^T
  M3 o;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o(i, j) = a(i, 0) * b(j, 0) + a(i, 1) * b(j, 1) + a(i, 2) * b(j, 2);
  return o;
}
__device__ __forceinline__ M3 mul_at(const M3& a, const M3& b) {  // a^T * b
  M3 o;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o(i, j) = a(0, i) * b(0, j) + a(1, i) * b(1, j) + a(2, i) * b(2, j);
  return o;
}

// wxyz quaternion (normalised here) -> rotation matrix.
__device__ __forceinline__ M3 quat_to_rotmat(float w, float x, float y, float z) {
  float inv = rsqrtf(w * w + x * x + y * y + z * z);
  w *= inv; x *= inv; y *= inv; z *= inv;
  M3 R;
  R(0, 0) = 1.f - 2.f * (y * y + z * z); R(0, 1) = 2.f * (x * y - w * z); R(0, 2) = 2.f * (x * z + w * y);
  R(1, 0) = 2.f * (x * y + w * z); R(1, 1) = 1.f - 2.f * (x * x + z * z); R(1, 2) = 2.f * (y * z - w * x);
  R(2, 0) = 2.f * (x * z - w * y); R(2, 1) = 2.f * (y * z + w * x); R(2, 2) = 1.f - 2.f * (x * x + y * y);
  return R;
}

// Sigma = (Rq S)(Rq S)^T
__device__ __forceinline__ M3 quat_scale_to_covar(const float q[4], const float s[3]) {
  M3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
  M3 M;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M(i, j) = R(i, j) * s[j];
  return mul_bt(M, M);
}

struct Cam {  // wave-uniform camera constants (scalar registers)
  M3 R;
  float t[3];
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ V, const float* __restrict__ K) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c.R(i, j) = V[i * 4 + j];
    c.t[i] = V[i * 4 + 3];
  }
  c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
  return c;
}

// Per-Gaussian records Q0/Q1/Q2: three arrays of float4.  (Measured alternatives, profiles/r02_backward_ablation.txt:
// one interleaved array of 64-byte rows made the random-order forward 10 % faster and projection / binning slower,
// no net gain; a workgroup -> tile map that gives each XCD one contiguous span of tiles changed nothing.)
#define GSL_Q(arr, g) (arr)[(g)]
// Pixel groups of an 8x8 quadrant in the compositing backward (raster_g16.hip): 4 = the four 4x4 blocks (one DPP row
// of 16 lanes each), 8 = the eight 4x2 half blocks (8 lanes each).  The forward's hit list carries one bit per group
// above the entry's list index: (group bits) << GSL_HIT_SHIFT | index.
#ifndef GSL_NG
#define GSL_NG 4
#endif
#define GSL_HIT_SHIFT (32 - GSL_NG)
#define GSL_HIT_INDEX_MASK ((1u << GSL_HIT_SHIFT) - 1u)
// Depths that reach the sort keys are positive, finite, normal floats: the window [near_plane, far_plane] of a projection
// is clamped to [FLT_MIN, FLT_MAX] on the host (a Gaussian at z <= 0 has no perspective projection anyway; the reference
// calls with near_plane = 0.01).  The per-tile sorts compare keys as doubles on that ground (sort_dev.h cswap).
#define GSL_CLAMP_DEPTH_WINDOW(nearp, farp)            \
  do {                                                 \
    if (!((nearp) >= 1.17549435e-38f)) (nearp) = 1.17549435e-38f; \
    if (!((farp) <= 3.40282347e+38f)) (farp) = 3.40282347e+38f;   \
  } while (0)
// Most Gaussians one call takes: the compositing backward addresses a Gaussian's 64-byte gradient row by a 32-bit byte
// offset (id << 6).  2^26 Gaussians are 4 GiB of rows; the largest configuration of BASELINE.json has 5 M.
#define GSL_MAX_GAUSSIANS (1 << 26)
// Workgroup -> work item.  Workgroups go to the eight XCDs round-robin (workgroup b runs on XCD b % 8), and every XCD has
// its own 4 MiB L2.  With GSL_XCD_SPANS the items (tiles, in raster order) are dealt so that XCD x gets ONE contiguous
// span of them: with the Gaussians stored in tile order (context.py:_choose_placement) the records an XCD gathers are
// then one eighth of the frame's instead of all of them.  (Round 2 tried the same map on randomly ordered records: no
// locality to keep, no effect.)
#ifndef GSL_XCD_SPANS
#define GSL_XCD_SPANS 0
#endif
__device__ __forceinline__ int xcd_span_item(int b, int n) {
#if GSL_XCD_SPANS
  const int x = b & 7, k = b >> 3, q = n >> 3, r = n & 7;
  return x * q + min(x, r) + k;
#else
  (void)n;
  return b;
#endif
}
#define GSL_TILE_OF_BLOCK() gsl::xcd_span_item((int)blockIdx.x, (int)gridDim.x)

// ---- fp16-staged records (workload X, "fp16 compositing": BASELINE.json configs[4]) --------------------
// One 32-byte record per Gaussian for the compositing kernels instead of three 16-byte ones: the centre stays float32
// (a 1920-px coordinate needs it), conic, cull radius, depth feature, opacity and colour are halves.  Everything the
// compositing loops accumulate (T, colour sums, gradient sums) stays float32: the records are widened when they are
// staged in LDS, so the loops themselves are the float32 ones.
//   dw0 x   dw1 y   dw2 (conic a | b)   dw3 (conic c | r_cull, rounded up)   dw4 (depth | opacity)   dw5 (r | g)   dw6 (b | 0)
__device__ __forceinline__ unsigned pack2h(float a, float b) {
  _Float16 ha = (_Float16)a, hb = (_Float16)b;
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ float h_lo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xFFFFu)); }
__device__ __forceinline__ float h_hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }

__device__ __forceinline__ void store_half_record(uint4* __restrict__ Qh, size_t i, float4 q0, float4 q1, float4 q2) {
  float rc = (q1.w > 0.f && q1.w < 6.0e4f) ? q1.w * 1.002f + 0.01f : q1.w;  // conservative radius: never rounded down
  Qh[2 * i] = make_uint4(__float_as_uint(q0.x), __float_as_uint(q0.y), pack2h(q1.x, q1.y), pack2h(q1.z, rc));
  Qh[2 * i + 1] = make_uint4(pack2h(q0.z, q0.w), pack2h(q2.x, q2.y), pack2h(q2.z, 0.f), 0u);
}

// Record g for the compositing kernels: the float32 arrays, or the fp16-staged array when Qh is given.
__device__ __forceinline__ void load_record(const float4* __restrict__ Q0, const float4* __restrict__ Q1,
                                            const float4* __restrict__ Q2, const uint4* __restrict__ Qh, int g,
                                            bool want_rgb, float4& r0, float4& r1, float4& r2) {
  if (Qh) {
    uint4 lo = Qh[2 * (size_t)g], hi = Qh[2 * (size_t)g + 1];
    r0 = make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), h_lo(hi.x), h_hi(hi.x));
    r1 = make_float4(h_lo(lo.z), h_hi(lo.z), h_lo(lo.w), h_hi(lo.w));
    r2 = make_float4(h_lo(hi.y), h_hi(hi.y), h_lo(hi.z), 0.f);
  } else {
    r0 = GSL_Q(Q0, g);
    r1 = GSL_Q(Q1, g);
    if (want_rgb) r2 = GSL_Q(Q2, g);
  }
}

// Tile-order placement of the Gaussians (RenderContext(reorder=True)): the caller stores its Gaussians
// sorted by the tile of their centre, once per frame, so that the record gathers of a tile's list fall on a few contiguous
// runs.  The list ORDER must not change with the placement -- depth ties break by Gaussian index -- so the
// low key word stays the Gaussian's ORIGINAL index (order_ids[storage slot], given to the projection) and the sorts translate
// it to the storage slot (storage_of[original index]) only when they write the list.  Both NULL: identity.
__device__ __forceinline__ int32_t list_id(const int32_t* __restrict__ storage_of, uint64_t key) {
  uint32_t g = (uint32_t)key;
  return storage_of ? storage_of[g] : (int32_t)g;
}

// DPP lane exchange (no LDS traffic).  Lanes a row_mask disables contribute 0.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}

// Sum over the 64 lanes of a wave in 6 DPP adds (quad swaps, row mirrors, row broadcasts);
// the total comes back wave-uniform (scalar register) from lane 63.  Fixed order => deterministic.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_get<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_get<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_get<0x141>(v);       // row_half_mirror
  v += dpp_get<0x140>(v);       // row_mirror: every lane holds its 16-lane row sum
  v += dpp_get<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
  v += dpp_get<0x143, 0xC>(v);  // row_bcast31 into rows 2,3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Inverse of the rotation block and the camera position -R^-1 t (what torch.inverse(viewmat)[:3,3] is).
__device__ __forceinline__ void cam_inverse(const Cam& cam, M3& Ri, float cp[3]) {
  const M3& R = cam.R;
  float c00 = R(1, 1) * R(2, 2) - R(1, 2) * R(2, 1);
  float c01 = R(1, 2) * R(2, 0) - R(1, 0) * R(2, 2);
  float c02 = R(1, 0) * R(2, 1) - R(1, 1) * R(2, 0);
  float det = R(0, 0) * c00 + R(0, 1) * c01 + R(0, 2) * c02;
  float id = 1.f / det;
  Ri(0, 0) = c00 * id; Ri(1, 0) = c01 * id; Ri(2, 0) = c02 * id;
  Ri(0, 1) = (R(0, 2) * R(2, 1) - R(0, 1) * R(2, 2)) * id;
  Ri(1, 1) = (R(0, 0) * R(2, 2) - R(0, 2) * R(2, 0)) * id;
  Ri(2, 1) = (R(0, 1) * R(2, 0) - R(0, 0) * R(2, 1)) * id;
  Ri(0, 2) = (R(0, 1) * R(1, 2) - R(0, 2) * R(1, 1)) * id;
  Ri(1, 2) = (R(0, 2) * R(1, 0) - R(0, 0) * R(1, 2)) * id;
  Ri(2, 2) = (R(0, 0) * R(1, 1) - R(0, 1) * R(1, 0)) * id;
#pragma unroll
  for (int k = 0; k < 3; ++k) cp[k] = -(Ri(k, 0) * cam.t[0] + Ri(k, 1) * cam.t[1] + Ri(k, 2) * cam.t[2]);
}

// Thread t < 16 of a workgroup: entry t of v_viewmat from the 15 summed rows [v_R 9 | v_t 3 | v_campos 3] (the SH view
// direction chained through the camera position, campos = -R^-1 t; row 3 = 0: that row is constant).
__device__ __forceinline__ float viewmat_from_totals(const float* tot, const float* __restrict__ V,
                                                     const float* __restrict__ Kmat) {
  float v = 0.f;
  if (threadIdx.x < 16) {
    int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    if (r < 3) {
      Cam cam = load_cam(V, Kmat);
      M3 Ri;
      float cp[3];
      cam_inverse(cam, Ri, cp);
      // w = R^-T v_campos
      float w = Ri(0, r) * tot[12] + Ri(1, r) * tot[13] + Ri(2, r) * tot[14];
      if (c < 3) v = tot[r * 3 + c] - w * cp[c];
      else v = tot[9 + r] - w;
    }
  }
  return v;
}

// The pose gradient leaves the projection backward as one row of 15 sums per workgroup
// ([v_R 9 | v_t 3 | v_campos 3], 16 floats apart).  Fixed-order sum of the rows by a 256-thread workgroup, chain of
// the SH view direction through the camera position (campos = -R^-1 t): thread t < 16 returns v_viewmat[t]
// (row 3 = 0: that row is constant).  Used by k_freduce_viewmat and, to save its launch, by the tracker's pose step.
__device__ __forceinline__ float reduce_viewmat_rows(const float* __restrict__ partials, int nb,
                                                     const float* __restrict__ V, const float* __restrict__ Kmat,
                                                     float (*red)[15], float* tot) {
  // thread (row r0 = tid >> 2, quarter q = tid & 3) adds quarter q of rows r0, r0 + 64, ...: 16-byte coalesced loads,
  // four rows in flight; then lanes of equal q are folded (xor 4 .. 32) and the four waves summed in wave order
  int q = threadIdx.x & 3;
  const float4* rows = reinterpret_cast<const float4*>(partials) + q;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int b = threadIdx.x >> 2;
  for (; b + 192 < nb; b += 256) {
    float4 x0 = rows[(size_t)b * 4], x1 = rows[(size_t)(b + 64) * 4], x2 = rows[(size_t)(b + 128) * 4],
           x3 = rows[(size_t)(b + 192) * 4];
    a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
    a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
    a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
    a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
  }
  for (; b < nb; b += 64) {
    float4 x0 = rows[(size_t)b * 4];
    a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
  }
  float v4[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                 (a0.w + a1.w) + (a2.w + a3.w)};
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = v4[c];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
    if (lane < 4 && 4 * q + c < 15) red[wv][4 * q + c] = x;
  }
  __syncthreads();
  if (threadIdx.x < 15) tot[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  __syncthreads();
  return viewmat_from_totals(tot, V, Kmat);
}

// The same reduction by a 1024-thread workgroup (k_freduce_viewmat: 3 907 rows at 1 M Gaussians; with 256 threads the
// launch was 9.5 us of dependent L2 round trips, 15 rounds of 4 loads in flight per thread).  Its own fixed order:
// thread (row mod 256, quarter), four rows in flight, lanes of equal quarter folded, the 16 waves summed in wave order.
__device__ __forceinline__ float reduce_viewmat_rows_wide(const float* __restrict__ partials, int nb,
                                                          const float* __restrict__ V, const float* __restrict__ Kmat,
                                                          float (*red)[15], float* tot) {
  int q = threadIdx.x & 3;
  const float4* rows = reinterpret_cast<const float4*>(partials) + q;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int b = threadIdx.x >> 2;
  for (; b + 768 < nb; b += 1024) {
    float4 x0 = rows[(size_t)b * 4], x1 = rows[(size_t)(b + 256) * 4], x2 = rows[(size_t)(b + 512) * 4],
           x3 = rows[(size_t)(b + 768) * 4];
    a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
    a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
    a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
    a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
  }
  for (; b < nb; b += 256) {
    float4 x0 = rows[(size_t)b * 4];
    a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
  }
  float v4[4] = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                 (a0.w + a1.w) + (a2.w + a3.w)};
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = v4[c];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) x += __shfl_xor(x, o, 64);
    if (lane < 4 && 4 * q + c < 15) red[wv][4 * q + c] = x;
  }
  __syncthreads();
  if (threadIdx.x < 15) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][threadIdx.x];
    tot[threadIdx.x] = s;
  }
  __syncthreads();
  return viewmat_from_totals(tot, V, Kmat);
}

// ---- tiny-splat backward, pass 2 (csrc/raster_px.hip has pass 1 and the story) ---------------------------------------
__device__ __forceinline__ int tiny_origin(float centre, float r) {  // first pixel index within r of centre
  return (int)ceilf(centre - r - 0.5f);
}

// Lane r (0..3) of a quad folds row r of Gaussian gid's 4x4 slab of (w, alpha*T) records into the gradient row
// [v_xy 2 | v_conic 3 | v_opacity 1 | v_colour D] (dx, dy rebuilt from the Gaussian's own record), clears the slab row,
// and the quad's four partial rows are added up: every lane of the quad returns the Gaussian's total.
// Split in two (round 4): tiny_slab_load issues every load that does not depend on another one -- radius, the slab row,
// both records -- and tiny_slab_fold works on them; the caller loads two items, consumes all loads at once (empty asm) and
// folds.  Written as one function called four times in a rolled loop, a thread went through radius -> slab -> records ->
// upstream pixels four times in a row: sixteen dependent memory round trips, half of the projection backward's 16 us in a
// tracker iteration at 102 k Gaussians.
struct TinySlabIn {
  int rad;
  float4 lo, hi, q0, qc;
};
__device__ __forceinline__ TinySlabIn tiny_slab_load(const int32_t* __restrict__ radii, const float4* __restrict__ Q0,
                                                     const float4* __restrict__ Q1, const float4* __restrict__ trec,
                                                     int gid, int r, int N) {
  TinySlabIn in;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  in.rad = 0; in.lo = z; in.hi = z; in.q0 = z; in.qc = z;
  if (gid < N) {
    const float4* row = trec + (size_t)gid * 8 + 2 * r;  // slab = 16 float2 = 8 float4; row r = float4 2r, 2r+1
    in.rad = radii[gid];
    in.lo = row[0];
    in.hi = row[1];
    in.q0 = Q0[gid];
    in.qc = Q1[gid];
  }
  return in;
}
#define GSL_TINY_SLAB_PIN(in)                                                                                         \
  asm volatile("" : "+v"((in).rad), "+v"((in).lo.x), "+v"((in).lo.y), "+v"((in).lo.z), "+v"((in).lo.w), "+v"((in).hi.x), \
               "+v"((in).hi.y), "+v"((in).hi.z), "+v"((in).hi.w), "+v"((in).q0.x), "+v"((in).q0.y), "+v"((in).q0.z),    \
               "+v"((in).q0.w), "+v"((in).qc.x), "+v"((in).qc.y), "+v"((in).qc.z), "+v"((in).qc.w)                       \
               :                                                                                                      \
               : "memory")
template <int D>
__device__ __forceinline__ void tiny_slab_fold(const TinySlabIn& in, int W, int H, float4* __restrict__ trec,
                                               const float* __restrict__ vcT, int gid, int r, float (&v)[6 + D]) {
  constexpr int A = 6 + D;
#pragma unroll
  for (int k = 0; k < A; ++k) v[k] = 0.f;
  if (in.rad > 0) {
    float4* row = trec + (size_t)gid * 8 + 2 * r;
    const float4 lo = in.lo, hi = in.hi;
    float w[4] = {lo.x, lo.z, hi.x, hi.z}, f[4] = {lo.y, lo.w, hi.y, hi.w};
    bool any = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) any = any || (w[c] != 0.f) || (f[c] != 0.f);
    if (any) {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      row[0] = z;
      row[1] = z;
      const float4 q0 = in.q0, qc = in.qc;
      int pcol0 = tiny_origin(q0.x, qc.w), prow = tiny_origin(q0.y, qc.w) + r;
      float dy = q0.y - ((float)prow + 0.5f);
      bool row_in = (unsigned)prow < (unsigned)H;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (w[c] != 0.f || f[c] != 0.f) {
          int pcol = pcol0 + c;
          float dx = q0.x - ((float)pcol + 0.5f);
          float gx = qc.x * dx + qc.y * dy, gy = qc.y * dx + qc.z * dy;
          float v_sigma = -q0.w * w[c], hs = 0.5f * v_sigma;
          v[0] += v_sigma * gx; v[1] += v_sigma * gy;
          v[2] += hs * dx * dx; v[3] += v_sigma * dx * dy; v[4] += hs * dy * dy;
          v[5] += w[c];
          if (f[c] != 0.f && row_in && (unsigned)pcol < (unsigned)W) {
            size_t pid = (size_t)prow * W + pcol;
#pragma unroll
            for (int k = 0; k < D; ++k) v[6 + k] += f[c] * vcT[pid * D + k];
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < A; ++k) {
    float x = v[k];
    x += dpp_get<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp_get<0x4E>(x);  // quad_perm [2,3,0,1]: every lane of the quad holds the Gaussian's total
    v[k] = x;
  }
}

// ---- long tile lists split over workgroups (raster_px.hip: forward; raster_g16.hip: backward) -----------------------------
// GSL_SEG: entries per compositing segment (the walk of a segment is serial per pixel, so its length is the latency of
// a pile frame: 512 -> 256 -> 128 took the pile frame's compositing from 0.50 to 0.37 to ... ms); GSL_SORT_SEG: keys per
// sorted run of the long-list sort (a multiple of GSL_SEG: 8 keys per lane in registers).
#ifndef GSL_SEG_LOG2
#define GSL_SEG_LOG2 7
#endif
#define GSL_SEG (1 << GSL_SEG_LOG2)
#define GSL_SORT_SEG_LOG2 9
#define GSL_SORT_SEG (1 << GSL_SORT_SEG_LOG2)

struct LongWs {  // views into the caller's long_ws (sized by gsl_long_ws_bytes)
  int32_t* n_seg;     // [4]: segments of this frame, max_seg overflow flag, -, -
  int32_t* seg_tile;  // [max_seg]
  int32_t* seg_idx;   // [max_seg]  segment number inside its tile
  int32_t* seg_cnt;   // [max_seg]  segments of that tile
  int32_t* seg_qcnt;  // [max_seg][4] length of the segment's hit list per quadrant (written by the forward's pass B)
  float* P;           // [max_seg][256]
  float* Tend;        // [max_seg][256]
  int32_t* last;      // [max_seg][256]
  float* C;           // [max_seg][256][4]
};
__host__ __device__ __forceinline__ LongWs long_ws_views(void* ws, int max_seg) {
  LongWs w;
  char* p = (char*)ws;
  w.n_seg = (int32_t*)p; p += 16;
  w.seg_tile = (int32_t*)p; p += (size_t)max_seg * 4;
  w.seg_idx = (int32_t*)p; p += (size_t)max_seg * 4;
  w.seg_cnt = (int32_t*)p; p += (size_t)max_seg * 4;
  w.seg_qcnt = (int32_t*)p; p += (size_t)max_seg * 16;
  p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
  w.P = (float*)p; p += (size_t)max_seg * 256 * 4;
  w.Tend = (float*)p; p += (size_t)max_seg * 256 * 4;
  w.last = (int32_t*)p; p += (size_t)max_seg * 256 * 4;
  w.C = (float*)p;
  return w;
}

// One workgroup: the (tile, segment) pairs of every tile of the strip whose list is longer than long_min.
static __global__ __launch_bounds__(1024) void k_long_map(const int32_t* __restrict__ tile_offsets, int tile_begin, int n_strip_tiles,
                                                   long long capacity, int long_min, int max_seg, int max_list, LongWs w) {
  __shared__ int wsum[16];
  __shared__ int carry_s, n_long_s;
  __shared__ int lt_tile[32], lt_first[32], lt_nseg[32];
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    carry_s = 0;
    n_long_s = 0;
  }
  __syncthreads();
  for (int base = 0; base < n_strip_tiles; base += 1024) {
    int q = base + tid;
    int nseg = 0, tile = tile_begin + q;
    if (q < n_strip_tiles) {
      long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
      if (re > capacity) re = capacity;
      long long len = re - rs;
      if (len > long_min) nseg = (int)((len + GSL_SEG - 1) / GSL_SEG);
      if (max_list > 0 && len > long_min && len > max_list) w.n_seg[2] = (int)len;  // longer than the merge passes cover
    }
    int x = nseg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    int first = carry_s + woff + x - nseg;
    // A tile is mapped with ALL its segments or not at all: the combine and the backward index a tile's segments
    // first .. first + nseg - 1, which must exist.  Slots of a tile that does not fit get tile -1 (every kernel that walks
    // the map returns on it); n_seg[1] then tells the host how many segments the frame needed.
    const bool fits = first + nseg <= max_seg;
    // the long tiles of this round (a handful at most) are noted in LDS and their slots written by all threads together
    // (one thread writing its tile's 180 slots alone was 7 us of a pile frame)
    if (nseg > 0) {
      int k = atomicAdd(&n_long_s, 1);
      if (k < 32) {
        lt_tile[k] = fits ? tile : -1;
        lt_first[k] = first;
        lt_nseg[k] = nseg;
      } else {  // (more long tiles than notes: this thread writes its own)
        for (int sgm = 0; sgm < nseg; ++sgm) {
          int g = first + sgm;
          if (g < max_seg) {
            w.seg_tile[g] = fits ? tile : -1;
            w.seg_idx[g] = fits ? sgm : -1;
            w.seg_cnt[g] = fits ? nseg : 0;
          }
        }
      }
    }
    __syncthreads();
    int nl = min(n_long_s, 32);
    for (int k = 0; k < nl; ++k) {
      int t = lt_tile[k], f = lt_first[k], c = lt_nseg[k];
      for (int sgm = tid; sgm < c; sgm += 1024) {
        int g = f + sgm;
        if (g < max_seg) {
          w.seg_tile[g] = t;
          w.seg_idx[g] = t >= 0 ? sgm : -1;
          w.seg_cnt[g] = t >= 0 ? c : 0;
        }
      }
    }
    __syncthreads();
    if (tid == 1023) {
      carry_s = carry_s + woff + x;
      n_long_s = 0;
    }
    __syncthreads();
  }
  if (tid == 0) {
    w.n_seg[0] = min(carry_s, max_seg);
    if (carry_s > max_seg) w.n_seg[1] = carry_s;  // sticky: more segments than the workspace holds (host polls)
  }
}

// Tile rectangle of a projected Gaussian: [xmin,xmax) x [ymin,ymax) in tiles.
__device__ __forceinline__ void tile_rect(float mx, float my, int radius, int tile_size, int tile_w,
                                          int tile_h, int& xmin, int& ymin, int& xmax, int& ymax) {
  float ts = (float)tile_size;
  float tr = (float)radius / ts;
  float tx = mx / ts, ty = my / ts;
  // clamp in float first: (uint32_t)floor(negative) saturates to 0 in the reference kernel
  xmin = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)tile_w);
  ymin = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)tile_h);
  xmax = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)tile_w);
  ymax = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)tile_h);
}

}  // namespace gsl
