// Front-to-back alpha compositing of depth-sorted tile lists and its backward replay.
// Replaces gsplat.rasterize_to_pixels fwd/bwd (IDX:14378, IDX:14279).
//
// MI355X mapping: one 256-thread workgroup per 16x16 tile = 4 wave64; each wave owns an
// 8x8 pixel quadrant.  The tile's sorted splat list is staged through LDS in batches of
// 256 (one splat per thread, gathered by index from the per-Gaussian arrays).  Pose-tracking
// splats are tiny (radius 2-4 px) so most splats of a tile miss a given quadrant: each wave
// first tests the staged batch against its quadrant (64 splats per ballot, conservative
// bounding box of the alpha >= 1/255 ellipse) and then walks only the set bits.  Skipped
// splats are exactly those the reference loop would `continue` over, so results are
// unchanged.  Termination (T <= 1e-4) is tracked per lane and voted per wave and per block.
#include "gsloc_common.h"

namespace gsl {

template <int D>
struct Batch {
  float x[256], y[256], ca[256], cb[256], cc[256], op[256];
  float hx[256], hy[256];  // conservative half extents of the alpha >= 1/255 region
  int32_t id[256];
  float col[256 * D];
};

// Conservative half extents: sigma <= tau  <=>  |dx| <= sqrt(2 tau Sxx), Sxx = cc/det(conic).
__device__ __forceinline__ void cull_extent(float ca, float cb, float cc, float op, float& hx, float& hy) {
  float tau = __logf(255.f * op) * 1.01f + 0.01f;
  float det = ca * cc - cb * cb;
  if (!(tau > 0.f) || !(det > 0.f) || !(ca > 0.f) || !(cc > 0.f)) {
    // never contributes (op < 1/255) or degenerate conic: do not cull on geometry, let the
    // per-pixel test decide (keeps reference behaviour for sigma < 0 conics)
    bool dead = !(tau > 0.f);
    hx = dead ? -1.f : 1e30f;
    hy = hx;
    return;
  }
  float inv = 2.f * tau / det;
  hx = sqrtf(inv * cc) * 1.0001f + 1e-3f;
  hy = sqrtf(inv * ca) * 1.0001f + 1e-3f;
}

template <int D>
__global__ __launch_bounds__(256) void k_raster_fwd(
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds, int W, int H, int tile_w, int ty0,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids, long long capacity,
    float* __restrict__ render_colors, float* __restrict__ render_alphas, int32_t* __restrict__ last_ids) {
  __shared__ Batch<D> sb;
  int tile = ty0 * tile_w + blockIdx.x;
  int tyi = tile / tile_w, txi = tile - tyi * tile_w;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int qx = txi * 16 + (wv & 1) * 8, qy = tyi * 16 + (wv >> 1) * 8;  // quadrant origin
  int j = qx + (lane & 7), i = qy + (lane >> 3);
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W);
  bool done = !inside;
  // quadrant bounds in pixel-centre coordinates
  float qx0 = (float)qx + 0.5f, qx1 = (float)qx + 7.5f, qy0 = (float)qy + 0.5f, qy1 = (float)qy + 7.5f;

  long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
  if (re > capacity) re = capacity;
  if (rs > re) rs = re;
  int nb = (int)((re - rs + 255) / 256);

  float T = 1.f;
  int cur_idx = 0;
  float pix[D];
#pragma unroll
  for (int k = 0; k < D; ++k) pix[k] = 0.f;

  for (int b = 0; b < nb; ++b) {
    if (__syncthreads_and(done)) break;
    long long bstart = rs + (long long)b * 256;
    int bsize = (int)min((long long)256, re - bstart);
    if (tid < bsize) {
      int g = flatten_ids[bstart + tid];
      sb.id[tid] = g;
      float x = means2d[2 * (size_t)g], y = means2d[2 * (size_t)g + 1];
      float ca = conics[3 * (size_t)g], cb = conics[3 * (size_t)g + 1], cc = conics[3 * (size_t)g + 2];
      float op = opacities[g];
      sb.x[tid] = x; sb.y[tid] = y; sb.ca[tid] = ca; sb.cb[tid] = cb; sb.cc[tid] = cc; sb.op[tid] = op;
      float hx, hy;
      cull_extent(ca, cb, cc, op, hx, hy);
      sb.hx[tid] = hx; sb.hy[tid] = hy;
#pragma unroll
      for (int k = 0; k < D; ++k) sb.col[tid * D + k] = colors[(size_t)g * D + k];
    }
    __syncthreads();
    bool wdone = __all(done);
    for (int c = 0; c < bsize && !wdone; c += 64) {
      int e = c + lane;
      bool hit = false;
      if (e < bsize) {
        float x = sb.x[e], y = sb.y[e], hx = sb.hx[e], hy = sb.hy[e];
        hit = (x + hx >= qx0) && (x - hx <= qx1) && (y + hy >= qy0) && (y - hy <= qy1);
      }
      unsigned long long m = __ballot(hit);
      while (m) {
        int t = c + (__ffsll((long long)m) - 1);
        m &= m - 1;
        float dx = sb.x[t] - px, dy = sb.y[t] - py;
        float ca = sb.ca[t], cb = sb.cb[t], cc = sb.cc[t];
        float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
        float alpha = fminf(GSL_ALPHA_MAX, sb.op[t] * __expf(-sigma));
        bool ok = !done && !(sigma < 0.f || alpha < GSL_ALPHA_MIN);
        if (ok) {
          float nT = T * (1.f - alpha);
          if (nT <= GSL_T_STOP) {
            done = true;
          } else {
            float vis = alpha * T;
#pragma unroll
            for (int k = 0; k < D; ++k) pix[k] += sb.col[t * D + k] * vis;
            cur_idx = (int)(bstart - 0) + t;  // absolute index into flatten_ids
            T = nT;
          }
        }
        if (__all(done)) { wdone = true; break; }
      }
    }
  }
  if (inside) {
    size_t pid = (size_t)i * W + j;
    render_alphas[pid] = 1.f - T;
#pragma unroll
    for (int k = 0; k < D; ++k)
      render_colors[pid * D + k] = backgrounds ? (pix[k] + T * backgrounds[k]) : pix[k];
    last_ids[pid] = cur_idx;
  }
}

// Per-Gaussian gradient accumulator row ("vacc"): [v_xy 2][v_conic 3][v_opacity 1][v_color D], padded
// to 16 floats (64 B, one cache line / one atomic request) for D <= 10, 48 floats otherwise.
__host__ __device__ constexpr int vacc_stride(int D) { return (6 + D <= 16) ? 16 : 48; }

template <int D>
struct BatchB {
  static constexpr int A = 6 + D;
  static constexpr int AP = A | 1;                  // odd row pitch: conflict-free LDS access
  static constexpr int NACC = (D <= 5) ? 4 : 1;     // one accumulator per wave while LDS allows
  float x[256], y[256], ca[256], cb[256], cc[256], op[256];
  float hx[256], hy[256];
  int32_t id[256];
  float col[256 * D];
  float acc[NACC][256 * AP];
  uint16_t list[4][64];
};

template <int D>
__global__ __launch_bounds__(256) void k_raster_bwd(
    const float* __restrict__ means2d, const float* __restrict__ conics, const float* __restrict__ colors,
    const float* __restrict__ opacities, const float* __restrict__ backgrounds, int W, int H, int tile_w, int ty0,
    const int32_t* __restrict__ tile_offsets, const int32_t* __restrict__ flatten_ids, long long capacity,
    const float* __restrict__ render_alphas, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_render_colors, const float* __restrict__ v_render_alphas,
    float* __restrict__ vacc) {
  constexpr int A = BatchB<D>::A;
  constexpr int AP = BatchB<D>::AP;
  constexpr int NACC = BatchB<D>::NACC;
  constexpr int AS = vacc_stride(D);
  __shared__ BatchB<D> sb;
  int tile = ty0 * tile_w + blockIdx.x;
  int tyi = tile / tile_w, txi = tile - tyi * tile_w;
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int qx = txi * 16 + (wv & 1) * 8, qy = tyi * 16 + (wv >> 1) * 8;
  int j = qx + (lane & 7), i = qy + (lane >> 3);
  float px = (float)j + 0.5f, py = (float)i + 0.5f;
  bool inside = (i < H) && (j < W);
  float qx0 = (float)qx + 0.5f, qx1 = (float)qx + 7.5f, qy0 = (float)qy + 0.5f, qy1 = (float)qy + 7.5f;

  long long rs = tile_offsets[tile], re = tile_offsets[tile + 1];
  if (re > capacity) re = capacity;
  if (rs >= re) return;

  size_t pid = inside ? ((size_t)i * W + j) : 0;
  float T_final = inside ? (1.f - render_alphas[pid]) : 1.f;
  float T = T_final;
  int bin_final = inside ? last_ids[pid] : -1;
  float vc[D], buf[D];
  float va = inside ? v_render_alphas[pid] : 0.f;
  float bg_dot = 0.f;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    vc[k] = inside ? v_render_colors[pid * D + k] : 0.f;
    buf[k] = 0.f;
    if (backgrounds) bg_dot += backgrounds[k] * vc[k];
  }
  // the deepest splat any pixel of the quadrant / tile composited bounds the work
  int wave_final = bin_final;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) wave_final = max(wave_final, __shfl_xor(wave_final, o, 64));
  __shared__ int s_block_final[4];
  if (lane == 0) s_block_final[wv] = wave_final;
  __syncthreads();
  int block_final = max(max(s_block_final[0], s_block_final[1]), max(s_block_final[2], s_block_final[3]));
  // nothing behind block_final was composited by any pixel: start there
  if ((long long)block_final + 1 < re) re = max((long long)block_final + 1, rs);
  if (rs >= re) return;
  int nb = (int)((re - rs + 255) / 256);

  for (int b = 0; b < nb; ++b) {
    // batch b covers absolute indices [bend-bsize+1, bend], staged back to front: slot t <-> bend - t
    long long bend = re - 1 - (long long)b * 256;
    int bsize = (int)min((long long)256, bend + 1 - rs);
    __syncthreads();
    if (tid < bsize) {
      int g = flatten_ids[bend - tid];
      sb.id[tid] = g;
      float x = means2d[2 * (size_t)g], y = means2d[2 * (size_t)g + 1];
      float ca = conics[3 * (size_t)g], cb = conics[3 * (size_t)g + 1], cc = conics[3 * (size_t)g + 2];
      float op = opacities[g];
      sb.x[tid] = x; sb.y[tid] = y; sb.ca[tid] = ca; sb.cb[tid] = cb; sb.cc[tid] = cc; sb.op[tid] = op;
      float hx, hy;
      cull_extent(ca, cb, cc, op, hx, hy);
      sb.hx[tid] = hx; sb.hy[tid] = hy;
#pragma unroll
      for (int k = 0; k < D; ++k) sb.col[tid * D + k] = colors[(size_t)g * D + k];
    }
#pragma unroll
    for (int w = 0; w < NACC; ++w)
#pragma unroll
      for (int k = 0; k < A; ++k) sb.acc[w][tid * AP + k] = 0.f;
    __syncthreads();
    int t_first = (int)max((long long)0, bend - (long long)wave_final);
    for (int c = (t_first / 64) * 64; c < bsize; c += 64) {
      int e = c + lane;
      bool hit = false;
      if (e < bsize && e >= t_first) {
        float x = sb.x[e], y = sb.y[e], hx = sb.hx[e], hy = sb.hy[e];
        hit = (x + hx >= qx0) && (x - hx <= qx1) && (y + hy >= qy0) && (y - hy <= qy1);
      }
      unsigned long long m = __ballot(hit);
      while (m) {
        int t = c + (__ffsll((long long)m) - 1);
        m &= m - 1;
        bool valid = inside && ((bend - t) <= (long long)bin_final);
        float dx = 0.f, dy = 0.f, ca = 0.f, cb = 0.f, cc = 0.f, op = 0.f, vis = 0.f, alpha = 0.f;
        if (valid) {
          dx = sb.x[t] - px; dy = sb.y[t] - py;
          ca = sb.ca[t]; cb = sb.cb[t]; cc = sb.cc[t]; op = sb.op[t];
          float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
          vis = __expf(-sigma);
          alpha = fminf(GSL_ALPHA_MAX, op * vis);
          if (sigma < 0.f || alpha < GSL_ALPHA_MIN) valid = false;
        }
        if (!__any(valid)) continue;
        float g_xy0 = 0.f, g_xy1 = 0.f, g_c0 = 0.f, g_c1 = 0.f, g_c2 = 0.f, g_op = 0.f;
        float g_col[D];
#pragma unroll
        for (int k = 0; k < D; ++k) g_col[k] = 0.f;
        if (valid) {
          float ra = 1.f / (1.f - alpha);
          T *= ra;
          float fac = alpha * T;
          float v_alpha = 0.f;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            float ck = sb.col[t * D + k];
            g_col[k] = fac * vc[k];
            v_alpha += (ck * T - buf[k] * ra) * vc[k];
            buf[k] += ck * fac;
          }
          v_alpha += T_final * ra * va;
          if (backgrounds) v_alpha += -T_final * ra * bg_dot;
          if (op * vis <= GSL_ALPHA_MAX) {
            float v_sigma = -op * vis * v_alpha;
            g_c0 = 0.5f * v_sigma * dx * dx;
            g_c1 = v_sigma * dx * dy;
            g_c2 = 0.5f * v_sigma * dy * dy;
            g_xy0 = v_sigma * (ca * dx + cb * dy);
            g_xy1 = v_sigma * (cb * dx + cc * dy);
            g_op = vis * v_alpha;
          }
        }
        g_xy0 = wave_sum(g_xy0); g_xy1 = wave_sum(g_xy1);
        g_c0 = wave_sum(g_c0); g_c1 = wave_sum(g_c1); g_c2 = wave_sum(g_c2);
        g_op = wave_sum(g_op);
#pragma unroll
        for (int k = 0; k < D; ++k) g_col[k] = wave_sum(g_col[k]);
        if (lane == 0) {
          if (NACC == 4) {
            float* a = sb.acc[wv] + t * AP;
            a[0] = g_xy0; a[1] = g_xy1; a[2] = g_c0; a[3] = g_c1; a[4] = g_c2; a[5] = g_op;
#pragma unroll
            for (int k = 0; k < D; ++k) a[6 + k] = g_col[k];
          } else {
            float* a = sb.acc[0] + t * AP;
            atomicAdd(&a[0], g_xy0); atomicAdd(&a[1], g_xy1);
            atomicAdd(&a[2], g_c0); atomicAdd(&a[3], g_c1); atomicAdd(&a[4], g_c2);
            atomicAdd(&a[5], g_op);
#pragma unroll
            for (int k = 0; k < D; ++k) atomicAdd(&a[6 + k], g_col[k]);
          }
        }
      }
    }
    __syncthreads();
    // Flush.  Thread t folds the per-wave partials of slot t in a fixed order (deterministic per
    // tile), then each wave packs its non-zero slots so that 16 consecutive lanes add one
    // Gaussian's 64-byte accumulator row: one atomic request per (tile, splat) pair.
    {
      bool nz = false;
      if (tid < bsize) {
#pragma unroll
        for (int k = 0; k < A; ++k) {
          float v = sb.acc[0][tid * AP + k];
#pragma unroll
          for (int w = 1; w < NACC; ++w) v += sb.acc[w][tid * AP + k];
          if (NACC > 1) sb.acc[0][tid * AP + k] = v;
          nz = nz || (v != 0.f);
        }
      }
      unsigned long long mask = __ballot(nz);
      int cnt = __popcll(mask);
      if (nz) sb.list[wv][__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)tid;
      __syncthreads();
      constexpr int LPG = (A <= 16) ? 16 : 64;  // lanes per accumulator row
      constexpr int GPW = 64 / LPG;
      int f = lane % LPG;
      for (int i0 = 0; i0 < cnt; i0 += GPW) {
        int gi = i0 + lane / LPG;
        if (gi < cnt && f < A) {
          int sl = sb.list[wv][gi];
          size_t g = (size_t)sb.id[sl];
          atomicAdd(&vacc[g * AS + f], sb.acc[0][sl * AP + f]);
        }
      }
    }
  }
}

// vacc rows -> the four gsplat-style gradient tensors.
__global__ __launch_bounds__(256) void k_vacc_unpack(const float* __restrict__ vacc, int n, int D, int AS,
                                                     float* __restrict__ v_means2d, float* __restrict__ v_conics,
                                                     float* __restrict__ v_colors, float* __restrict__ v_opacities) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* r = vacc + (size_t)i * AS;
  v_means2d[2 * (size_t)i] = r[0];
  v_means2d[2 * (size_t)i + 1] = r[1];
  v_conics[3 * (size_t)i] = r[2];
  v_conics[3 * (size_t)i + 1] = r[3];
  v_conics[3 * (size_t)i + 2] = r[4];
  v_opacities[i] = r[5];
  for (int k = 0; k < D; ++k) v_colors[(size_t)i * D + k] = r[6 + k];
}

template <int D>
static int launch_fwd(const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, int W, int H, int tile_w, int ty0, int ty1,
                      const int32_t* tile_offsets, const int32_t* flatten_ids, long long capacity,
                      float* render_colors, float* render_alphas, int32_t* last_ids, hipStream_t st) {
  int nblk = (ty1 - ty0) * tile_w;
  hipLaunchKernelGGL(k_raster_fwd<D>, dim3(nblk), dim3(256), 0, st, means2d, conics, colors, opacities, backgrounds,
                     W, H, tile_w, ty0, tile_offsets, flatten_ids, capacity, render_colors, render_alphas, last_ids);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

template <int D>
static int launch_bwd(const float* means2d, const float* conics, const float* colors, const float* opacities,
                      const float* backgrounds, int W, int H, int tile_w, int ty0, int ty1,
                      const int32_t* tile_offsets, const int32_t* flatten_ids, long long capacity,
                      const float* render_alphas, const int32_t* last_ids, const float* v_render_colors,
                      const float* v_render_alphas, float* vacc, hipStream_t st) {
  int nblk = (ty1 - ty0) * tile_w;
  hipLaunchKernelGGL(k_raster_bwd<D>, dim3(nblk), dim3(256), 0, st, means2d, conics, colors, opacities, backgrounds,
                     W, H, tile_w, ty0, tile_offsets, flatten_ids, capacity, render_alphas, last_ids,
                     v_render_colors, v_render_alphas, vacc);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

}  // namespace gsl

#define GSL_DISPATCH_D(D, CALL)          \
  switch (D) {                           \
    case 1: return CALL(1);              \
    case 2: return CALL(2);              \
    case 3: return CALL(3);              \
    case 4: return CALL(4);              \
    case 5: return CALL(5);              \
    case 8: return CALL(8);              \
    case 16: return CALL(16);            \
    case 32: return CALL(32);            \
    default: return GSL_ERR_BAD_ARG;     \
  }

extern "C" int gsl_rasterize_fwd(const float* means2d, const float* conics, const float* colors,
                                 const float* opacities, const float* backgrounds, int channels, int width,
                                 int height, int tile_size, int tile_w, int tile_h, int ty0, int ty1,
                                 const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                                 float* render_colors, float* render_alphas, int32_t* last_ids, void* stream) {
  if (tile_size != 16) return GSL_ERR_BAD_ARG;  // the wave64 quadrant mapping is built for 16x16 tiles
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0)
    return GSL_ERR_BAD_ARG;
  if (tile_w * 16 < width || tile_h * 16 < height) return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render_colors || !render_alphas || !last_ids) return GSL_ERR_BAD_ARG;
  if (capacity > 0 && (!means2d || !conics || !colors || !opacities || !flatten_ids)) return GSL_ERR_BAD_ARG;
  if (ty0 == ty1) return GSL_OK;
  hipStream_t st = (hipStream_t)stream;
#define CALL_FWD(DD)                                                                                             \
  gsl::launch_fwd<DD>(means2d, conics, colors, opacities, backgrounds, width, height, tile_w, ty0, ty1,         \
                      tile_offsets, flatten_ids, (long long)capacity, render_colors, render_alphas, last_ids, st)
  GSL_DISPATCH_D(channels, CALL_FWD)
#undef CALL_FWD
}

extern "C" size_t gsl_vacc_bytes(int n_gaussians, int channels) {
  return (size_t)(n_gaussians > 0 ? n_gaussians : 0) * gsl::vacc_stride(channels) * sizeof(float);
}

extern "C" int gsl_rasterize_bwd(const float* means2d, const float* conics, const float* colors,
                                 const float* opacities, const float* backgrounds, int channels, int width,
                                 int height, int tile_size, int tile_w, int tile_h, int ty0, int ty1,
                                 const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                                 const float* render_alphas, const int32_t* last_ids,
                                 const float* v_render_colors, const float* v_render_alphas, float* vacc,
                                 void* stream) {
  if (tile_size != 16) return GSL_ERR_BAD_ARG;
  if (width <= 0 || height <= 0 || tile_w <= 0 || tile_h <= 0 || ty0 < 0 || ty1 > tile_h || ty0 > ty1 ||
      capacity < 0)
    return GSL_ERR_BAD_ARG;
  if (tile_w * 16 < width || tile_h * 16 < height) return GSL_ERR_BAD_ARG;
  if (!tile_offsets || !render_alphas || !last_ids || !v_render_colors || !v_render_alphas) return GSL_ERR_BAD_ARG;
  if (capacity == 0 || ty0 == ty1) return GSL_OK;
  if (!means2d || !conics || !colors || !opacities || !flatten_ids || !vacc) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
#define CALL_BWD(DD)                                                                                             \
  gsl::launch_bwd<DD>(means2d, conics, colors, opacities, backgrounds, width, height, tile_w, ty0, ty1,         \
                      tile_offsets, flatten_ids, (long long)capacity, render_alphas, last_ids, v_render_colors,  \
                      v_render_alphas, vacc, st)
  GSL_DISPATCH_D(channels, CALL_BWD)
#undef CALL_BWD
}

extern "C" int gsl_vacc_unpack(const float* vacc, int n_gaussians, int channels, float* v_means2d, float* v_conics,
                               float* v_colors, float* v_opacities, void* stream) {
  if (n_gaussians < 0 || channels <= 0 || channels > 32) return GSL_ERR_BAD_ARG;
  if (n_gaussians == 0) return GSL_OK;
  if (!vacc || !v_means2d || !v_conics || !v_colors || !v_opacities) return GSL_ERR_BAD_ARG;
  hipLaunchKernelGGL(gsl::k_vacc_unpack, dim3((n_gaussians + 255) / 256), dim3(256), 0, (hipStream_t)stream, vacc,
                     n_gaussians, channels, gsl::vacc_stride(channels), v_means2d, v_conics, v_colors, v_opacities);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
