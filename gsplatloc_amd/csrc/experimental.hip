// Kernels written after round 1's GPU access had ended: they compile for gfx950 and are reachable through
// the C ABI, but they have NOT run on hardware yet and nothing selects them by default (RenderContext uses
// them only under an explicit environment switch).  Kept in their own translation unit so the objects of the
// measured kernels are untouched.
#include "gsloc_common.h"
#include "fproject_bwd.h"

namespace gsl {

// first pixel index whose centre lies within r of `centre` (same definition as raster_px.hip)
__device__ __forceinline__ int xp_tiny_origin(float centre, float r) { return (int)ceilf(centre - r - 0.5f); }

// Pass 2 of the tiny-splat backward with FOUR lanes per Gaussian instead of sixteen (k_tiny_gather):
// lane r of a quad owns slab row r -- four (w, alpha*T) records, 32 contiguous bytes, two 16-byte loads -- and
// accumulates the row's gradient terms in registers; two quad-permute DPP adds fold the four rows.  16 Gaussians
// per wave at about the instruction count k_tiny_gather spends on 4 (profiles/r01_isa_mix.txt: that kernel is
// instruction-bound, 216 instructions per wave).  Same inputs, same outputs, same slab clearing.
template <int D>
__global__ __launch_bounds__(256) void k_tiny_gather4(const float4* __restrict__ Q0, const float4* __restrict__ Q1,
                                                      const int32_t* __restrict__ radii, int N, int W, int H,
                                                      float4* __restrict__ trec, const float* __restrict__ vcT,
                                                      float4* __restrict__ vacc) {
  constexpr int A = 6 + D;
  int t = blockIdx.x * 256 + threadIdx.x;
  int gid = t >> 2, r = t & 3;
  bool live = gid < N && radii[gid] > 0;
  float v[A];
#pragma unroll
  for (int k = 0; k < A; ++k) v[k] = 0.f;
  if (live) {
    float4* row = trec + (size_t)gid * 8 + 2 * r;  // slab = 16 float2 = 8 float4; row r = float4 2r, 2r+1
    float4 lo = row[0], hi = row[1];
    float w[4] = {lo.x, lo.z, hi.x, hi.z}, f[4] = {lo.y, lo.w, hi.y, hi.w};
    bool any = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) any = any || (w[c] != 0.f) || (f[c] != 0.f);
    if (any) {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      row[0] = z;
      row[1] = z;
      float4 q0 = GSL_Q(Q0, gid), qc = GSL_Q(Q1, gid);
      int pcol0 = xp_tiny_origin(q0.x, qc.w), prow = xp_tiny_origin(q0.y, qc.w) + r;
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
  if (live) {
    float pad[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) pad[k] = (k < A) ? v[k] : 0.f;
    if (r == 0) vacc[4 * (size_t)gid] = make_float4(pad[0], pad[1], pad[2], pad[3]);
    if (r == 1) vacc[4 * (size_t)gid + 1] = make_float4(pad[4], pad[5], pad[6], pad[7]);
    if (r == 2) vacc[4 * (size_t)gid + 2] = make_float4(pad[8], pad[9], pad[10], pad[11]);
  }
}

// Pass 2 of the tiny-splat backward FUSED with the projection backward: phase A folds the slabs of the block's
// 256 Gaussians into shared memory (four lanes per Gaussian as in k_tiny_gather4, four rounds of 64 Gaussians),
// phase B is the per-Gaussian projection/colour vjp (fproject_bwd.h) reading its gradient row from there.  Saves
// the 64-byte row round trip through memory (write, read, clear) and one launch.
template <bool FULL, int D>
__global__ __launch_bounds__(256) void k_tiny_project_bwd(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ colors, int sh_degree, int K_sh,
    const float* __restrict__ V, const float* __restrict__ Kmat, int N, int W, int H, float eps2d, int antialiased,
    const int32_t* __restrict__ radii, const float4* __restrict__ Q0, const float4* __restrict__ Q1,
    const float* __restrict__ comps, float4* __restrict__ trec, const float* __restrict__ vcT,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_opacities, float* __restrict__ v_colors, float* __restrict__ partials) {
  constexpr int A = 6 + D;
  __shared__ float4 s_rows[256][3];
  int quad = threadIdx.x >> 2, r = threadIdx.x & 3;
#pragma unroll 1
  for (int round = 0; round < 4; ++round) {
    int local = round * 64 + quad, gid = blockIdx.x * 256 + local;
    bool live = gid < N && radii[gid] > 0;
    float v[A];
#pragma unroll
    for (int k = 0; k < A; ++k) v[k] = 0.f;
    if (live) {
      float4* row = trec + (size_t)gid * 8 + 2 * r;
      float4 lo = row[0], hi = row[1];
      float w[4] = {lo.x, lo.z, hi.x, hi.z}, f[4] = {lo.y, lo.w, hi.y, hi.w};
      bool any = false;
#pragma unroll
      for (int c = 0; c < 4; ++c) any = any || (w[c] != 0.f) || (f[c] != 0.f);
      if (any) {
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        row[0] = z;
        row[1] = z;
        float4 q0 = GSL_Q(Q0, gid), qc = GSL_Q(Q1, gid);
        int pcol0 = xp_tiny_origin(q0.x, qc.w), prow = xp_tiny_origin(q0.y, qc.w) + r;
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
    float pad[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      float x = (k < A) ? v[k < A ? k : 0] : 0.f;
      x += dpp_get<0xB1>(x);  // quad_perm [1,0,3,2]
      x += dpp_get<0x4E>(x);  // quad_perm [2,3,0,1]
      pad[k] = x;
    }
    if (r == 0) s_rows[local][0] = make_float4(pad[0], pad[1], pad[2], pad[3]);
    if (r == 1) s_rows[local][1] = make_float4(pad[4], pad[5], pad[6], pad[7]);
    if (r == 2) s_rows[local][2] = make_float4(pad[8], pad[9], pad[10], pad[11]);
  }
  __syncthreads();
  auto rows = [&](int, float4& r0, float4& r1, float4& r2) {  // Gaussian blockIdx.x*256 + threadIdx.x
    r0 = s_rows[threadIdx.x][0];
    r1 = s_rows[threadIdx.x][1];
    r2 = s_rows[threadIdx.x][2];
  };
  fproject_bwd_thread<FULL, D>(means, quats, scales, opacities, colors, sh_degree, K_sh, V, Kmat, N, W, H, eps2d,
                               antialiased, radii, Q1, comps, rows, v_means, v_quats, v_scales, v_opacities, v_colors,
                               partials);
}

// Fixed-order sum of the per-block partial rows into v_viewmat (as k_freduce_viewmat in fused.hip).
__global__ __launch_bounds__(256) void k_xp_reduce_viewmat(const float* __restrict__ partials, int nb,
                                                          const float* __restrict__ V, const float* __restrict__ Kmat,
                                                          float* __restrict__ v_viewmat) {
  __shared__ float red[4][15];
  __shared__ float tot[15];
  float acc[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) acc[k] = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
    for (int k = 0; k < 15; ++k) acc[k] += partials[(size_t)b * 16 + k];
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    float s = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 15) tot[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  __syncthreads();
  if (threadIdx.x < 16) {
    int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    float v = 0.f;
    if (r < 3) {
      Cam cam = load_cam(V, Kmat);
      M3 Ri;
      float cp[3];
      cam_inverse(cam, Ri, cp);
      float w = Ri(0, r) * tot[12] + Ri(1, r) * tot[13] + Ri(2, r) * tot[14];  // R^-T v_campos
      if (c < 3) v = tot[r * 3 + c] - w * cp[c];
      else v = tot[9 + r] - w;
    }
    v_viewmat[threadIdx.x] = v;
  }
}

}  // namespace gsl

// Same contract as gsl_tiny_gather (include/gsloc_hip.h); see the note at the top of this file.
extern "C" int gsl_tiny_gather4(const float* Q0, const float* Q1, const int32_t* radii, int N, int channels, int width,
                                int height, float* trec, const float* vcT, float* vacc, void* stream) {
  if (N < 0 || width <= 0 || height <= 0) return GSL_ERR_BAD_ARG;
  if (channels != 1 && channels != 3 && channels != 4) return GSL_ERR_BAD_ARG;
  if (N == 0) return GSL_OK;
  if (!Q0 || !Q1 || !radii || !trec || !vcT || !vacc) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  unsigned grid = (unsigned)(((size_t)N * 4 + 255) / 256);
#define CALL_TG4(DD)                                                                                               \
  hipLaunchKernelGGL((gsl::k_tiny_gather4<DD>), dim3(grid), dim3(256), 0, st, (const float4*)Q0, (const float4*)Q1, \
                     radii, N, width, height, (float4*)trec, vcT, (float4*)vacc)
  if (channels == 1) CALL_TG4(1); else if (channels == 3) CALL_TG4(3); else CALL_TG4(4);
#undef CALL_TG4
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// gsl_tiny_gather + gsl_fused_project_bwd in one kernel: same arguments as gsl_fused_project_bwd with the gradient
// rows replaced by what the gather reads (Q0, trec, vcT).  `partials` live in `ws` exactly as there.
extern "C" int gsl_tiny_project_bwd(const float* means, const float* quats, const float* scales, const float* opacities,
                                    const float* colors, int sh_degree, int K_sh, const float* viewmat, const float* K,
                                    int N, int width, int height, float eps2d, int antialiased, int channels,
                                    const int32_t* radii, const float* Q0, const float* Q1, const float* compensations,
                                    float* trec, const float* vcT, float* v_means, float* v_quats, float* v_scales,
                                    float* v_opacities, float* v_colors, float* v_viewmat, void* ws, size_t ws_bytes,
                                    int n_tiles, void* stream) {
  if (N < 0 || width <= 0 || height <= 0 || n_tiles <= 0) return GSL_ERR_BAD_ARG;
  if (channels != 1 && channels != 3 && channels != 4) return GSL_ERR_BAD_ARG;
  bool full = v_means != nullptr;
  if (full != (v_quats != nullptr) || full != (v_scales != nullptr) || full != (v_opacities != nullptr))
    return GSL_ERR_BAD_ARG;
  if (full && channels >= 3 && !v_colors) return GSL_ERR_BAD_ARG;
  if (antialiased && !compensations) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    if (v_viewmat && hipMemsetAsync(v_viewmat, 0, 16 * sizeof(float), st) != hipSuccess) return GSL_ERR_HIP;
    return GSL_OK;
  }
  if (!means || !quats || !scales || !opacities || !viewmat || !K || !radii || !Q0 || !Q1 || !trec || !vcT)
    return GSL_ERR_BAD_ARG;
  if (channels >= 3 && !colors) return GSL_ERR_BAD_ARG;
  size_t nb = ((size_t)N + 255) / 256;
  size_t need = (size_t)2 * (size_t)n_tiles * sizeof(int32_t) + nb * 16 * sizeof(float);  // == gsl_fused_ws_bytes
  if (!ws || ws_bytes < need) return GSL_ERR_WORKSPACE;
  float* partials = v_viewmat ? (float*)((int32_t*)ws + 2 * (size_t)n_tiles) : nullptr;
  int grid = (int)nb;
#define CALL_TPB(FF, DD)                                                                                            \
  hipLaunchKernelGGL((gsl::k_tiny_project_bwd<FF, DD>), dim3(grid), dim3(256), 0, st, means, quats, scales, opacities, \
                     colors, sh_degree, K_sh, viewmat, K, N, width, height, eps2d, antialiased, radii,                 \
                     (const float4*)Q0, (const float4*)Q1, compensations, (float4*)trec, vcT, v_means, v_quats,        \
                     v_scales, v_opacities, v_colors, partials)
  if (full) {
    if (channels == 1) CALL_TPB(true, 1); else if (channels == 3) CALL_TPB(true, 3); else CALL_TPB(true, 4);
  } else {
    if (channels == 1) CALL_TPB(false, 1); else if (channels == 3) CALL_TPB(false, 3); else CALL_TPB(false, 4);
  }
#undef CALL_TPB
  GSL_CHECK_LAUNCH();
  if (v_viewmat) {
    hipLaunchKernelGGL(gsl::k_xp_reduce_viewmat, dim3(1), dim3(256), 0, st, partials, grid, viewmat, K, v_viewmat);
    GSL_CHECK_LAUNCH();
  }
  return GSL_OK;
}
