// Per-Gaussian projection (EWA splatting) forward and its vjp, one camera.
// Replaces gsplat.fully_fused_projection fwd/bwd (IDX:14351, IDX:14270) at the call
// sites /root/reference/src/my_gsplat/model.py:195-213, geometry.py:117-132.
// One thread per Gaussian, 256-thread blocks (4 x wave64).  The [N,3]/[N,4] inputs are
// read as consecutive dwords per lane: a wave touches one contiguous 768/1024-byte span,
// so every fetched cache line is fully used.  Camera constants are wave-uniform and live
// in scalar registers.  HBM-bound: 40 B read + 28 B written per Gaussian (fwd).
#include "project_dev.h"

namespace gsl {

__global__ __launch_bounds__(256) void k_project_fwd(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ V, const float* __restrict__ K, int N, int W, int H, float eps2d, float near_plane,
    float far_plane, float radius_clip, int32_t* __restrict__ radii, float* __restrict__ means2d,
    float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ comps) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  Cam cam = load_cam(V, K);
  ProjMid p;
  float q[4], s[3];
  load_gaussian(means, quats, scales, i, cam, p, q, s);
  ProjOut o;
  o.radius = 0; o.mx = o.my = o.depth = o.ca = o.cb = o.cc = o.comp = 0.f;
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
      float radius = ceilf(3.f * sqrtf(v1));
      float mx = cam.fx * p.mc[0] * p.rz + cam.cx;
      float my = cam.fy * p.mc[1] * p.rz + cam.cy;
      bool ok = radius > radius_clip;
      ok = ok && !(mx + radius <= 0.f || mx - radius >= (float)W || my + radius <= 0.f || my - radius >= (float)H);
      if (ok) {
        float inv = 1.f / det;
        o.radius = (int)radius;
        o.mx = mx; o.my = my; o.depth = p.mc[2];
        o.ca = c * inv; o.cb = -b * inv; o.cc = a * inv;
        o.comp = sqrtf(fmaxf(0.f, det_orig / det));
      }
    }
  }
  radii[i] = o.radius;
  means2d[2 * (size_t)i] = o.mx;
  means2d[2 * (size_t)i + 1] = o.my;
  depths[i] = o.depth;
  conics[3 * (size_t)i] = o.ca;
  conics[3 * (size_t)i + 1] = o.cb;
  conics[3 * (size_t)i + 2] = o.cc;
  if (comps) comps[i] = o.comp;
}

template <bool FULL>
__global__ __launch_bounds__(256) void k_project_bwd(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ V, const float* __restrict__ K, int N, int W, int H, float eps2d,
    const int32_t* __restrict__ radii, const float* __restrict__ conics, const float* __restrict__ comps,
    const float* __restrict__ v_means2d, const float* __restrict__ v_depths, const float* __restrict__ v_conics,
    const float* __restrict__ v_comps, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales, float* __restrict__ partials /* [gridDim.x][12] or null */) {
  int i = blockIdx.x * 256 + threadIdx.x;
  Cam cam = load_cam(V, K);
  float vRt[12];  // v_R (9, row-major) then v_t (3)
#pragma unroll
  for (int k = 0; k < 12; ++k) vRt[k] = 0.f;
  float vmean[3] = {0.f, 0.f, 0.f}, vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
  bool live = (i < N) && (radii[i] > 0);
  if (live) {
    ProjMid p;
    float q[4], s[3];
    load_gaussian(means, quats, scales, i, cam, p, q, s);
    p.covar = quat_scale_to_covar(q, s);
    p.covar_c = mul_bt(mul(cam.R, p.covar), cam.R);
    persp_mid(cam, W, H, p);
    float vm2x = v_means2d[2 * (size_t)i], vm2y = v_means2d[2 * (size_t)i + 1];
    bool has_comp = v_comps != nullptr;
    project_vjp<FULL>(cam, eps2d, p, q, s, conics[3 * (size_t)i], conics[3 * (size_t)i + 1],
                      conics[3 * (size_t)i + 2], vm2x, vm2y, v_depths[i], v_conics[3 * (size_t)i],
                      v_conics[3 * (size_t)i + 1], v_conics[3 * (size_t)i + 2], has_comp,
                      has_comp ? comps[i] : 0.f, has_comp ? v_comps[i] : 0.f, vRt, vmean, vq, vs);
  }
  if (FULL && i < N) {
#pragma unroll
    for (int k = 0; k < 3; ++k) v_means[3 * (size_t)i + k] = vmean[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quats[4 * (size_t)i + k] = vq[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) v_scales[3 * (size_t)i + k] = vs[k];
  }
  if (partials != nullptr) {
    // deterministic block reduction: wave butterfly -> LDS -> 12 lanes write one partial row
    __shared__ float red[4][12];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      float sum = wave_sum(vRt[k]);
      if (lane == 0) red[wv][k] = sum;
    }
    __syncthreads();
    if (threadIdx.x < 12)
      partials[(size_t)blockIdx.x * 12 + threadIdx.x] =
          red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

// Sum partial rows [nb][12] in a fixed order -> v_viewmat[16] (row 3 zero).
__global__ __launch_bounds__(256) void k_reduce_viewmat(const float* __restrict__ partials, int nb,
                                                       float* __restrict__ v_viewmat) {
  __shared__ float red[4][12];
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256)
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] += partials[(size_t)b * 12 + k];
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    float s = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float v = 0.f;
    int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    if (r < 3) {
      int k = (c < 3) ? (r * 3 + c) : (9 + r);
      v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
    v_viewmat[threadIdx.x] = v;
  }
}

}  // namespace gsl

extern "C" int gsl_project_fwd(const float* means, const float* quats, const float* scales, const float* viewmat,
                               const float* K, int N, int width, int height, float eps2d, float near_plane,
                               float far_plane, float radius_clip, int32_t* radii, float* means2d, float* depths,
                               float* conics, float* compensations, void* stream) {
  if (N < 0 || width <= 0 || height <= 0) return GSL_ERR_BAD_ARG;
  if (N == 0) return GSL_OK;
  if (!means || !quats || !scales || !viewmat || !K || !radii || !means2d || !depths || !conics)
    return GSL_ERR_BAD_ARG;
  int grid = (N + 255) / 256;
  GSL_CLAMP_DEPTH_WINDOW(near_plane, far_plane);
  hipLaunchKernelGGL(gsl::k_project_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, means, quats, scales,
                     viewmat, K, N, width, height, eps2d, near_plane, far_plane, radius_clip, radii, means2d, depths,
                     conics, compensations);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" size_t gsl_project_bwd_ws_bytes(int N) {
  size_t nb = (size_t)((N > 0 ? N : 1) + 255) / 256;
  return nb * 12 * sizeof(float);
}

extern "C" int gsl_project_bwd(const float* means, const float* quats, const float* scales, const float* viewmat,
                               const float* K, int N, int width, int height, float eps2d, const int32_t* radii,
                               const float* conics, const float* compensations, const float* v_means2d,
                               const float* v_depths, const float* v_conics, const float* v_compensations,
                               float* v_means, float* v_quats, float* v_scales, float* v_viewmat, void* ws,
                               size_t ws_bytes, void* stream) {
  if (N < 0 || width <= 0 || height <= 0) return GSL_ERR_BAD_ARG;
  bool full = v_means != nullptr;
  if (full != (v_quats != nullptr) || full != (v_scales != nullptr)) return GSL_ERR_BAD_ARG;
  if (v_compensations && !compensations) return GSL_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    if (v_viewmat) {
      if (gsl::zero_u32(v_viewmat, 16, st) != GSL_OK) return GSL_ERR_HIP;
    }
    return GSL_OK;
  }
  if (!means || !quats || !scales || !viewmat || !K || !radii || !conics || !v_means2d || !v_depths || !v_conics)
    return GSL_ERR_BAD_ARG;
  if (!full && !v_viewmat) return GSL_OK;
  int grid = (N + 255) / 256;
  float* partials = nullptr;
  if (v_viewmat) {
    if (!ws || ws_bytes < gsl_project_bwd_ws_bytes(N)) return GSL_ERR_WORKSPACE;
    partials = (float*)ws;
  }
  if (full)
    hipLaunchKernelGGL(gsl::k_project_bwd<true>, dim3(grid), dim3(256), 0, st, means, quats, scales, viewmat, K, N,
                       width, height, eps2d, radii, conics, compensations, v_means2d, v_depths, v_conics,
                       v_compensations, v_means, v_quats, v_scales, partials);
  else
    hipLaunchKernelGGL(gsl::k_project_bwd<false>, dim3(grid), dim3(256), 0, st, means, quats, scales, viewmat, K, N,
                       width, height, eps2d, radii, conics, compensations, v_means2d, v_depths, v_conics,
                       v_compensations, v_means, v_quats, v_scales, partials);
  GSL_CHECK_LAUNCH();
  if (v_viewmat) {
    hipLaunchKernelGGL(gsl::k_reduce_viewmat, dim3(1), dim3(256), 0, st, partials, grid, v_viewmat);
    GSL_CHECK_LAUNCH();
  }
  return GSL_OK;
}
