// Per-Gaussian projection (EWA splatting) forward and its vjp, one camera.
// Replaces gsplat.fully_fused_projection fwd/bwd (IDX:14351, IDX:14270) at the call
// sites /root/reference/src/my_gsplat/model.py:195-213, geometry.py:117-132.
// One thread per Gaussian, 256-thread blocks (4 x wave64).  The [N,3]/[N,4] inputs are
// read as consecutive dwords per lane: a wave touches one contiguous 768/1024-byte span,
// so every fetched cache line is fully used.  Camera constants are wave-uniform and live
// in scalar registers.  HBM-bound: 40 B read + 28 B written per Gaussian (fwd).
#include "gsloc_common.h"

namespace gsl {

struct ProjOut {
  float mx, my, depth;
  float ca, cb, cc;  // conic
  float comp;
  int radius;  // 0 => culled
};

// Intermediates the backward needs again.
struct ProjMid {
  float mean[3];
  float mc[3];
  M3 covar;    // world
  M3 covar_c;  // camera
  float J[6];  // 2x3 row-major
  float tx, ty, rz;
  bool x_in, y_in;
};

__device__ __forceinline__ void persp_mid(const Cam& cam, int W, int H, ProjMid& p) {
  float x = p.mc[0], y = p.mc[1], z = p.mc[2];
  float lim_x = 1.3f * (0.5f * (float)W / cam.fx);
  float lim_y = 1.3f * (0.5f * (float)H / cam.fy);
  float rz = 1.f / z;
  float rz2 = rz * rz;
  float xr = x * rz, yr = y * rz;
  p.x_in = (xr <= lim_x) && (xr >= -lim_x);
  p.y_in = (yr <= lim_y) && (yr >= -lim_y);
  p.tx = z * fminf(lim_x, fmaxf(-lim_x, xr));
  p.ty = z * fminf(lim_y, fmaxf(-lim_y, yr));
  p.rz = rz;
  p.J[0] = cam.fx * rz; p.J[1] = 0.f; p.J[2] = -cam.fx * p.tx * rz2;
  p.J[3] = 0.f; p.J[4] = cam.fy * rz; p.J[5] = -cam.fy * p.ty * rz2;
}

// cov2d = J * Sc * J^T (symmetric 2x2: a, b, c)
__device__ __forceinline__ void cov2d_from(const float J[6], const M3& Sc, float& a, float& b, float& c) {
  float r0[3], r1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r0[j] = J[0] * Sc(0, j) + J[1] * Sc(1, j) + J[2] * Sc(2, j);
    r1[j] = J[3] * Sc(0, j) + J[4] * Sc(1, j) + J[5] * Sc(2, j);
  }
  a = r0[0] * J[0] + r0[1] * J[1] + r0[2] * J[2];
  b = r0[0] * J[3] + r0[1] * J[4] + r0[2] * J[5];
  c = r1[0] * J[3] + r1[1] * J[4] + r1[2] * J[5];
}

__device__ __forceinline__ void load_gaussian(const float* __restrict__ means, const float* __restrict__ quats,
                                              const float* __restrict__ scales, int i, const Cam& cam,
                                              ProjMid& p, float q[4], float s[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) p.mean[k] = means[3 * (size_t)i + k];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    p.mc[k] = cam.R(k, 0) * p.mean[0] + cam.R(k, 1) * p.mean[1] + cam.R(k, 2) * p.mean[2] + cam.t[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = quats[4 * (size_t)i + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) s[k] = scales[3 * (size_t)i + k];
}

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

// vjp wrt the unit quaternion components of a rotation matrix gradient vR (row-major).
__device__ __forceinline__ void quat_vjp(const float qin[4], const M3& vR, float vq[4]) {
  float inv = rsqrtf(qin[0] * qin[0] + qin[1] * qin[1] + qin[2] * qin[2] + qin[3] * qin[3]);
  float w = qin[0] * inv, x = qin[1] * inv, y = qin[2] * inv, z = qin[3] * inv;
  float g[4];
  g[0] = 2.f * (x * (vR(2, 1) - vR(1, 2)) + y * (vR(0, 2) - vR(2, 0)) + z * (vR(1, 0) - vR(0, 1)));
  g[1] = 2.f * (-2.f * x * (vR(1, 1) + vR(2, 2)) + y * (vR(1, 0) + vR(0, 1)) + z * (vR(2, 0) + vR(0, 2)) +
                w * (vR(2, 1) - vR(1, 2)));
  g[2] = 2.f * (x * (vR(1, 0) + vR(0, 1)) - 2.f * y * (vR(0, 0) + vR(2, 2)) + z * (vR(2, 1) + vR(1, 2)) +
                w * (vR(0, 2) - vR(2, 0)));
  g[3] = 2.f * (x * (vR(2, 0) + vR(0, 2)) + y * (vR(2, 1) + vR(1, 2)) - 2.f * z * (vR(0, 0) + vR(1, 1)) +
                w * (vR(1, 0) - vR(0, 1)));
  float d = g[0] * w + g[1] * x + g[2] * y + g[3] * z;
  vq[0] = (g[0] - d * w) * inv;
  vq[1] = (g[1] - d * x) * inv;
  vq[2] = (g[2] - d * y) * inv;
  vq[3] = (g[3] - d * z) * inv;
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
    // vjp of the 2x2 inverse: v_cov2 = -C * Vc * C, C = conic matrix
    float ca = conics[3 * (size_t)i], cb = conics[3 * (size_t)i + 1], cc = conics[3 * (size_t)i + 2];
    float va = v_conics[3 * (size_t)i], vb = 0.5f * v_conics[3 * (size_t)i + 1], vc = v_conics[3 * (size_t)i + 2];
    // T = C * Vc
    float t00 = ca * va + cb * vb, t01 = ca * vb + cb * vc;
    float t10 = cb * va + cc * vb, t11 = cb * vb + cc * vc;
    float g00 = -(t00 * ca + t01 * cb), g01 = -(t00 * cb + t01 * cc);
    float g10 = -(t10 * ca + t11 * cb), g11 = -(t10 * cb + t11 * cc);
    if (v_comps != nullptr) {  // add_blur vjp (rasterize_mode "antialiased")
      float comp = comps[i];
      float vcomp = v_comps[i];
      float det_conic = ca * cc - cb * cb;
      float v_sqr = vcomp * 0.5f / (comp + 1e-6f);
      float om = 1.f - comp * comp;
      g00 += v_sqr * (om * ca - eps2d * det_conic);
      g01 += v_sqr * (om * cb);
      g10 += v_sqr * (om * cb);
      g11 += v_sqr * (om * cc - eps2d * det_conic);
    }
    const float* J = p.J;
    // v_covar_c = J^T G J
    float GJ0[3], GJ1[3];  // G*J rows
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      GJ0[j] = g00 * J[j] + g01 * J[3 + j];
      GJ1[j] = g10 * J[j] + g11 * J[3 + j];
    }
    M3 vSc;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) vSc(r, c) = J[r] * GJ0[c] + J[3 + r] * GJ1[c];
    // v_J = G J Sc^T + G^T J Sc
    float GtJ0[3], GtJ1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      GtJ0[j] = g00 * J[j] + g10 * J[3 + j];
      GtJ1[j] = g01 * J[j] + g11 * J[3 + j];
    }
    float vJ[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a0 += GJ0[k] * p.covar_c(c, k) + GtJ0[k] * p.covar_c(k, c);
        a1 += GJ1[k] * p.covar_c(c, k) + GtJ1[k] * p.covar_c(k, c);
      }
      vJ[c] = a0;
      vJ[3 + c] = a1;
    }
    float x = p.mc[0], y = p.mc[1];
    float rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
    float vm2x = v_means2d[2 * (size_t)i], vm2y = v_means2d[2 * (size_t)i + 1];
    float vmc[3];
    vmc[0] = cam.fx * rz * vm2x;
    vmc[1] = cam.fy * rz * vm2y;
    vmc[2] = -(cam.fx * x * vm2x + cam.fy * y * vm2y) * rz2;
    if (p.x_in) vmc[0] += -cam.fx * rz2 * vJ[2];
    else vmc[2] += -cam.fx * rz3 * vJ[2] * p.tx;
    if (p.y_in) vmc[1] += -cam.fy * rz2 * vJ[5];
    else vmc[2] += -cam.fy * rz3 * vJ[5] * p.ty;
    vmc[2] += -cam.fx * rz2 * vJ[0] - cam.fy * rz2 * vJ[4] + 2.f * cam.fx * p.tx * rz3 * vJ[2] +
              2.f * cam.fy * p.ty * rz3 * vJ[5];
    vmc[2] += v_depths[i];
    // world->camera vjp: v_R = v_mc mean^T + vSc R Sigma^T + vSc^T R Sigma ; v_t = v_mc
    M3 RS = mul(cam.R, p.covar);  // Sigma symmetric
    M3 A = mul(vSc, RS);
    M3 B = mul_at(vSc, RS);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) vRt[r * 3 + c] = vmc[r] * p.mean[c] + A(r, c) + B(r, c);
      vRt[9 + r] = vmc[r];
    }
    if (FULL) {
#pragma unroll
      for (int k = 0; k < 3; ++k) vmean[k] = cam.R(0, k) * vmc[0] + cam.R(1, k) * vmc[1] + cam.R(2, k) * vmc[2];
      M3 vS = mul(mul_at(cam.R, vSc), cam.R);  // R^T vSc R
      M3 Rq = quat_to_rotmat(q[0], q[1], q[2], q[3]);
      M3 M;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M(r, c) = Rq(r, c) * s[c];
      M3 vSs;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) vSs(r, c) = vS(r, c) + vS(c, r);
      M3 vM = mul(vSs, M);
      M3 vRq;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) vRq(r, c) = vM(r, c) * s[c];
      quat_vjp(q, vRq, vq);
#pragma unroll
      for (int c = 0; c < 3; ++c) vs[c] = Rq(0, c) * vM(0, c) + Rq(1, c) * vM(1, c) + Rq(2, c) * vM(2, c);
    }
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
      if (hipMemsetAsync(v_viewmat, 0, 16 * sizeof(float), st) != hipSuccess) return GSL_ERR_HIP;
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
