// Per-Gaussian backward of projection + colour (one thread per Gaussian) as a device function whose gradient row
// comes from a caller-supplied source.  It is the body of k_fproject_bwd (fused.hip) verbatim; fused.hip keeps its
// own copy because routing the measured kernel through this function changed its generated code, and the measured
// objects stay untouched until the experiment in experimental.hip has run on hardware.  Used by experimental.hip
// only.
#pragma once
#include "gsloc_common.h"
#include "project_dev.h"
#include "sh_dev.h"

namespace gsl {

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

// `rows(i, r0, r1, r2)` delivers Gaussian i's gradient row [vx vy | va vb vc | vop | colours ...] as three float4.
template <bool FULL, int D, typename Rows>
__device__ __forceinline__ void fproject_bwd_thread(
    const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, const float* __restrict__ colors, int sh_degree, int K_sh,
    const float* __restrict__ V, const float* __restrict__ Kmat, int N, int W, int H, float eps2d, int antialiased,
    const int32_t* __restrict__ radii, const float4* __restrict__ Q1, const float* __restrict__ comps, Rows rows,
    float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_opacities, float* __restrict__ v_colors, float* __restrict__ partials) {
  constexpr bool RGB = D >= 3;
  int i = blockIdx.x * 256 + threadIdx.x;
  Cam cam = load_cam(V, Kmat);
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
    rows(i, r0, r1, r2);
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
        v_colors[3 * (size_t)i] = vrgb[0]; v_colors[3 * (size_t)i + 1] = vrgb[1]; v_colors[3 * (size_t)i + 2] = vrgb[2];
      } else {
        for (int k = 0; k < K_sh * 3; ++k) v_colors[(size_t)i * K_sh * 3 + k] = 0.f;
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

}  // namespace gsl
