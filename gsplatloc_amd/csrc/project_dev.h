// Device-side projection math shared by the stage kernels (project.hip) and the fused
// pipeline (fused.hip).  Formulas: SURVEY.md A.1 / A.5 (gsplat fully_fused_projection fwd/bwd).
#pragma once
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


// vjp of one valid Gaussian.  p must hold mean, mc, covar, covar_c, J, tx, ty, rz, x_in, y_in.
// Inputs: conic (ca,cb,cc) of the blurred 2D covariance, upstream grads of mean2d / depth / conic
// (+ compensation when has_comp).  Outputs: vRt[12] = v_R (row-major 9) then v_t (3); when FULL
// also v_mean[3], v_quat[4], v_scale[3].
template <bool FULL>
__device__ __forceinline__ void project_vjp(const Cam& cam, float eps2d, const ProjMid& p, const float q[4],
                                            const float s[3], float ca, float cb, float cc, float vm2x, float vm2y,
                                            float vdepth, float v_ca, float v_cb, float v_cc, bool has_comp,
                                            float comp, float vcomp, float vRt[12], float vmean[3], float vq[4],
                                            float vs[3]) {
  {
    float va = v_ca, vb = 0.5f * v_cb, vc = v_cc;
    // vjp of the 2x2 inverse: v_cov2 = -C * Vc * C, C = conic matrix
    // T = C * Vc
    float t00 = ca * va + cb * vb, t01 = ca * vb + cb * vc;
    float t10 = cb * va + cc * vb, t11 = cb * vb + cc * vc;
    float g00 = -(t00 * ca + t01 * cb), g01 = -(t00 * cb + t01 * cc);
    float g10 = -(t10 * ca + t11 * cb), g11 = -(t10 * cb + t11 * cc);
    if (has_comp) {  // add_blur vjp (rasterize_mode "antialiased")
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
    vmc[2] += vdepth;
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
}

}  // namespace gsl
