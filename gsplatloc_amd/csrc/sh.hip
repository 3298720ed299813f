// Real spherical harmonics (degree 0..3) -> RGB, forward and vjp.
// Replaces gsplat.spherical_harmonics (IDX:14306, autograd IDX:14297); GsplatLoc calls it with
// sh_degree=1 through gsplat.rasterization (/root/reference/src/my_gsplat/model.py:127,195-213).
// One thread per (camera, Gaussian) element; a wave reads a contiguous K*3*4*64-byte span.
#include "gsloc_common.h"

namespace gsl {

// Basis values Y[0..nK) at the normalised direction (x,y,z).
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* Y) {
  Y[0] = 0.2820947917738781f;
  if (deg < 1) return;
  Y[1] = -0.48860251190292f * y;
  Y[2] = 0.48860251190292f * z;
  Y[3] = -0.48860251190292f * x;
  if (deg < 2) return;
  float z2 = z * z;
  float fTmpB = -1.092548430592079f * z;
  float fC1 = x * x - y * y;
  float fS1 = 2.f * x * y;
  Y[4] = 0.5462742152960395f * fS1;
  Y[5] = fTmpB * y;
  Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
  Y[7] = fTmpB * x;
  Y[8] = 0.5462742152960395f * fC1;
  if (deg < 3) return;
  float fTmpC = -2.285228997322329f * z2 + 0.4570457994644658f;
  float fTmpBb = 1.445305721320277f * z;
  float fC2 = x * fC1 - y * fS1;
  float fS2 = x * fS1 + y * fC1;
  Y[9] = -0.5900435899266435f * fS2;
  Y[10] = fTmpBb * fS1;
  Y[11] = fTmpC * y;
  Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
  Y[13] = fTmpC * x;
  Y[14] = fTmpBb * fC1;
  Y[15] = -0.5900435899266435f * fC2;
}

// Gradient of sum_k s[k]*Y_k wrt (x,y,z) treated as independent variables.
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, const float* s, float g[3]) {
  g[0] = g[1] = g[2] = 0.f;
  if (deg < 1) return;
  const float C1 = 0.48860251190292f;
  g[1] += -C1 * s[1];
  g[2] += C1 * s[2];
  g[0] += -C1 * s[3];
  if (deg < 2) return;
  const float c2 = 0.5462742152960395f, b2 = -1.092548430592079f, a2 = 0.9461746957575601f;
  g[0] += s[4] * c2 * 2.f * y + s[7] * b2 * z + s[8] * c2 * 2.f * x;
  g[1] += s[4] * c2 * 2.f * x + s[5] * b2 * z - s[8] * c2 * 2.f * y;
  g[2] += s[5] * b2 * y + s[6] * 2.f * a2 * z + s[7] * b2 * x;
  if (deg < 3) return;
  const float e = -0.5900435899266435f, f = 1.445305721320277f, tc = -2.285228997322329f;
  float z2 = z * z;
  float fTmpC = tc * z2 + 0.4570457994644658f;
  float x2 = x * x, y2 = y * y;
  g[0] += s[9] * e * 6.f * x * y + s[10] * f * 2.f * y * z + s[13] * fTmpC + s[14] * f * 2.f * x * z +
          s[15] * e * 3.f * (x2 - y2);
  g[1] += s[9] * e * 3.f * (x2 - y2) + s[10] * f * 2.f * x * z + s[11] * fTmpC - s[14] * f * 2.f * y * z -
          s[15] * e * 6.f * x * y;
  g[2] += s[10] * f * 2.f * x * y + s[11] * 2.f * tc * z * y + s[12] * (3.f * 1.865881662950577f * z2 - 1.119528997770346f) +
          s[13] * 2.f * tc * z * x + s[14] * f * (x2 - y2);
}

__global__ __launch_bounds__(256) void k_sh_fwd(int deg, const float* __restrict__ dirs,
                                                const float* __restrict__ coeffs,
                                                const uint8_t* __restrict__ masks, int M, int K,
                                                float* __restrict__ colors) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  if (!masks || masks[i]) {
    float x = dirs[3 * (size_t)i], y = dirs[3 * (size_t)i + 1], z = dirs[3 * (size_t)i + 2];
    float inorm = rsqrtf(x * x + y * y + z * z);
    x *= inorm; y *= inorm; z *= inorm;
    float Y[16];
    sh_basis(deg, x, y, z, Y);
    int nK = (deg + 1) * (deg + 1);
    const float* c = coeffs + (size_t)i * K * 3;
    for (int k = 0; k < nK; ++k) {
      c0 += Y[k] * c[3 * k];
      c1 += Y[k] * c[3 * k + 1];
      c2 += Y[k] * c[3 * k + 2];
    }
  }
  colors[3 * (size_t)i] = c0;
  colors[3 * (size_t)i + 1] = c1;
  colors[3 * (size_t)i + 2] = c2;
}

__global__ __launch_bounds__(256) void k_sh_bwd(int deg, const float* __restrict__ dirs,
                                                const float* __restrict__ coeffs,
                                                const uint8_t* __restrict__ masks, int M, int K,
                                                const float* __restrict__ v_colors, float* __restrict__ v_coeffs,
                                                float* __restrict__ v_dirs) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  int nK = (deg + 1) * (deg + 1);
  float* vc = v_coeffs + (size_t)i * K * 3;
  bool on = !masks || masks[i];
  float gd[3] = {0.f, 0.f, 0.f};
  if (on) {
    float rx = dirs[3 * (size_t)i], ry = dirs[3 * (size_t)i + 1], rz = dirs[3 * (size_t)i + 2];
    float inorm = rsqrtf(rx * rx + ry * ry + rz * rz);
    float x = rx * inorm, y = ry * inorm, z = rz * inorm;
    float Y[16];
    sh_basis(deg, x, y, z, Y);
    float v0 = v_colors[3 * (size_t)i], v1 = v_colors[3 * (size_t)i + 1], v2 = v_colors[3 * (size_t)i + 2];
    float s[16];
    const float* c = coeffs + (size_t)i * K * 3;
    for (int k = 0; k < nK; ++k) {
      vc[3 * k] = Y[k] * v0;
      vc[3 * k + 1] = Y[k] * v1;
      vc[3 * k + 2] = Y[k] * v2;
      s[k] = c[3 * k] * v0 + c[3 * k + 1] * v1 + c[3 * k + 2] * v2;
    }
    for (int k = nK; k < K; ++k) vc[3 * k] = vc[3 * k + 1] = vc[3 * k + 2] = 0.f;
    if (v_dirs) {
      float g[3];
      sh_basis_grad(deg, x, y, z, s, g);
      float d = g[0] * x + g[1] * y + g[2] * z;
      gd[0] = (g[0] - d * x) * inorm;
      gd[1] = (g[1] - d * y) * inorm;
      gd[2] = (g[2] - d * z) * inorm;
    }
  } else {
    for (int k = 0; k < K * 3; ++k) vc[k] = 0.f;
  }
  if (v_dirs) {
    v_dirs[3 * (size_t)i] = gd[0];
    v_dirs[3 * (size_t)i + 1] = gd[1];
    v_dirs[3 * (size_t)i + 2] = gd[2];
  }
}

}  // namespace gsl

extern "C" int gsl_sh_fwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks, int M, int K,
                          float* colors, void* stream) {
  if (degree < 0 || degree > 3 || M < 0 || K < (degree + 1) * (degree + 1)) return GSL_ERR_BAD_ARG;
  if (M == 0) return GSL_OK;
  if (!dirs || !coeffs || !colors) return GSL_ERR_BAD_ARG;
  hipLaunchKernelGGL(gsl::k_sh_fwd, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, degree, dirs, coeffs,
                     masks, M, K, colors);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" int gsl_sh_bwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks, int M, int K,
                          const float* v_colors, float* v_coeffs, float* v_dirs, void* stream) {
  if (degree < 0 || degree > 3 || M < 0 || K < (degree + 1) * (degree + 1)) return GSL_ERR_BAD_ARG;
  if (M == 0) return GSL_OK;
  if (!dirs || !coeffs || !v_colors || !v_coeffs) return GSL_ERR_BAD_ARG;
  hipLaunchKernelGGL(gsl::k_sh_bwd, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, degree, dirs, coeffs,
                     masks, M, K, v_colors, v_coeffs, v_dirs);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
