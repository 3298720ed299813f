// Real spherical harmonics (degree 0..3) -> RGB, forward and vjp.
// Replaces gsplat.spherical_harmonics (IDX:14306, autograd IDX:14297); GsplatLoc calls it with
// sh_degree=1 through gsplat.rasterization (/root/reference/src/my_gsplat/model.py:127,195-213).
// One thread per (camera, Gaussian) element; a wave reads a contiguous K*3*4*64-byte span.
#include "sh_dev.h"

namespace gsl {

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
