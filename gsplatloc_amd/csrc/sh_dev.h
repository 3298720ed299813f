// Real spherical-harmonics basis (degree 0..3) and its gradient, shared by sh.hip and fused.hip.
#pragma once
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

}  // namespace gsl
