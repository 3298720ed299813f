// Kernels written after round 1's GPU access had ended: they compile for gfx950 and are reachable through
// the C ABI, but they have NOT run on hardware yet and nothing selects them by default (RenderContext uses
// them only under an explicit environment switch).  Kept in their own translation unit so the objects of the
// measured kernels are untouched.
#include "gsloc_common.h"

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
