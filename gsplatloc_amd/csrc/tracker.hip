// Device-side tail of one pose-tracking iteration: the depth + Sobel-edge loss with its gradient, and the
// pose update (pose chain, two Adam optimisers, exponential LR decay, early-stop bookkeeping).
// Replaces ~40 small PyTorch/kornia launches and three blocking .item() reads per iteration of
// /root/reference/src/my_gsplat/gs_trainer_total.py:105-185,265-267 (loss.py:10-59, model.py:79-116,
// eval/utils.py:122-168) so that a whole iteration is a fixed launch sequence (HIP graph).
//
//   loss  = 0.8 * mean|d*m - g*m| + 0.2 * mean|S(d*m) - S(g*m)|,  m = (d != 0) (no gradient),
//   S(x)  = sqrt(gx^2 + gy^2 + 1e-6), gx/gy = 3x3 Sobel / 8 with replicate padding (kornia.filters.sobel).
#include "gsloc_common.h"
#include "loss_dev.h"

namespace gsl {

// Sobel of the masked image at (i,j) with replicate padding.  img(i,j) = v * (d != 0).
template <typename F>
__device__ __forceinline__ void sobel_at(F img, int i, int j, int H, int W, float& gx, float& gy) {
  float p[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) p[a][b] = img(clampi(i + a - 1, 0, H - 1), clampi(j + b - 1, 0, W - 1));
  gx = ((p[0][2] - p[0][0]) + 2.f * (p[1][2] - p[1][0]) + (p[2][2] - p[2][0])) * 0.125f;
  gy = ((p[2][0] - p[0][0]) + 2.f * (p[2][1] - p[0][1]) + (p[2][2] - p[0][2])) * 0.125f;
}

// One workgroup per 16x16 pixel block of rows [h0,h1) (owned rows [r0,r1) plus the one-row halo whose depth the owned
// rows' Sobel reads).  The block's depth values with a two-pixel apron go through LDS once; the edge-loss weights
// wx = s*gx/S, wy = s*gy/S (s = sign(S_d - S_g)) of the block and its one-pixel ring are computed there, and every
// pixel then gathers the adjoint of the replicate-padded Sobel from its 3x3 neighbourhood -- value and gradient of the
// loss in ONE launch.  render is [H,W,D], depth = channel D-1.  partial[block] = (sum|a-b|, sum|Sa-Sb|) over the
// block's owned pixels; v_render[...,D-1] is written for every pixel of rows [h0,h1).
__global__ __launch_bounds__(256) void k_loss_fused(const float* __restrict__ render, int D,
                                                    const float* __restrict__ gt, int W, int H, int r0, int r1, int h0,
                                                    int h1, float depth_w, float edge_w, float inv_P,
                                                    float* __restrict__ v_render, float* __restrict__ partial) {
  __shared__ LossLds L;
  int tid = threadIdx.x;
  int x0 = blockIdx.x * 16, y0 = h0 + blockIdx.y * 16;
  float g, l1, le;
  bool has_pixel;
  loss_block(L, render, D, gt, W, H, r0, r1, h1, depth_w, edge_w, inv_P, x0, y0, tid, g, has_pixel, l1, le);
  if (has_pixel) v_render[((size_t)(y0 + (tid >> 4)) * W + (x0 + (tid & 15))) * D + (D - 1)] = g;
  auto& red = L.red;
  int lane = tid & 63, wv = tid >> 6;
  float s1 = wave_sum(l1), s2 = wave_sum(le);
  if (lane == 0) { red[wv][0] = s1; red[wv][1] = s2; }
  __syncthreads();
  if (tid < 2)
    partial[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ------------------------------------------------------------------------------------------------
// Normal-consistency term (/root/reference/src/my_gsplat/loss.py:62-101 "cosine", geometry.py:164-197; the call the
// reference keeps commented out at gs_trainer_total.py:138-143 with normal_lambda = 0, data/base.py:28):
//   P(i,j)  = depth(i,j) * ((j - cx)/fx, (i - cy)/fy, 1)             back-projection of the masked depth image
//   n(i,j)  = normalize(dx x dy),  dx/dy = central differences of P with replicated borders, eps 1e-12
//   loss    = 1 - mean_{i,c} cos_i,c,   cos_i,c = <a_i.c, b_i.c> / (max(|a_i.c|, 1e-8) max(|b_i.c|, 1e-8))
// where a = normals of the rendered depth, b = normals of the target depth and -- as coded in the reference:
// F.cosine_similarity(..., dim=1) on [H,W,3] maps -- the similarity runs ALONG EACH IMAGE ROW per component.
// Four deterministic kernels: row statistics, their sum, the gradient w.r.t. dx/dy per pixel, its gather to depth.
// ------------------------------------------------------------------------------------------------
struct Intrin { float fx, fy, cx, cy; };

template <typename F>
__device__ __forceinline__ void surface_diffs(F img, int i, int j, int H, int W, Intrin k, float dx[3], float dy[3]) {
  // No fused multiply-add in the differences and the cross product: the reference (torch, one kernel per operation)
  // subtracts ROUNDED products, so on a plane facing the camera (zr == zl) yi * zr - yi * zl is exactly 0 and so is the
  // normal's x component; fma(yi, zr, -(yi * zl)) would leave the product's rounding error there instead, and the
  // per-row cosine of that component -- noise over noise -- moved the loss by 1e-3 (found when a compiler flag changed
  // which of these expressions got contracted).
#pragma clang fp contract(off)
  int jl = clampi(j - 1, 0, W - 1), jr = clampi(j + 1, 0, W - 1), iu = clampi(i - 1, 0, H - 1), id = clampi(i + 1, 0, H - 1);
  float zl = img(i, jl), zr = img(i, jr), zu = img(iu, j), zd = img(id, j);
  float yi = ((float)i - k.cy) / k.fy, xj = ((float)j - k.cx) / k.fx;
  dx[0] = ((float)jr - k.cx) / k.fx * zr - ((float)jl - k.cx) / k.fx * zl;
  dx[1] = yi * zr - yi * zl;
  dx[2] = zr - zl;
  dy[0] = xj * zd - xj * zu;
  dy[1] = ((float)id - k.cy) / k.fy * zd - ((float)iu - k.cy) / k.fy * zu;
  dy[2] = zd - zu;
}

__device__ __forceinline__ float unit_normal(const float dx[3], const float dy[3], float n[3]) {
#pragma clang fp contract(off)  // (see surface_diffs)
  float c0 = dx[1] * dy[2] - dx[2] * dy[1], c1 = dx[2] * dy[0] - dx[0] * dy[2], c2 = dx[0] * dy[1] - dx[1] * dy[0];
  float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
  float inv = 1.f / fmaxf(len, 1e-12f);
  n[0] = c0 * inv; n[1] = c1 * inv; n[2] = c2 * inv;
  return len;
}

// One workgroup per owned row: rowstat[i][0..2] = <a,b>, [3..5] = <a,a>, [6..8] = <b,b> per component, [9] = sum_c cos.
__global__ __launch_bounds__(256) void k_normal_rows(const float* __restrict__ render, int D,
                                                     const float* __restrict__ gt, int W, int H, int r0, Intrin k,
                                                     float* __restrict__ rowstat) {
  int i = r0 + blockIdx.x;
  auto A = [&](int ii, int jj) { return render[((size_t)ii * W + jj) * D + (D - 1)]; };
  auto B = [&](int ii, int jj) {
    float d = render[((size_t)ii * W + jj) * D + (D - 1)];
    return d != 0.f ? gt[(size_t)ii * W + jj] : 0.f;
  };
  float acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) acc[q] = 0.f;
  for (int j = threadIdx.x; j < W; j += 256) {
    float dx[3], dy[3], a[3], b[3];
    surface_diffs(A, i, j, H, W, k, dx, dy);
    unit_normal(dx, dy, a);
    surface_diffs(B, i, j, H, W, k, dx, dy);
    unit_normal(dx, dy, b);
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc[c] += a[c] * b[c]; acc[3 + c] += a[c] * a[c]; acc[6 + c] += b[c] * b[c]; }
  }
  __shared__ float red[4][9];
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    float s = wave_sum(acc[q]);
    if (lane == 0) red[wv][q] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float st[9];
    for (int q = 0; q < 9; ++q) { st[q] = red[0][q] + red[1][q] + red[2][q] + red[3][q]; rowstat[(size_t)i * 10 + q] = st[q]; }
    float cs = 0.f;
    for (int c = 0; c < 3; ++c) cs += st[c] / (fmaxf(sqrtf(st[3 + c]), 1e-8f) * fmaxf(sqrtf(st[6 + c]), 1e-8f));
    rowstat[(size_t)i * 10 + 9] = cs;
  }
}

__global__ __launch_bounds__(256) void k_normal_sum(const float* __restrict__ rowstat, int r0, int r1,
                                                    float* __restrict__ normal_sum) {
  float a = 0.f;
  for (int i = r0 + threadIdx.x; i < r1; i += 256) a += rowstat[(size_t)i * 10 + 9];
  __shared__ float red[4];
  float s = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) normal_sum[0] = red[0] + red[1] + red[2] + red[3];
}

// Per owned pixel: d loss / d dx and d loss / d dy of the rendered surface (6 floats).  coef = normal_lambda / (3 H).
__global__ __launch_bounds__(256) void k_normal_grad(const float* __restrict__ render, int D,
                                                     const float* __restrict__ gt, int W, int H, int r0, int r1,
                                                     Intrin k, const float* __restrict__ rowstat, float coef,
                                                     float* __restrict__ gdxy) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= (r1 - r0) * W) return;
  int i = r0 + idx / W, j = idx - (idx / W) * W;
  auto A = [&](int ii, int jj) { return render[((size_t)ii * W + jj) * D + (D - 1)]; };
  auto B = [&](int ii, int jj) {
    float d = render[((size_t)ii * W + jj) * D + (D - 1)];
    return d != 0.f ? gt[(size_t)ii * W + jj] : 0.f;
  };
  float dx[3], dy[3], a[3], b[3], bx[3], by[3];
  surface_diffs(B, i, j, H, W, k, bx, by);
  unit_normal(bx, by, b);
  surface_diffs(A, i, j, H, W, k, dx, dy);
  float len = unit_normal(dx, dy, a);
  const float* st = rowstat + (size_t)i * 10;
  float ga[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float na = sqrtf(st[3 + c]), nb = sqrtf(st[6 + c]);
    float da = fmaxf(na, 1e-8f), db = fmaxf(nb, 1e-8f);
    // torch clamps the norms in place outside autograd: the value uses max(|a|, eps), the norm's own derivative a/|a|
    float dcos = b[c] / (da * db) - ((na > 0.f) ? st[c] / (da * da * db) * (a[c] / na) : 0.f);
    ga[c] = -coef * dcos;
  }
  float gc[3];
  if (len > 1e-12f) {
    float dot = a[0] * ga[0] + a[1] * ga[1] + a[2] * ga[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) gc[c] = (ga[c] - a[c] * dot) / len;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) gc[c] = ga[c] * 1e12f;
  }
  // c = dx x dy  =>  g_dx = dy x g_c,  g_dy = g_c x dx
  float* o = gdxy + 6 * ((size_t)i * W + j);
  o[0] = dy[1] * gc[2] - dy[2] * gc[1];
  o[1] = dy[2] * gc[0] - dy[0] * gc[2];
  o[2] = dy[0] * gc[1] - dy[1] * gc[0];
  o[3] = gc[1] * dx[2] - gc[2] * dx[1];
  o[4] = gc[2] * dx[0] - gc[0] * dx[2];
  o[5] = gc[0] * dx[1] - gc[1] * dx[0];
}

// Per pixel of rows [h0,h1): gather the differences that read its point (replicated borders included) and chain to
// depth through P = depth * ray; ADDED to v_render[...,D-1] (gsl_tracking_loss wrote the other two terms there).
__global__ __launch_bounds__(256) void k_normal_gather(const float* __restrict__ render, int D, int W, int H, int r0,
                                                       int r1, int h0, int h1, Intrin k,
                                                       const float* __restrict__ gdxy, float* __restrict__ v_render) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= (h1 - h0) * W) return;
  int i = h0 + idx / W, j = idx - (idx / W) * W;
  size_t q = (size_t)i * W + j;
  if (render[q * D + (D - 1)] == 0.f) return;  // masked pixel: the mask carries no gradient
  float g[3] = {0.f, 0.f, 0.f};
  if (i >= r0 && i < r1) {
    for (int pj = max(j - 1, 0); pj <= min(j + 1, W - 1); ++pj) {
      const float* o = gdxy + 6 * ((size_t)i * W + pj);
      float sgn = (clampi(pj + 1, 0, W - 1) == j ? 1.f : 0.f) - (clampi(pj - 1, 0, W - 1) == j ? 1.f : 0.f);
      g[0] += sgn * o[0]; g[1] += sgn * o[1]; g[2] += sgn * o[2];
    }
  }
  for (int pi = max(i - 1, r0); pi <= min(i + 1, r1 - 1); ++pi) {
    const float* o = gdxy + 6 * ((size_t)pi * W + j) + 3;
    float sgn = (clampi(pi + 1, 0, H - 1) == i ? 1.f : 0.f) - (clampi(pi - 1, 0, H - 1) == i ? 1.f : 0.f);
    g[0] += sgn * o[0]; g[1] += sgn * o[1]; g[2] += sgn * o[2];
  }
  float xr = ((float)j - k.cx) / k.fx, yr = ((float)i - k.cy) / k.fy;
  v_render[q * D + (D - 1)] += g[0] * xr + g[1] * yr + g[2];
}

// ------------------------------------------------------------------------------------------------
// Pose state (device resident, one per tracker):
//   f[0..3] quat (wxyz)  f[4..6] t      f[7..13] Adam exp_avg   f[14..20] Adam exp_avg_sq
//   f[21] lr_quat f[22] lr_trans        f[23] best_loss f[24] best_depth f[25] best_edge
//   f[26] best_eT f[27] best_eR         f[28] last_loss f[29] last_eT f[30] last_eR
//   i[0] step  i[1] counter  i[2] stopped  i[3] best_step
// ------------------------------------------------------------------------------------------------
struct PoseHyper {
  float beta1, beta2, eps, wd_quat, wd_trans, gamma, depth_w, edge_w, inv_P, normal_w, inv_3H;
  int min_step, patience, early_stop, max_steps;
};

__device__ __forceinline__ void quat_to_R(const float q[4], float R[9], float qh[4], float& qn) {
  qn = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float inv = 1.f / fmaxf(qn, 1e-12f);
  float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
  qh[0] = w; qh[1] = x; qh[2] = y; qh[3] = z;
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// c2w = [R(q^)|t]; viewmat = c2w^-1 = [R^T | -R^T t] (rows 0..2; row 3 = 0 0 0 1)
__device__ __forceinline__ void write_pose(const float q[4], const float t[3], float* c2w, float* viewmat) {
  float R[9], qh[4], qn;
  quat_to_R(q, R, qh, qn);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      c2w[r * 4 + c] = R[r * 3 + c];
      viewmat[r * 4 + c] = R[c * 3 + r];
    }
    c2w[r * 4 + 3] = t[r];
    viewmat[r * 4 + 3] = -(R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1] + R[2 * 3 + r] * t[2]);
  }
  for (int c = 0; c < 4; ++c) { c2w[12 + c] = (c == 3) ? 1.f : 0.f; viewmat[12 + c] = (c == 3) ? 1.f : 0.f; }
}

__global__ void k_pose_init(float* __restrict__ f, int* __restrict__ istate, const float* __restrict__ init_c2w,
                            float lr_quat, float lr_trans, float* __restrict__ c2w, float* __restrict__ viewmat) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // rotation matrix -> wxyz quaternion (kornia rotation_matrix_to_quaternion, eps = 1e-8)
  const float* m = init_c2w;
  float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[4], m11 = m[5], m12 = m[6], m20 = m[8], m21 = m[9], m22 = m[10];
  float trace = m00 + m11 + m22, q[4];
  const float eps = 1e-8f;
  if (trace > 0.f) {
    float sq = sqrtf(trace + 1.f + eps) * 2.f;
    q[0] = 0.25f * sq; q[1] = (m21 - m12) / sq; q[2] = (m02 - m20) / sq; q[3] = (m10 - m01) / sq;
  } else if (m00 > m11 && m00 > m22) {
    float sq = sqrtf(1.f + m00 - m11 - m22 + eps) * 2.f;
    q[0] = (m21 - m12) / sq; q[1] = 0.25f * sq; q[2] = (m01 + m10) / sq; q[3] = (m02 + m20) / sq;
  } else if (m11 > m22) {
    float sq = sqrtf(1.f + m11 - m00 - m22 + eps) * 2.f;
    q[0] = (m02 - m20) / sq; q[1] = (m01 + m10) / sq; q[2] = 0.25f * sq; q[3] = (m12 + m21) / sq;
  } else {
    float sq = sqrtf(1.f + m22 - m00 - m11 + eps) * 2.f;
    q[0] = (m10 - m01) / sq; q[1] = (m02 + m20) / sq; q[2] = (m12 + m21) / sq; q[3] = 0.25f * sq;
  }
  float t[3] = {m[3], m[7], m[11]};
  for (int k = 0; k < 4; ++k) f[k] = q[k];
  for (int k = 0; k < 3; ++k) f[4 + k] = t[k];
  for (int k = 7; k < 21; ++k) f[k] = 0.f;
  f[21] = lr_quat; f[22] = lr_trans;
  const float inf = __builtin_huge_valf();
  for (int k = 23; k < 31; ++k) f[k] = inf;
  istate[0] = 0; istate[1] = 0; istate[2] = 0; istate[3] = -1;
  write_pose(q, t, c2w, viewmat);
}

// One optimisation step.  v_viewmat[16] = d loss / d viewmat (rows 0..2) summed over ranks already -- or, one rank,
// vm_rows[n_vm][16]: the partial rows the projection backward left (gsl_fused_project_bwd, reduce_viewmat = 0), summed
// here in a fixed order (saves the launch of the separate reduction); Kmat and the CURRENT viewmat are then read too.
// partial[nb][2] loss sums of this rank (loss_sums_in != null: already reduced (sum_l1, sum_edge) over ranks).
__global__ __launch_bounds__(256) void k_pose_step(float* __restrict__ f, int* __restrict__ istate,
                                                   const float* __restrict__ v_viewmat,
                                                   const float* __restrict__ vm_rows, int n_vm,
                                                   const float* __restrict__ Kmat,
                                                   const float* __restrict__ partial, int nb,
                                                   const float* __restrict__ loss_sums_in,
                                                   const float* __restrict__ normal_sum,
                                                   const float* __restrict__ gt_c2w, PoseHyper hp,
                                                   float* __restrict__ c2w, float* __restrict__ viewmat,
                                                   float* __restrict__ loss_hist) {
  __shared__ float red[4][2];
  __shared__ float sums[2];
  __shared__ float vred[4][15];
  __shared__ float vtot[15];
  __shared__ float svm[16];
  // the state thread 0 works on (pose, Adam moments, learning rates, best-so-far, ground truth, counters) is fetched by
  // three groups of lanes while the rows are being summed: read one after the other by the lone thread these were
  // a dozen dependent global loads of the 9 us this kernel takes
  __shared__ float sF[32], sG[16];
  __shared__ int sI[4];
  if (threadIdx.x >= 64 && threadIdx.x < 96) sF[threadIdx.x - 64] = f[threadIdx.x - 64];
  else if (threadIdx.x >= 96 && threadIdx.x < 112) sG[threadIdx.x - 96] = gt_c2w[threadIdx.x - 96];
  else if (threadIdx.x >= 112 && threadIdx.x < 116) sI[threadIdx.x - 112] = istate[threadIdx.x - 112];
  if (vm_rows) {  // `viewmat` still holds the pose this iteration rendered with (thread 0 overwrites it at the very end)
    float v = reduce_viewmat_rows(vm_rows, n_vm, viewmat, Kmat, vred, vtot);
    if (threadIdx.x < 16) svm[threadIdx.x] = v;
  } else if (threadIdx.x < 16) {
    svm[threadIdx.x] = threadIdx.x < 12 ? v_viewmat[threadIdx.x] : 0.f;
  }
  float a0 = 0.f, a1 = 0.f;
  if (loss_sums_in == nullptr) {
    for (int b = threadIdx.x; b < nb; b += 256) { a0 += partial[2 * (size_t)b]; a1 += partial[2 * (size_t)b + 1]; }
  }
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float s0 = wave_sum(a0), s1 = wave_sum(a1);
  if (lane == 0) { red[wv][0] = s0; red[wv][1] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (loss_sums_in) { sums[0] = loss_sums_in[0]; sums[1] = loss_sums_in[1]; }
    else { sums[0] = red[0][0] + red[1][0] + red[2][0] + red[3][0]; sums[1] = red[0][1] + red[1][1] + red[2][1] + red[3][1]; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (sI[2]) return;  // stopped: keep the state frozen (graph replays may still arrive)
  int step = sI[0];
  float depth_loss = sums[0] * hp.inv_P, edge_loss = sums[1] * hp.inv_P;
  float total = hp.depth_w * depth_loss + hp.edge_w * edge_loss;
  if (hp.normal_w != 0.f) {  // sum of the row cosines: this rank's (normal_sum) or all ranks' (loss_sums_in[2])
    float cs = loss_sums_in ? loss_sums_in[2] : (normal_sum ? normal_sum[0] : 0.f);
    total += hp.normal_w * (1.f - cs * hp.inv_3H);
  }
  float q[4] = {sF[0], sF[1], sF[2], sF[3]}, t[3] = {sF[4], sF[5], sF[6]};
  float R[9], qh[4], qn;
  quat_to_R(q, R, qh, qn);
  // pose errors of the CURRENT pose vs ground truth (eval/utils.py:122-168)
  float dt0 = t[0] - sG[3], dt1 = t[1] - sG[7], dt2 = t[2] - sG[11];
  float eT = sqrtf(dt0 * dt0 + dt1 * dt1 + dt2 * dt2);
  // rotation angle of R_est R_gt^T (eval/utils.py:144-168 takes acos((trace - 1) / 2)).  In float32 that form
  // resolves angles only down to sqrt(2 * 6e-8) rad = 0.02 degrees -- the level of the reference's own AAE table --
  // so the same angle is computed from the difference of the matrices:  |R_est - R_gt|_F^2 = 8 sin^2(theta / 2).
  float fro = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float d = R[i * 3 + j] - sG[i * 4 + j];
      fro += d * d;
    }
  float eR = 2.f * asinf(fminf(1.f, sqrtf(fro * 0.125f))) * 57.29577951308232f;
  f[28] = total; f[29] = eT; f[30] = eR;
  if (loss_hist && step < hp.max_steps) loss_hist[step] = total;
  if (hp.early_stop && step > hp.min_step) {
    if (total < sF[23]) {
      f[23] = total; f[24] = depth_loss; f[25] = edge_loss; f[26] = eT; f[27] = eR;
      sI[1] = 0; istate[3] = step;
    } else {
      sI[1] += 1;
    }
    istate[1] = sI[1];
  }
  istate[0] = step + 1;
  if (hp.early_stop && sI[1] >= hp.patience) { istate[2] = 1; return; }  // stop BEFORE the optimiser step
  if (step + 1 >= hp.max_steps) istate[2] = 1;  // last iteration still takes its optimiser step
  // ---- pose chain: viewmat = inv(c2w) => v_c2w = -V^T v_V V^T
  float V[16], vV[16];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) V[r * 4 + c] = R[c * 3 + r];
    V[r * 4 + 3] = -(R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1] + R[2 * 3 + r] * t[2]);
  }
  V[12] = V[13] = V[14] = 0.f; V[15] = 1.f;
  for (int k = 0; k < 12; ++k) vV[k] = svm[k];
  vV[12] = vV[13] = vV[14] = vV[15] = 0.f;
  float M1[16];  // V^T vV
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) {
      float s = 0.f;
      for (int i = 0; i < 4; ++i) s += V[i * 4 + a] * vV[i * 4 + b];
      M1[a * 4 + b] = s;
    }
  float vC[16];  // -(M1 V^T)
  for (int a = 0; a < 4; ++a)
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      for (int b = 0; b < 4; ++b) s += M1[a * 4 + b] * V[c * 4 + b];
      vC[a * 4 + c] = -s;
    }
  float vR[9], vt[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) vR[r * 3 + c] = vC[r * 4 + c];
    vt[r] = vC[r * 4 + 3];
  }
  float w = qh[0], x = qh[1], y = qh[2], z = qh[3];
  float gq[4];
  gq[0] = 2.f * (-z * vR[1] + y * vR[2] + z * vR[3] - x * vR[5] - y * vR[6] + x * vR[7]);
  gq[1] = 2.f * (y * vR[1] + z * vR[2] + y * vR[3] - 2.f * x * vR[4] - w * vR[5] + z * vR[6] + w * vR[7] - 2.f * x * vR[8]);
  gq[2] = 2.f * (-2.f * y * vR[0] + x * vR[1] + w * vR[2] + x * vR[3] + z * vR[5] - w * vR[6] + z * vR[7] - 2.f * y * vR[8]);
  gq[3] = 2.f * (-2.f * z * vR[0] - w * vR[1] + x * vR[2] + w * vR[3] - 2.f * z * vR[4] + y * vR[5] + x * vR[6] + y * vR[7]);
  float dq = gq[0] * w + gq[1] * x + gq[2] * y + gq[3] * z;
  float grad[7];
  for (int k = 0; k < 4; ++k) grad[k] = (gq[k] - dq * qh[k]) / fmaxf(qn, 1e-12f);
  for (int k = 0; k < 3; ++k) grad[4 + k] = vt[k];
  // ---- Adam (torch.optim.Adam semantics, weight decay added to the gradient), per-group lr
  float b1p = powf(hp.beta1, (float)(step + 1)), b2p = powf(hp.beta2, (float)(step + 1));
  float bc1 = 1.f - b1p, bc2s = sqrtf(1.f - b2p);
  for (int k = 0; k < 7; ++k) {
    float p = (k < 4) ? q[k] : t[k - 4];
    float wd = (k < 4) ? hp.wd_quat : hp.wd_trans;
    float lr = (k < 4) ? sF[21] : sF[22];
    float g = grad[k] + wd * p;
    float m = sF[7 + k] * hp.beta1 + (1.f - hp.beta1) * g;
    float v = sF[14 + k] * hp.beta2 + (1.f - hp.beta2) * g * g;
    f[7 + k] = m; f[14 + k] = v;
    float denom = sqrtf(v) / bc2s + hp.eps;
    p -= (lr / bc1) * (m / denom);
    if (k < 4) q[k] = p; else t[k - 4] = p;
  }
  for (int k = 0; k < 4; ++k) f[k] = q[k];
  for (int k = 0; k < 3; ++k) f[4 + k] = t[k];
  f[21] = sF[21] * hp.gamma; f[22] = sF[22] * hp.gamma;  // ExponentialLR
  write_pose(q, t, c2w, viewmat);
}

// The 16 floats one rank contributes to the per-iteration all-reduce: 12 pose-gradient entries (rows 0..2 of
// d loss / d viewmat), its two loss sums (fixed-order sum of its block partials) and, in entry 14, its sum of row
// cosines of the normal-consistency term (0 when that term is off); entry 15 is zero.
__global__ __launch_bounds__(256) void k_pack_pose_reduce(const float* __restrict__ v_viewmat,
                                                          const float* __restrict__ vm_rows, int n_vm,
                                                          const float* __restrict__ V, const float* __restrict__ Kmat,
                                                          const float* __restrict__ partial, int nb,
                                                          const float* __restrict__ normal_sum,
                                                          float* __restrict__ out16) {
  __shared__ float red[4][2];
  __shared__ float vred[4][15];
  __shared__ float vtot[15];
  float vm = 0.f;
  if (vm_rows) vm = reduce_viewmat_rows(vm_rows, n_vm, V, Kmat, vred, vtot);
  else if (threadIdx.x < 12) vm = v_viewmat[threadIdx.x];
  float a0 = 0.f, a1 = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) { a0 += partial[2 * (size_t)b]; a1 += partial[2 * (size_t)b + 1]; }
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float s0 = wave_sum(a0), s1 = wave_sum(a1);
  if (lane == 0) { red[wv][0] = s0; red[wv][1] = s1; }
  __syncthreads();
  if (threadIdx.x < 12) out16[threadIdx.x] = vm;
  else if (threadIdx.x < 14)
    out16[threadIdx.x] = red[0][threadIdx.x - 12] + red[1][threadIdx.x - 12] + red[2][threadIdx.x - 12] + red[3][threadIdx.x - 12];
  else if (threadIdx.x == 14) out16[14] = normal_sum ? normal_sum[0] : 0.f;
  else if (threadIdx.x == 15) out16[15] = 0.f;
}

}  // namespace gsl

extern "C" int gsl_pack_pose_reduce(const float* v_viewmat, const float* vm_rows, int n_vm_rows, const float* viewmat,
                                    const float* K, const float* loss_partials, int n_partials,
                                    const float* normal_sum, float* out16, void* stream) {
  if (!out16 || n_partials < 0 || (n_partials > 0 && !loss_partials)) return GSL_ERR_BAD_ARG;
  if (!v_viewmat && !vm_rows) return GSL_ERR_BAD_ARG;
  if (vm_rows && (n_vm_rows < 0 || !viewmat || !K)) return GSL_ERR_BAD_ARG;
  hipLaunchKernelGGL(gsl::k_pack_pose_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, v_viewmat, vm_rows,
                     n_vm_rows, viewmat, K, loss_partials, n_partials, normal_sum, out16);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// number of (sum|a-b|, sum|Sa-Sb|) rows gsl_tracking_loss writes for the owned rows [row0,row1)
extern "C" int gsl_loss_n_partials(int width, int height, int row0, int row1) {
  if (width <= 0 || height <= 0 || row0 < 0 || row1 > height || row0 >= row1) return 0;
  int h0 = row0 > 0 ? row0 - 1 : 0, h1 = row1 < height ? row1 + 1 : height;
  return ((width + 15) / 16) * ((h1 - h0 + 15) / 16);
}

extern "C" size_t gsl_loss_ws_bytes(int width, int height) {
  int nb = gsl_loss_n_partials(width, height, 0, height);
  return (size_t)(nb > 0 ? nb : 1) * 2 * sizeof(float);
}

extern "C" int gsl_tracking_loss(const float* render, int channels, const float* depth_gt, int width, int height,
                                 int row0, int row1, float depth_lambda, float edge_lambda, float* v_render,
                                 float* loss_partials, int* n_partials_host, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (width <= 0 || height <= 0 || channels <= 0 || row0 < 0 || row1 > height || row0 > row1) return GSL_ERR_BAD_ARG;
  if (!render || !depth_gt || !v_render || (!ws && !loss_partials)) return GSL_ERR_BAD_ARG;
  if (!loss_partials && ws_bytes < gsl_loss_ws_bytes(width, height)) return GSL_ERR_WORKSPACE;
  float* partial = loss_partials ? loss_partials : (float*)ws;
  int nb = gsl_loss_n_partials(width, height, row0, row1);
  if (n_partials_host) *n_partials_host = nb;
  if (nb == 0) return GSL_OK;
  int h0 = row0 > 0 ? row0 - 1 : 0, h1 = row1 < height ? row1 + 1 : height;
  dim3 grid((width + 15) / 16, (h1 - h0 + 15) / 16);
  hipLaunchKernelGGL(gsl::k_loss_fused, grid, dim3(256), 0, (hipStream_t)stream, render, channels, depth_gt, width,
                     height, row0, row1, h0, h1, depth_lambda, edge_lambda, 1.0f / ((float)width * (float)height),
                     v_render, partial);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" size_t gsl_normal_ws_bytes(int width, int height) {
  size_t P = (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0);
  return (P * 6 + (size_t)(height > 0 ? height : 0) * 10) * sizeof(float);
}

extern "C" int gsl_normal_loss(const float* render, int channels, const float* depth_gt, int width, int height,
                               int row0, int row1, float fx, float fy, float cx, float cy, float normal_lambda,
                               float* v_render, float* normal_sum, void* ws, size_t ws_bytes, void* stream) {
  if (width <= 0 || height <= 0 || channels <= 0 || row0 < 0 || row1 > height || row0 > row1) return GSL_ERR_BAD_ARG;
  if (!render || !depth_gt || !v_render || !normal_sum || !ws || fx == 0.f || fy == 0.f) return GSL_ERR_BAD_ARG;
  if (ws_bytes < gsl_normal_ws_bytes(width, height)) return GSL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  size_t P = (size_t)width * height;
  float* gdxy = (float*)ws;
  float* rowstat = gdxy + 6 * P;
  gsl::Intrin k{fx, fy, cx, cy};
  if (row1 == row0) {  // empty strip: contributes 0
    hipLaunchKernelGGL(gsl::k_normal_sum, dim3(1), dim3(256), 0, st, rowstat, 0, 0, normal_sum);
    GSL_CHECK_LAUNCH();
    return GSL_OK;
  }
  hipLaunchKernelGGL(gsl::k_normal_rows, dim3(row1 - row0), dim3(256), 0, st, render, channels, depth_gt, width, height,
                     row0, k, rowstat);
  GSL_CHECK_LAUNCH();
  hipLaunchKernelGGL(gsl::k_normal_sum, dim3(1), dim3(256), 0, st, rowstat, row0, row1, normal_sum);
  GSL_CHECK_LAUNCH();
  int nown = (row1 - row0) * width;
  hipLaunchKernelGGL(gsl::k_normal_grad, dim3((nown + 255) / 256), dim3(256), 0, st, render, channels, depth_gt, width,
                     height, row0, row1, k, rowstat, normal_lambda / (3.0f * (float)height), gdxy);
  GSL_CHECK_LAUNCH();
  int h0 = row0 > 0 ? row0 - 1 : 0, h1 = row1 < height ? row1 + 1 : height;
  int n2 = (h1 - h0) * width;
  hipLaunchKernelGGL(gsl::k_normal_gather, dim3((n2 + 255) / 256), dim3(256), 0, st, render, channels, width, height,
                     row0, row1, h0, h1, k, gdxy, v_render);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" int gsl_pose_init(float* pose_f, int* pose_i, const float* init_c2w, float lr_quat, float lr_trans,
                             float* c2w, float* viewmat, void* stream) {
  if (!pose_f || !pose_i || !init_c2w || !c2w || !viewmat) return GSL_ERR_BAD_ARG;
  hipLaunchKernelGGL(gsl::k_pose_init, dim3(1), dim3(64), 0, (hipStream_t)stream, pose_f, pose_i, init_c2w, lr_quat,
                     lr_trans, c2w, viewmat);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

extern "C" int gsl_pose_step(float* pose_f, int* pose_i, const float* v_viewmat, const float* vm_rows, int n_vm_rows,
                             const float* K, const float* loss_partials, int n_partials, const float* loss_sums, const float* normal_sum, const float* gt_c2w,
                             int width, int height, float depth_lambda, float edge_lambda, float normal_lambda,
                             float beta1, float beta2, float eps,
                             float wd_quat, float wd_trans, float gamma, int min_step, int patience, int early_stop,
                             int max_steps, float* c2w, float* viewmat, float* loss_hist, void* stream) {
  if (!pose_f || !pose_i || !gt_c2w || !c2w || !viewmat) return GSL_ERR_BAD_ARG;
  if (!v_viewmat && !vm_rows) return GSL_ERR_BAD_ARG;
  if (vm_rows && (n_vm_rows < 0 || !K)) return GSL_ERR_BAD_ARG;
  if (!loss_partials && !loss_sums) return GSL_ERR_BAD_ARG;
  if (width <= 0 || height <= 0 || n_partials < 0) return GSL_ERR_BAD_ARG;
  gsl::PoseHyper hp;
  hp.beta1 = beta1; hp.beta2 = beta2; hp.eps = eps; hp.wd_quat = wd_quat; hp.wd_trans = wd_trans; hp.gamma = gamma;
  hp.depth_w = depth_lambda; hp.edge_w = edge_lambda; hp.inv_P = 1.0f / ((float)width * (float)height);
  hp.normal_w = normal_lambda; hp.inv_3H = 1.0f / (3.0f * (float)height);
  hp.min_step = min_step; hp.patience = patience; hp.early_stop = early_stop; hp.max_steps = max_steps;
  hipLaunchKernelGGL(gsl::k_pose_step, dim3(1), dim3(256), 0, (hipStream_t)stream, pose_f, pose_i, v_viewmat,
                     vm_rows, n_vm_rows, K, loss_partials, n_partials, loss_sums, normal_sum, gt_c2w, hp, c2w, viewmat, loss_hist);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
