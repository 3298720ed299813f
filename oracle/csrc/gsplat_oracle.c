/* C restatement of the gsplat 1.3.0 operators on GsplatLoc's hot path -- forward AND hand-derived backward.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg, never by the product.  PARITY UNPINNED: gsplat's sources are not in the
 * reference tree and none of its tests holds a fixture for this path; the algorithm is restated from the
 * published kernels and pinned against oracle/gsplat_oracle.py (autograd, float64).
 *
 * What it stands in for, one camera (reference call sites /root/reference/src/my_gsplat/model.py:195-213,
 * geometry.py:117-132; operator signatures in /root/reference/.vscode/PythonImportHelper-v2-Completion.json):
 *   fully_fused_projection fwd/bwd (IDX:14351 / 14270), spherical_harmonics fwd/bwd (14306 / 14297),
 *   isect_tiles + isect_offset_encode (14360 / 14369), rasterize_to_pixels fwd/bwd (14378 / 14279),
 *   the expected-depth normalisation at the tail of rasterization (14954).
 *
 * Built twice from this file: -DGSO_REAL=double (the checker) and -DGSO_REAL=float (the CPU baseline that
 * bench.py times).  OpenMP over Gaussians / tiles; gradient sums use per-tile partial rows added with
 * atomics, so float64 results agree to rounding, not bit for bit, between runs.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef GSO_REAL
#define GSO_REAL double
#endif
typedef GSO_REAL real;

#define R_(x) ((real)(x))
static inline real r_sqrt(real x) { return sizeof(real) == 4 ? (real)sqrtf((float)x) : (real)sqrt((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_ceil(real x) { return sizeof(real) == 4 ? (real)ceilf((float)x) : (real)ceil((double)x); }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_max(real a, real b) { return a > b ? a : b; }

/* gsplat's kernels spell these thresholds as float32 literals (0.999f, 1.f / 255.f): the float64 build uses the
 * float32-representable values, so that a pixel at the clamp takes 1 - alpha = 1 - 0.999f like every float32
 * implementation does (1 - 0.999 differs from it by 1.3e-5 relative, which would show up as a transmittance error) */
#define ALPHA_MAX ((GSO_REAL)0.999f)
#define ALPHA_MIN ((GSO_REAL)(1.0f / 255.0f))
#define T_STOP R_(1e-4)
#define RADIUS_LAMBDA_FLOOR R_(0.01)
#define FOV_LIM R_(1.3)
#define ED_CLAMP R_(1e-10)

enum { GSO_OK = 0, GSO_BAD_ARG = -1, GSO_NO_MEM = -2 };
enum { GSO_COLOR_NONE = 0, GSO_COLOR_SH = 1, GSO_COLOR_RGB = 2 };
enum { GSO_DEPTH_NONE = 0, GSO_DEPTH_ACC = 1, GSO_DEPTH_EXPECTED = 2 };

int gso_real_bytes(void) { return (int)sizeof(real); }

/* GSO_TIMING=1 in the environment: stage times of gso_rasterization on stderr. */
#include <stdio.h>
static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}
static void lap(const char* what, double* t0) {
  if (!getenv("GSO_TIMING")) return;
  double t = now_s();
  fprintf(stderr, "[gso] %-14s %8.1f ms\n", what, (t - *t0) * 1e3);
  *t0 = t;
}
void gso_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------ small linear algebra */
static void mat3_mul(const real* a, const real* b, real* c) { /* c = a b, row-major 3x3 */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static void mat3_mul_bt(const real* a, const real* b, real* c) { /* c = a b^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      c[3 * i + j] = a[3 * i] * b[3 * j] + a[3 * i + 1] * b[3 * j + 1] + a[3 * i + 2] * b[3 * j + 2];
}
static void mat3_mul_at(const real* a, const real* b, real* c) { /* c = a^T b */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
}
static int mat3_inv(const real* m, real* o) {
  real c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  real det = m[0] * c0 + m[1] * c1 + m[2] * c2;
  if (det == 0) return 0;
  real id = R_(1.0) / det;
  o[0] = c0 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c1 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c2 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return 1;
}
static void quat_to_rotmat(const real* q, real* R, real* qn_out, real* norm_out) {
  real n = r_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  real w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
  if (qn_out) { qn_out[0] = w; qn_out[1] = x; qn_out[2] = y; qn_out[3] = z; }
  if (norm_out) *norm_out = n;
}

/* Everything the projection of one Gaussian computes on the way (shared by forward and backward). */
typedef struct {
  real Rq[9], M[9], S[9]; /* rotation of the Gaussian, R*diag(s), covariance */
  real mc[3], Sc[9];      /* camera-frame mean and covariance */
  real J[6];              /* 2x3 EWA Jacobian */
  real tx, ty;            /* clamped z*x/z, z*y/z */
  int in_x, in_y;         /* tangent inside the frustum clamp */
  real c00, c01, c11;     /* cov2d before the blur */
  real a, b, c, det, det_orig, comp;
  real m2[2];
} Proj;

static void project_one(const real* mean, const real* quat, const real* scale, const real* V, const real* K, int W,
                        int H, real eps2d, Proj* p) {
  const real Rv[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
  const real t[3] = {V[3], V[7], V[11]};
  quat_to_rotmat(quat, p->Rq, 0, 0);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) p->M[3 * i + j] = p->Rq[3 * i + j] * scale[j];
  mat3_mul_bt(p->M, p->M, p->S);
  for (int i = 0; i < 3; ++i) p->mc[i] = Rv[3 * i] * mean[0] + Rv[3 * i + 1] * mean[1] + Rv[3 * i + 2] * mean[2] + t[i];
  real tmp[9];
  mat3_mul(Rv, p->S, tmp);
  mat3_mul_bt(tmp, Rv, p->Sc);
  real fx = K[0], fy = K[4], cx = K[2], cy = K[5];
  real x = p->mc[0], y = p->mc[1], z = p->mc[2];
  real limx = FOV_LIM * (R_(0.5) * (real)W / fx), limy = FOV_LIM * (R_(0.5) * (real)H / fy);
  real rz = R_(1.0) / z, rz2 = rz * rz;
  real xr = x * rz, yr = y * rz;
  p->in_x = (xr >= -limx) && (xr <= limx);
  p->in_y = (yr >= -limy) && (yr <= limy);
  p->tx = z * r_min(limx, r_max(-limx, xr));
  p->ty = z * r_min(limy, r_max(-limy, yr));
  real* J = p->J;
  J[0] = fx * rz; J[1] = 0; J[2] = -fx * p->tx * rz2;
  J[3] = 0; J[4] = fy * rz; J[5] = -fy * p->ty * rz2;
  /* cov2d = J Sc J^T */
  real JS[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) JS[3 * i + j] = J[3 * i] * p->Sc[j] + J[3 * i + 1] * p->Sc[3 + j] + J[3 * i + 2] * p->Sc[6 + j];
  p->c00 = JS[0] * J[0] + JS[1] * J[1] + JS[2] * J[2];
  p->c01 = JS[0] * J[3] + JS[1] * J[4] + JS[2] * J[5];
  p->c11 = JS[3] * J[3] + JS[4] * J[4] + JS[5] * J[5];
  p->det_orig = p->c00 * p->c11 - p->c01 * p->c01;
  p->a = p->c00 + eps2d;
  p->c = p->c11 + eps2d;
  p->b = p->c01;
  p->det = p->a * p->c - p->b * p->b;
  real ratio = p->det_orig / p->det;
  p->comp = r_sqrt(ratio > 0 ? ratio : 0);
  p->m2[0] = fx * x * rz + cx;
  p->m2[1] = fy * y * rz + cy;
}

/* fully_fused_projection forward (A.1).  Outputs are zero where radii == 0.  comps may be NULL. */
int gso_project_fwd(const real* means, const real* quats, const real* scales, const real* viewmat, const real* K,
                    int N, int W, int H, real eps2d, real near_plane, real far_plane, real radius_clip,
                    int32_t* radii, real* means2d, real* depths, real* conics, real* comps) {
  if (N < 0 || W <= 0 || H <= 0 || !viewmat || !K || (N > 0 && (!means || !quats || !scales || !radii || !means2d || !depths || !conics)))
    return GSO_BAD_ARG;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    radii[i] = 0;
    means2d[2 * i] = means2d[2 * i + 1] = 0;
    depths[i] = 0;
    conics[3 * i] = conics[3 * i + 1] = conics[3 * i + 2] = 0;
    if (comps) comps[i] = 0;
    const real* m = means + 3 * (size_t)i;
    real z = viewmat[8] * m[0] + viewmat[9] * m[1] + viewmat[10] * m[2] + viewmat[11];
    if (!(z >= near_plane && z <= far_plane)) continue;
    Proj p;
    project_one(m, quats + 4 * (size_t)i, scales + 3 * (size_t)i, viewmat, K, W, H, eps2d, &p);
    if (!(p.det > 0)) continue;
    real bb = R_(0.5) * (p.a + p.c);
    real v1 = bb + r_sqrt(r_max(bb * bb - p.det, RADIUS_LAMBDA_FLOOR));
    real radius = r_ceil(R_(3.0) * r_sqrt(v1));
    if (!(radius > radius_clip)) continue;
    if (p.m2[0] + radius <= 0 || p.m2[0] - radius >= (real)W || p.m2[1] + radius <= 0 || p.m2[1] - radius >= (real)H)
      continue;
    radii[i] = (int32_t)radius;
    means2d[2 * i] = p.m2[0];
    means2d[2 * i + 1] = p.m2[1];
    depths[i] = p.mc[2];
    conics[3 * i] = p.c / p.det;
    conics[3 * i + 1] = -p.b / p.det;
    conics[3 * i + 2] = p.a / p.det;
    if (comps) comps[i] = p.comp;
  }
  return GSO_OK;
}

/* vjp of Sigma = (R diag(s))(R diag(s))^T to (quaternion, scale); quaternion normalised inside. */
static void covar_vjp(const real* quat, const real* scale, const Proj* p, const real* vS, real* v_quat, real* v_scale) {
  real qn[4], norm;
  real Rq[9];
  quat_to_rotmat(quat, Rq, qn, &norm);
  real sym[9], vM[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) sym[3 * i + j] = vS[3 * i + j] + vS[3 * j + i];
  mat3_mul(sym, p->M, vM);
  real vR[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) vR[3 * i + j] = vM[3 * i + j] * scale[j];
  for (int k = 0; k < 3; ++k) v_scale[k] = Rq[k] * vM[k] + Rq[3 + k] * vM[3 + k] + Rq[6 + k] * vM[6 + k];
  real w = qn[0], x = qn[1], y = qn[2], z = qn[3];
#define VR(i, j) vR[3 * (i) + (j)]
  real g[4];
  g[0] = 2 * (x * (VR(2, 1) - VR(1, 2)) + y * (VR(0, 2) - VR(2, 0)) + z * (VR(1, 0) - VR(0, 1)));
  g[1] = 2 * (-2 * x * (VR(1, 1) + VR(2, 2)) + y * (VR(1, 0) + VR(0, 1)) + z * (VR(2, 0) + VR(0, 2)) + w * (VR(2, 1) - VR(1, 2)));
  g[2] = 2 * (x * (VR(1, 0) + VR(0, 1)) - 2 * y * (VR(0, 0) + VR(2, 2)) + z * (VR(2, 1) + VR(1, 2)) + w * (VR(0, 2) - VR(2, 0)));
  g[3] = 2 * (x * (VR(2, 0) + VR(0, 2)) + y * (VR(2, 1) + VR(1, 2)) - 2 * z * (VR(0, 0) + VR(1, 1)) + w * (VR(1, 0) - VR(0, 1)));
#undef VR
  real d = g[0] * qn[0] + g[1] * qn[1] + g[2] * qn[2] + g[3] * qn[3];
  for (int k = 0; k < 4; ++k) v_quat[k] = (g[k] - d * qn[k]) / norm;
}

/* fully_fused_projection backward (A.5): gradients of (means2d, depths, conics, compensations) pulled back to
 * means [N,3], quats [N,4], scales [N,3] (any may be NULL) and the view matrix (v_viewmat[16], row 3 zero;
 * ACCUMULATED into, so the caller zeroes it).  v_comps may be NULL. */
int gso_project_bwd(const real* means, const real* quats, const real* scales, const real* viewmat, const real* K,
                    int N, int W, int H, real eps2d, const int32_t* radii, const real* v_means2d,
                    const real* v_depths, const real* v_conics, const real* v_comps, real* v_means, real* v_quats,
                    real* v_scales, real* v_viewmat) {
  if (N < 0 || !viewmat || !K || !v_viewmat || (N > 0 && (!means || !quats || !scales || !radii || !v_means2d || !v_depths || !v_conics)))
    return GSO_BAD_ARG;
  const real Rv[9] = {viewmat[0], viewmat[1], viewmat[2], viewmat[4], viewmat[5], viewmat[6], viewmat[8], viewmat[9], viewmat[10]};
  real fx = K[0], fy = K[4];
  real accR[9] = {0}, acct[3] = {0};
#pragma omp parallel
  {
    real locR[9] = {0}, loct[3] = {0};
#pragma omp for schedule(static)
    for (int i = 0; i < N; ++i) {
      if (v_means) v_means[3 * (size_t)i] = v_means[3 * (size_t)i + 1] = v_means[3 * (size_t)i + 2] = 0;
      if (v_quats) for (int k = 0; k < 4; ++k) v_quats[4 * (size_t)i + k] = 0;
      if (v_scales) for (int k = 0; k < 3; ++k) v_scales[3 * (size_t)i + k] = 0;
      if (radii[i] <= 0) continue;
      const real* m = means + 3 * (size_t)i;
      Proj p;
      project_one(m, quats + 4 * (size_t)i, scales + 3 * (size_t)i, viewmat, K, W, H, eps2d, &p);
      /* conic = inverse of [[a b][b c]]: G = -C Vc C with Vc = [[v0, v1/2],[v1/2, v2]] */
      real A = p.c / p.det, B = -p.b / p.det, Cc = p.a / p.det;
      real v0 = v_conics[3 * (size_t)i], v1 = R_(0.5) * v_conics[3 * (size_t)i + 1], v2 = v_conics[3 * (size_t)i + 2];
      real t00 = A * v0 + B * v1, t01 = A * v1 + B * v2, t10 = B * v0 + Cc * v1, t11 = B * v1 + Cc * v2;
      real G00 = -(t00 * A + t01 * B), G01 = -(t00 * B + t01 * Cc), G10 = -(t10 * A + t11 * B), G11 = -(t10 * B + t11 * Cc);
      if (v_comps) {
        real ratio = p.det_orig / p.det;
        if (ratio > 0) {
          real vr = v_comps[i] * R_(0.5) / p.comp;
          real id = R_(1.0) / p.det, r2 = p.det_orig * id * id;
          G00 += vr * (p.c11 * id - r2 * p.c);
          G11 += vr * (p.c00 * id - r2 * p.a);
          real off = vr * (-p.c01 * id + r2 * p.c01);
          G01 += off;
          G10 += off;
        }
      }
      const real* J = p.J;
      /* v_Sc = J^T G J (3x3) ; v_J = G J Sc^T + G^T J Sc (2x3) */
      real GJ[6], GtJ[6];
      for (int j = 0; j < 3; ++j) {
        GJ[j] = G00 * J[j] + G01 * J[3 + j];
        GJ[3 + j] = G10 * J[j] + G11 * J[3 + j];
        GtJ[j] = G00 * J[j] + G10 * J[3 + j];
        GtJ[3 + j] = G01 * J[j] + G11 * J[3 + j];
      }
      real vSc[9];
      for (int a2 = 0; a2 < 3; ++a2)
        for (int b2 = 0; b2 < 3; ++b2) vSc[3 * a2 + b2] = J[a2] * GJ[b2] + J[3 + a2] * GJ[3 + b2];
      real vJ[6];
      for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) {
          real s = 0;
          for (int k = 0; k < 3; ++k) s += GJ[3 * r + k] * p.Sc[3 * j + k] + GtJ[3 * r + k] * p.Sc[3 * k + j];
          vJ[3 * r + j] = s;
        }
      real x = p.mc[0], y = p.mc[1], z = p.mc[2];
      real rz = R_(1.0) / z, rz2 = rz * rz, rz3 = rz2 * rz;
      real vm2x = v_means2d[2 * (size_t)i], vm2y = v_means2d[2 * (size_t)i + 1];
      real vmc[3] = {fx * rz * vm2x, fy * rz * vm2y, -(fx * x * vm2x + fy * y * vm2y) * rz2};
      if (p.in_x) vmc[0] += -fx * rz2 * vJ[2]; else vmc[2] += -fx * rz3 * vJ[2] * p.tx;
      if (p.in_y) vmc[1] += -fy * rz2 * vJ[5]; else vmc[2] += -fy * rz3 * vJ[5] * p.ty;
      vmc[2] += -fx * rz2 * vJ[0] - fy * rz2 * vJ[4] + 2 * fx * p.tx * rz3 * vJ[2] + 2 * fy * p.ty * rz3 * vJ[5];
      vmc[2] += v_depths[i];
      /* world: mc = R m + t ; Sc = R S R^T */
      real T1[9], T2[9], vScT[9];
      for (int a2 = 0; a2 < 3; ++a2)
        for (int b2 = 0; b2 < 3; ++b2) vScT[3 * a2 + b2] = vSc[3 * b2 + a2];
      mat3_mul(vSc, Rv, T1);
      mat3_mul_bt(T1, p.S, T2); /* vSc R S^T */
      for (int k = 0; k < 9; ++k) locR[k] += T2[k];
      mat3_mul(vScT, Rv, T1);
      mat3_mul(T1, p.S, T2); /* vSc^T R S */
      for (int k = 0; k < 9; ++k) locR[k] += T2[k];
      for (int a2 = 0; a2 < 3; ++a2)
        for (int b2 = 0; b2 < 3; ++b2) locR[3 * a2 + b2] += vmc[a2] * m[b2];
      for (int k = 0; k < 3; ++k) loct[k] += vmc[k];
      if (v_means)
        for (int k = 0; k < 3; ++k) v_means[3 * (size_t)i + k] = Rv[k] * vmc[0] + Rv[3 + k] * vmc[1] + Rv[6 + k] * vmc[2];
      if (v_quats || v_scales) {
        real vS[9], vq[4], vs[3];
        mat3_mul_at(Rv, vSc, T1);
        mat3_mul(T1, Rv, vS); /* R^T vSc R */
        covar_vjp(quats + 4 * (size_t)i, scales + 3 * (size_t)i, &p, vS, vq, vs);
        if (v_quats) for (int k = 0; k < 4; ++k) v_quats[4 * (size_t)i + k] = vq[k];
        if (v_scales) for (int k = 0; k < 3; ++k) v_scales[3 * (size_t)i + k] = vs[k];
      }
    }
#pragma omp critical
    {
      for (int k = 0; k < 9; ++k) accR[k] += locR[k];
      for (int k = 0; k < 3; ++k) acct[k] += loct[k];
    }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) v_viewmat[4 * i + j] += accR[3 * i + j];
    v_viewmat[4 * i + 3] += acct[i];
  }
  return GSO_OK;
}

/* ----------------------------------------------------------------------------------------------------------- SH */
/* Real SH basis up to degree 3 in gsplat's evaluation order, and its gradient w.r.t. the unit direction. */
static void sh_basis(int deg, real x, real y, real z, real* B, real* dBx, real* dBy, real* dBz) {
  const real C0 = R_(0.2820947917738781), C1 = R_(0.48860251190292);
  int K = (deg + 1) * (deg + 1);
  for (int k = 0; k < K; ++k) dBx[k] = dBy[k] = dBz[k] = 0;
  B[0] = C0;
  if (deg < 1) return;
  B[1] = -C1 * y; dBy[1] = -C1;
  B[2] = C1 * z; dBz[2] = C1;
  B[3] = -C1 * x; dBx[3] = -C1;
  if (deg < 2) return;
  const real a1 = R_(1.092548430592079), a2 = R_(0.5462742152960395), a3 = R_(0.9461746957575601), a4 = R_(0.3153915652525201);
  real z2 = z * z, fC1 = x * x - y * y, fS1 = 2 * x * y, fTmpB = -a1 * z;
  B[4] = a2 * fS1; dBx[4] = a2 * 2 * y; dBy[4] = a2 * 2 * x;
  B[5] = fTmpB * y; dBy[5] = fTmpB; dBz[5] = -a1 * y;
  B[6] = a3 * z2 - a4; dBz[6] = 2 * a3 * z;
  B[7] = fTmpB * x; dBx[7] = fTmpB; dBz[7] = -a1 * x;
  B[8] = a2 * fC1; dBx[8] = a2 * 2 * x; dBy[8] = -a2 * 2 * y;
  if (deg < 3) return;
  const real b1 = R_(2.285228997322329), b2 = R_(0.4570457994644658), b3 = R_(1.445305721320277),
             b4 = R_(0.5900435899266435), b5 = R_(1.865881662950577), b6 = R_(1.119528997770346);
  real fTmpC = -b1 * z2 + b2, fTmpBb = b3 * z;
  real fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
  B[9] = -b4 * fS2; dBx[9] = -b4 * 6 * x * y; dBy[9] = -b4 * (3 * x * x - 3 * y * y);
  B[10] = fTmpBb * fS1; dBx[10] = b3 * 2 * y * z; dBy[10] = b3 * 2 * x * z; dBz[10] = b3 * fS1;
  B[11] = fTmpC * y; dBy[11] = fTmpC; dBz[11] = -2 * b1 * z * y;
  B[12] = z * (b5 * z2 - b6); dBz[12] = 3 * b5 * z2 - b6;
  B[13] = fTmpC * x; dBx[13] = fTmpC; dBz[13] = -2 * b1 * z * x;
  B[14] = fTmpBb * fC1; dBx[14] = b3 * 2 * x * z; dBy[14] = -b3 * 2 * y * z; dBz[14] = b3 * fC1;
  B[15] = -b4 * fC2; dBx[15] = -b4 * (3 * x * x - 3 * y * y); dBy[15] = b4 * 6 * x * y;
}

/* colours[N,3] = spherical_harmonics(degree, dirs[N,3], coeffs[N,K,3]); zero where mask (radii) == 0. */
int gso_sh_fwd(int degree, const real* dirs, const real* coeffs, const int32_t* radii, int N, int K, real* colors) {
  if (degree < 0 || degree > 3 || K < (degree + 1) * (degree + 1) || N < 0 || (N > 0 && (!dirs || !coeffs || !colors)))
    return GSO_BAD_ARG;
  int nK = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    real* c = colors + 3 * (size_t)i;
    c[0] = c[1] = c[2] = 0;
    if (radii && radii[i] <= 0) continue;
    const real* d = dirs + 3 * (size_t)i;
    real inorm = R_(1.0) / r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    real B[16], gx[16], gy[16], gz[16];
    sh_basis(degree, d[0] * inorm, d[1] * inorm, d[2] * inorm, B, gx, gy, gz);
    const real* cf = coeffs + (size_t)i * K * 3;
    for (int k = 0; k < nK; ++k) { c[0] += B[k] * cf[3 * k]; c[1] += B[k] * cf[3 * k + 1]; c[2] += B[k] * cf[3 * k + 2]; }
  }
  return GSO_OK;
}

/* vjp of gso_sh_fwd: v_coeffs[N,K,3] (bands above the degree get zero) and v_dirs[N,3] (either may be NULL). */
int gso_sh_bwd(int degree, const real* dirs, const real* coeffs, const int32_t* radii, int N, int K,
               const real* v_colors, real* v_coeffs, real* v_dirs) {
  if (degree < 0 || degree > 3 || K < (degree + 1) * (degree + 1) || N < 0 || (N > 0 && (!dirs || !coeffs || !v_colors)))
    return GSO_BAD_ARG;
  int nK = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i) {
    if (v_coeffs) for (int k = 0; k < 3 * K; ++k) v_coeffs[(size_t)i * K * 3 + k] = 0;
    if (v_dirs) v_dirs[3 * (size_t)i] = v_dirs[3 * (size_t)i + 1] = v_dirs[3 * (size_t)i + 2] = 0;
    if (radii && radii[i] <= 0) continue;
    const real* d = dirs + 3 * (size_t)i;
    const real* vc = v_colors + 3 * (size_t)i;
    real inorm = R_(1.0) / r_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    real u[3] = {d[0] * inorm, d[1] * inorm, d[2] * inorm};
    real B[16], gx[16], gy[16], gz[16];
    sh_basis(degree, u[0], u[1], u[2], B, gx, gy, gz);
    const real* cf = coeffs + (size_t)i * K * 3;
    real vu[3] = {0, 0, 0};
    for (int k = 0; k < nK; ++k) {
      if (v_coeffs) for (int ch = 0; ch < 3; ++ch) v_coeffs[((size_t)i * K + k) * 3 + ch] = B[k] * vc[ch];
      real s = cf[3 * k] * vc[0] + cf[3 * k + 1] * vc[1] + cf[3 * k + 2] * vc[2];
      vu[0] += gx[k] * s; vu[1] += gy[k] * s; vu[2] += gz[k] * s;
    }
    if (v_dirs) {
      real dd = vu[0] * u[0] + vu[1] * u[1] + vu[2] * u[2];
      for (int k = 0; k < 3; ++k) v_dirs[3 * (size_t)i + k] = (vu[k] - dd * u[k]) * inorm;
    }
  }
  return GSO_OK;
}

/* ------------------------------------------------------------------------------------------------------ binning */
static void tile_rect(const real* m2, int32_t radius, int tile_size, int tw, int th, int* x0, int* y0, int* x1, int* y1) {
  /* float32 arithmetic as in the kernels: (uint)floor(negative) saturates to 0, then min(., tile_{w,h}) */
  float ts = (float)tile_size, tr = (float)radius / ts, cx = (float)m2[0] / ts, cy = (float)m2[1] / ts;
  float fx0 = floorf(cx - tr), fy0 = floorf(cy - tr), fx1 = ceilf(cx + tr), fy1 = ceilf(cy + tr);
  *x0 = (int)fminf(fmaxf(fx0, 0.f), (float)tw); *y0 = (int)fminf(fmaxf(fy0, 0.f), (float)th);
  *x1 = (int)fminf(fmaxf(fx1, 0.f), (float)tw); *y1 = (int)fminf(fmaxf(fy1, 0.f), (float)th);
}

typedef struct { uint32_t dbits; int32_t id; } Entry;
static int entry_cmp(const void* a, const void* b) {
  const Entry* x = (const Entry*)a; const Entry* y = (const Entry*)b;
  if (x->dbits != y->dbits) return x->dbits < y->dbits ? -1 : 1;
  return (x->id > y->id) - (x->id < y->id);
}

/* isect_tiles (sorted) + isect_offset_encode for one camera.  Pass 1: flatten_ids == NULL returns the number of
 * intersections and fills tiles_per_gauss[N] (may be NULL).  Pass 2: fills isect_ids[I] (may be NULL),
 * flatten_ids[I], offsets[tw*th]; `capacity` guards the buffers.  Order within a tile: float32 depth bits, then
 * Gaussian id (what a stable sort of keys emitted in id order gives). */
int64_t gso_isect(const real* means2d, const int32_t* radii, const real* depths, int N, int tile_size, int tw, int th,
                  int32_t* tiles_per_gauss, int64_t capacity, int64_t* isect_ids, int32_t* flatten_ids, int32_t* offsets) {
  if (N < 0 || tile_size <= 0 || tw <= 0 || th <= 0 || (N > 0 && (!means2d || !radii || !depths))) return GSO_BAD_ARG;
  int n_tiles = tw * th;
  int64_t* counts = (int64_t*)calloc((size_t)n_tiles + 1, sizeof(int64_t));
  if (!counts) return GSO_NO_MEM;
  int64_t total = 0;
  for (int i = 0; i < N; ++i) {
    int c = 0;
    if (radii[i] > 0) {
      int x0, y0, x1, y1;
      tile_rect(means2d + 2 * (size_t)i, radii[i], tile_size, tw, th, &x0, &y0, &x1, &y1);
      c = (x1 - x0) * (y1 - y0);
      for (int ty = y0; ty < y1; ++ty)
        for (int tx = x0; tx < x1; ++tx) counts[ty * tw + tx + 1]++;
    }
    if (tiles_per_gauss) tiles_per_gauss[i] = c;
    total += c;
  }
  if (!flatten_ids) { free(counts); return total; }
  if (!offsets || total > capacity) { free(counts); return GSO_BAD_ARG; }
  for (int t = 0; t < n_tiles; ++t) counts[t + 1] += counts[t];
  for (int t = 0; t < n_tiles; ++t) offsets[t] = (int32_t)counts[t];
  Entry* ent = (Entry*)malloc((size_t)(total > 0 ? total : 1) * sizeof(Entry));
  int64_t* cursor = (int64_t*)malloc((size_t)n_tiles * sizeof(int64_t));
  if (!ent || !cursor) { free(counts); free(ent); free(cursor); return GSO_NO_MEM; }
  memcpy(cursor, counts, (size_t)n_tiles * sizeof(int64_t));
  for (int i = 0; i < N; ++i) {
    if (radii[i] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2d + 2 * (size_t)i, radii[i], tile_size, tw, th, &x0, &y0, &x1, &y1);
    float df = (float)depths[i];
    uint32_t bits;
    memcpy(&bits, &df, 4);
    for (int ty = y0; ty < y1; ++ty)
      for (int tx = x0; tx < x1; ++tx) {
        int64_t pos = cursor[ty * tw + tx]++;
        ent[pos].dbits = bits;
        ent[pos].id = i;
      }
  }
  int nbits = 0;
  while ((1 << nbits) <= n_tiles) ++nbits; /* floor(log2(n_tiles)) + 1 */
  (void)nbits;
#pragma omp parallel for schedule(dynamic, 8)
  for (int t = 0; t < n_tiles; ++t) {
    int64_t s = counts[t], e = counts[t + 1];
    if (e - s > 1) qsort(ent + s, (size_t)(e - s), sizeof(Entry), entry_cmp);
    for (int64_t k = s; k < e; ++k) {
      flatten_ids[k] = ent[k].id;
      if (isect_ids) isect_ids[k] = ((int64_t)t << 32) | (int64_t)ent[k].dbits;
    }
  }
  free(ent); free(cursor); free(counts);
  return total;
}

/* -------------------------------------------------------------------------------------------------- compositing */
/* rasterize_to_pixels forward (A.3): colors[N,D] composited front to back per pixel.
 * render[H,W,D], alphas[H,W], last_ids[H,W] (index into flatten_ids of the last composited entry, 0 if none). */
int gso_raster_fwd(const real* means2d, const real* conics, const real* colors, const real* opacities, int N, int D,
                   int W, int H, int tile_size, int tw, int th, const int32_t* offsets, const int32_t* flatten_ids,
                   int64_t n_isects, real* render, real* alphas, int32_t* last_ids) {
  if (D < 1 || D > 64 || W <= 0 || H <= 0 || !offsets || !render || !alphas || !last_ids || (n_isects > 0 && !flatten_ids))
    return GSO_BAD_ARG;
  (void)N;
  int n_tiles = tw * th;
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < n_tiles; ++t) {
    int tyi = t / tw, txi = t - tyi * tw;
    int64_t s = offsets[t], e = (t + 1 < n_tiles) ? offsets[t + 1] : n_isects;
    for (int i = tyi * tile_size; i < (tyi + 1) * tile_size && i < H; ++i)
      for (int j = txi * tile_size; j < (txi + 1) * tile_size && j < W; ++j) {
        real px = (real)j + R_(0.5), py = (real)i + R_(0.5), T = 1;
        real acc[64];
        for (int k = 0; k < D; ++k) acc[k] = 0;
        int32_t cur = 0;
        for (int64_t idx = s; idx < e; ++idx) {
          int g = flatten_ids[idx];
          real dx = means2d[2 * (size_t)g] - px, dy = means2d[2 * (size_t)g + 1] - py;
          const real* cn = conics + 3 * (size_t)g;
          real sigma = R_(0.5) * (cn[0] * dx * dx + cn[2] * dy * dy) + cn[1] * dx * dy;
          real al = r_min(ALPHA_MAX, opacities[g] * r_exp(-sigma));
          if (sigma < 0 || al < ALPHA_MIN) continue;
          real nT = T * (1 - al);
          if (nT <= T_STOP) break;
          real vis = al * T;
          const real* c = colors + (size_t)g * D;
          for (int k = 0; k < D; ++k) acc[k] += c[k] * vis;
          cur = (int32_t)idx;
          T = nT;
        }
        size_t pid = (size_t)i * W + j;
        for (int k = 0; k < D; ++k) render[pid * D + k] = acc[k];
        alphas[pid] = 1 - T;
        last_ids[pid] = cur;
      }
  }
  return GSO_OK;
}

/* rasterize_to_pixels backward (A.4): back-to-front replay per pixel.  Outputs are overwritten. */
int gso_raster_bwd(const real* means2d, const real* conics, const real* colors, const real* opacities, int N, int D,
                   int W, int H, int tile_size, int tw, int th, const int32_t* offsets, const int32_t* flatten_ids,
                   int64_t n_isects, const real* alphas, const int32_t* last_ids, const real* v_render,
                   const real* v_alphas, real* v_means2d, real* v_conics, real* v_colors, real* v_opacities) {
  if (D < 1 || D > 64 || N < 0 || !offsets || !alphas || !last_ids || !v_render || !v_alphas || !v_means2d || !v_conics ||
      !v_colors || !v_opacities || (n_isects > 0 && !flatten_ids))
    return GSO_BAD_ARG;
  memset(v_means2d, 0, sizeof(real) * 2 * (size_t)N);
  memset(v_conics, 0, sizeof(real) * 3 * (size_t)N);
  memset(v_colors, 0, sizeof(real) * (size_t)D * (size_t)N);
  memset(v_opacities, 0, sizeof(real) * (size_t)N);
  int n_tiles = tw * th;
  const int A = 6 + D;
  int failed = 0;
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < n_tiles; ++t) {
    int tyi = t / tw, txi = t - tyi * tw;
    int64_t s = offsets[t], e = (t + 1 < n_tiles) ? offsets[t + 1] : n_isects;
    if (e <= s) continue;
    real* part = (real*)calloc((size_t)(e - s) * A, sizeof(real)); /* per entry: vx vy | va vb vc | vo | colours */
    if (!part) { failed = 1; continue; }
    for (int i = tyi * tile_size; i < (tyi + 1) * tile_size && i < H; ++i)
      for (int j = txi * tile_size; j < (txi + 1) * tile_size && j < W; ++j) {
        size_t pid = (size_t)i * W + j;
        real px = (real)j + R_(0.5), py = (real)i + R_(0.5);
        real T_final = 1 - alphas[pid], T = T_final, va = v_alphas[pid];
        const real* vc = v_render + pid * D;
        real buf[64];
        for (int k = 0; k < D; ++k) buf[k] = 0;
        int64_t start = last_ids[pid];
        if (start > e - 1) start = e - 1;
        for (int64_t idx = start; idx >= s; --idx) {
          int g = flatten_ids[idx];
          real dx = means2d[2 * (size_t)g] - px, dy = means2d[2 * (size_t)g + 1] - py;
          const real* cn = conics + 3 * (size_t)g;
          real sigma = R_(0.5) * (cn[0] * dx * dx + cn[2] * dy * dy) + cn[1] * dx * dy;
          real vis = r_exp(-sigma), o = opacities[g];
          real al = r_min(ALPHA_MAX, o * vis);
          if (sigma < 0 || al < ALPHA_MIN) continue;
          real ra = 1 / (1 - al);
          T = T * ra;
          real fac = al * T;
          real* row = part + (size_t)(idx - s) * A;
          const real* c = colors + (size_t)g * D;
          real v_al = T_final * ra * va;
          for (int k = 0; k < D; ++k) {
            row[6 + k] += fac * vc[k];
            v_al += (c[k] * T - buf[k] * ra) * vc[k];
          }
          if (o * vis <= ALPHA_MAX) {
            real v_sigma = -o * vis * v_al;
            row[2] += R_(0.5) * v_sigma * dx * dx;
            row[3] += v_sigma * dx * dy;
            row[4] += R_(0.5) * v_sigma * dy * dy;
            row[0] += v_sigma * (cn[0] * dx + cn[1] * dy);
            row[1] += v_sigma * (cn[1] * dx + cn[2] * dy);
            row[5] += vis * v_al;
          }
          for (int k = 0; k < D; ++k) buf[k] += c[k] * fac;
        }
      }
    for (int64_t idx = s; idx < e; ++idx) {
      int g = flatten_ids[idx];
      const real* row = part + (size_t)(idx - s) * A;
      real* dst[4] = {v_means2d + 2 * (size_t)g, v_conics + 3 * (size_t)g, v_opacities + g, v_colors + (size_t)g * D};
      const int off[4] = {0, 2, 5, 6}, len[4] = {2, 3, 1, D};
      for (int q = 0; q < 4; ++q)
        for (int k = 0; k < len[q]; ++k) {
          real v = row[off[q] + k];
          if (v != 0) {
#pragma omp atomic
            dst[q][k] += v;
          }
        }
    }
    free(part);
  }
  return failed ? GSO_NO_MEM : GSO_OK;
}

/* ----------------------------------------------------------------------------------------------------- end to end */
/* gsplat.rasterization, one camera, forward then backward from (v_render, v_alphas).
 *   color_mode: GSO_COLOR_NONE | GSO_COLOR_SH (coeffs[N,K,3], sh_degree) | GSO_COLOR_RGB (colors[N,3])
 *   depth_mode: GSO_DEPTH_NONE | GSO_DEPTH_ACC ("D") | GSO_DEPTH_EXPECTED ("ED")
 *   channels D = (color_mode ? 3 : 0) + (depth_mode ? 1 : 0), depth last.
 * Outputs: render[H,W,D], alphas[H,W]; with do_backward: v_means[N,3], v_quats[N,4], v_scales[N,3],
 * v_opacities[N], v_colors ([N,K,3] or [N,3]; NULL without colours), v_viewmat[16]; *n_isects_out.
 * antialiased: opacities are multiplied by the compensation factor (rasterize_mode="antialiased"). */
int gso_rasterization(const real* means, const real* quats, const real* scales, const real* opacities,
                      const real* colors, int color_mode, int sh_degree, int K_sh, const real* viewmat, const real* K,
                      int N, int W, int H, int tile_size, real eps2d, real near_plane, real far_plane,
                      real radius_clip, int antialiased, int depth_mode, int do_backward, const real* v_render,
                      const real* v_alphas, real* render, real* alphas, real* v_means, real* v_quats, real* v_scales,
                      real* v_opacities, real* v_colors, real* v_viewmat, int64_t* n_isects_out) {
  int nc = color_mode ? 3 : 0, D = nc + (depth_mode ? 1 : 0);
  if (D < 1 || N < 0 || !render || !alphas || (color_mode && !colors)) return GSO_BAD_ARG;
  if (do_backward && (!v_render || !v_alphas || !v_means || !v_quats || !v_scales || !v_opacities || !v_viewmat || (color_mode && !v_colors)))
    return GSO_BAD_ARG;
  int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
  size_t n = (size_t)(N > 0 ? N : 1), P = (size_t)W * H;
  int rc = GSO_NO_MEM;
  int32_t* radii = (int32_t*)malloc(n * sizeof(int32_t));
  real* m2 = (real*)malloc(n * 2 * sizeof(real));
  real* dep = (real*)malloc(n * sizeof(real));
  real* con = (real*)malloc(n * 3 * sizeof(real));
  real* comp = antialiased ? (real*)malloc(n * sizeof(real)) : NULL;
  real* opa = (real*)malloc(n * sizeof(real));
  real* col = (real*)malloc(n * (size_t)D * sizeof(real));
  real* dirs = (color_mode == GSO_COLOR_SH) ? (real*)malloc(n * 3 * sizeof(real)) : NULL;
  real* rgb = (color_mode == GSO_COLOR_SH) ? (real*)malloc(n * 3 * sizeof(real)) : NULL;
  int32_t* offs = (int32_t*)malloc((size_t)tw * th * sizeof(int32_t));
  int32_t* last = (int32_t*)malloc(P * sizeof(int32_t));
  int32_t* fids = NULL;
  real *acc = NULL, *va_tot = NULL, *vr = NULL, *v_m2 = NULL, *v_con = NULL, *v_col = NULL, *v_opa = NULL, *v_dep = NULL,
       *v_rgb = NULL, *v_dirs = NULL;
  real campos[3] = {0, 0, 0}, Ainv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (!radii || !m2 || !dep || !con || !opa || !col || !offs || !last || (antialiased && !comp) ||
      (color_mode == GSO_COLOR_SH && (!dirs || !rgb)))
    goto done;
  double t0 = now_s();
  rc = gso_project_fwd(means, quats, scales, viewmat, K, N, W, H, eps2d, near_plane, far_plane, radius_clip, radii, m2,
                       dep, con, comp);
  if (rc) goto done;
  lap("project fwd", &t0);
  if (color_mode == GSO_COLOR_SH) {
    const real Rv[9] = {viewmat[0], viewmat[1], viewmat[2], viewmat[4], viewmat[5], viewmat[6], viewmat[8], viewmat[9], viewmat[10]};
    if (!mat3_inv(Rv, Ainv)) { rc = GSO_BAD_ARG; goto done; }
    for (int k = 0; k < 3; ++k) campos[k] = -(Ainv[3 * k] * viewmat[3] + Ainv[3 * k + 1] * viewmat[7] + Ainv[3 * k + 2] * viewmat[11]);
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) dirs[3 * (size_t)i + k] = means[3 * (size_t)i + k] - campos[k];
    rc = gso_sh_fwd(sh_degree, dirs, colors, radii, N, K_sh, rgb);
    if (rc) goto done;
  }
  for (int i = 0; i < N; ++i) {
    opa[i] = antialiased ? opacities[i] * comp[i] : opacities[i];
    real* c = col + (size_t)i * D;
    if (color_mode == GSO_COLOR_SH)
      for (int k = 0; k < 3; ++k) c[k] = r_max(rgb[3 * (size_t)i + k] + R_(0.5), 0);
    else if (color_mode == GSO_COLOR_RGB)
      for (int k = 0; k < 3; ++k) c[k] = colors[3 * (size_t)i + k];
    if (depth_mode) c[nc] = dep[i];
  }
  lap("colours", &t0);
  int64_t I = gso_isect(m2, radii, dep, N, tile_size, tw, th, NULL, 0, NULL, NULL, NULL);
  if (I < 0) { rc = (int)I; goto done; }
  fids = (int32_t*)malloc((size_t)(I > 0 ? I : 1) * sizeof(int32_t));
  if (!fids) { rc = GSO_NO_MEM; goto done; }
  if (gso_isect(m2, radii, dep, N, tile_size, tw, th, NULL, I, NULL, fids, offs) != I) { rc = GSO_BAD_ARG; goto done; }
  if (n_isects_out) *n_isects_out = I;
  lap("binning", &t0);
  rc = gso_raster_fwd(m2, con, col, opa, N, D, W, H, tile_size, tw, th, offs, fids, I, render, alphas, last);
  if (rc) goto done;
  lap("raster fwd", &t0);
  if (depth_mode == GSO_DEPTH_EXPECTED && do_backward) {
    acc = (real*)malloc(P * sizeof(real)); /* accumulated depth before normalisation */
    if (!acc) { rc = GSO_NO_MEM; goto done; }
    for (size_t p = 0; p < P; ++p) acc[p] = render[p * D + nc];
  }
  if (depth_mode == GSO_DEPTH_EXPECTED)
    for (size_t p = 0; p < P; ++p) render[p * D + nc] = render[p * D + nc] / r_max(alphas[p], ED_CLAMP);
  if (!do_backward) { rc = GSO_OK; goto done; }

  rc = GSO_NO_MEM;
  va_tot = (real*)malloc(P * sizeof(real));
  vr = (real*)malloc(P * (size_t)D * sizeof(real));
  v_m2 = (real*)malloc(n * 2 * sizeof(real));
  v_con = (real*)malloc(n * 3 * sizeof(real));
  v_col = (real*)malloc(n * (size_t)D * sizeof(real));
  v_opa = (real*)malloc(n * sizeof(real));
  v_dep = (real*)calloc(n, sizeof(real));
  if (!va_tot || !vr || !v_m2 || !v_con || !v_col || !v_opa || !v_dep) goto done;
  memcpy(vr, v_render, P * (size_t)D * sizeof(real));
  memcpy(va_tot, v_alphas, P * sizeof(real));
  if (depth_mode == GSO_DEPTH_EXPECTED)
    for (size_t p = 0; p < P; ++p) { /* depth = acc / max(alpha, clamp) */
      real a = alphas[p], den = r_max(a, ED_CLAMP), g = v_render[p * D + nc];
      vr[p * D + nc] = g / den;
      if (a > ED_CLAMP) va_tot[p] += -g * acc[p] / (den * den);
    }
  rc = gso_raster_bwd(m2, con, col, opa, N, D, W, H, tile_size, tw, th, offs, fids, I, alphas, last, vr, va_tot, v_m2,
                      v_con, v_col, v_opa);
  if (rc) goto done;
  lap("raster bwd", &t0);
  for (int k = 0; k < 16; ++k) v_viewmat[k] = 0;
  if (depth_mode)
    for (int i = 0; i < N; ++i) v_dep[i] = v_col[(size_t)i * D + nc];
  real* v_comp = NULL;
  if (antialiased) {
    v_comp = (real*)malloc(n * sizeof(real));
    if (!v_comp) { rc = GSO_NO_MEM; goto done; }
    for (int i = 0; i < N; ++i) { v_comp[i] = v_opa[i] * opacities[i]; v_opacities[i] = v_opa[i] * comp[i]; }
  } else {
    for (int i = 0; i < N; ++i) v_opacities[i] = v_opa[i];
  }
  rc = gso_project_bwd(means, quats, scales, viewmat, K, N, W, H, eps2d, radii, v_m2, v_dep, v_con, v_comp, v_means,
                       v_quats, v_scales, v_viewmat);
  free(v_comp);
  if (rc) goto done;
  lap("project bwd", &t0);
  if (color_mode == GSO_COLOR_RGB) {
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) v_colors[3 * (size_t)i + k] = v_col[(size_t)i * D + k];
  } else if (color_mode == GSO_COLOR_SH) {
    rc = GSO_NO_MEM;
    v_rgb = (real*)malloc(n * 3 * sizeof(real));
    v_dirs = (real*)malloc(n * 3 * sizeof(real));
    if (!v_rgb || !v_dirs) goto done;
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) /* clamp_min(x + 0.5, 0) passes the gradient where x + 0.5 > 0 */
        v_rgb[3 * (size_t)i + k] = (rgb[3 * (size_t)i + k] + R_(0.5) > 0) ? v_col[(size_t)i * D + k] : 0;
    rc = gso_sh_bwd(sh_degree, dirs, colors, radii, N, K_sh, v_rgb, v_colors, v_dirs);
    if (rc) goto done;
    real vcp[3] = {0, 0, 0};
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) { v_means[3 * (size_t)i + k] += v_dirs[3 * (size_t)i + k]; vcp[k] -= v_dirs[3 * (size_t)i + k]; }
    /* campos = -A t, A = R^-1:  v_t = -A^T v_c ;  v_R = -A^T v_c campos^T */
    real Atv[3];
    for (int k = 0; k < 3; ++k) Atv[k] = Ainv[k] * vcp[0] + Ainv[3 + k] * vcp[1] + Ainv[6 + k] * vcp[2];
    for (int i = 0; i < 3; ++i) {
      v_viewmat[4 * i + 3] += -Atv[i];
      for (int j = 0; j < 3; ++j) v_viewmat[4 * i + j] += -Atv[i] * campos[j];
    }
  }
  lap("colours bwd", &t0);
  rc = GSO_OK;
done:
  free(radii); free(m2); free(dep); free(con); free(comp); free(opa); free(col); free(dirs); free(rgb); free(offs);
  free(last); free(fids); free(acc); free(va_tot); free(vr); free(v_m2); free(v_con); free(v_col); free(v_opa);
  free(v_dep); free(v_rgb); free(v_dirs);
  return rc;
}
