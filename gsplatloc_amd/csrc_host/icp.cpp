// CPU scan-to-scan ICP baseline behind include/gsloc_icp.h (host only; see the header for what it replaces).
// Self-contained: no Eigen.  Parallel loops use OpenMP with a fixed block order so sums are reproducible
// for a given thread count.
#include "../../include/gsloc_icp.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include <omp.h>

namespace {

using V3 = std::array<double, 3>;
using M3 = std::array<double, 9>;  // row-major

inline V3 sub(const V3& a, const V3& b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline V3 mulv(const M3& m, const V3& v) {
  return {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
          m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
}
inline M3 mulm(const M3& a, const M3& b) {
  M3 c{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  return c;
}
inline M3 transpose(const M3& a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
inline M3 hat(const V3& v) { return {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}; }

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations; eigenvalues ascending, columns of `vec`.
void eigen_sym3(const M3& a_in, double val[3], M3& vec) {
  M3 a = a_in;
  vec = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int sweep = 0; sweep < 32; ++sweep) {
    double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = a[3 * p + q];
        if (std::fabs(apq) < 1e-300) continue;
        double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A * G
          double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq;
          a[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- G^T * A
          double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = c * apk - s * aqk;
          a[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = vec[3 * k + p], vkq = vec[3 * k + q];
          vec[3 * k + p] = c * vkp - s * vkq;
          vec[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {a[0], a[4], a[8]};
  std::sort(order, order + 3, [&](int i, int j) { return d[i] < d[j]; });
  M3 v2;
  for (int j = 0; j < 3; ++j) {
    val[j] = d[order[j]];
    for (int k = 0; k < 3; ++k) v2[3 * k + j] = vec[3 * k + order[j]];
  }
  vec = v2;
}

bool inverse3(const M3& m, M3& out) {
  double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
  if (std::fabs(det) < 1e-300) return false;
  double id = 1.0 / det;
  out = {c0 * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id,
         c1 * id, (m[0] * m[8] - m[2] * m[6]) * id, (m[2] * m[3] - m[0] * m[5]) * id,
         c2 * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id};
  return true;
}

// Solve the symmetric positive-definite 6x6 system A x = rhs (LDL^T without pivoting, as Eigen's ldlt on an
// LM-damped normal matrix).  Returns false if a pivot vanishes.
bool solve6(const double A[36], const double rhs[6], double x[6]) {
  double L[36] = {0}, D[6];
  for (int j = 0; j < 6; ++j) {
    double d = A[6 * j + j];
    for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
    if (!(std::fabs(d) > 1e-300)) return false;
    D[j] = d;
    L[6 * j + j] = 1.0;
    for (int i = j + 1; i < 6; ++i) {
      double v = A[6 * i + j];
      for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
      L[6 * i + j] = v / d;
    }
  }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double v = rhs[i];
    for (int k = 0; k < i; ++k) v -= L[6 * i + k] * y[k];
    y[i] = v;
  }
  for (int i = 0; i < 6; ++i) y[i] /= D[i];
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
    for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * x[k];
    x[i] = v;
  }
  return true;
}

struct Iso {  // rigid transform
  M3 R{1, 0, 0, 0, 1, 0, 0, 0, 1};
  V3 t{0, 0, 0};
  V3 apply(const V3& p) const {
    V3 q = mulv(R, p);
    return {q[0] + t[0], q[1] + t[1], q[2] + t[2]};
  }
};

Iso compose(const Iso& a, const Iso& b) {
  Iso c;
  c.R = mulm(a.R, b.R);
  V3 rt = mulv(a.R, b.t);
  c.t = {rt[0] + a.t[0], rt[1] + a.t[1], rt[2] + a.t[2]};
  return c;
}

// exp of a twist (rotation vector first, then translation).
Iso se3_exp(const double a[6]) {
  V3 w{a[0], a[1], a[2]}, v{a[3], a[4], a[5]};
  double th2 = dot(w, w), th = std::sqrt(th2);
  M3 W = hat(w), W2 = mulm(W, W);
  double A, B, C;
  if (th < 1e-10) {
    A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0;
  } else {
    A = std::sin(th) / th; B = (1.0 - std::cos(th)) / th2; C = (th - std::sin(th)) / (th2 * th);
  }
  Iso T;
  M3 V;
  for (int i = 0; i < 9; ++i) {
    double I = (i % 4 == 0) ? 1.0 : 0.0;
    T.R[i] = I + A * W[i] + B * W2[i];
    V[i] = I + B * W[i] + C * W2[i];
  }
  T.t = mulv(V, v);
  return T;
}

// ------------------------------------------------------------------------------------------- kd-tree
struct Node {
  int32_t left = -1, right = -1;  // children (inner node) ...
  int32_t first = 0, last = 0;    // ... or index range into `order` (leaf)
  int32_t axis = -1;
  double split = 0;
};

struct KdTree {
  static constexpr int kLeaf = 20;
  const std::vector<V3>* pts = nullptr;
  std::vector<int32_t> order;
  std::vector<Node> nodes;

  void build(const std::vector<V3>& points) {
    pts = &points;
    order.resize(points.size());
    std::iota(order.begin(), order.end(), 0);
    nodes.clear();
    nodes.reserve(points.size() / (kLeaf / 2) + 8);
    if (!points.empty()) build_node(0, (int32_t)points.size());
  }

  int32_t build_node(int32_t first, int32_t last) {
    int32_t id = (int32_t)nodes.size();
    nodes.emplace_back();
    if (last - first <= kLeaf) {
      nodes[id].first = first; nodes[id].last = last;
      return id;
    }
    // split the axis of largest variance at the median
    double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
    for (int32_t i = first; i < last; ++i)
      for (int k = 0; k < 3; ++k) mean[k] += (*pts)[order[i]][k];
    for (int k = 0; k < 3; ++k) mean[k] /= (last - first);
    for (int32_t i = first; i < last; ++i)
      for (int k = 0; k < 3; ++k) {
        double d = (*pts)[order[i]][k] - mean[k];
        var[k] += d * d;
      }
    int axis = (var[0] >= var[1] && var[0] >= var[2]) ? 0 : (var[1] >= var[2] ? 1 : 2);
    int32_t mid = first + (last - first) / 2;
    std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + last,
                     [&](int32_t a, int32_t b) { return (*pts)[a][axis] < (*pts)[b][axis]; });
    double split = (*pts)[order[mid]][axis];
    int32_t l = build_node(first, mid);
    int32_t r = build_node(mid, last);
    nodes[id].axis = axis; nodes[id].split = split; nodes[id].left = l; nodes[id].right = r;
    return id;
  }

  // k nearest: (squared distance, index) ascending in `best` (size k, padded with inf / -1)
  void knn(const V3& q, int k, double* best_d, int64_t* best_i) const {
    for (int j = 0; j < k; ++j) { best_d[j] = std::numeric_limits<double>::infinity(); best_i[j] = -1; }
    if (nodes.empty()) return;
    search(0, q, k, best_d, best_i);
  }

  void search(int32_t id, const V3& q, int k, double* best_d, int64_t* best_i) const {
    const Node& n = nodes[id];
    if (n.axis < 0) {
      for (int32_t i = n.first; i < n.last; ++i) {
        int32_t p = order[i];
        V3 d = sub((*pts)[p], q);
        double d2 = dot(d, d);
        // ties on distance resolve to the smaller index so the result does not depend on the tree shape
        if (d2 < best_d[k - 1] || (d2 == best_d[k - 1] && p < best_i[k - 1])) {
          int j = k - 1;
          while (j > 0 && (best_d[j - 1] > d2 || (best_d[j - 1] == d2 && best_i[j - 1] > p))) {
            best_d[j] = best_d[j - 1]; best_i[j] = best_i[j - 1];
            --j;
          }
          best_d[j] = d2; best_i[j] = p;
        }
      }
      return;
    }
    double diff = q[n.axis] - n.split;
    int32_t near = diff < 0 ? n.left : n.right, far = diff < 0 ? n.right : n.left;
    search(near, q, k, best_d, best_i);
    if (diff * diff <= best_d[k - 1]) search(far, q, k, best_d, best_i);
  }
};

}  // namespace

struct gsl_icp_cloud {
  std::vector<V3> points;
  std::vector<V3> normals;
  std::vector<M3> covs;
  KdTree tree;
  bool has_tree = false;
};

namespace {

int threads_or_default(int n) { return n > 0 ? n : omp_get_max_threads(); }

struct Sys {
  double H[36], b[6], e;
  int64_t inliers;
  void zero() { std::memset(this, 0, sizeof(*this)); }
  void add(const Sys& o) {
    for (int i = 0; i < 36; ++i) H[i] += o.H[i];
    for (int i = 0; i < 6; ++i) b[i] += o.b[i];
    e += o.e; inliers += o.inliers;
  }
};

// One correspondence's contribution.  J = d(residual)/d(twist), residual = q - T p, right perturbation
// T <- T exp(twist): d/d(rot) = R [p]x, d/d(trans) = -R.
inline void accumulate(int type, const Iso& T, const V3& p, const V3& q, const V3* normal, const M3* cov_t,
                       const M3* cov_s, bool with_jacobian, Sys& s) {
  V3 r = sub(q, T.apply(p));
  M3 Jr = mulm(T.R, hat(p));
  double J[3][6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { J[i][j] = Jr[3 * i + j]; J[i][3 + j] = -T.R[3 * i + j]; }
  M3 Wm{1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (type == GSL_ICP_PLANE) {
    // small_gicp's point-to-plane factor weights residual and Jacobian rows element-wise by the normal
    for (int i = 0; i < 3; ++i) {
      r[i] *= (*normal)[i];
      for (int j = 0; j < 6; ++j) J[i][j] *= (*normal)[i];
    }
  } else if (type == GSL_ICP_GICP) {
    M3 RCR = mulm(mulm(T.R, *cov_s), transpose(T.R));
    M3 sum;
    for (int i = 0; i < 9; ++i) sum[i] = (*cov_t)[i] + RCR[i];
    if (!inverse3(sum, Wm)) return;
  }
  V3 Wr = mulv(Wm, r);
  s.e += 0.5 * dot(r, Wr);
  s.inliers += 1;
  if (!with_jacobian) return;
  double WJ[3][6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 6; ++j) WJ[i][j] = Wm[3 * i] * J[0][j] + Wm[3 * i + 1] * J[1][j] + Wm[3 * i + 2] * J[2][j];
  for (int a = 0; a < 6; ++a) {
    s.b[a] += J[0][a] * Wr[0] + J[1][a] * Wr[1] + J[2][a] * Wr[2];
    for (int c = 0; c < 6; ++c) s.H[6 * a + c] += J[0][a] * WJ[0][c] + J[1][a] * WJ[1][c] + J[2][a] * WJ[2][c];
  }
}

// Fixed-size blocks summed in block order: the totals do not depend on how OpenMP schedules the blocks.
template <class F>
Sys block_reduce(int64_t n, int num_threads, F&& body) {
  const int64_t block = 4096;
  int64_t nb = (n + block - 1) / block;
  std::vector<Sys> parts((size_t)nb);
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
  for (int64_t bidx = 0; bidx < nb; ++bidx) {
    Sys s;
    s.zero();
    int64_t hi = std::min(n, (bidx + 1) * block);
    for (int64_t i = bidx * block; i < hi; ++i) body(i, s);
    parts[(size_t)bidx] = s;
  }
  Sys total;
  total.zero();
  for (const Sys& p : parts) total.add(p);
  return total;
}

}  // namespace

extern "C" {

const char* gsl_icp_version(void) { return "gsloc_icp 0.1.0"; }

gsl_icp_cloud* gsl_icp_cloud_create(const double* points, int64_t n, int stride) {
  if (n < 0 || stride < 3 || (n > 0 && !points)) return nullptr;
  auto* c = new gsl_icp_cloud();
  c->points.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) c->points[(size_t)i] = {points[i * stride], points[i * stride + 1], points[i * stride + 2]};
  return c;
}

void gsl_icp_cloud_destroy(gsl_icp_cloud* cloud) { delete cloud; }

int64_t gsl_icp_cloud_size(const gsl_icp_cloud* cloud) { return cloud ? (int64_t)cloud->points.size() : GSL_ICP_BAD_ARG; }

int gsl_icp_cloud_read(const gsl_icp_cloud* cloud, int what, double* out) {
  if (!cloud || !out) return GSL_ICP_BAD_ARG;
  size_t n = cloud->points.size();
  if (what == 0) {
    for (size_t i = 0; i < n; ++i) std::memcpy(out + 3 * i, cloud->points[i].data(), 24);
  } else if (what == 1) {
    if (cloud->normals.size() != n) return GSL_ICP_NO_ATTR;
    for (size_t i = 0; i < n; ++i) std::memcpy(out + 3 * i, cloud->normals[i].data(), 24);
  } else if (what == 2) {
    if (cloud->covs.size() != n) return GSL_ICP_NO_ATTR;
    for (size_t i = 0; i < n; ++i) std::memcpy(out + 9 * i, cloud->covs[i].data(), 72);
  } else {
    return GSL_ICP_BAD_ARG;
  }
  return GSL_ICP_OK;
}

int gsl_icp_build_tree(gsl_icp_cloud* cloud, int /*num_threads*/) {
  if (!cloud) return GSL_ICP_BAD_ARG;
  cloud->tree.build(cloud->points);
  cloud->has_tree = true;
  return GSL_ICP_OK;
}

int gsl_icp_knn(const gsl_icp_cloud* cloud, const double* queries, int64_t m, int stride, int k, int64_t* indices,
                double* sq_dists, int num_threads) {
  if (!cloud || m < 0 || stride < 3 || k < 1 || k > 256 || (m > 0 && (!queries || !indices || !sq_dists)))
    return GSL_ICP_BAD_ARG;
  if (!cloud->has_tree) return GSL_ICP_NO_TREE;
  int nt = threads_or_default(num_threads);
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
  for (int64_t i = 0; i < m; ++i) {
    V3 q{queries[i * stride], queries[i * stride + 1], queries[i * stride + 2]};
    cloud->tree.knn(q, k, sq_dists + i * k, indices + i * k);
  }
  return GSL_ICP_OK;
}

int gsl_icp_estimate_normals_covariances(gsl_icp_cloud* cloud, int num_neighbors, int num_threads) {
  if (!cloud || num_neighbors < 3 || num_neighbors > 256) return GSL_ICP_BAD_ARG;
  if (!cloud->has_tree) return GSL_ICP_NO_TREE;
  int64_t n = (int64_t)cloud->points.size();
  cloud->normals.assign((size_t)n, V3{0, 0, 0});
  cloud->covs.assign((size_t)n, M3{0, 0, 0, 0, 0, 0, 0, 0, 0});
  int nt = threads_or_default(num_threads);
  int k = num_neighbors;
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
  for (int64_t i = 0; i < n; ++i) {
    double d[256];
    int64_t idx[256];
    cloud->tree.knn(cloud->points[(size_t)i], k, d, idx);
    int found = 0;
    double mean[3] = {0, 0, 0}, m2[9] = {0};
    for (int j = 0; j < k; ++j) {
      if (idx[j] < 0) break;
      const V3& p = cloud->points[(size_t)idx[j]];
      for (int a = 0; a < 3; ++a) {
        mean[a] += p[a];
        for (int b = 0; b < 3; ++b) m2[3 * a + b] += p[a] * p[b];
      }
      ++found;
    }
    if (found < 5) {  // too few neighbours: identity covariance, zero normal (small_gicp does the same)
      cloud->covs[(size_t)i] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      continue;
    }
    M3 cov;
    for (int a = 0; a < 3; ++a) mean[a] /= found;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) cov[3 * a + b] = m2[3 * a + b] / found - mean[a] * mean[b];
    double val[3];
    M3 vec;
    eigen_sym3(cov, val, vec);
    V3 nrm{vec[0], vec[3], vec[6]};  // eigenvector of the smallest eigenvalue
    if (dot(cloud->points[(size_t)i], nrm) > 0) nrm = {-nrm[0], -nrm[1], -nrm[2]};  // towards the sensor origin
    cloud->normals[(size_t)i] = nrm;
    const double reg[3] = {1e-3, 1.0, 1.0};
    M3 out{};
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        out[3 * a + b] = reg[0] * vec[3 * a] * vec[3 * b] + reg[1] * vec[3 * a + 1] * vec[3 * b + 1] +
                         reg[2] * vec[3 * a + 2] * vec[3 * b + 2];
    cloud->covs[(size_t)i] = out;
  }
  return GSL_ICP_OK;
}

gsl_icp_cloud* gsl_icp_voxel_downsample(const gsl_icp_cloud* cloud, double leaf, int /*num_threads*/) {
  if (!cloud || !(leaf > 0)) return nullptr;
  const double inv = 1.0 / leaf;
  const int64_t offset = 1 << 20, mask = (1 << 21) - 1;  // 21 bits per axis
  size_t n = cloud->points.size();
  std::vector<std::pair<uint64_t, size_t>> keyed;
  keyed.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    const V3& p = cloud->points[i];
    int64_t c[3];
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      c[a] = (int64_t)std::floor(p[a] * inv) + offset;
      ok = ok && c[a] >= 0 && c[a] <= mask;
    }
    if (!ok) continue;  // outside the addressable grid: dropped
    keyed.emplace_back(((uint64_t)c[0]) | ((uint64_t)c[1] << 21) | ((uint64_t)c[2] << 42), i);
  }
  std::sort(keyed.begin(), keyed.end());
  auto* out = new gsl_icp_cloud();
  size_t i = 0;
  while (i < keyed.size()) {
    size_t j = i;
    V3 sum{0, 0, 0};
    while (j < keyed.size() && keyed[j].first == keyed[i].first) {
      const V3& p = cloud->points[keyed[j].second];
      sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2];
      ++j;
    }
    double c = (double)(j - i);
    out->points.push_back({sum[0] / c, sum[1] / c, sum[2] / c});
    i = j;
  }
  return out;
}

int gsl_icp_align(const gsl_icp_cloud* target, const gsl_icp_cloud* source, const double* init_T,
                  double max_corr, int type, int max_iterations, int num_threads, gsl_icp_result* result) {
  if (!target || !source || !result || type < GSL_ICP_POINT || type > GSL_ICP_GICP || !(max_corr > 0))
    return GSL_ICP_BAD_ARG;
  if (!target->has_tree) return GSL_ICP_NO_TREE;
  size_t nt_pts = target->points.size(), ns = source->points.size();
  if (type == GSL_ICP_PLANE && target->normals.size() != nt_pts) return GSL_ICP_NO_ATTR;
  if (type == GSL_ICP_GICP && (target->covs.size() != nt_pts || source->covs.size() != ns)) return GSL_ICP_NO_ATTR;
  if (max_iterations <= 0) max_iterations = 20;
  const int max_inner = 10;
  const double lambda_factor = 10.0, rot_eps = 0.1 * M_PI / 180.0, trans_eps = 1e-3;
  double lambda = 1e-3;
  int nt = threads_or_default(num_threads);

  Iso T;
  if (init_T) {
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T.R[3 * i + j] = init_T[4 * i + j];
      T.t[i] = init_T[4 * i + 3];
    }
  }
  std::vector<int64_t> corr(ns, -1);
  const double max_d2 = max_corr * max_corr;
  std::memset(result, 0, sizeof(*result));

  auto factor_args = [&](size_t i, int64_t j, const V3*& nrm, const M3*& ct, const M3*& cs) {
    nrm = (type == GSL_ICP_PLANE) ? &target->normals[(size_t)j] : nullptr;
    ct = (type == GSL_ICP_GICP) ? &target->covs[(size_t)j] : nullptr;
    cs = (type == GSL_ICP_GICP) ? &source->covs[i] : nullptr;
  };

  bool converged = false;
  Sys sys;
  sys.zero();
  int it = 0;
  for (; it < max_iterations && !converged; ++it) {
    // linearise: nearest target point of every transformed source point, rejected beyond max_corr
    sys = block_reduce((int64_t)ns, nt, [&](int64_t i, Sys& s) {
      V3 tp = T.apply(source->points[(size_t)i]);
      double d;
      int64_t j;
      target->tree.knn(tp, 1, &d, &j);
      if (j < 0 || d > max_d2) { corr[(size_t)i] = -1; return; }
      corr[(size_t)i] = j;
      const V3* nrm; const M3* ct; const M3* cs;
      factor_args((size_t)i, j, nrm, ct, cs);
      accumulate(type, T, source->points[(size_t)i], target->points[(size_t)j], nrm, ct, cs, true, s);
    });
    bool success = false;
    for (int inner = 0; inner < max_inner; ++inner) {
      double A[36], rhs[6], delta[6];
      for (int i = 0; i < 36; ++i) A[i] = sys.H[i] + ((i % 7 == 0) ? lambda : 0.0);
      for (int i = 0; i < 6; ++i) rhs[i] = -sys.b[i];
      if (!solve6(A, rhs, delta)) { lambda *= lambda_factor; continue; }
      Iso Tn = compose(T, se3_exp(delta));
      Sys trial = block_reduce((int64_t)ns, nt, [&](int64_t i, Sys& s) {  // same correspondences, new pose
        int64_t j = corr[(size_t)i];
        if (j < 0) return;
        const V3* nrm; const M3* ct; const M3* cs;
        factor_args((size_t)i, j, nrm, ct, cs);
        accumulate(type, Tn, source->points[(size_t)i], target->points[(size_t)j], nrm, ct, cs, false, s);
      });
      if (trial.e <= sys.e) {
        double rn = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
        double tn = std::sqrt(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]);
        converged = rn <= rot_eps && tn <= trans_eps;
        T = Tn;
        lambda /= lambda_factor;
        sys.e = trial.e;
        success = true;
        break;
      }
      lambda *= lambda_factor;
    }
    if (!success) { ++it; break; }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) result->T_target_source[4 * i + j] = T.R[3 * i + j];
    result->T_target_source[4 * i + 3] = T.t[i];
  }
  result->T_target_source[15] = 1.0;
  std::memcpy(result->H, sys.H, sizeof(sys.H));
  std::memcpy(result->b, sys.b, sizeof(sys.b));
  result->error = sys.e;
  result->converged = converged ? 1 : 0;
  result->iterations = it;
  result->num_inliers = sys.inliers;
  return GSL_ICP_OK;
}

}  // extern "C"
