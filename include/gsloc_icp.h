/* gsloc_icp.h -- C ABI of the CPU scan-to-scan ICP baseline (libgsloc_icp.so, host only: g++ + OpenMP).
 *
 * Stands in for the third-party small_gicp package (C++17/OpenMP behind pybind11, un-vendored, sources absent)
 * at the call sites of the reference:
 *   /root/reference/src/component/tracker.py:94-127   small_gicp.PointCloud / KdTree /
 *                                                     estimate_normals_covariances / preprocess_points / align
 *   /root/reference/src/my_gsplat/utils.py:16-22      small_gicp.KdTree(...).batch_knn_search
 * The algorithm is restated from small_gicp's published design (kd-tree with nearest-neighbour search, k-NN
 * covariance + normal estimation, ICP / point-to-plane / GICP factors, Levenberg-Marquardt on SE(3)); no
 * fixture of the reference pins it: PARITY UNPINNED.  This is the comparison baseline of BASELINE.json
 * configs[0] ("plumbing, no GPU"), not part of the GPU hot path.
 *
 * Conventions: double precision; points are rows of `stride` doubles (>= 3, the first three are xyz);
 * transforms are 4x4 row-major, T_target_source maps source coordinates into the target frame;
 * every function returns 0 or a negative gsl_icp_status; handles are owned by the caller.
 */
#ifndef GSLOC_ICP_H
#define GSLOC_ICP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum gsl_icp_status { GSL_ICP_OK = 0, GSL_ICP_BAD_ARG = -1, GSL_ICP_NO_TREE = -2, GSL_ICP_NO_ATTR = -3 };
enum gsl_icp_type { GSL_ICP_POINT = 0, GSL_ICP_PLANE = 1, GSL_ICP_GICP = 2 };

typedef struct gsl_icp_cloud gsl_icp_cloud; /* points (+ normals, covariances once estimated) + kd-tree */

typedef struct gsl_icp_result {
  double T_target_source[16]; /* row-major 4x4 */
  double H[36];               /* last linearised system */
  double b[6];
  double error;
  int32_t converged;
  int32_t iterations;
  int64_t num_inliers;
} gsl_icp_result;

const char* gsl_icp_version(void);

/* small_gicp.PointCloud(points): copies n rows. */
gsl_icp_cloud* gsl_icp_cloud_create(const double* points, int64_t n, int stride);
void gsl_icp_cloud_destroy(gsl_icp_cloud* cloud);
int64_t gsl_icp_cloud_size(const gsl_icp_cloud* cloud);
/* what: 0 points [n,3], 1 normals [n,3], 2 covariances [n,9]. */
int gsl_icp_cloud_read(const gsl_icp_cloud* cloud, int what, double* out);

/* small_gicp.KdTree(cloud, num_threads). */
int gsl_icp_build_tree(gsl_icp_cloud* cloud, int num_threads);
/* KdTree.batch_knn_search(queries, k): indices [m,k] int64 (-1 where fewer than k points), squared distances
 * [m,k], ascending. */
int gsl_icp_knn(const gsl_icp_cloud* cloud, const double* queries, int64_t m, int stride, int k, int64_t* indices,
                double* sq_dists, int num_threads);

/* small_gicp.estimate_normals_covariances(cloud, tree, num_neighbors, num_threads): per point the covariance of
 * its k nearest neighbours; normal = eigenvector of the smallest eigenvalue, flipped towards the origin;
 * covariance regularised to eigenvalues (1e-3, 1, 1) in its own eigenbasis. */
int gsl_icp_estimate_normals_covariances(gsl_icp_cloud* cloud, int num_neighbors, int num_threads);

/* small_gicp.voxelgrid_sampling (first half of preprocess_points): one averaged point per occupied voxel of edge
 * `leaf`, voxels in ascending key order.  Returns a new cloud (NULL on bad arguments). */
gsl_icp_cloud* gsl_icp_voxel_downsample(const gsl_icp_cloud* cloud, double leaf, int num_threads);

/* small_gicp.align(target, source, target_tree, init_T_target_source, max_correspondence_distance,
 * registration_type, num_threads, max_iterations): Levenberg-Marquardt (lambda 1e-3, factor 10, 10 inner
 * trials) until the update is below 0.1 degree and 1e-3 in translation or max_iterations (20 when <= 0).
 * The target needs its tree (and normals for PLANE, covariances for GICP); the source needs covariances for
 * GICP. */
int gsl_icp_align(const gsl_icp_cloud* target, const gsl_icp_cloud* source, const double* init_T_target_source,
                  double max_correspondence_distance, int registration_type, int max_iterations, int num_threads,
                  gsl_icp_result* result);

#ifdef __cplusplus
}
#endif
#endif
