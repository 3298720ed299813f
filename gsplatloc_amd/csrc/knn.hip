// Exact k-nearest-neighbour distances of a point cloud to itself on the GPU (per-frame set-up of the
// tracker).  Replaces the host-side small_gicp KdTree of /root/reference/src/my_gsplat/utils.py:16-22
// (called twice per frame through init_gs_scales, geometry.py:44-66; ~0.4 s per call for 816 k points
// on the host, more than the 200 optimisation iterations it prepares).
//
// Uniform grid of 128^3 cubic cells over the bounding cube: count -> (host-free) offsets -> fill, then one
// thread per query point scans shells of cells around its own cell and keeps the k smallest squared
// distances in registers.  The result is exact: the search radius grows until the k-th distance is not
// larger than the distance to the nearest unsearched cell face.
#include "gsloc_common.h"

namespace gsl {

#define GSL_KNN_G 128
#define GSL_KNN_MAXK 8

struct GridGeom {
  float ox, oy, oz, inv_h, h;
};

__device__ __forceinline__ GridGeom grid_geom(const float* __restrict__ bbox) {
  GridGeom g;
  float ex = bbox[3] - bbox[0], ey = bbox[4] - bbox[1], ez = bbox[5] - bbox[2];
  float ext = fmaxf(fmaxf(ex, ey), fmaxf(ez, 1e-12f));
  g.h = ext * (1.0f / (GSL_KNN_G - 1)) * 1.0001f;  // points on the max face still land in cell G-1
  g.inv_h = 1.0f / g.h;
  g.ox = bbox[0]; g.oy = bbox[1]; g.oz = bbox[2];
  return g;
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h) {
  int c = (int)floorf((v - o) * inv_h);
  return c < 0 ? 0 : (c > GSL_KNN_G - 1 ? GSL_KNN_G - 1 : c);
}

__global__ __launch_bounds__(256) void k_knn_count(const float* __restrict__ pts, int N, const float* __restrict__ bbox,
                                                   int32_t* __restrict__ cell_of, int32_t* __restrict__ counts) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  GridGeom g = grid_geom(bbox);
  int cx = cell_coord(pts[3 * (size_t)i], g.ox, g.inv_h), cy = cell_coord(pts[3 * (size_t)i + 1], g.oy, g.inv_h),
      cz = cell_coord(pts[3 * (size_t)i + 2], g.oz, g.inv_h);
  int c = (cz * GSL_KNN_G + cy) * GSL_KNN_G + cx;
  cell_of[i] = c;
  atomicAdd(&counts[c], 1);
}

// offsets = inclusive cumsum of counts (computed by the caller); start(c) = offsets[c] - counts[c]
__global__ __launch_bounds__(256) void k_knn_fill(const float* __restrict__ pts, int N,
                                                  const int32_t* __restrict__ cell_of,
                                                  const int32_t* __restrict__ counts,
                                                  const int32_t* __restrict__ incl, int32_t* __restrict__ cursors,
                                                  float4* __restrict__ sorted) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int c = cell_of[i];
  int pos = incl[c] - counts[c] + atomicAdd(&cursors[c], 1);
  sorted[pos] = make_float4(pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2], __int_as_float(i));
}

template <int K>
__global__ __launch_bounds__(256) void k_knn_query(const float* __restrict__ pts, int N,
                                                   const float* __restrict__ bbox, const int32_t* __restrict__ counts,
                                                   const int32_t* __restrict__ incl,
                                                   const float4* __restrict__ sorted, float* __restrict__ dists) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  GridGeom g = grid_geom(bbox);
  float px = pts[3 * (size_t)i], py = pts[3 * (size_t)i + 1], pz = pts[3 * (size_t)i + 2];
  int cx = cell_coord(px, g.ox, g.inv_h), cy = cell_coord(py, g.oy, g.inv_h), cz = cell_coord(pz, g.oz, g.inv_h);
  float best[K];
#pragma unroll
  for (int k = 0; k < K; ++k) best[k] = 3.0e38f;
  // position inside the own cell: distance to the searched block's faces after radius R is
  // R*h + min(offset to the cell's faces)
  float fx = (px - g.ox) - cx * g.h, fy = (py - g.oy) - cy * g.h, fz = (pz - g.oz) - cz * g.h;
  float inner = fminf(fminf(fminf(fx, g.h - fx), fminf(fy, g.h - fy)), fminf(fz, g.h - fz));
  inner = fmaxf(inner, 0.f);
  for (int R = 0; R < GSL_KNN_G; ++R) {
    // visit the shell of Chebyshev radius R around (cx,cy,cz)
    int z0 = max(cz - R, 0), z1 = min(cz + R, GSL_KNN_G - 1);
    int y0 = max(cy - R, 0), y1 = min(cy + R, GSL_KNN_G - 1);
    int x0 = max(cx - R, 0), x1 = min(cx + R, GSL_KNN_G - 1);
    auto visit = [&](int x, int y, int z) {
      int c = (z * GSL_KNN_G + y) * GSL_KNN_G + x;
      int n = counts[c];
      if (n == 0) return;
      int s = incl[c] - n;
      for (int t = 0; t < n; ++t) {
        float4 q = sorted[s + t];
        float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        float d = dx * dx + dy * dy + dz * dz;
        if (d < best[K - 1]) {
          best[K - 1] = d;
#pragma unroll
          for (int k = K - 1; k > 0; --k) {
            if (best[k] < best[k - 1]) { float tmp = best[k]; best[k] = best[k - 1]; best[k - 1] = tmp; }
          }
        }
      }
    };
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        if ((abs(z - cz) == R) || (abs(y - cy) == R)) {
          for (int x = x0; x <= x1; ++x) visit(x, y, z);   // a face of the shell: the whole row
        } else {                                            // interior row: only its two end cells
          if (cx - R >= 0) visit(cx - R, y, z);
          if (cx + R <= GSL_KNN_G - 1) visit(cx + R, y, z);
        }
      }
    float reach = (float)R * g.h + inner;  // every unsearched cell is at least this far away
    if (best[K - 1] <= reach * reach) break;
    if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == GSL_KNN_G - 1 && y1 == GSL_KNN_G - 1 && z1 == GSL_KNN_G - 1) break;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) dists[(size_t)i * K + k] = best[k];
}

}  // namespace gsl

extern "C" size_t gsl_knn_ws_bytes(int N) {
  size_t cells = (size_t)GSL_KNN_G * GSL_KNN_G * GSL_KNN_G;
  // [counts cells][cursors cells][cell_of N] int32 + [sorted N] float4   (the inclusive scan lives with the caller)
  return cells * 2 * sizeof(int32_t) + (size_t)(N > 0 ? N : 0) * (sizeof(int32_t) + sizeof(float) * 4) + 16;
}

extern "C" int gsl_knn_cells(void) { return GSL_KNN_G * GSL_KNN_G * GSL_KNN_G; }

// Phase 1: per-cell counts (ws[0 .. cells)).  bbox[6] = (min xyz, max xyz) on the device.
extern "C" int gsl_knn_count(const float* points, int N, const float* bbox, void* ws, size_t ws_bytes, void* stream) {
  if (N < 0 || !bbox || !ws) return GSL_ERR_BAD_ARG;
  if (ws_bytes < gsl_knn_ws_bytes(N)) return GSL_ERR_WORKSPACE;
  size_t cells = (size_t)gsl_knn_cells();
  int32_t* counts = (int32_t*)ws;
  int32_t* cell_of = counts + 2 * cells;
  hipStream_t st = (hipStream_t)stream;
  if (gsl::zero_u32(counts, (size_t)cells * 2, st) != GSL_OK) return GSL_ERR_HIP;
  if (N == 0) return GSL_OK;
  if (!points) return GSL_ERR_BAD_ARG;
  hipLaunchKernelGGL(gsl::k_knn_count, dim3((N + 255) / 256), dim3(256), 0, st, points, N, bbox, cell_of, counts);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// Phase 2: incl_offsets[cells] = inclusive cumulative sum of the counts (caller computes it on the device);
// squared distances to the k nearest points (self included), ascending, into dists[N,k].  k <= 8.
extern "C" int gsl_knn_query(const float* points, int N, const float* bbox, const int32_t* incl_offsets, int k,
                             float* dists, void* ws, size_t ws_bytes, void* stream) {
  if (N < 0 || k < 1 || k > GSL_KNN_MAXK || !bbox || !ws || !incl_offsets) return GSL_ERR_BAD_ARG;
  if (ws_bytes < gsl_knn_ws_bytes(N)) return GSL_ERR_WORKSPACE;
  if (N == 0) return GSL_OK;
  if (!points || !dists) return GSL_ERR_BAD_ARG;
  size_t cells = (size_t)gsl_knn_cells();
  int32_t* counts = (int32_t*)ws;
  int32_t* cursors = counts + cells;
  int32_t* cell_of = counts + 2 * cells;
  float4* sorted = (float4*)(cell_of + N + ((4 - (N & 3)) & 3));  // 16-byte aligned
  hipStream_t st = (hipStream_t)stream;
  int grid = (N + 255) / 256;
  hipLaunchKernelGGL(gsl::k_knn_fill, dim3(grid), dim3(256), 0, st, points, N, cell_of, counts, incl_offsets, cursors,
                     sorted);
  GSL_CHECK_LAUNCH();
#define KNN_CALL(KK)                                                                                            \
  hipLaunchKernelGGL(gsl::k_knn_query<KK>, dim3(grid), dim3(256), 0, st, points, N, bbox, counts, incl_offsets, \
                     sorted, dists)
  switch (k) {
    case 1: KNN_CALL(1); break;
    case 2: KNN_CALL(2); break;
    case 3: KNN_CALL(3); break;
    case 4: KNN_CALL(4); break;
    case 5: KNN_CALL(5); break;
    case 6: KNN_CALL(6); break;
    case 7: KNN_CALL(7); break;
    default: KNN_CALL(8); break;
  }
#undef KNN_CALL
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
