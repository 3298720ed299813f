/*
 * gsloc_hip.h -- C ABI of libgsloc_hip.so: the MI355X (gfx950) Gaussian-splat
 * rasterizer behind GsplatLoc's pose-tracking loop.
 *
 * Boundary being replaced.  The reference reaches this arithmetic through the
 * third-party pybind11/torch extension `gsplat.csrc` (`_C`, IDX:14216 of
 * /root/reference/.vscode/PythonImportHelper-v2-Completion.json) whose Python
 * wrappers are called from /root/reference/src/my_gsplat/model.py:195-213 and
 * /root/reference/src/my_gsplat/geometry.py:117-132.  Nothing C-callable exists
 * in the reference; every entry point below names the gsplat operator (IDX line)
 * whose work it performs, and INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - all floating point data is fp32, row-major, contiguous; indices are int32,
 *     intersection keys int64;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - no allocation, no host synchronisation, no global state inside any call
 *     (safe to capture in a hipGraph); scratch comes from the caller through the
 *     `*_ws` arguments, sized by the matching `*_ws_bytes` query;
 *   - return value: GSL_OK (0) or a negative gsl_status; the launch error of the
 *     last kernel (hipGetLastError) is reported as GSL_ERR_HIP.
 */
#ifndef GSLOC_HIP_H
#define GSLOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gsl_status {
  GSL_OK = 0,
  GSL_ERR_BAD_ARG = -1,   /* null pointer / non-positive size / unsupported channel count */
  GSL_ERR_WORKSPACE = -2, /* workspace too small */
  GSL_ERR_HIP = -3        /* HIP runtime reported an error for a launch */
} gsl_status;

/* Library identification: returns "gsloc_hip <version> gfx950". */
const char* gsl_version(void);
/* Text for a gsl_status. */
const char* gsl_status_string(int status);
/* Diagnostics (tests): fill the LDS of every CU with `pattern` (e.g. 0xFFFFFFFF, a NaN), to show that no kernel of the
 * library reads LDS it has not written. */
int gsl_dev_poison_lds(uint32_t pattern, void* stream);
/* ---- projection: gsplat.fully_fused_projection fwd/bwd (IDX:14351, IDX:14270) ----
 * One camera.  viewmat[16] world->camera row-major, K[9] intrinsics, both on device.
 * Outputs for culled Gaussians: radii = 0, other outputs 0.
 * compensations may be NULL (rasterize_mode "classic"). */
int gsl_project_fwd(const float* means, const float* quats, const float* scales,
                    const float* viewmat, const float* K, int N, int width, int height,
                    float eps2d, float near_plane, float far_plane, float radius_clip,
                    int32_t* radii, float* means2d, float* depths, float* conics,
                    float* compensations, void* stream);

/* vjp of the above.  v_means/v_quats/v_scales may be NULL together (pose-only
 * mode); v_viewmat may be NULL.  v_viewmat[16] is OVERWRITTEN with this camera's
 * gradient (rows 0..2 used; row 3 zero).  v_compensations may be NULL.
 * ws: gsl_project_bwd_ws_bytes(N) bytes. */
size_t gsl_project_bwd_ws_bytes(int N);
int gsl_project_bwd(const float* means, const float* quats, const float* scales,
                    const float* viewmat, const float* K, int N, int width, int height,
                    float eps2d, const int32_t* radii, const float* conics,
                    const float* compensations, const float* v_means2d, const float* v_depths,
                    const float* v_conics, const float* v_compensations, float* v_means,
                    float* v_quats, float* v_scales, float* v_viewmat, void* ws, size_t ws_bytes,
                    void* stream);

/* ---- spherical harmonics: gsplat.spherical_harmonics fwd/bwd (IDX:14306, IDX:14297) ----
 * dirs[M,3], coeffs[M,K,3] with K >= (degree+1)^2, masks[M] (uint8, may be NULL).
 * degree 0..3.  v_dirs may be NULL. */
int gsl_sh_fwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks, int M,
               int K, float* colors, void* stream);
int gsl_sh_bwd(int degree, const float* dirs, const float* coeffs, const uint8_t* masks, int M,
               int K, const float* v_colors, float* v_coeffs, float* v_dirs, void* stream);

/* ---- tile binning: gsplat.isect_tiles + isect_offset_encode (IDX:14360, IDX:14369) ----
 * Tiles are tile_size x tile_size pixels; tile rows [ty0, ty1) of the tile_w x tile_h grid
 * are binned (ty0=0, ty1=tile_h for the whole image; a strip for tile-parallel ranks).
 * Tile ids in keys and offsets are GLOBAL (ty*tile_w+tx).
 *
 * gsl_isect_count: tiles_per_gauss[N] (may be NULL), tile_offsets[n_tiles+1] exclusive scan over
 *   all tile_w*tile_h tiles (tiles outside the strip are empty), n_isects[1] = total.
 *   ws: gsl_isect_ws_bytes(tile_w*tile_h).
 * gsl_isect_fill: after gsl_isect_count with the same arguments.  Writes, for every tile,
 *   its intersections sorted by (float32 depth bits, Gaussian index) ascending:
 *   flatten_ids[I] and, if not NULL, isect_ids[I] = (cam_id << (32+tile_n_bits)) | (tile_id << 32) | depth bits.
 *   sort_keys[capacity] is scratch (8 bytes per intersection).  Intersections with
 *   position >= capacity are dropped (caller compares n_isects with capacity). */
size_t gsl_isect_ws_bytes(int n_tiles);
int gsl_isect_count(const float* means2d, const int32_t* radii, int N, int tile_size, int tile_w,
                    int tile_h, int ty0, int ty1, int32_t* tiles_per_gauss, int32_t* tile_offsets,
                    int32_t* n_isects, void* ws, size_t ws_bytes, void* stream);
int gsl_isect_fill(const float* means2d, const int32_t* radii, const float* depths, int N,
                   int tile_size, int tile_w, int tile_h, int ty0, int ty1, int cam_id,
                   int tile_n_bits, const int32_t* tile_offsets, int64_t capacity,
                   uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, void* ws,
                   size_t ws_bytes, void* stream);

/* Sort every tile bucket of tiles [tile_begin, tile_begin+n_strip_tiles) on (depth bits, id) and
 * write flatten_ids / isect_ids (second half of gsl_isect_fill; cam_enc is OR-ed into isect_ids). */
int gsl_tile_sort(const int32_t* tile_offsets, int tile_begin, int n_strip_tiles, int64_t capacity,
                  uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, int64_t cam_enc,
                  void* stream);

/* Unsorted emit in Gaussian order (isect_tiles(sort=False)): cum_tiles[N] is the
 * INCLUSIVE cumulative sum of tiles_per_gauss (int64). */
int gsl_isect_emit(const float* means2d, const int32_t* radii, const float* depths,
                   const int64_t* cum_tiles, int N, int tile_size, int tile_w, int tile_h,
                   int cam_id, int tile_n_bits, int id_offset, int64_t* isect_ids,
                   int32_t* flatten_ids, void* stream);

/* Tile start offsets from sorted keys (isect_offset_encode): offsets[n_cameras*n_tiles]. */
int gsl_isect_offsets(const int64_t* isect_ids, int64_t n_isects, int n_cameras, int n_tiles,
                      int tile_n_bits, int32_t* offsets, void* stream);

/* ---- compositing: gsplat.rasterize_to_pixels fwd/bwd (IDX:14378, IDX:14279) ----
 * One camera.  channels in {1,2,3,4,5,8,16,32}.  tile_offsets[n_tiles+1] (global tile ids),
 * flatten_ids index the per-Gaussian arrays directly.  Rows of tile rows [ty0,ty1) are
 * rendered into full-size images (pixels outside the strip are left untouched).
 * backgrounds[channels] may be NULL. */
int gsl_rasterize_fwd(const float* means2d, const float* conics, const float* colors,
                      const float* opacities, const float* backgrounds, int channels, int width,
                      int height, int tile_size, int tile_w, int tile_h, int ty0, int ty1,
                      const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                      float* render_colors, float* render_alphas, int32_t* last_ids, void* stream);

/* vjp.  Per-Gaussian gradients are ACCUMULATED INTO `vacc`, one padded row per Gaussian:
 * [v_means2d 2][v_conics 3][v_opacity 1][v_colors channels], row pitch 16 floats for
 * channels <= 10 and 48 floats otherwise (gsl_vacc_bytes gives the size).  The caller zeroes
 * vacc before the first camera; rows are indexed by the same ids as flatten_ids.
 * gsl_vacc_unpack splits the rows into the four gsplat gradient tensors. */
size_t gsl_vacc_bytes(int n_gaussians, int channels);
int gsl_rasterize_bwd(const float* means2d, const float* conics, const float* colors,
                      const float* opacities, const float* backgrounds, int channels, int width,
                      int height, int tile_size, int tile_w, int tile_h, int ty0, int ty1,
                      const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                      const float* render_alphas, const int32_t* last_ids,
                      const float* v_render_colors, const float* v_render_alphas, float* vacc,
                      void* stream);
int gsl_vacc_unpack(const float* vacc, int n_gaussians, int channels, float* v_means2d,
                    float* v_conics, float* v_colors, float* v_opacities, void* stream);

/* ---- fused single-camera pipeline: gsplat.rasterization end to end (IDX:14954) ----
 * The hot path of GsplatLoc's tracker (model.py:195-213).  16x16 tiles.  Per-Gaussian state lives
 * in 16-byte records: Q0[N] = (x, y, depth, opacity_eff), Q1[N] = (conic a, b, c, r_cull),
 * Q2[N] = (r, g, b, 0) (NULL for depth-only rendering).  channels: 1 = depth, 3 = RGB,
 * 4 = RGB + depth; ed != 0 divides the depth channel by alpha (render modes "ED"/"RGB+ED").
 * colors: SH coefficients [N,K_sh,3] when sh_degree in 0..3, direct RGB [N,3] when sh_degree < 0.
 * ws: gsl_fused_ws_bytes(N, tile_w*tile_h), shared by the five calls of one render.
 *
 * gsl_fused_project : projection + colour + pack + tile histogram + scan
 *                     -> radii, Q0, Q1, Q2, tile_offsets[n_tiles+1], n_isects[1]
 * gsl_fused_bin     : scatter + per-tile sort -> flatten_ids (isect_ids optional)
 * gsl_fused_raster_fwd / _bwd : compositing and its vjp; the vjp ACCUMULATES into vacc, 16 floats
 *                     per Gaussian ([v_xy 2][v_conic 3][v_opacity 1][v_colour channels]), zero on entry.
 *                     Forward: every lane walks the candidate list of its own pixel (csrc/raster_px.hip);
 *                     backward: each 16-lane DPP row of a wave walks the list of its own 4x4 pixel block
 *                     (csrc/raster_g16.hip; deterministic mode: quadrant walk with per-splat pixel sums on the
 *                     matrix cores, v_mfma_f32_16x16x4_f32, exact f32, csrc/fused.hip).  isect_hits (uint32[4 x capacity])
 *                     + isect_hit_counts (int32[4 x n_tiles + 1]), may be NULL together in both calls: the forward leaves, per
 *                     tile and 8x8 quadrant, the list of the entries at least one of the quadrant's pixels composited --
 *                     (nibble of its four 4x4 blocks that did) << 28 | list index, in list order, at
 *                     isect_hits[4 start + quadrant x length ...] for the tile's list [start, start + length), its
 *                     length in isect_hit_counts[4 tile + quadrant] (the extra last int is a flag between the
 *                     backward's two launches) -- and the backward given the same arrays walks
 *                     exactly those (block, entry) pairs instead of scanning the tile's list and testing every splat's
 *                     alpha >= 1/255 disc against the blocks.  (Segments of long lists: same layout per segment, the
 *                     lengths live in long_ws.)  Only pixel rows [row0,row1)
 *                     of the tile rows [ty0,ty1) are rendered / back-propagated (whole strip: 0,height):
 *                     a strip's one-pixel Sobel halo costs one pixel row, not a tile row.
 * fp16 staging (Qh, may be NULL everywhere): gsl_fused_project additionally packs one 32-byte record per Gaussian
 *                     into Qh[N][8 dwords] -- centre float32, conic / cull radius / depth / opacity / colour as
 *                     halves -- and the compositing calls given Qh gather that record instead of Q0/Q1/Q2 (which may
 *                     then be NULL there; gsl_fused_project given Qh does not write Q2 at all).  Transmittance and every accumulator stay float32 (BASELINE.json
 *                     configs[4] "fp16 compositing").  Binning and gsl_fused_project_bwd keep reading
 *                     the float32 records, so the list order is the float32 order.
 * Deterministic mode (vrow, may be NULL): by default the backward accumulates with float atomics (LDS across the
 *                     four waves of a tile, memory across the tiles of a Gaussian), whose order varies from run to
 *                     run.  With write_sorted_keys = 1 in gsl_fused_bin (sort_keys then holds the sorted
 *                     (depth bits << 32 | id) keys) and vrow[capacity][16] given to gsl_fused_raster_bwd (vacc may be
 *                     NULL) and gsl_fused_project_bwd (with sorted_keys, tile_offsets, Q0 and the strip), every sum
 *                     has a fixed order: waves in wave order, a Gaussian's tiles in tile order (its entry in a tile
 *                     list is found by binary search).  Bit-identical gradients run to run.
 * Binned mode (bins, may be NULL): when the caller knows an upper bound of a tile's list length (a tracker renders the
 *                     same Gaussians hundreds of times per frame), gsl_fused_project given bins[n_tiles][bin_cap]
 *                     (uint64) writes every (depth bits << 32 | id) key straight into its tile's bin and leaves
 *                     the tile sizes in ws (it does NOT write tile_offsets / n_isects then); gsl_fused_bin given the
 *                     same bins adds the sizes up inside its sort kernel -- tile_offsets[n_tiles + 1] and n_isects
 *                     are its OUTPUTS in this mode -- and sorts; gsl_fused_raster_fwd given binned_ws = ws clears
 *                     the counters for the next projection.  The three calls belong together: the scatter pass,
 *                     its second read of the records, the scan launch and the counter-clearing launch disappear.
 *                     ws must be zero-filled once before the first call.  A tile that outgrows bin_cap keeps its
 *                     first bin_cap entries, raises flags[1] = 1 and leaves the largest count in flags[2]
 *                     (flags: 4 ints, may be NULL): poll it and re-run with larger bins.  Counter contract: the
 *                     counters in ws must be zero when gsl_fused_project starts; gsl_fused_raster_fwd(binned_ws = ws)
 *                     leaves them so.  A projection that follows another one without a compositing forward in
 *                     between (skipped, or failed) finds the state word in ws dirty and raises flags[3] = 1: zero-fill
 *                     ws and run the iteration again.
 * Tile-order placement (order_ids / storage_of, int32[N] each, may be NULL together everywhere): a caller that renders the
 *                     same Gaussians many times (a tracker: /root/reference/src/my_gsplat/model.py:137-175 builds them
 *                     once per frame) may STORE them sorted by the tile of their centre, so that the record gathers of a
 *                     tile's list touch a few contiguous runs.  The list order must not depend on the placement -- depth
 *                     ties break by Gaussian index -- so gsl_fused_project (and gsl_fused_bin's scatter pass) put
 *                     order_ids[slot], the Gaussian's ORIGINAL index, into the low key word, and every sort
 *                     (gsl_fused_bin, gsl_long_sort, the sorting gsl_fused_raster_fwd) writes storage_of[original] into
 *                     flatten_ids: the lists are the unpermuted run's lists with every id relabelled, images and
 *                     last_ids bit-identical.  Everything per Gaussian (records, vacc, v_means ...) is then in storage
 *                     order.  Not with write_sorted_keys (the deterministic backward searches the keys by id).
 * Sort keys: (depth bits << 32) | Gaussian index, the depth a positive finite normal float -- gsl_fused_project and
 *                     gsl_project_fwd clamp their depth window to [FLT_MIN, FLT_MAX], so every key this library makes is
 *                     one; the per-tile sorts (gsl_fused_bin, gsl_tile_sort_keys, gsl_long_sort, the sorting forward)
 *                     compare keys as doubles on that ground and order other bit patterns differently from an unsigned
 *                     compare (depths handed to the stage-wise isect entry points must be positive and finite).
 * Limits: N <= 2^26 Gaussians per call (gsl_fused_project returns GSL_ERR_BAD_ARG beyond: the compositing backward
 *                     addresses the 64-byte gradient rows of vacc by 32-bit byte offsets).
 * gsl_fused_project_bwd : consumes AND CLEARS vacc; v_means/v_quats/v_scales/v_opacities (and
 *                     v_colors, shaped like colors) may be NULL together (pose-only);
 *                     v_viewmat[16] is overwritten (row 3 = 0).  tiny_trec / tiny_vcT (may be NULL): the slabs
 *                     gsl_tiny_raster_bwd filled; the kernel then folds them itself (pass 2 of the tiny-splat
 *                     backward fused in: the gradient rows never leave LDS, vacc is not touched, Q0 required)
 *                     instead of reading vacc.  reduce_viewmat = 0 skips that last reduction
 *                     launch and leaves the pose gradient as ceil(N/256) partial rows of 16 floats at
 *                     gsl_fused_viewmat_rows(ws, n_tiles) for gsl_pose_step / gsl_pack_pose_reduce to sum
 *                     (same fixed order, same result; v_viewmat may then be NULL). */
size_t gsl_fused_ws_bytes(int N, int n_tiles);
int gsl_fused_project(const float* means, const float* quats, const float* scales,
                      const float* opacities, const float* colors, int sh_degree, int K_sh,
                      const float* viewmat, const float* K, int N, int width, int height,
                      float eps2d, float near_plane, float far_plane, float radius_clip,
                      int antialiased, int tile_w, int tile_h, int ty0, int ty1, int32_t* radii,
                      float* Q0, float* Q1, float* Q2, float* compensations,
                      int32_t* tiles_per_gauss, int32_t* tile_offsets, int32_t* n_isects, void* ws,
                      size_t ws_bytes, void* Qh, void* bins, int bin_cap, int32_t* flags,
                      const int32_t* order_ids, void* stream);
int gsl_fused_bin(const float* Q0, const int32_t* radii, int N, int tile_w, int tile_h, int ty0,
                  int ty1, int tile_n_bits, int32_t* tile_offsets, int64_t capacity,
                  uint64_t* sort_keys, int32_t* flatten_ids, int64_t* isect_ids, void* ws,
                  size_t ws_bytes, int write_sorted_keys, void* bins, int bin_cap, int32_t* n_isects,
                  int32_t* flags, int long_min, const int32_t* order_ids, const int32_t* storage_of, void* stream);
int gsl_fused_raster_fwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed,
                         int width, int height, int tile_w, int tile_h, int ty0, int ty1,
                         const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                         float* render, float* alphas, int32_t* last_ids, int row0, int row1, const void* Qh,
                         void* binned_ws, uint32_t* isect_hits, int32_t* isect_hit_counts, int long_min,
                         void* sort_bins, int bin_cap, int32_t* n_isects, int32_t* flags,
                         const int32_t* storage_of, void* stream);
/* sort_bins != NULL (binned projection, whole frame, bin_cap <= 2048, long_min == 0): every workgroup first does
 * gsl_fused_bin's work for its own tile -- adds up the sizes of the tiles before it, sorts its bin in LDS, writes
 * tile_offsets / flatten_ids (outputs then, despite the const) / n_isects and raises flags like gsl_fused_bin -- and
 * gsl_fused_bin is NOT called: the tracker's iteration goes without the sort launch.  The tile counters in binned_ws then
 * stay set until a compositing backward that is given clear_ws (gsl_fused_raster_bwd / gsl_tiny_raster_bwd) clears
 * them; after a forward nobody back-propagates the caller zeroes binned_ws before the next gsl_fused_project. */
int gsl_fused_raster_bwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed,
                         int width, int height, int tile_w, int tile_h, int ty0, int ty1,
                         const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                         const float* render, const float* alphas, const int32_t* last_ids,
                         const float* v_render, const float* v_alphas, float* vacc, int row0, int row1,
                         const void* Qh, float* vrow, const uint32_t* isect_hits,
                         const int32_t* isect_hit_counts, int long_min, void* clear_ws, void* stream);
/* clear_ws (may be NULL): the binned_ws whose tile counters this backward clears (see gsl_fused_raster_fwd, sort_bins). */
/* Long tile lists split over workgroups (long_min > 0 in the calls above and in gsl_tiny_raster_bwd: tiles whose list
 * is longer than long_min entries are skipped there and handled here).  A pile of splats in one tile -- the invalid
 * pixels of a TUM depth frame, /root/reference/src/data/Image.py:29-35 -- is cut into segments of gsl_long_segment() entries, one
 * workgroup each: the forward computes per-pixel segment transmittances, restarts every segment from the product of
 * the earlier ones and combines the partial images; the backward restarts every segment from the stored state.
 * long_ws: gsl_long_ws_bytes(max_seg) bytes, zero-filled once; max_seg bounds the (tile, segment) pairs of a frame
 * (if a frame has more, long_ws[1] (int32) is set to the number needed: poll it, grow, re-run).  Call
 * gsl_long_raster_fwd after gsl_fused_raster_fwd, gsl_long_raster_bwd after gsl_fused_raster_bwd / gsl_tiny_raster_bwd
 * (it adds into vacc; gsl_fused_project_bwd given both tiny_trec and vacc consumes both). */
size_t gsl_long_ws_bytes(int max_seg);
int gsl_long_segment(void);      /* entries per compositing segment (callers size max_seg with it) */
int gsl_long_sort_segment(void); /* keys per sorted run of gsl_long_sort before its merge passes (sizes `passes`) */
/* Binned mode only: gsl_fused_bin(long_min > 0) leaves the lists longer than long_min unsorted, and gsl_long_sort (call it
 * right after) sorts them with one wave per gsl_long_sort_segment() keys (registers) + `passes` merge-path passes (2^passes such runs per
 * list at most; a longer list sets long_ws[2] (int32)), ping-ponging between sort_keys (scratch, packed) and the tile's
 * bin; flatten_ids receives the result.  One workgroup used to take 1.6 ms for a 23 k-entry list. */
int gsl_long_sort(const int32_t* tile_offsets, int tile_w, int tile_h, int ty0, int ty1, int64_t capacity,
                  uint64_t* bins, int bin_cap, uint64_t* sort_keys, int32_t* flatten_ids, int long_min,
                  void* long_ws, size_t long_ws_bytes, int max_seg, int passes, const int32_t* storage_of,
                  void* stream);
int gsl_long_raster_fwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                        int height, int tile_w, int tile_h, int ty0, int ty1, const int32_t* tile_offsets,
                        const int32_t* flatten_ids, int64_t capacity, float* render, float* alphas,
                        int32_t* last_ids, int row0, int row1, const void* Qh, uint32_t* isect_hits,
                        int long_min, void* long_ws, size_t long_ws_bytes, int max_seg, int map_ready, void* stream);
/* map_ready != 0: gsl_long_sort has already listed this frame's (tile, segment) pairs for the same strip, long_min and
 * max_seg in long_ws (binned mode); 0: list them here. */
int gsl_long_raster_bwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed, int width,
                        int height, int tile_w, int tile_h, int ty0, int ty1, const int32_t* tile_offsets,
                        const int32_t* flatten_ids, int64_t capacity, const float* render,
                        const float* alphas, const int32_t* last_ids, const float* v_render,
                        const float* v_alphas, float* vacc, int row0, int row1, const void* Qh,
                        const uint32_t* isect_hits, int long_min, void* long_ws, int max_seg, void* stream);
int gsl_fused_project_bwd(const float* means, const float* quats, const float* scales,
                          const float* opacities, const float* colors, int sh_degree, int K_sh,
                          const float* viewmat, const float* K, int N, int width, int height,
                          float eps2d, int antialiased, int channels, const int32_t* radii,
                          const float* Q1, const float* compensations, float* vacc, float* v_means,
                          float* v_quats, float* v_scales, float* v_opacities, float* v_colors,
                          float* v_viewmat, void* ws, size_t ws_bytes, int n_tiles, const float* vrow,
                          const uint64_t* sorted_keys, const int32_t* tile_offsets, const float* Q0, int tile_w,
                          int tile_h, int ty0, int ty1, int64_t capacity, float* tiny_trec, const float* tiny_vcT,
                          int reduce_viewmat, int32_t* v_colors_state, void* stream);
/* v_colors_state (int32[1], may be NULL): the caller's promise that v_colors is the SAME buffer in every call that is
 * given this state word; initialise it to 1 together with a zero-filled v_colors (or to 0 for a buffer of unknown
 * content).  While it reads 1 a Gaussian whose colour gradient is zero -- all of them under a depth-only loss,
 * /root/reference/src/my_gsplat/gs_trainer_total.py:111-123 -- skips its zero stores (48 of the ~92 bytes per Gaussian the
 * call writes at SH degree 1); a launch that writes a real colour gradient marks the buffer dirty and the following
 * launch stores everything again.  Maintained by the reduction kernel (reduce_viewmat = 1). */
const float* gsl_fused_viewmat_rows(const void* ws, int n_tiles);

/* "Tiny splat" backward: valid when every r_cull (Q1[:,3]) is < 2 px, i.e. no splat reaches more than 4x4
 * pixel centres (GsplatLoc's as-coded scales).  gsl_tiny_raster_bwd replaces gsl_fused_raster_bwd: instead of
 * reducing and accumulating gradient rows it stores per (splat, pixel) records into trec[N][16][2] (zero on
 * entry) and the chained upstream gradient of every pixel into vcT[H,W,channels]; gsl_fused_project_bwd given
 * tiny_trec / tiny_vcT then sums each Gaussian's 4x4 slab into its gradient row (in LDS, four lanes per Gaussian)
 * and clears the slab.  flags (int32[1], may be NULL): flags[0] is set to 1 when a
 * (pixel, splat) pair fell outside its slab, i.e. the precondition did not hold and the gradients are
 * incomplete -- the caller polls it and re-runs the iteration with gsl_fused_raster_bwd. */
int gsl_tiny_raster_bwd(const float* Q0, const float* Q1, const float* Q2, int channels, int ed,
                        int width, int height, int tile_w, int tile_h, int ty0, int ty1,
                        const int32_t* tile_offsets, const int32_t* flatten_ids, int64_t capacity,
                        const float* render, const float* alphas, const int32_t* last_ids,
                        const float* v_render, const float* v_alphas, float* trec, float* vcT,
                        int row0, int row1, int32_t* flags, int long_min, const float* loss_depth_gt,
                        float depth_lambda, float edge_lambda, float* loss_partials, void* clear_ws, void* stream);
/* loss_depth_gt != NULL (whole frame, channels 1 or 4): the kernel computes the tracking loss of gsl_tracking_loss for
 * its tile itself -- the tile is that kernel's 16x16 block -- WRITES v_render's depth channel and
 * loss_partials[tiles][2], and back-propagates from that gradient (v_render's other channels count as zero): the
 * tracker's iteration then goes without the separate loss launch.  NULL: v_render is read as given. */

/* ---- tracker tail: loss + pose update on device (csrc/tracker.hip) ----
 * Replaces the PyTorch/kornia glue of one iteration of GsplatLoc's Runner.train
 * (/root/reference/src/my_gsplat/gs_trainer_total.py:105-185,265-267; loss.py:10-59; model.py:79-116).
 *
 * gsl_tracking_loss: depth L1 + Sobel-edge L1 of render[...,channels-1] against depth_gt[H,W] over the owned
 *   pixel rows [row0,row1) (whole image: 0,height), normalised by width*height; writes d loss / d depth into
 *   v_render[...,channels-1] for rows [row0-1,row1+1) and the block sums (sum|a-b|, sum|Sa-Sb|) into
 *   loss_partials[gsl_loss_n_partials(width,height,row0,row1)][2] (inside ws when NULL; ws may be NULL otherwise).
 *   One launch: 16x16 pixel blocks with a two-pixel apron through LDS.  ws: gsl_loss_ws_bytes.
 * gsl_pose_init : pose state <- init_c2w (wxyz quaternion + translation), zero Adam moments, step 0; writes
 *   c2w[16] and viewmat[16] = c2w^-1.  pose_f[32] floats, pose_i[4] ints (layout: csrc/tracker.hip).
 * gsl_pose_step : finish the loss, pose errors vs gt_c2w, early-stop bookkeeping (best loss after min_step,
 *   patience), pose chain viewmat->(quat,t), two Adam updates (weight decay in the gradient), lr *= gamma,
 *   new c2w / viewmat; appends the loss to loss_hist[max_steps] (may be NULL).  Does nothing once stopped
 *   (pose_i[2]).  The pose gradient is v_viewmat[16], or -- vm_rows not NULL -- the n_vm_rows partial rows a
 *   gsl_fused_project_bwd(reduce_viewmat = 0) left (K and the viewmat buffer, which still holds the rendered pose, are
 *   read for the camera-position chain).  loss_sums[3] (already summed over ranks: the two sums and the row-cosine sum) overrides
 *   loss_partials / normal_sum when not NULL.  normal_lambda != 0 adds normal_lambda * (1 - normal_sum / (3 height)).
 * gsl_normal_loss: the normal-consistency term the reference defines and keeps switched off (loss.py:62-101 "cosine",
 *   geometry.py:164-197; call commented out at gs_trainer_total.py:138-143, normal_lambda = 0 at data/base.py:28):
 *   unit normals of the back-projected masked depth images (central differences, replicated borders), cosine
 *   similarity ALONG EACH IMAGE ROW per component as the reference codes it, loss = 1 - mean.  Adds
 *   normal_lambda * d loss / d depth to v_render[...,channels-1] for rows [row0-1,row1+1) (call it after
 *   gsl_tracking_loss, which writes that channel) and writes the sum of the owned rows' cosines to normal_sum[0].
 *   ws: gsl_normal_ws_bytes. */
size_t gsl_loss_ws_bytes(int width, int height);
int gsl_loss_n_partials(int width, int height, int row0, int row1);
size_t gsl_normal_ws_bytes(int width, int height);
int gsl_normal_loss(const float* render, int channels, const float* depth_gt, int width, int height, int row0,
                    int row1, float fx, float fy, float cx, float cy, float normal_lambda, float* v_render,
                    float* normal_sum, void* ws, size_t ws_bytes, void* stream);
int gsl_tracking_loss(const float* render, int channels, const float* depth_gt, int width, int height,
                      int row0, int row1, float depth_lambda, float edge_lambda, float* v_render,
                      float* loss_partials, int* n_partials_host, void* ws, size_t ws_bytes, void* stream);
int gsl_pose_init(float* pose_f, int* pose_i, const float* init_c2w, float lr_quat, float lr_trans,
                  float* c2w, float* viewmat, void* stream);
int gsl_pose_step(float* pose_f, int* pose_i, const float* v_viewmat, const float* vm_rows, int n_vm_rows,
                  const float* K, const float* loss_partials, int n_partials, const float* loss_sums, const float* normal_sum, const float* gt_c2w,
                  int width, int height, float depth_lambda, float edge_lambda, float normal_lambda,
                  float beta1, float beta2, float eps,
                  float wd_quat, float wd_trans, float gamma, int min_step, int patience, int early_stop,
                  int max_steps, float* c2w, float* viewmat, float* loss_hist, void* stream);
/* Several GPUs: what one rank contributes to the ONE all-reduce of an iteration.  out16[0..11] =
 * v_viewmat[0..11] of its strip (given reduced, or as vm_rows with the viewmat and K they were computed at), out16[12..13] = its (sum |d - g|, sum |S(d) - S(g)|) over loss_partials[n][2]
 * (fixed order), out16[14] = normal_sum[0] (0 when NULL), out16[15] = 0.  After the all-reduce (sum) gsl_pose_step
 * takes v_viewmat = out16 and loss_sums = out16 + 12.  Written by a kernel of this library so that a captured
 * iteration holds no foreign node. */
int gsl_pack_pose_reduce(const float* v_viewmat, const float* vm_rows, int n_vm_rows, const float* viewmat,
                         const float* K, const float* loss_partials, int n_partials, const float* normal_sum,
                         float* out16, void* stream);

/* ---- per-frame set-up: exact k nearest neighbours on the device (csrc/knn.hip) ----
 * Stands in for the small_gicp KdTree search of /root/reference/src/my_gsplat/utils.py:16-22.
 * points[N,3]; bbox[6] = (min x,y,z, max x,y,z) on the device; uniform grid of gsl_knn_cells() cells.
 * gsl_knn_count fills the per-cell counts at the start of ws (int32[cells]); the caller turns them into an
 * inclusive cumulative sum incl_offsets[cells] (any device scan); gsl_knn_query then writes the SQUARED
 * distances to the k <= 8 nearest points (the point itself included), ascending, into dists[N,k]. */
size_t gsl_knn_ws_bytes(int N);
int gsl_knn_cells(void);
int gsl_knn_count(const float* points, int N, const float* bbox, void* ws, size_t ws_bytes, void* stream);
int gsl_knn_query(const float* points, int N, const float* bbox, const int32_t* incl_offsets, int k,
                  float* dists, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSLOC_HIP_H */
