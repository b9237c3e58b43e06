/* stardist_hip.h -- C ABI of libstardist_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the native layer of the StarDist prediction path.  Every entry point
 * replaces one function of the reference's native modules `stardist.lib.stardist2d` /
 * `stardist.lib.stardist3d` (CPython, positional args) or of its experimental plain-C ABI
 * `libstardist3d` (stardist/lib/stardist3d_lib.h:52-79) and keeps that function's argument
 * meaning, array layout (row-major, dense) and ownership rules:
 *
 *   - caller owns every buffer; inputs are never modified;
 *   - `*_host` entry points (and the two `_LIB_*` names kept verbatim from the reference ABI)
 *     take HOST pointers and do H2D/D2H themselves on the null stream;
 *   - `*_device` entry points take DEVICE pointers plus a `hipStream_t` (passed as void*) and
 *     enqueue all work on that stream; they synchronise the stream only where documented
 *     (the NMS entry points do, because the greedy scan is driven from the host);
 *   - return value: 0 on success, -1 on error (message via sd_last_error()); the two `_LIB_*`
 *     functions return void like the reference and report errors on stderr.
 *
 * No torch / numpy types appear here.  The reference-side bindings (CPython / ctypes / JNA)
 * are shown in INTEGRATION.md.
 */
#ifndef STARDIST_HIP_H
#define STARDIST_HIP_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------ */
const char* sd_last_error(void);          /* message of the calling thread's last failed call  */
int sd_version(void);                     /* ABI version, currently 1                          */
int sd_device_count(void);                /* number of visible HIP devices (0 without a GPU)   */
int sd_release_workspace(void);           /* free the cached device workspace                  */
/* Switches between EQUIVALENT formulations (same results; the parity suite runs both settings against the reference):
 *   "nms3d_volume_bounds" 1|0  decide 3D pairs from rigorous volume bounds where possible / always compute the exact volume
 *   "nms3d_cone_map"      1|0  stage-5 voxel tests through the cone map / over every face as the reference does
 *   "nms3d_refine_mesh"   2|1|0  direction meshes of the volume bounds: the ray mesh only (0); refined once for the pairs it leaves
 *                              undecided (1); refined twice for the pairs that reach the exact-volume kernels (2, default)
 *   "nms3d_tail_batch"    1|0  late greedy rounds of the 3D NMS (once at most max(N/32, 512) candidates are undecided) as one speculative
 *                              batch + replay on the device / as plain rounds; a value d >= 2 sets the threshold to N/d (tuning)
 *   "nms3d_split_exact"   3|2|1|0  exact volumes of the pairs the bounds leave undecided by four waves per pair in a second pass (3,
 *                              default: for every launch; 2: stage 4 only up to 16 384 pairs per launch; 1: both stages only for small
 *                              launches, full-size workspace in the bounds pass) / 0: by the wave that evaluated the bounds
 *                              (bit-identical volumes)
 *   "nms3d_bounds_reuse"  1|0  the once-refined direction mesh keeps the ray mesh's vertices in front: their boundary points are taken
 *                              from the coarse pass that has just run over the same planes (bit for bit what a second cast would store) /
 *                              every direction of the refined mesh is cast
 *   "nms3d_bounds_lean"   1|0  bounds-only launches of stages 3 / 4 (every launch in the two-pass form) are made without the adjacency-seed
 *                              table, and the cull's index tables lie in the ray-cast workspace: 21.9 instead of 25.6 KB of LDS per wave,
 *                              seven waves per CU instead of six / the layout of the one-pass kernels
 *   "nms3d_defer_exact"   r|0  from greedy round r on, the pairs the bounds of stages 3 / 4 leave undecided are not integrated in their
 *                              round (a launch of the exact-volume kernel costs one exact volume's latency however few pairs it holds):
 *                              they are queued, the suppressed-or-not candidate stays undecided, and the tail batch evaluates the queue
 *                              in its one pass (needs "nms3d_tail_batch" and "nms3d_split_exact"; same survivors; default 3: measured
 *                              on the 256^3 bench set 23.0 -> 21.3 ms, from round 1 or 2 on the pending candidates stall the rounds
 *                              behind them: 23.2 ms) / 0: every round integrates its own
 *   "nms2d_area_bounds"   1|0  2D pairs far from the threshold are decided from an enclosure of the intersection area (regular arithmetic,
 *                              area_bounds.h) / every pair runs the Clipper-exact sweep
 *   "nms2d_defer_undecided" r|0  (with "nms2d_area_bounds") from greedy round r on (default 2), a round that leaves at most 16 384 pairs undecided
 *                              does not sweep them itself: they are swept by the tail batch's one launch (a sweep launch costs one sweep's
 *                              latency however few pairs it holds) / 0: every round sweeps its own; "nms2d_defer_max" = that pair limit (16 384;
 *                              measured on the 2048^2 bench set: deferring round 1's 55 000 pairs as well stalls the rounds behind it -- 13.5 instead of 7.7 ms)
 *   "nms2d_strict"        0|1  1 = bit-exact BY CONSTRUCTION: every 2D pair runs the Clipper-exact sweep (overrides "nms2d_area_bounds"); the
 *                              default decides the pairs far from the threshold from an enclosure of Clipper's area whose band is validated
 *                              empirically and adversarially, not proven
 *   "nms2d_neighbours_single_pass", "nms3d_neighbours_single_pass" 1|0  neighbour lists of the NMS written in one pass into slots sized
 *                              from the cell table / counted, scanned and filled in two passes (same lists up to order)
 *   "probe_tier"          1|2  capacity tier sd_clip_pairs_device runs first;  "probe_no_general" 1: do not fall back to the general path
 *   "trace"               1    print per-round counters to stdout
 * Nothing in the library reads the process environment.  sd_get_option returns -1 for an unknown name. */
int sd_set_option(const char* name, int value);
int sd_get_option(const char* name);

/* ---- 2D non-maximum suppression ------------------------------------------------------------
 * replaces stardist.lib.stardist2d.c_non_max_suppression_inds
 *   (stardist/lib/stardist2d.cpp:390-615; caller stardist/nms.py:186-227)
 * dist   (n_polys, n_rays) float32, points (n_polys, 2) float32 (y, x), both sorted by score
 * descending.  use_kdtree / use_bbox / verbose / threshold as in the reference ("O!O!iiif").
 * keep   (n_polys,) bytes: 1 = survivor, 0 = suppressed (the reference returns NPY_BOOL).
 * stats  optional int64[16] (may be NULL): {0 pairs evaluated, 1 pairs re-run on the exact-join path,
 *        2 greedy rounds, 3 neighbour entries, 4 pair-kernel time ns (HIP events on `stream`),
 *        5 pair-kernel launches, 6 exact-join kernel ns, 7 build+bin+neighbour kernels ns, 8 capacity spills,
 *        9 pairs decided by the area enclosure (counted in 0 as well), 10 undecided pairs deferred to the tail batch, 11.. 0}.
 */
int sd_nms2d_host(const float* dist, const float* points, int n_polys, int n_rays,
                  int use_kdtree, int use_bbox, int verbose, float threshold,
                  uint8_t* keep, int64_t* stats);
int sd_nms2d_device(const float* d_dist, const float* d_points, int n_polys, int n_rays,
                    int use_kdtree, int use_bbox, int verbose, float threshold,
                    uint8_t* d_keep, int64_t* stats, void* stream);

/* replaces stardist.lib.stardist2d.c_non_max_suppression_inds_old
 *   (stardist/lib/stardist2d.cpp:173-386, "O!O!fiiii"; caller stardist/nms.py:20-74 _non_maximum_suppression_old; the reference keeps
 *   it as a second statement of the NMS, tests/test_nms2D.py:78-110 "old == new")
 * polys   (n_polys, 2, n_rays) int32 vertex coordinates (row 0 = y, row 1 = x), sorted by score descending;
 * mapping (height, width) int32: id of the score-sorted polygon at that pixel of the prediction grid, -1 = none (only read with
 *         max_bbox_search != 0: polygon i is compared with the polygons j > i found in the window
 *         [(bbox - max bbox size) / grid, (bbox + max bbox size) / grid) of the map, :285-311; else with every j > i, :337);
 * keep    (n_polys,) bytes: 1 = survivor. */
int sd_nms2d_old_host(const int32_t* polys, int n_polys, int n_rays, const int32_t* mapping, int height, int width,
                      float threshold, int max_bbox_search, int grid_y, int grid_x, int verbose, uint8_t* keep);
int sd_nms2d_old_device(const int32_t* d_polys, int n_polys, int n_rays, const int32_t* d_mapping, int height, int width,
                        float threshold, int max_bbox_search, int grid_y, int grid_x, int verbose, uint8_t* d_keep, void* stream);

/* Test probe: Clipper::AddPath (clipper.cpp:1045-1221) once per polygon -- the prepared-polygon records the 2D NMS
 * builds per candidate (stardist_amd/csrc/clip_beam.h, PolyPrep<MAXV>, MAXV = 32/64/128/256 for n_verts). */
int sd_prepare_polys_device(const int32_t* d_x, const int32_t* d_y, int n_polys, int n_verts, void* d_out,
                            int64_t out_bytes, void* stream);

/* pair-level probe used by the parity tests: intersection area of integer polygons exactly as
 * poly_intersection_area (stardist2d.cpp:152-165) computes it.  xa..yb are device int32
 * arrays of shape (n_pairs, n_verts); out_twice_area int64 (2*area), out_flags int32
 * (bit0.. see clip_sweep.h status bits, bit8 = pair needed the join path). */
int sd_clip_pairs_device(const int32_t* d_xa, const int32_t* d_ya, const int32_t* d_xb,
                         const int32_t* d_yb, int n_pairs, int n_verts,
                         int64_t* d_out_twice_area, int32_t* d_out_flags, void* stream);

/* Test probe of the decision shortcut of the 2D NMS (stardist_amd/csrc/area_bounds.h): for the same inputs as sd_clip_pairs_device
 * (n_verts 3..32) the exact area of the intersection by boundary integration (float32), the half-width of the band that encloses
 * the area ClipperLib returns for the pair, and info = bit 0: the enclosure may be used for a decision (both polygons simple, equally
 * oriented, small enough for exact float predicates), bits 8..17: number of boundary crossings, bits 18..28: number of edge pairs
 * closer than one lattice step. */
int sd_area_bounds_pairs_device(const int32_t* d_xa, const int32_t* d_ya, const int32_t* d_xb, const int32_t* d_yb, int n_pairs,
                                int n_verts, float* d_out_area, float* d_out_band, int32_t* d_out_info, void* stream);

/* ---- star-convex distances (training targets; same native module) --------------------------
 * replaces stardist.lib.stardist2d.c_star_dist (stardist2d.cpp:55-124)
 * src (H, W) uint16 labels; dst (ceil(H/gy), ceil(W/gx), n_rays) float32. */
int sd_star_dist2d_host(const uint16_t* src, int H, int W, int n_rays, int grid_y, int grid_x,
                        float* dst);
int sd_star_dist2d_device(const uint16_t* d_src, int H, int W, int n_rays, int grid_y,
                          int grid_x, float* d_dst, void* stream);
/* replaces stardist.lib.stardist3d.c_star_dist3d (stardist3d.cpp:245-346)
 * src (Z, Y, X) uint16; dz/dy/dx (n_rays,) float32 ray unit vectors;
 * dst (ceil(Z/gz), ceil(Y/gy), ceil(X/gx), n_rays) float32. */
int sd_star_dist3d_host(const uint16_t* src, int Z, int Y, int X, const float* dz,
                        const float* dy, const float* dx, int n_rays, int grid_z, int grid_y,
                        int grid_x, float* dst);
int sd_star_dist3d_device(const uint16_t* d_src, int Z, int Y, int X, const float* d_dz,
                          const float* d_dy, const float* d_dx, int n_rays, int grid_z,
                          int grid_y, int grid_x, float* d_dst, void* stream);

/* ---- training target: per-object normalised distance transform ----------------------------
 * replaces stardist.utils.edt_prob (stardist/utils.py:71-125; called by the data generators).
 * lbl (Z, Y, X) int32 label image (2D: Z = 1), labels 1..max_label are objects, everything else background;
 * (sz, sy, sx) axis spacing (the reference's `anisotropy`); prob (Z, Y, X) float32: for an object pixel the Euclidean distance
 * (float64) to the nearest pixel inside the image with another label, divided by (the object's maximum + 1e-10); else 0. */
int sd_edt_prob_device(const int32_t* d_lbl, int Z, int Y, int X, double sz, double sy, double sx, int max_label,
                       float* d_prob, void* stream);

/* ---- 2D label rasteriser --------------------------------------------------------------------
 * replaces the Python loop stardist.geometry.geom2d.polygons_to_label_coord
 *   (stardist/geometry/geom2d.py:149-166: skimage.draw.polygon per object, later objects
 *   overwrite earlier ones).
 * coord (n_polys, 2, n_rays) float32 (row 0 = y/r, row 1 = x/c), painted in array order;
 * labels (n_polys,) int32: value written is labels[i]+1 (geom2d.py:164);
 * result (H, W) int32, fully written (background 0). */
int sd_polygons_to_label_host(const float* coord, const int32_t* labels, int n_polys, int n_rays,
                              int H, int W, int32_t* result);
int sd_polygons_to_label_device(const float* d_coord, const int32_t* d_labels, int n_polys,
                                int n_rays, int H, int W, int32_t* d_result, void* stream);
/* ... only the window [y0, y0 + H) x [x0, x0 + W) of an HI x WI image (d_result (H, W)): the pixels are the ones the whole-image
 * call writes there -- each rank of a block-sharded prediction renders the write regions of its own blocks (stardist/big.py:319-326
 * block.write) from the global polygon list. */
int sd_polygons_to_label_window_device(const float* d_coord, const int32_t* d_labels, int n_polys, int n_rays, int HI, int WI,
                                       int y0, int x0, int H, int W, int32_t* d_result, void* stream);

/* polar -> cartesian for a list of polygons: replaces stardist.geometry.geom2d.dist_to_coord (geom2d.py:130-146) in numpy's own
 * arithmetic (products and the final sum in float64, rounded to float32 where numpy rounds): d_dist (n, R) float32, d_points (n, 2)
 * float64 (y, x), d_sincos (2, R) float64 = (sin, cos) of linspace(0, 2 pi, R, endpoint=False) computed by the caller's libm,
 * scale (y, x) = `scale_dist`; d_coord (n, 2, R) float32. */
int sd_dist_to_coord_device(const float* d_dist, const double* d_points, const double* d_sincos, long long n_polys, int n_rays,
                            double scale_y, double scale_x, float* d_coord, void* stream);

/* ---- behind the 2D NMS on predict_instances (stardist/models/model2d.py:536-561) --------------
 * positions of the non-zero keep flags, ascending (the `inds` of non_maximum_suppression_sparse, nms.py:175-183: boolean-mask indexing):
 * d_positions (n) int64 receives *d_count entries. */
int sd_survivor_positions_device(const unsigned char* d_keep, long long n, long long* d_positions, int32_t* d_count, void* stream);
/* for candidates in SCORE order (descending, as the NMS takes them): rows of the m survivors at d_positions -- prob (m), points (m, 2)
 * int64 -- their polygon coordinates d_out_coord (m, 2, R) = dist_to_coord (geom2d.py:130-146, arithmetic of sd_dist_to_coord_device,
 * scale 1), and the same coordinates in the order polygons_to_label paints them (ascending score, stable: geom2d.py:186-197) with
 * their label ids minus one, d_out_labels_paint (m) -- the inputs of sd_polygons_to_label_device.  d_out_coord_paint may be NULL. */
int sd_survivors2d_device(const long long* d_positions, int m, const float* d_prob, const long long* d_points, const float* d_dist,
                          int n_rays, const double* d_sincos, float* d_out_prob, long long* d_out_points, float* d_out_coord,
                          float* d_out_coord_paint, int32_t* d_out_labels_paint, void* stream);

/* ---- 3D non-maximum suppression -------------------------------------------------------------
 * name, signature and semantics of the reference's C ABI
 *   (stardist/lib/stardist3d_lib.h:52-66 -> _COMMON_non_maximum_suppression_sparse,
 *    stardist/lib/stardist3d_impl.cpp:956-1385; Python caller stardist/nms.py:327-384)
 * host pointers; result (n_polys,) bool: true = survivor. */
void _LIB_non_maximum_suppression_sparse(const float* scores, const float* dist,
                                         const float* points, const int n_polys,
                                         const int n_rays, const int n_faces, const float* verts,
                                         const int* faces, const float threshold,
                                         const int use_bbox, const int use_kdtree,
                                         const int verbose, bool* result);
/* stats: optional int64[16]: {0 upper-bound tests, 1 lower-bound tests, 2 kernel-volume calls, 3 render calls,
 * 4 greedy rounds, 5 neighbour entries, 6 suppressed by kernel stage, 7 suppressed by render stage,
 * 8 stage-3 kernel ns, 9 stage-4 ns, 10 stage-5 ns, 11 hull-volume calls, 12 kept by hull stage,
 * 13 exact-volume decisions (stages 3 / 4) whose ratio lies within 1e-6 of the threshold -- the volumes agree with Qhull's to 1e-9
 * relative, so only such a pair could be decided differently; the count makes that observable --, 14 faces that needed the
 * large-capacity fallback, 15 broad-phase ns} */
int sd_nms3d_device(const float* d_scores, const float* d_dist, const float* d_points,
                    int n_polys, int n_rays, int n_faces, const float* d_verts,
                    const int* d_faces, float threshold, int use_bbox, int use_kdtree,
                    int verbose, uint8_t* d_keep, int64_t* stats, void* stream);

/* Test probe of the two volume stages of the 3D cascade: for every pair (i, j) of d_pairs (int32 [n_pairs][2], indices into the
 * n_polys candidates) the intersection volume of the two KERNELS (replaces qhull_overlap_kernel, stardist3d_impl.cpp:830-869,
 * 0 if the midpoint of the centres is not interior) and of the two CONVEX HULLS (replaces qhull_overlap_convex_hulls, :872-939,
 * 1e10 on failure), as float64.  Either output pointer may be NULL. */
int sd_hiv_pairs_device(const float* d_dist, const float* d_points, int n_polys, int n_rays, int n_faces,
                        const float* d_verts, const int* d_faces, const int32_t* d_pairs, int n_pairs,
                        double* d_vol_kernel, double* d_vol_hull, void* stream);

/* ---- 3D label rasteriser --------------------------------------------------------------------
 * name, signature and semantics of the reference's C ABI
 *   (stardist/lib/stardist3d_lib.h:69-77 -> _COMMON_polyhedron_to_label,
 *    stardist/lib/stardist3d_impl.cpp:1404-1525; Python caller geom3d.py:100-198)
 * result (nz, ny, nx) int32 must be zero-initialised by the caller (the reference only writes
 * inside polyhedra). Polyhedra are given in painting order (first writer keeps the voxel). */
void _LIB_polyhedron_to_label(const float* dist, const float* points, const float* verts,
                              const int* faces, const int n_polys, const int n_rays,
                              const int n_faces, const int* labels, const int nz, const int ny,
                              const int nx, const int render_mode, const int verbose,
                              const int use_overlap_label, const int overlap_label, int* result);
int sd_polyhedron_to_label_device(const float* d_dist, const float* d_points,
                                  const float* d_verts, const int* d_faces, int n_polys,
                                  int n_rays, int n_faces, const int* d_labels, int nz, int ny,
                                  int nx, int render_mode, int verbose, int use_overlap_label,
                                  int overlap_label, int* d_result, void* stream);
/* ... only the window [z0, z0 + nz) x [y0, y0 + ny) x [x0, x0 + nx) of an NZ x NY x NX volume; d_result (nz, ny, nx) must be
 * zero-initialised by the caller like the whole-volume result (the implementation only writes inside polyhedra). */
int sd_polyhedron_to_label_window_device(const float* d_dist, const float* d_points, const float* d_verts, const int* d_faces,
                                         int n_polys, int n_rays, int n_faces, const int* d_labels, int NZ, int NY, int NX,
                                         int z0, int y0, int x0, int nz, int ny, int nx, int render_mode, int verbose,
                                         int use_overlap_label, int overlap_label, int* d_result, void* stream);

/* ---- candidate selection --------------------------------------------------------------------
 * replaces the numpy glue stardist.nms._ind_prob_thresh + np.where + gather
 *   (stardist/nms.py:6-17, stardist/models/base.py:553-610) on device.
 * prob (n_pix,) float32 and dist (n_pix, n_rays) float32 are the network heads over a grid of
 * `ndim` (2 or 3) dims `shape`; selects pixels with prob > thresh that are at least b[2*d] /
 * b[2*d+1] grid cells from the low / high face of dim d, in C order (== np.where order).
 * Writes out_prob (cap,), out_dist (cap, n_rays) = max(dist, 1e-3f) (skipped when d_dist is NULL), out_points (cap, ndim)
 * int32 grid indices (NOT yet multiplied by the grid) and *d_count (int32) = number found
 * (may exceed cap: then only the first cap are written). */
int sd_select_candidates_device(const float* d_prob, const float* d_dist, int ndim,
                                const int* shape, const int* b, int n_rays, float thresh,
                                int cap, float* d_out_prob, float* d_out_dist,
                                int32_t* d_out_points, int32_t* d_count, void* stream);

/* np.argsort(scores)[::-1] of stardist/nms.py:114,167 on the device, stated as a stable ascending sort reversed (best score first, equal scores
 * in descending order of their position): d_scores (n,) float32 (finite) -> d_sorted (n,) float32 = the scores in that order, d_order (n,) int64 =
 * the position of the k-th best in d_scores. */
int sd_sort_scores_desc_device(const float* d_scores, int n, float* d_sorted, int64_t* d_order, void* stream);

/* Candidates in score order (the argsort of stardist/nms.py:167 applied to the selection above): d_points (n_sel, ndim) int32 grid indices as
 * written by sd_select_candidates_device, d_order (n,) int64 = index of the k-th best candidate (NULL: identity).  Writes, per k:
 * d_rows[k] = C-order linear index of (point + origin) in the grid `full_shape` (the row of the channels-last feature matrix the
 * distance head is evaluated on, sd_head_rows_device), d_points_f32[k] / d_points_i64[k] = point * grid (pixel coordinates: float32 as the
 * NMS natives take them, int64 as the result dict returns them).  Any output pointer may be NULL. */
int sd_sorted_rows_device(const int32_t* d_points, const int64_t* d_order, int n, int ndim, const int* full_shape, const int* origin,
                          const int* grid, int64_t* d_rows, float* d_points_f32, int64_t* d_points_i64, void* stream);

/* ---- reference-style C ABI for the natives that have none in the reference --------------------
 * (stardist/lib/stardist3d_lib.h:52-79 covers only the two 3D functions above).  Same conventions: host pointers, caller
 * owns all buffers, no return code (errors abort with a message on stderr).  Argument meaning as the CPython functions:
 *   _LIB_non_maximum_suppression_2d  <- c_non_max_suppression_inds      stardist2d.cpp:390-615  (dist, points sorted by score)
 *   _LIB_polygon_to_label            <- polygons_to_label_coord         geom2d.py:149-166       (coord (n,2,n_rays); result (ny,nx))
 *   _LIB_star_dist                   <- c_star_dist                     stardist2d.cpp:55-124
 *   _LIB_star_dist3d                 <- c_star_dist3d                   stardist3d.cpp:245-346 */
void _LIB_non_maximum_suppression_2d(const float* dist, const float* points, const int n_polys,
                                     const int n_rays, const float threshold, const int use_bbox,
                                     const int use_kdtree, const int verbose, bool* result);
void _LIB_polygon_to_label(const float* coord, const int* labels, const int n_polys,
                           const int n_rays, const int ny, const int nx, int* result);
void _LIB_star_dist(const unsigned short* src, const int ny, const int nx, const int n_rays,
                    const int grid_y, const int grid_x, float* dst);
void _LIB_star_dist3d(const unsigned short* src, const int nz, const int ny, const int nx,
                      const float* pdz, const float* pdy, const float* pdx, const int n_rays,
                      const int grid_z, const int grid_y, const int grid_x, float* dst);

/* ---- network epilogue ------------------------------------------------------------------------
 * bias + activation of a convolution output in one in-place pass (the reference's Keras Conv layers do both inside the
 * layer: csbdeep unet_block / resnet_block as called from stardist/models/model2d.py:310-349, model3d.py:360-447).
 * x is viewed as [n_outer][n_channels][inner] float32 (channels-last tensors: inner = 1, n_outer = pixels);
 * act: 0 = linear, 1 = relu. */
int sd_bias_act_device(float* d_x, const float* d_bias, long long n_outer, int n_channels,
                       long long inner, int act, void* stream);
/* x = act((x + addend) + bias): epilogue of a convolution over a channel concatenation that is evaluated as two
 * convolutions over the two sources (Concatenate + Conv of csbdeep's unet_block up path, blocks.py) -- the
 * concatenated tensor is never written. addend has the layout of x. */
int sd_add_bias_act_device(float* d_x, const float* d_addend, const float* d_bias, long long n_outer,
                           int n_channels, long long inner, int act, void* stream);

/* Keras MaxPooling2D / 3D(pool), 'valid', stride = pool, channels-last float32 (csbdeep unet_block between its levels; the grid > 1
 * stages of stardist/models/model2d.py:317-325): d_in [D][H][W][C] -> d_out [D/pz][H/py][W/px][C]; 2D: D = pz = 1.  C % 4 == 0. */
int sd_maxpool_ndhwc_device(const float* d_in, int n_channels, int D, int H, int W, int pz, int py, int px, float* d_out, void* stream);

/* point-level probe of the reference's inside_polyhedron (stardist/lib/stardist3d_impl.cpp:153-191) for ONE polyhedron
 * (d_dist: n_rays, d_centre: 3, zyx) on n points (d_points: n x 3, zyx): d_out[t] = 1 if inside.  use_cone_map != 0 evaluates
 * the predicate only on the faces whose cone can contain the point's direction (csrc/geom3d.h), as stage 5 of the NMS does;
 * both settings give identical results. */
int sd_inside_polyhedron_device(const float* d_dist, const float* d_centre, int n_rays, int n_faces,
                                const float* d_verts, const int* d_faces, const float* d_points,
                                long long n, int use_cone_map, uint8_t* d_out, void* stream);

/* features epilogue + one-channel head: out = act(in + bias) over channels-last [n_pix][n_channels] float32 (in place when
 * d_out == d_in; n_channels 32, 64, 128 or 256) and, when d_w is given, d_dot[p] = sum_c out[p][c] * d_w[c] + d_wbias[0], through the
 * logistic function when sigmoid != 0: the object-probability head of the reference's models (Conv 1x1, sigmoid:
 * stardist/models/model2d.py:338-341, model3d.py:436-439) evaluated while the features are in registers.
 * d_bias == NULL: no bias; d_out == NULL (d_w given): the features are only read and just d_dot is written. */
int sd_bias_act_dot_device(const float* d_in, float* d_out, const float* d_bias, long long n_pix, int n_channels,
                           int act, const float* d_w, const float* d_wbias, int sigmoid, float* d_dot, void* stream);
/* distance head on selected pixels: d_out[i][r] = max(clamp_min, d_bias[r] + sum_k d_feat[d_rows[i]][k] * d_w[r][k]) for
 * i < n_rows, r < n_out (d_rows == NULL: row i itself, i.e. the dense head).  d_feat is channels-last [n_pix][n_channels],
 * d_w the Conv 1x1 kernel [n_out][n_channels] (model2d.py:342-343, model3d.py:440-441).  With d_rows = the candidate pixels
 * (prob > prob_thresh) this is what predict_sparse (stardist/models/base.py:553-610) keeps of the dense prediction.
 * fp32 MFMA, one fixed fma chain per output whatever rows are batched together. */
int sd_head_rows_device(const float* d_feat, int n_channels, const long long* d_rows, long long n_rows,
                        const float* d_w, const float* d_bias, int n_out, float clamp_min, float* d_out,
                        void* stream);

/* ---- network convolutions (hand-written, f32 matrix cores) --------------------------------------------------------------
 * One Keras Conv2D(3x3) / Conv3D(3x3x3), padding='same', stride 1, + bias + activation of the reference's U-Net (csbdeep unet_block as
 * built by stardist/models/model2d.py:310-349 and model3d.py:360-399), channels-last float32:
 *     out[z][y][x][co] = act(bias[co] + sum_{ci,kz,ky,kx} in[z+kz-1][y+ky-1][x+kx-1][ci] * w[co][ci][kz][ky][kx])   (zero padding)
 * kz = 1: 2D (D must be 1), kz = 3: 3D.  The input channels are the concatenation [source 0 (c0 channels), source 1 (c1 channels)]
 * -- Concatenate([up, skip]) of the up path without the concatenated tensor; d_src1 == NULL: one source.  `up` is a bit mask of the
 * axes (1: x, 2: y, 4: z) along which a source has half the output resolution and is read through nearest-neighbour 2x up-sampling
 * (UpSampling2D/3D folded into the operand fetch).  stride = floats per pixel of a source.
 * Supported: c0, c1 multiples of 32 with c0 + c1 <= 512 and c_out a multiple of 32; and the first layer c0 = 1 (c_out % 4 == 0).
 * d_wpacked: the kernel in the device layout written by sd_conv3_pack_weights_host (sd_conv3_packed_floats floats).
 * act: 0 linear, 1 relu.  Exact float32: each output is one fma chain in a fixed order (bias first), repeatable bit for bit. */
long long sd_conv3_packed_floats(int c_in, int c_out, int kz);
int sd_conv3_pack_weights_host(const float* w /* [c_out][c_in][kz][3][3] */, int c_in, int c_out, int kz, float* packed);
int sd_conv3_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                          int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out, int act,
                          float* d_out, void* stream);
/* ... with a residual: out = act(conv + bias + res), res channels-last [D][H][W][res_stride] -- Add([shortcut, x]) + Activation that
 * closes a csbdeep resnet_block (stardist/models/model3d.py:417-422), folded into the convolution's epilogue.  d_res == NULL: none. */
int sd_conv3_res_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                              int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, const float* d_res,
                              int res_stride, int c_out, int act, float* d_out, void* stream);

/* The same layer with every f32 product evaluated as six bf16 x bf16 products (operands split into three bf16 terms each, f32
 * accumulation) on the bf16 matrix cores: f32-level accuracy (the dropped cross terms are below 2^-24 of a product) at 6/16 of the
 * f32-MFMA time.  Same arguments; c0, c1 multiples of 32 only (the one-channel first layer stays on sd_conv3_ndhwc_device); the
 * weights are packed by sd_conv3_bf16x6_pack_weights_host.  Opt-in: the exact-f32 kernel above is the default network path. */
long long sd_conv3_bf16x6_packed_floats(int c_in, int c_out, int kz);
int sd_conv3_bf16x6_pack_weights_host(const float* w /* [c_out][c_in][kz][3][3] */, int c_in, int c_out, int kz, float* packed);
int sd_conv3_bf16x6_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                 int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out,
                                 int act, float* d_out, void* stream);
int sd_conv3_bf16x6_res_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                     int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias,
                                     const float* d_res, int res_stride, int c_out, int act, float* d_out, void* stream);

/* The same layer with every f32 product evaluated as three fp16 x fp16 products: each operand x = hi + lo' * 2^-11 (hi = fp16(x),
 * lo' = fp16((x - hi) * 2^11)), a * b ~ hi*hi + 2^-11 (hi*lo' + lo'*hi), f32 accumulation in two accumulators on the fp16 matrix cores:
 * f32-level accuracy (the dropped term is below 2^-22 of a product; networks within 3e-6 of a float64 evaluation like the two forms
 * above) with half the matrix instructions of the bf16 form -- the default network path since round 4.  Same arguments as
 * sd_conv3_bf16x6_*, plus d_range_flag (device int, may be NULL): OR-ed with 1 when an input value lies outside the fp16 range
 * (|x| > 65504, infinities included; a NaN is not flagged -- it propagates into the output as in any f32 evaluation), in which case the output is not valid and the caller must re-evaluate the layer with the bf16x6 form.
 * sd_conv3_f16x3_pack_weights_host returns -2 (weights still packed) when a weight is outside that range.
 * Option "conv_f16_workgroups_per_cu" (sd_set_option): 2 (default) or 1, an A/B probe of the launch geometry; same results. */
long long sd_conv3_f16x3_packed_floats(int c_in, int c_out, int kz);
int sd_conv3_f16x3_pack_weights_host(const float* w /* [c_out][c_in][kz][3][3] */, int c_in, int c_out, int kz, float* packed);
int sd_conv3_f16x3_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out,
                                int act, float* d_out, int* d_range_flag, void* stream);
int sd_conv3_f16x3_res_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                    int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias,
                                    const float* d_res, int res_stride, int c_out, int act, float* d_out, int* d_range_flag,
                                    void* stream);

/* ... with the ONE-CHANNEL HEAD behind the layer fused into its epilogue (the object-probability head behind the features layer:
 * Conv 1x1 + sigmoid, stardist/models/model2d.py:338-341, model3d.py:436-439).  Besides d_out the kernel writes, per pixel and per group of
 * four consecutive output channels, the sum (in channel order) of their products (after bias + activation) with d_dot_w[c_out]:
 * d_dot_partial[D * H * W][c_out / 4] -- the per-lane term of sd_bias_act_dot_device, taken while the tile is in registers.
 * sd_dot_combine_device (groups = c_out / 32: 1, 2, 4 or 8) adds a pixel's terms in the order of that function's reduction, then the
 * head's bias d_wbias[0] (NULL: none), then the logistic function (sigmoid != 0): the result equals sd_bias_act_dot_device on the same
 * features BIT FOR BIT, without reading the features a second time. */
int sd_conv3_f16x3_dot_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                    int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out,
                                    int act, float* d_out, int* d_range_flag, const float* d_dot_w, float* d_dot_partial, void* stream);
int sd_dot_combine_device(const float* d_partial, int groups, long long n_pix, const float* d_wbias, int sigmoid, float* d_out, void* stream);

/* ---- split16 activation tensors (round 6) ---------------------------------------------------------------------------------------
 * What the split-fp16 kernel's consumer side derives from every f32 value it reads -- hi = fp16(x), lo' = fp16((x - hi) * 2^11) -- made
 * ONCE by the producing layer instead of once per consumer workgroup and unit.  A split16 tensor has the shape, strides and addresses
 * of the channels-last f32 tensor it stands for (C a multiple of 32, dense); per pixel and 32-channel chunk its 128 bytes hold 8
 * elements of 16 bytes: element p * 4 + o = the fp16 terms of plane p (0: hi, 1: lo') of channels o * 8 .. o * 8 + 7.
 * Value = hi + lo' * 2^-11 (exact in f32): the 22 bits the f32-tensor entry points keep of x, so every layer's result is bit-identical
 * whichever form carries the activations between the layers of the reference's U-Net (csbdeep unet_block, model2d.py:310-349,
 * model3d.py:360-399).
 *   sd_conv3_f16x3_fmt_ndhwc_device   the layer of sd_conv3_f16x3_ndhwc_device / _dot_ with either side in split16 form
 *                                     (in_split16: all sources; out_split16: no fused head then).  d_range_flag |= 1: an f32 INPUT was
 *                                     outside the fp16 range; |= 2: a value of the split16 OUTPUT was (the output is not valid).
 *   sd_conv3_c1x32_split16_device     the one-channel first layer (1 -> 32, weights packed by sd_conv3_pack_weights_host) writing split16
 *   sd_maxpool_split16_ndhwc_device   MaxPooling on a split16 tensor: == split16(maxpool(f32 tensor)) bit for bit (x -> (hi, lo') is monotone)
 *   sd_split16_pack_device / _unpack_device   f32 <-> split16 as their own passes (tests; consumers that only take f32 tensors) */
int sd_conv3_f16x3_fmt_ndhwc_device(const float* d_src0, int c0, int up0, const float* d_src1, int c1, int up1, int D, int H, int W, int kz,
                                    const float* d_wpacked, const float* d_bias, int c_out, int act, float* d_out, int in_split16,
                                    int out_split16, int* d_range_flag, const float* d_dot_w, float* d_dot_partial, void* stream);
/* ... on SELECTED pixels: d_out[r][0 .. c_out) = act(bias + conv(src))[d_rows[r]] (d_rows: linear pixel indices into [D][H][W]), bit-identical to
 * what the dense entry points store there.  The sparse prediction path (stardist/models/base.py:553-610 predict_sparse keeps the candidates of
 * a dense prediction) runs the features layer densely WITHOUT its store -- sd_conv3_f16x3_fmt_ndhwc_device with d_out == NULL and the fused
 * probability head: only the head's partial sums leave the kernel -- and then evaluates the layer here for the candidate pixels only. */
int sd_conv3_f16x3_rows_device(const float* d_src, int c_in, int in_split16, int D, int H, int W, int kz, const float* d_wpacked,
                               const float* d_bias, int c_out, int act, const long long* d_rows, long long n_rows, float* d_out, void* stream);
int sd_conv3_c1x32_split16_device(const float* d_src, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int act,
                                  float* d_out, int* d_range_flag, void* stream);
int sd_maxpool_split16_ndhwc_device(const float* d_in, int n_channels, int D, int H, int W, int pz, int py, int px, float* d_out, void* stream);
int sd_split16_pack_device(const float* d_in, long long n_pix, int n_channels, float* d_out, int* d_range_flag, void* stream);
int sd_split16_unpack_device(const float* d_in, long long n_pix, int n_channels, float* d_out, void* stream);

/* UpSampling2D/3D (nearest, x2 along the axes of `up`: bit 0 x, 1 y, 2 z) + Concatenate([up-sampled a, b]) of a csbdeep unet_block up
 * level as one channels-last tensor [D][H][W][ca + cb] (a: [D >> z][H >> y][W >> x][ca]).  Only the coverage path needs it -- up
 * levels whose channel counts are not multiples of 32 (e.g. n_filter_base = 48) run this + sd_convg_ndhwc_device; the fused 3x3 kernels
 * above never materialise the concatenation.  ca, cb multiples of 4. */
int sd_upcat_ndhwc_device(const float* d_a, int ca, int up, const float* d_b, int cb, int D, int H, int W, float* d_out, void* stream);

/* ---- general convolution (any kernel size, stride, padding, channel counts) ------------------------------------------------
 * Every other convolution of the reference's networks, channels-last float32, exact f32 on the matrix cores with one fixed fma chain
 * per output (bias first, residual last): the 7x7x7 stem, the strided first convolution and the strided 1x1x1 shortcut projection of
 * csbdeep's resnet_block (stardist/models/model3d.py:400-447), first layers with n_channel_in = 3 (model2d.py:310-316), 1x1 heads
 * with few channels (prob_class, model2d.py:345-347).
 *     out[zo][yo][xo][co] = act(bias[co] + sum in[zo*sz - pz + dz][yo*sy - py + dy][xo*sx - px + dx][ci] * w[co][ci][dz][dy][dx] (+ res))
 * with zeros outside the input; (pz, py, px) = padding BEFORE the first element (TensorFlow 'same' with a stride pads asymmetrically,
 * so the caller states it); (Do, Ho, Wo) = output extent.  2D: kz = sz = 1, pz = 0, D = Do = 1.  c_in a multiple of 32, or
 * taps * c_in <= 6144 (small-channel form); any c_out.  src_stride / res_stride / out_stride = floats per pixel of those tensors.
 * d_wpacked: written by sd_convg_pack_weights_host (sd_convg_packed_floats floats; -1 = unsupported shape). */
long long sd_convg_packed_floats(int c_in, int c_out, int kz, int ky, int kx);
int sd_convg_pack_weights_host(const float* w /* [c_out][c_in][kz][ky][kx] */, int c_in, int c_out, int kz, int ky, int kx, float* packed);
int sd_convg_ndhwc_device(const float* d_src, int c_in, int src_stride, int D, int H, int W, int kz, int ky, int kx, int sz, int sy,
                          int sx, int pz, int py, int px, int Do, int Ho, int Wo, const float* d_wpacked, const float* d_bias,
                          const float* d_res, int res_stride, int c_out, int act, float* d_out, int out_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STARDIST_HIP_H */
