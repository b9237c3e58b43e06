// lib_abi.hip -- reference-style plain C ABI for the natives that have none in the reference
// (stardist/lib/stardist3d_lib.h:52-79 only covers the two 3D functions).  Same conventions as `_LIB_non_maximum_suppression_sparse`
// / `_LIB_polyhedron_to_label`: host pointers, caller owns every buffer, no return code (errors abort with a message on stderr,
// since the reference ABI has no way to report them).  Thin wrappers over the `sd_*_host` entry points.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"

#include "stardist_hip.h"

namespace {
void die(const char* who) {
  fprintf(stderr, "%s failed: %s\n", who, sd_last_error());
  abort();
}
}  // namespace

extern "C" void _LIB_non_maximum_suppression_2d(const float* dist, const float* points, const int n_polys, const int n_rays,
                                                const float threshold, const int use_bbox, const int use_kdtree, const int verbose,
                                                bool* result) {
  if (n_polys <= 0) return;
  std::vector<uint8_t> keep((size_t)n_polys);
  if (sd_nms2d_host(dist, points, n_polys, n_rays, use_kdtree, use_bbox, verbose, threshold, keep.data(), nullptr)) die("_LIB_non_maximum_suppression_2d");
  for (int i = 0; i < n_polys; ++i) result[i] = keep[i] != 0;
}

extern "C" void _LIB_polygon_to_label(const float* coord, const int* labels, const int n_polys, const int n_rays, const int ny, const int nx,
                                      int* result) {
  if (sd_polygons_to_label_host(coord, labels, n_polys, n_rays, ny, nx, result)) die("_LIB_polygon_to_label");
}

extern "C" void _LIB_star_dist(const unsigned short* src, const int ny, const int nx, const int n_rays, const int grid_y, const int grid_x,
                               float* dst) {
  if (sd_star_dist2d_host(src, ny, nx, n_rays, grid_y, grid_x, dst)) die("_LIB_star_dist");
}

extern "C" void _LIB_star_dist3d(const unsigned short* src, const int nz, const int ny, const int nx, const float* pdz, const float* pdy,
                                 const float* pdx, const int n_rays, const int grid_z, const int grid_y, const int grid_x, float* dst) {
  if (sd_star_dist3d_host(src, nz, ny, nx, pdz, pdy, pdx, n_rays, grid_z, grid_y, grid_x, dst)) die("_LIB_star_dist3d");
}
