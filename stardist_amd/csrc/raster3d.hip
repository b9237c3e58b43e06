// raster3d.hip -- polyhedron -> label volume rasteriser.
//
// Replaces _COMMON_polyhedron_to_label (stardist/lib/stardist3d_impl.cpp:1404-1525; C ABI
// stardist3d_lib.h:69-77; Python caller stardist/geometry/geom3d.py:100-198).
// Semantics: polyhedra are painted in the given order; a voxel keeps the FIRST label written
// to it (or becomes overlap_label when more than one polyhedron covers it).
//
// GPU formulation: one workgroup per polyhedron.  Ray vertices (center + dist*ray, fp32, no
// FMA), the face list and the kernel half-spaces (fp32 normals widened to double, :744-764)
// are staged in LDS; threads sweep the polyhedron's clipped integer bounding box (:536-567,
// lrint) and resolve the paint order with atomicMin on the polyhedron index plus a coverage
// counter; a second pass maps index -> labels[index] with the first-writer / overlap rule.
//
// render_mode 0 ("full") in the reference is   kernel OR (hull AND union-of-tetrahedra)
// (:1474-1477) where `hull` are Qhull's convex-hull half-spaces of the same vertices; render_mode 2
// ("hull") paints the hull itself.  The hulls come from the gift-wrapping kernel of the 3D NMS (nms3d.hip
// k_hull, fp64, facets verified against every vertex), not from Qhull: same facets, planes not normalised
// (the sign test is scale invariant), so a voxel can differ only if it lies within rounding distance
// (~1e-12 voxel) of a hull facet -- see DESIGN.md ("3D rasteriser: hull test").  A polyhedron whose hull
// cannot be built (degenerate vertex set; Qhull would throw) is treated as hull = all space in mode 0
// and fails loudly in mode 2.
#include "common.h"
#include "geom3d.h"
#include "../../include/stardist_hip.h"
#include <limits.h>
#include <vector>

namespace sd {
int hull_planes(const float* d_dist, const float* d_points, const float* d_verts, int n, int R, double** planes, int** count, int* cap_out,
                hipStream_t s);
int cone_map(const float* d_verts, const int* d_faces, int F, sd3::ConeMap* out, hipStream_t s);
}

namespace {

__global__ void __launch_bounds__(256) k_paint3d(const float* __restrict__ dist, const float* __restrict__ points,
                                                 const float* __restrict__ verts, const int* __restrict__ faces, int p0, int p1,
                                                 int n_rays, int n_faces, int NZ, int NY, int NX, int z0, int y0, int x0, int nz, int ny, int nx,
                                                 int render_mode,
                                                 int* __restrict__ first, int* __restrict__ count, int* __restrict__ result_dbg,
                                                 const double* __restrict__ hullPlanes, const int* __restrict__ hullCount, int hullCap,
                                                 int* __restrict__ hullFail, sd3::ConeMap cm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* hs = (double*)smem;                          // n_faces * 4
  double* hh = hs + 4 * n_faces;                       // hullCap * 4 (modes 0 and 2)
  float* pv = (float*)(hh + 4 * hullCap);              // n_rays * 3
  int* fc = (int*)(pv + 3 * n_rays);                   // n_faces * 3
  __shared__ int sb[6];
  __shared__ int s_unsafe;        // cone map preconditions (geom3d.h): every dist >= 1, coordinates below 8192
  for (int k = threadIdx.x; k < 3 * n_faces; k += blockDim.x) fc[k] = faces[k];
  for (int p = p0 + blockIdx.x; p < p1; p += gridDim.x) {
    __syncthreads();
    const float cz = points[3 * p], cy = points[3 * p + 1], cx = points[3 * p + 2];
    if (threadIdx.x < 6) sb[threadIdx.x] = (threadIdx.x & 1) ? -1 : INT_MAX;   // polyhedron_bbox :541-543
    if (threadIdx.x == 0) s_unsafe = (cm.list && fabsf(cz) < 8192.f && fabsf(cy) < 8192.f && fabsf(cx) < 8192.f) ? 0 : 1;
    __syncthreads();
    int z1 = INT_MAX, z2 = -1, y1 = INT_MAX, y2 = -1, x1 = INT_MAX, x2 = -1;
    for (int j = threadIdx.x; j < n_rays; j += blockDim.x) {
      const float d = dist[(size_t)p * n_rays + j];
      const float z = cz + d * verts[3 * j], y = cy + d * verts[3 * j + 1], x = cx + d * verts[3 * j + 2];   // :547-549, :577-579
      pv[3 * j] = z; pv[3 * j + 1] = y; pv[3 * j + 2] = x;
      if (!(d >= 1.f) || !(fmaxf(fmaxf(fabsf(z), fabsf(y)), fabsf(x)) < 8192.f)) s_unsafe = 1;
      const int rz = sd3::round_to_int(z), ry = sd3::round_to_int(y), rx = sd3::round_to_int(x);
      z1 = min(z1, rz); z2 = max(z2, rz); y1 = min(y1, ry); y2 = max(y2, ry); x1 = min(x1, rx); x2 = max(x2, rx);
    }
    atomicMin(&sb[0], z1); atomicMax(&sb[1], z2); atomicMin(&sb[2], y1); atomicMax(&sb[3], y2);
    atomicMin(&sb[4], x1); atomicMax(&sb[5], x2);
    __syncthreads();
    for (int f = threadIdx.x; f < n_faces; f += blockDim.x)
      sd3::build_halfspace(&pv[3 * fc[3 * f]], &pv[3 * fc[3 * f + 1]], &pv[3 * fc[3 * f + 2]], &hs[4 * f]);   // :804-812
    int nh = 0;                                                       // hull facets of this polyhedron (halfspaces_convex :767-795)
    if (hullPlanes) {
      nh = hullCount[p];
      if (nh < 0) { nh = 0; if (threadIdx.x == 0 && render_mode == 2) atomicAdd(hullFail, 1); }
      for (int k = threadIdx.x; k < 4 * nh; k += blockDim.x) hh[k] = hullPlanes[(size_t)p * hullCap * 4 + k];
    }
    __syncthreads();
    // clipped to the volume (:1461-1463) and to the window [z0, z0 + nz) x [y0, y0 + ny) x [x0, x0 + nx) that is rendered
    const bool safe = !s_unsafe;
    const int zlo = max(max(0, sb[0]), z0), zhi = min(min(NZ - 1, sb[1]), z0 + nz - 1);
    const int ylo = max(max(0, sb[2]), y0), yhi = min(min(NY - 1, sb[3]), y0 + ny - 1);
    const int xlo = max(max(0, sb[4]), x0), xhi = min(min(NX - 1, sb[5]), x0 + nx - 1);
    if (zhi < zlo || yhi < ylo || xhi < xlo) continue;
    const long long bz = zhi - zlo + 1, by = yhi - ylo + 1, bx = xhi - xlo + 1;
    const long long nvox = bz * by * bx;
    for (long long t = threadIdx.x; t < nvox; t += blockDim.x) {
      const int xi = xlo + (int)(t % bx);
      const long long r = t / bx;
      const int yi = ylo + (int)(r % by), zi = zlo + (int)(r / by);
      const float z = (float)zi, y = (float)yi, x = (float)xi;     // uint64 -> float at the call sites :1475-1477
      bool inside = false;
      if (render_mode == 0 || render_mode == 1) {
        bool in_kernel = true;                                      // point_in_halfspaces :818-827
        for (int f = 0; f < n_faces; ++f) {
          if (hs[4 * f] * (double)z + hs[4 * f + 1] * (double)y + hs[4 * f + 2] * (double)x + hs[4 * f + 3] > 0) { in_kernel = false; break; }
        }
        inside = in_kernel;
        if (!inside && render_mode == 0) {                            // (in convex hull AND in rendered) :1474-1477
          bool in_hull = true;
          for (int f = 0; f < nh; ++f)
            if (hh[4 * f] * (double)z + hh[4 * f + 1] * (double)y + hh[4 * f + 2] * (double)x + hh[4 * f + 3] > 0) { in_hull = false; break; }
          // union of the tetrahedra (centre, face): only the faces whose cone can hold the voxel's direction (same result, geom3d.h)
          inside = in_hull && sd3::inside_polyhedron_mapped(z, y, x, cz, cy, cx, pv, fc, n_faces, cm, safe);
        }
      } else if (render_mode == 2) {                                  // "convex" :1485-1487
        bool in_hull = true;
        for (int f = 0; f < nh; ++f)
          if (hh[4 * f] * (double)z + hh[4 * f + 1] * (double)y + hh[4 * f + 2] * (double)x + hh[4 * f + 3] > 0) { in_hull = false; break; }
        inside = in_hull && nh >= 4;
      } else if (render_mode == 3) {
        inside = true;
      } else if (render_mode == 4) {                                // "debug": flag kernel-but-not-polyhedron voxels with -1
        if (sd3::inside_polyhedron_kernel(z, y, x, pv, fc, n_faces) && !sd3::inside_polyhedron(z, y, x, cz, cy, cx, pv, fc, n_faces))
          result_dbg[((size_t)(zi - z0) * ny + (yi - y0)) * nx + (xi - x0)] = -1;
      }
      if (inside) {
        const size_t off = ((size_t)(zi - z0) * ny + (yi - y0)) * nx + (xi - x0);
        atomicMin(&first[off], p);
        atomicAdd(&count[off], 1);
      }
    }
  }
}

__global__ void k_resolve3d(int* __restrict__ result, const int* __restrict__ first, const int* __restrict__ count, long long n,
                            const int* __restrict__ labels, int use_overlap, int overlap_label) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = count[i];
  if (c == 0) return;
  const int old = result[i];
  int v;
  if (old == 0) { v = labels[first[i]]; if (use_overlap && c >= 2) v = overlap_label; }   // :1511-1517
  else v = use_overlap ? overlap_label : old;
  result[i] = v;
}

// exact sequential semantics for the degenerate label values (a label or overlap_label equal to 0)
__global__ void k_resolve3d_seq(int* __restrict__ result, const int* __restrict__ first, const int* __restrict__ count, long long n,
                                int label, int use_overlap, int overlap_label) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (count[i] == 0) return;
  const int old = result[i];
  result[i] = (old == 0) ? label : (use_overlap ? overlap_label : old);
}

__global__ void k_fill(int* a, long long n, int v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

}  // namespace

extern "C" int sd_polyhedron_to_label_window_device(const float* d_dist, const float* d_points, const float* d_verts, const int* d_faces,
                                                    int n_polys, int n_rays, int n_faces, const int* d_labels, int NZ, int NY, int NX, int z0,
                                                    int y0, int x0, int nz, int ny, int nx, int render_mode, int verbose,
                                                    int use_overlap_label, int overlap_label, int* d_result, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + nz > NZ || y0 + ny > NY || x0 + nx > NX) {
    sd::set_error("sd_polyhedron_to_label_window: window outside the volume");
    return -1;
  }
  if (verbose >= 1) {
    printf("+++++++++++++++ polyhedra to label +++++++++++++++ \n");
    printf("n_polys           = %d \nn_rays            = %d \nn_faces           = %d \nnz, ny, nx        = %d %d %d \n", n_polys, n_rays, n_faces, nz, ny, nx);
    printf("use_overlap_label = %d \noverlap_label     = %d \n", use_overlap_label, overlap_label);
    fflush(stdout);
  }
  if (n_polys <= 0 || nz <= 0 || ny <= 0 || nx <= 0) return 0;
  if (render_mode < 0 || render_mode > 4) { sd::set_error("sd_polyhedron_to_label: unknown render_mode %d", render_mode); return -1; }
  const bool need_hull = (render_mode == 0 || render_mode == 2);
  const int hullCapL = need_hull ? 2 * n_rays : 0;
  const size_t lds = (size_t)(n_faces + hullCapL) * 4 * sizeof(double) + (size_t)n_rays * 3 * sizeof(float) + (size_t)n_faces * 3 * sizeof(int);
  if (lds > 150 * 1024) { sd::set_error("sd_polyhedron_to_label: n_rays/n_faces too large for LDS staging"); return -1; }
  if (lds > 64 * 1024) SD_CHECK(hipFuncSetAttribute((const void*)k_paint3d, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long nvox = (long long)nz * ny * nx;
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  double* hullPlanes = nullptr; int* hullCount = nullptr; int hullCap = 0;
  int* hullFail = A.take_n<int>(1);
  if (!hullFail) return -1;
  SD_CHECK(hipMemsetAsync(hullFail, 0, sizeof(int), s));
  if (need_hull && n_rays >= 4) {
    if (sd::hull_planes(d_dist, d_points, d_verts, n_polys, n_rays, &hullPlanes, &hullCount, &hullCap, s)) return -1;
  } else if (render_mode == 2) { sd::set_error("sd_polyhedron_to_label: render_mode 'hull' needs n_rays >= 4"); return -1; }
  sd3::ConeMap cmap{nullptr, nullptr};
  if (render_mode == 0 && sd::cone_map(d_verts, d_faces, n_faces, &cmap, s)) return -1;
  int* first = A.take_n<int>(nvox);
  int* count = A.take_n<int>(nvox);
  if (!first || !count) return -1;
  std::vector<int> h_labels(n_polys);
  SD_CHECK(hipMemcpyAsync(h_labels.data(), d_labels, (size_t)n_polys * sizeof(int), hipMemcpyDeviceToHost, s));
  SD_CHECK(hipStreamSynchronize(s));
  bool degenerate = (use_overlap_label && overlap_label == 0);
  for (int i = 0; i < n_polys && !degenerate; ++i) degenerate = (h_labels[i] == 0);
  const unsigned gb = (unsigned)((nvox + 255) / 256);
  if (!degenerate) {
    hipLaunchKernelGGL(k_fill, dim3(gb), dim3(256), 0, s, first, nvox, INT_MAX);
    SD_CHECK(hipMemsetAsync(count, 0, nvox * sizeof(int), s));
    const int blocks = n_polys < 8192 ? n_polys : 8192;
    hipLaunchKernelGGL(k_paint3d, dim3(blocks), dim3(256), lds, s, d_dist, d_points, d_verts, d_faces, 0, n_polys, n_rays, n_faces,
                       NZ, NY, NX, z0, y0, x0, nz, ny, nx, render_mode, first, count, d_result, hullPlanes, hullCount, hullCapL, hullFail, cmap);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_resolve3d, dim3(gb), dim3(256), 0, s, d_result, first, count, nvox, d_labels, use_overlap_label, overlap_label);
    SD_LAUNCH_CHECK();
  } else {
    // a zero label re-opens the voxel for later polyhedra: keep the reference's strictly sequential order
    for (int p = 0; p < n_polys; ++p) {
      hipLaunchKernelGGL(k_fill, dim3(gb), dim3(256), 0, s, first, nvox, INT_MAX);
      SD_CHECK(hipMemsetAsync(count, 0, nvox * sizeof(int), s));
      hipLaunchKernelGGL(k_paint3d, dim3(1), dim3(256), lds, s, d_dist, d_points, d_verts, d_faces, p, p + 1, n_rays, n_faces, NZ, NY, NX,
                         z0, y0, x0, nz, ny, nx, render_mode, first, count, d_result, hullPlanes, hullCount, hullCapL, hullFail, cmap);
      hipLaunchKernelGGL(k_resolve3d_seq, dim3(gb), dim3(256), 0, s, d_result, first, count, nvox, h_labels[p], use_overlap_label, overlap_label);
      SD_LAUNCH_CHECK();
    }
  }
  if (render_mode == 2) {
    int hf = 0;
    SD_CHECK(hipMemcpyAsync(&hf, hullFail, sizeof(int), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    if (hf) { sd::set_error("sd_polyhedron_to_label: the convex hull of %d polyhedra could not be built (degenerate vertices; Qhull raises here)", hf); return -1; }
  }
  return 0;
}

extern "C" int sd_polyhedron_to_label_device(const float* d_dist, const float* d_points, const float* d_verts, const int* d_faces,
                                             int n_polys, int n_rays, int n_faces, const int* d_labels, int nz, int ny, int nx,
                                             int render_mode, int verbose, int use_overlap_label, int overlap_label,
                                             int* d_result, void* stream) {
  return sd_polyhedron_to_label_window_device(d_dist, d_points, d_verts, d_faces, n_polys, n_rays, n_faces, d_labels, nz, ny, nx, 0, 0, 0, nz,
                                              ny, nx, render_mode, verbose, use_overlap_label, overlap_label, d_result, stream);
}

extern "C" void _LIB_polyhedron_to_label(const float* dist, const float* points, const float* verts, const int* faces,
                                         const int n_polys, const int n_rays, const int n_faces, const int* labels, const int nz,
                                         const int ny, const int nx, const int render_mode, const int verbose,
                                         const int use_overlap_label, const int overlap_label, int* result) {
  if (n_polys <= 0 || nz <= 0 || ny <= 0 || nx <= 0) return;
  const size_t nvox = (size_t)nz * ny * nx;
  float *d_dist = nullptr, *d_pts = nullptr, *d_verts = nullptr;
  int *d_faces = nullptr, *d_labels = nullptr, *d_res = nullptr;
  bool ok = hipMalloc(&d_dist, (size_t)n_polys * n_rays * 4) == hipSuccess && hipMalloc(&d_pts, (size_t)n_polys * 12) == hipSuccess &&
            hipMalloc(&d_verts, (size_t)n_rays * 12) == hipSuccess && hipMalloc(&d_faces, (size_t)n_faces * 12) == hipSuccess &&
            hipMalloc(&d_labels, (size_t)n_polys * 4) == hipSuccess && hipMalloc(&d_res, nvox * 4) == hipSuccess;
  if (ok) {
    ok = hipMemcpy(d_dist, dist, (size_t)n_polys * n_rays * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_pts, points, (size_t)n_polys * 12, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_verts, verts, (size_t)n_rays * 12, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_faces, faces, (size_t)n_faces * 12, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_labels, labels, (size_t)n_polys * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_res, result, nvox * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) sd::set_error("_LIB_polyhedron_to_label: H2D failed");
  } else sd::set_error("_LIB_polyhedron_to_label: hipMalloc failed");
  if (ok) ok = sd_polyhedron_to_label_device(d_dist, d_pts, d_verts, d_faces, n_polys, n_rays, n_faces, d_labels, nz, ny, nx, render_mode,
                                             verbose, use_overlap_label, overlap_label, d_res, nullptr) == 0;
  if (ok) ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(result, d_res, nvox * 4, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d_dist); (void)hipFree(d_pts); (void)hipFree(d_verts); (void)hipFree(d_faces); (void)hipFree(d_labels); (void)hipFree(d_res);
  if (!ok) {   // the reference ABI has no return code: fail loudly
    fprintf(stderr, "_LIB_polyhedron_to_label failed: %s\n", sd::err_buf());
    abort();
  }
}
