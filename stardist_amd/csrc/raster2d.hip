// raster2d.hip -- polygon -> label image rasteriser (2D).
//
// The reference has no native for this: stardist/geometry/geom2d.py:149-166 loops in Python
// over skimage.draw.polygon(r, c, shape) and assigns lbl[rr,cc] = label+1, so later polygons
// overwrite earlier ones.  scikit-image is not vendored; its published rule is restated:
//   bbox   minr=int(max(0,r.min())), maxr=min(H-1,int(ceil(r.max()))) (same for c)   [_draw.pyx _polygon]
//   inside point_in_polygon(xp=c, yp=r, x=col, y=row) != 0                            [_shared/geometry.pyx]
//          O'Rourke crossing test in float64 returning OUTSIDE/INSIDE/VERTEX/EDGE; vertex and
//          edge points count as inside.
// GPU formulation: one workgroup per polygon, vertices staged in LDS as float64, threads
// sweep the clipped bbox; overwrite order is resolved with atomicMax on the paint position
// (polygon index + 1), then a second pass maps position -> labels[pos-1] + 1.
#include "common.h"
#include "../../include/stardist_hip.h"
#include <math.h>

namespace {

__device__ __forceinline__ int point_in_polygon(const double* xp, const double* yp, int n, double x, double y) {
  unsigned int l_cross = 0, r_cross = 0;
  const double eps = (double)1e-12f;
  double x1 = xp[n - 1] - x, y1 = yp[n - 1] - y;
  for (int i = 0; i < n; ++i) {
    const double x0 = xp[i] - x, y0 = yp[i] - y;
    if ((-eps < x0 && x0 < eps) && (-eps < y0 && y0 < eps)) return 2;            // VERTEX
    if ((y0 > 0) != (y1 > 0)) { if (((x0 * y1 - x1 * y0) / (y1 - y0)) > 0) r_cross++; }
    if ((y0 < 0) != (y1 < 0)) { if (((x0 * y1 - x1 * y0) / (y1 - y0)) < 0) l_cross++; }
    x1 = x0; y1 = y0;
  }
  if ((r_cross & 1) != (l_cross & 1)) return 3;                                    // EDGE
  return (r_cross & 1) ? 1 : 0;
}

// paints the window [y0, y0 + H) x [x0, x0 + W) of an HI x WI image: pixel tests in image coordinates, stores window-relative
__global__ void __launch_bounds__(256) k_paint(const float* __restrict__ coord, int n_polys, int R, int HI, int WI, int y0, int x0, int H, int W,
                                               int* __restrict__ img) {
  extern __shared__ double sv[];   // r[R] | c[R]
  double* sr = sv; double* scol = sv + R;
  __shared__ float red[4][4];
  for (int p = blockIdx.x; p < n_polys; p += gridDim.x) {
    const float* cr = coord + (size_t)p * 2 * R;
    float rmin = INFINITY, rmax = -INFINITY, cmin = INFINITY, cmax = -INFINITY;
    __syncthreads();
    for (int k = threadIdx.x; k < R; k += blockDim.x) {
      const float r = cr[k], c = cr[R + k];
      sr[k] = (double)r; scol[k] = (double)c;
      rmin = fminf(rmin, r); rmax = fmaxf(rmax, r); cmin = fminf(cmin, c); cmax = fmaxf(cmax, c);
    }
    for (int o = 32; o; o >>= 1) {
      rmin = fminf(rmin, __shfl_xor(rmin, o)); rmax = fmaxf(rmax, __shfl_xor(rmax, o));
      cmin = fminf(cmin, __shfl_xor(cmin, o)); cmax = fmaxf(cmax, __shfl_xor(cmax, o));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = rmin; red[wave][1] = rmax; red[wave][2] = cmin; red[wave][3] = cmax; }
    __syncthreads();
    rmin = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
    rmax = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
    cmin = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
    cmax = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    // int(max(0, min)) truncates a non-negative value; ceil on the max; clip to the image
    long long minr = (long long)fmaxf(0.f, rmin), minc = (long long)fmaxf(0.f, cmin);
    long long maxr = (long long)ceilf(rmax), maxc = (long long)ceilf(cmax);
    if (maxr > HI - 1) maxr = HI - 1;
    if (maxc > WI - 1) maxc = WI - 1;
    if (minr < y0) minr = y0;
    if (minc < x0) minc = x0;
    if (maxr > y0 + H - 1) maxr = y0 + H - 1;
    if (maxc > x0 + W - 1) maxc = x0 + W - 1;
    if (maxr < minr || maxc < minc) continue;
    const long long nr = maxr - minr + 1, nc = maxc - minc + 1;
    for (long long t = threadIdx.x; t < nr * nc; t += blockDim.x) {
      const int r_i = (int)(minr + t / nc), c_i = (int)(minc + t % nc);
      if (point_in_polygon(scol, sr, R, (double)c_i, (double)r_i)) atomicMax(&img[(size_t)(r_i - y0) * W + (c_i - x0)], p + 1);
    }
  }
}

__global__ void k_map_labels(int* __restrict__ img, long long n, const int* __restrict__ labels) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int v = img[i]; if (v > 0) img[i] = labels[v - 1] + 1; }
}

}  // namespace

// dist_to_coord (stardist/geometry/geom2d.py:130-146) in numpy's arithmetic, one launch: coord[i][a][k] =
//   float32( float64( float32( float64(dist[i][k]) * sc[a][k] ) [* scale[a] -> float32] ) + float64(points[i][a]) )
// sc = (sin, cos)(linspace(0, 2 pi, R, endpoint=False)) in float64 from the HOST's libm (a device table uploaded by the caller):
// `(dist[:, None] * sc).astype(float32)`, `coord *= scale` (only when scale != 1: float32 * float64 rounded to float32), then
// `coord += points[..., None]` (added in float64, rounded once).
__global__ void __launch_bounds__(256) k_dist_to_coord(const float* __restrict__ dist, const double* __restrict__ points, const double* __restrict__ sc,
                                                       long long n, int R, double sy, double sx, int scaled, float* __restrict__ coord) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * 2 * R) return;
  const int k = (int)(t % R), a = (int)((t / R) % 2);
  const long long i = t / (2 * R);
  float c = (float)((double)dist[i * R + k] * sc[a * R + k]);
  if (scaled) c = (float)((double)c * (a ? sx : sy));
  coord[t] = (float)((double)c + points[i * 2 + a]);
}

extern "C" int sd_dist_to_coord_device(const float* d_dist, const double* d_points, const double* d_sincos, long long n_polys, int n_rays,
                                       double scale_y, double scale_x, float* d_coord, void* stream) {
  if (n_polys <= 0 || n_rays <= 0) return 0;
  const long long tot = n_polys * 2 * n_rays;
  hipLaunchKernelGGL(k_dist_to_coord, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_dist, d_points, d_sincos, n_polys, n_rays,
                     scale_y, scale_x, (scale_y != 1.0 || scale_x != 1.0) ? 1 : 0, d_coord);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_polygons_to_label_window_device(const float* d_coord, const int32_t* d_labels, int n_polys, int n_rays, int HI, int WI,
                                                  int y0, int x0, int H, int W, int32_t* d_result, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (H <= 0 || W <= 0) return 0;
  if (y0 < 0 || x0 < 0 || y0 + H > HI || x0 + W > WI) { sd::set_error("sd_polygons_to_label_window: window outside the image"); return -1; }
  SD_CHECK(hipMemsetAsync(d_result, 0, (size_t)H * W * sizeof(int32_t), s));
  if (n_polys <= 0 || n_rays <= 0) return 0;
  if ((size_t)n_rays * 2 * sizeof(double) > 60000) { sd::set_error("sd_polygons_to_label: n_rays=%d too large", n_rays); return -1; }
  const int blocks = n_polys < 65535 * 16 ? n_polys : 65535 * 16;
  hipLaunchKernelGGL(k_paint, dim3(blocks), dim3(256), 2 * n_rays * sizeof(double), s, d_coord, n_polys, n_rays, HI, WI, y0, x0, H, W, d_result);
  SD_LAUNCH_CHECK();
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(k_map_labels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_result, n, d_labels);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_polygons_to_label_device(const float* d_coord, const int32_t* d_labels, int n_polys, int n_rays,
                                           int H, int W, int32_t* d_result, void* stream) {
  return sd_polygons_to_label_window_device(d_coord, d_labels, n_polys, n_rays, H, W, 0, 0, H, W, d_result, stream);
}

extern "C" int sd_polygons_to_label_host(const float* coord, const int32_t* labels, int n_polys, int n_rays, int H, int W,
                                         int32_t* result) {
  if (H <= 0 || W <= 0) return 0;
  float* d_coord = nullptr; int* d_lab = nullptr; int* d_res = nullptr;
  const size_t nc = (size_t)(n_polys > 0 ? n_polys : 1) * 2 * (n_rays > 0 ? n_rays : 1) * sizeof(float);
  SD_CHECK(hipMalloc(&d_coord, nc));
  SD_CHECK(hipMalloc(&d_lab, (size_t)(n_polys > 0 ? n_polys : 1) * sizeof(int)));
  SD_CHECK(hipMalloc(&d_res, (size_t)H * W * sizeof(int)));
  int rc = -1;
  do {
    if (n_polys > 0) {
      if (hipMemcpy(d_coord, coord, (size_t)n_polys * 2 * n_rays * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
      if (hipMemcpy(d_lab, labels, (size_t)n_polys * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    }
    if (sd_polygons_to_label_device(d_coord, d_lab, n_polys, n_rays, H, W, d_res, nullptr)) break;
    if (hipMemcpy(result, d_res, (size_t)H * W * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { sd::set_error("D2H failed"); break; }
    rc = 0;
  } while (0);
  (void)hipFree(d_coord); (void)hipFree(d_lab); (void)hipFree(d_res);
  return rc;
}
