// star_dist.hip -- star-convex radial distances of a label image (2D / 3D).
//
// Replaces c_star_dist (stardist/lib/stardist2d.cpp:55-124) and c_star_dist3d
// (stardist/lib/stardist3d.cpp:245-346).  One thread per (grid pixel, ray), ray index
// fastest, so the (.., n_rays) float32 output is written fully coalesced; label reads along
// a ray are gathers served by L1/L2 (a 2048^2 uint16 image is 8 MiB, L2/MALL resident).
// Ray direction tables are computed by the HOST libm and staged in LDS, so the device never
// evaluates sinf/cosf (bit parity with the reference's glibc).
// Float arithmetic order follows the reference line by line; compiled with -ffp-contract=off.
#include "common.h"
#include "../../include/stardist_hip.h"
#include <math.h>
#include <vector>

namespace {

__device__ __forceinline__ int round_to_int(float r) { return __float2int_rn(r); }   // lrint, round-half-even

__global__ void __launch_bounds__(256) k_star_dist2d(const unsigned short* __restrict__ src, int H, int W, int R,
                                                     int gy, int gx, int Ho, int Wo, const float2* __restrict__ dirs,
                                                     float* __restrict__ dst) {
  extern __shared__ float2 sdir[];
  for (int k = threadIdx.x; k < R; k += blockDim.x) sdir[k] = dirs[k];
  __syncthreads();
  const long long total = (long long)Ho * Wo * R;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t % R);
    const long long pix = t / R;
    const int j = (int)(pix % Wo), i = (int)(pix / Wo);
    const unsigned short value = src[(size_t)(i * gy) * W + (size_t)j * gx];
    float out = 0.f;
    if (value != 0) {
      const float dx = sdir[k].x, dy = sdir[k].y;     // dx = sin(phi), dy = cos(phi)   :93-95
      float x = 0.f, y = 0.f;
      const int bi = i * gy, bj = j * gx;
      while (1) {
        x += dx; y += dy;
        const int ii = round_to_int((float)bi + x), jj = round_to_int((float)bj + y);   // :101
        if (ii < 0 || ii >= H || jj < 0 || jj >= W || value != src[(size_t)ii * W + jj]) {
          const float t_corr = .5f / fmaxf(fabsf(dx), fabsf(dy));                       // :108
          x += (t_corr - 1.f) * dx;
          y += (t_corr - 1.f) * dy;
          out = sqrtf(x * x + y * y);
          break;
        }
      }
    }
    dst[t] = out;
  }
}

__global__ void __launch_bounds__(256) k_star_dist3d(const unsigned short* __restrict__ src, int Z, int Y, int X, int R,
                                                     int gz, int gy, int gx, int Zo, int Yo, int Xo,
                                                     const float* __restrict__ pdz, const float* __restrict__ pdy,
                                                     const float* __restrict__ pdx, float* __restrict__ dst) {
  extern __shared__ float sray[];   // 3*R : dz | dy | dx
  for (int k = threadIdx.x; k < R; k += blockDim.x) { sray[k] = pdz[k]; sray[R + k] = pdy[k]; sray[2 * R + k] = pdx[k]; }
  __syncthreads();
  const long long total = (long long)Zo * Yo * Xo * R;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(t % R);
    long long pix = t / R;
    const int k = (int)(pix % Xo); pix /= Xo;
    const int j = (int)(pix % Yo);
    const int i = (int)(pix / Yo);
    const unsigned short value = src[((size_t)(i * gz) * Y + (size_t)j * gy) * X + (size_t)k * gx];
    float out = 0.f;
    if (value != 0) {
      const float dz = sray[n], dy = sray[R + n], dx = sray[2 * R + n];
      float x = 0.f, y = 0.f, z = 0.f;
      const int bi = i * gz, bj = j * gy, bk = k * gx;
      while (1) {
        x += dx; y += dy; z += dz;
        const int ii = round_to_int((float)bi + z), jj = round_to_int((float)bj + y), kk = round_to_int((float)bk + x);  // :307
        if (ii < 0 || ii >= Z || jj < 0 || jj >= Y || kk < 0 || kk >= X ||
            value != src[((size_t)ii * Y + jj) * X + kk]) {
          const int x2 = round_to_int(x), y2 = round_to_int(y), z2 = round_to_int(z);
          out = (float)sqrt((double)(x2 * x2 + y2 * y2 + z2 * z2));                     // :317-320 (sqrt(int) -> double)
          break;
        }
      }
    }
    dst[t] = out;
  }
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  const long long cap = 256 * 32;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

extern "C" int sd_star_dist2d_device(const uint16_t* d_src, int H, int W, int n_rays, int grid_y, int grid_x,
                                     float* d_dst, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (H <= 0 || W <= 0 || n_rays <= 0) return 0;
  if (grid_y < 1 || grid_x < 1) { sd::set_error("sd_star_dist2d: grid must be >= 1"); return -1; }
  const int R = n_rays;
  // host libm table: dy = cos(phi), dx = sin(phi), phi = k * (float)(2*pi/n_rays)   :91-95
  std::vector<float2> dirs(R);
  const float st_rays = (2 * M_PI) / R;
  for (int k = 0; k < R; ++k) { const float phi = k * st_rays; dirs[k].y = cosf(phi); dirs[k].x = sinf(phi); }
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  float2* d_dirs = A.take_n<float2>(R);
  if (!d_dirs) return -1;
  SD_CHECK(hipMemcpyAsync(d_dirs, dirs.data(), R * sizeof(float2), hipMemcpyHostToDevice, s));
  const int Ho = (H - 1) / grid_y + 1, Wo = (W - 1) / grid_x + 1;
  const long long total = (long long)Ho * Wo * R;
  hipLaunchKernelGGL(k_star_dist2d, dim3(grid_for(total)), dim3(256), R * sizeof(float2), s, d_src, H, W, R, grid_y, grid_x,
                     Ho, Wo, d_dirs, d_dst);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipStreamSynchronize(s));   // dirs (host vector) must outlive the async copy
  return 0;
}

extern "C" int sd_star_dist3d_device(const uint16_t* d_src, int Z, int Y, int X, const float* d_dz, const float* d_dy,
                                     const float* d_dx, int n_rays, int grid_z, int grid_y, int grid_x, float* d_dst,
                                     void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (Z <= 0 || Y <= 0 || X <= 0 || n_rays <= 0) return 0;
  if (grid_z < 1 || grid_y < 1 || grid_x < 1) { sd::set_error("sd_star_dist3d: grid must be >= 1"); return -1; }
  const int Zo = (Z - 1) / grid_z + 1, Yo = (Y - 1) / grid_y + 1, Xo = (X - 1) / grid_x + 1;
  const long long total = (long long)Zo * Yo * Xo * n_rays;
  hipLaunchKernelGGL(k_star_dist3d, dim3(grid_for(total)), dim3(256), 3 * n_rays * sizeof(float), s, d_src, Z, Y, X, n_rays,
                     grid_z, grid_y, grid_x, Zo, Yo, Xo, d_dz, d_dy, d_dx, d_dst);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_star_dist2d_host(const uint16_t* src, int H, int W, int n_rays, int grid_y, int grid_x, float* dst) {
  if (H <= 0 || W <= 0 || n_rays <= 0) return 0;
  if (grid_y < 1 || grid_x < 1) { sd::set_error("sd_star_dist2d: grid must be >= 1"); return -1; }
  const int Ho = (H - 1) / grid_y + 1, Wo = (W - 1) / grid_x + 1;
  const size_t nin = (size_t)H * W * sizeof(uint16_t), nout = (size_t)Ho * Wo * n_rays * sizeof(float);
  uint16_t* d_src = nullptr; float* d_dst = nullptr;
  SD_CHECK(hipMalloc(&d_src, nin));
  SD_CHECK(hipMalloc(&d_dst, nout));
  int rc = -1;
  do {
    if (hipMemcpy(d_src, src, nin, hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (sd_star_dist2d_device(d_src, H, W, n_rays, grid_y, grid_x, d_dst, nullptr)) break;
    if (hipMemcpy(dst, d_dst, nout, hipMemcpyDeviceToHost) != hipSuccess) { sd::set_error("D2H failed"); break; }
    rc = 0;
  } while (0);
  (void)hipFree(d_src); (void)hipFree(d_dst);
  return rc;
}

extern "C" int sd_star_dist3d_host(const uint16_t* src, int Z, int Y, int X, const float* dz, const float* dy,
                                   const float* dx, int n_rays, int grid_z, int grid_y, int grid_x, float* dst) {
  if (Z <= 0 || Y <= 0 || X <= 0 || n_rays <= 0) return 0;
  if (grid_z < 1 || grid_y < 1 || grid_x < 1) { sd::set_error("sd_star_dist3d: grid must be >= 1"); return -1; }
  const int Zo = (Z - 1) / grid_z + 1, Yo = (Y - 1) / grid_y + 1, Xo = (X - 1) / grid_x + 1;
  const size_t nin = (size_t)Z * Y * X * sizeof(uint16_t), nout = (size_t)Zo * Yo * Xo * n_rays * sizeof(float);
  uint16_t* d_src = nullptr; float *d_dst = nullptr, *d_r = nullptr;
  SD_CHECK(hipMalloc(&d_src, nin));
  SD_CHECK(hipMalloc(&d_dst, nout));
  SD_CHECK(hipMalloc(&d_r, 3 * (size_t)n_rays * sizeof(float)));
  int rc = -1;
  do {
    if (hipMemcpy(d_src, src, nin, hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (hipMemcpy(d_r, dz, n_rays * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (hipMemcpy(d_r + n_rays, dy, n_rays * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (hipMemcpy(d_r + 2 * n_rays, dx, n_rays * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (sd_star_dist3d_device(d_src, Z, Y, X, d_r, d_r + n_rays, d_r + 2 * n_rays, n_rays, grid_z, grid_y, grid_x, d_dst, nullptr)) break;
    if (hipMemcpy(dst, d_dst, nout, hipMemcpyDeviceToHost) != hipSuccess) { sd::set_error("D2H failed"); break; }
    rc = 0;
  } while (0);
  (void)hipFree(d_src); (void)hipFree(d_dst); (void)hipFree(d_r);
  return rc;
}
