// Pair-level probe of the area enclosure the 2D NMS decides most of its pairs with (area_bounds.h): tests compare it with the exact
// Clipper-equivalent area of sd_clip_pairs_device on the same polygons.
#include "common.h"
#include "area_bounds.h"
#include "../../include/stardist_hip.h"

namespace {
__global__ void __launch_bounds__(256) k_area_probe(const int* __restrict__ xa, const int* __restrict__ ya, const int* __restrict__ xb,
                                                    const int* __restrict__ yb, int n, int R, const sdarea::PolyProps* __restrict__ pa,
                                                    const sdarea::PolyProps* __restrict__ pb, float* __restrict__ out_area, float* __restrict__ out_band,
                                                    int* __restrict__ out_info) {
  __shared__ float2 sq[4][2][32];
  const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31, wv = threadIdx.x >> 6;
  const long long nw = (long long)gridDim.x * 4;
  for (long long base = 2ll * ((long long)blockIdx.x * 4 + wv); base < n; base += 2ll * nw) {
    const long long t = base + half;
    const bool active = t < n;
    const long long tt = active ? t : 0;
    const sdarea::Enclosure E = sdarea::pair_enclosure(xa + tt * R, ya + tt * R, xb + tt * R, yb + tt * R, R, pa[tt], pb[tt], active, sq[wv][half], l, half);
    if (active && l == 0) { out_area[t] = E.area; out_band[t] = E.band; out_info[t] = (E.usable ? 1 : 0) | ((E.crossings & 0x3FF) << 8) | ((E.near & 0x7FF) << 18); }
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace

extern "C" int sd_area_bounds_pairs_device(const int32_t* d_xa, const int32_t* d_ya, const int32_t* d_xb, const int32_t* d_yb, int n_pairs, int n_verts,
                                           float* d_out_area, float* d_out_band, int32_t* d_out_info, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_pairs <= 0) return 0;
  const int R = n_verts;
  if (R < 3 || R > 32) { sd::set_error("sd_area_bounds_pairs: n_verts=%d unsupported (3..32)", R); return -1; }
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  sdarea::PolyProps* pa = (sdarea::PolyProps*)A.take((size_t)n_pairs * sizeof(sdarea::PolyProps));
  sdarea::PolyProps* pb = (sdarea::PolyProps*)A.take((size_t)n_pairs * sizeof(sdarea::PolyProps));
  if (!pa || !pb) return -1;
  hipLaunchKernelGGL(sdarea::k_poly_props, dim3(sd::div_up(n_pairs, 8)), dim3(256), 0, s, d_xa, d_ya, n_pairs, R, pa);
  hipLaunchKernelGGL(sdarea::k_poly_props, dim3(sd::div_up(n_pairs, 8)), dim3(256), 0, s, d_xb, d_yb, n_pairs, R, pb);
  const int blocks = sd::div_up(n_pairs, 8) < 2048 ? sd::div_up(n_pairs, 8) : 2048;
  hipLaunchKernelGGL(k_area_probe, dim3(blocks), dim3(256), 0, s, d_xa, d_ya, d_xb, d_yb, n_pairs, R, pa, pb, d_out_area, d_out_band, d_out_info);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipStreamSynchronize(s));
  return 0;
}
