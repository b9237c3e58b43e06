// survivors.hip -- what follows the 2D NMS on predict_instances(): the rows of the survivors, their polygon coordinates and the order in
// which the rasteriser paints them, in two calls instead of a dozen framework launches (round 6: the chain nonzero -> 3 gathers ->
// dist_to_coord -> stable sort by score -> gather took 0.45 ms of a 14.7 ms step, nearly all of it launch gaps).
//   sd_survivor_positions_device   positions of the non-zero keep flags, ascending (what torch.nonzero / boolean-mask indexing compute)
//   sd_survivors2d_device          model2d.py:536-561 for candidates in SCORE order: points / prob of the survivors, coord = dist_to_coord
//                                  (geom2d.py:130-146, the arithmetic of k_dist_to_coord in raster2d.hip), and the same coordinates once more
//                                  in PAINTING order with their label ids: polygons_to_label paints in ascending score order, stable
//                                  (geom2d.py:186-197: `ind = np.argsort(prob, kind='stable')`), label id = position in NMS order + 1
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {
enum { ITEMS = 4, BLOCK = 256, TILE = ITEMS * BLOCK };

__global__ void __launch_bounds__(BLOCK) k_keep_count(const unsigned char* __restrict__ keep, long long n, int* __restrict__ blockCount) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.x * TILE;
  int c = 0;
  for (int r = 0; r < ITEMS; ++r) {
    const long long idx = base + r * BLOCK + threadIdx.x;
    c += __popcll(__ballot(idx < n && keep[idx] != 0));
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blockCount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(BLOCK) k_keep_write(const unsigned char* __restrict__ keep, long long n, const int* __restrict__ blockStart,
                                                      long long* __restrict__ pos, int* __restrict__ count, int nBlocks) {
  __shared__ int wcnt[ITEMS][4];
  const long long base = (long long)blockIdx.x * TILE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long masks[ITEMS];
  for (int r = 0; r < ITEMS; ++r) {
    const long long idx = base + r * BLOCK + threadIdx.x;
    masks[r] = __ballot(idx < n && keep[idx] != 0);
    if (lane == 0) wcnt[r][wave] = __popcll(masks[r]);
  }
  __syncthreads();
  int run = blockStart[blockIdx.x];
  for (int r = 0; r < ITEMS; ++r) {
    for (int w = 0; w < 4; ++w) {
      if (w == wave && ((masks[r] >> lane) & 1ull)) {
        const int o = run + __popcll(masks[r] & ((1ull << lane) - 1ull));
        pos[o] = base + r * BLOCK + threadIdx.x;
      }
      run += wcnt[r][w];
    }
  }
  if (blockIdx.x == nBlocks - 1 && threadIdx.x == 0) *count = run;
}

// one thread per survivor r (NMS order = descending score): its row, and its place in the painting order.  Ascending and stable over a
// descending list = the groups of equal scores in reverse, each group in its own order: rank = (m - end of r's group) + (r - start of the group)
__global__ void __launch_bounds__(256) k_survivor_rows(const long long* __restrict__ pos, int m, const float* __restrict__ prob,
                                                       const long long* __restrict__ points, float* __restrict__ oprob,
                                                       long long* __restrict__ opoints, int* __restrict__ rank, int* __restrict__ labelsPaint) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  const long long i = pos[r];
  const float p = prob[i];
  oprob[r] = p;
  opoints[2 * r] = points[2 * i]; opoints[2 * r + 1] = points[2 * i + 1];
  int gs = r, ge = r + 1;
  while (gs > 0 && prob[pos[gs - 1]] == p) --gs;
  while (ge < m && prob[pos[ge]] == p) ++ge;
  const int rk = (m - ge) + (r - gs);
  rank[r] = rk;
  labelsPaint[rk] = r;
}

// k_dist_to_coord (raster2d.hip) on the survivors' rows, written twice: NMS order (the result dict) and painting order (the rasteriser)
__global__ void __launch_bounds__(256) k_survivor_coord(const long long* __restrict__ pos, int m, const float* __restrict__ dist,
                                                        const long long* __restrict__ points, const double* __restrict__ sc, int R,
                                                        const int* __restrict__ rank, float* __restrict__ coord, float* __restrict__ coordPaint) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)m * 2 * R) return;
  const int k = (int)(t % R), a = (int)((t / R) % 2);
  const int r = (int)(t / (2 * R));
  const long long i = pos[r];
  const float c = (float)((double)dist[i * R + k] * sc[a * R + k]);
  const float v = (float)((double)c + (double)points[2 * i + a]);
  coord[t] = v;
  if (coordPaint) coordPaint[((long long)rank[r] * 2 + a) * R + k] = v;
}
}  // namespace

extern "C" int sd_survivor_positions_device(const unsigned char* d_keep, long long n, long long* d_positions, int32_t* d_count, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  SD_CHECK(hipMemsetAsync(d_count, 0, sizeof(int), s));
  if (n <= 0) return 0;
  if (n > (1ll << 31) - TILE) { sd::set_error("sd_survivor_positions: more than 2^31 flags"); return -1; }
  const int nBlocks = (int)((n + TILE - 1) / TILE);
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  int* blockCount = A.take_n<int>(nBlocks + 1);
  int* blockStart = A.take_n<int>(nBlocks + 1);
  size_t tmpBytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, blockCount, blockStart, nBlocks, s);
  void* tmp = A.take(tmpBytes + 256);
  if (!blockCount || !blockStart || !tmp) return -1;
  hipLaunchKernelGGL(k_keep_count, dim3(nBlocks), dim3(BLOCK), 0, s, d_keep, n, blockCount);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmpBytes, blockCount, blockStart, nBlocks, s));
  hipLaunchKernelGGL(k_keep_write, dim3(nBlocks), dim3(BLOCK), 0, s, d_keep, n, blockStart, d_positions, d_count, nBlocks);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_survivors2d_device(const long long* d_positions, int m, const float* d_prob, const long long* d_points, const float* d_dist,
                                     int n_rays, const double* d_sincos, float* d_out_prob, long long* d_out_points, float* d_out_coord,
                                     float* d_out_coord_paint, int32_t* d_out_labels_paint, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (m <= 0) return 0;
  if (n_rays <= 0) { sd::set_error("sd_survivors2d: n_rays must be positive"); return -1; }
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  int* rank = A.take_n<int>(m);
  int* lab = d_out_labels_paint ? d_out_labels_paint : A.take_n<int>(m);
  if (!rank || !lab) return -1;
  hipLaunchKernelGGL(k_survivor_rows, dim3((m + 255) / 256), dim3(256), 0, s, d_positions, m, d_prob, d_points, d_out_prob, d_out_points, rank, lab);
  SD_LAUNCH_CHECK();
  const long long tot = (long long)m * 2 * n_rays;
  hipLaunchKernelGGL(k_survivor_coord, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, d_positions, m, d_dist, d_points, d_sincos, n_rays, rank,
                     d_out_coord, d_out_coord_paint);
  SD_LAUNCH_CHECK();
  return 0;
}
