// select.hip -- candidate selection on device: threshold + border mask + ordered compaction.
//
// Replaces the numpy glue between the network heads and the NMS natives:
//   prob > prob_thresh with a b-pixel border excluded   (stardist/nms.py:6-17)
//   points = np.where(mask), dist = max(1e-3, dist)[mask] (stardist/models/base.py:553-610)
// so the dense (H, W, n_rays) distance map never leaves HBM.  Output order == np.where order
// (C order of the flat index), obtained with a two-pass block-count / scan / ordered-write
// compaction (wave ballot + popcount for the in-wave rank).
#include "common.h"
#include "../../include/stardist_hip.h"
#include <hipcub/hipcub.hpp>

namespace {

struct SelP { int ndim; int shape[3]; int lo[3], hi[3]; };

__device__ __forceinline__ bool selected(const SelP p, long long idx, float pr, float thr) {
  if (!(pr > thr)) return false;
  long long r = idx;
  for (int d = p.ndim - 1; d >= 0; --d) {
    const int c = (int)(r % p.shape[d]); r /= p.shape[d];
    if (c < p.lo[d] || c >= p.shape[d] - p.hi[d]) return false;
  }
  return true;
}

enum { ITEMS = 4, BLOCK = 256, TILE = ITEMS * BLOCK };

__global__ void __launch_bounds__(BLOCK) k_count(const float* __restrict__ prob, long long n, SelP p, float thr, int* __restrict__ blockCount) {
  __shared__ int wsum[4];
  const long long base = (long long)blockIdx.x * TILE;
  int c = 0;
  for (int r = 0; r < ITEMS; ++r) {
    const long long idx = base + r * BLOCK + threadIdx.x;
    const bool s = idx < n && selected(p, idx, prob[idx], thr);
    c += __popcll(__ballot(s));
  }
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blockCount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void __launch_bounds__(BLOCK) k_write(const float* __restrict__ prob, const float* __restrict__ dist, long long n, SelP p,
                                                 float thr, int R, const int* __restrict__ blockStart, int cap,
                                                 float* __restrict__ oprob, float* __restrict__ odist, int* __restrict__ opts,
                                                 int* __restrict__ count, int nBlocks) {
  __shared__ int wcnt[ITEMS][4];
  const long long base = (long long)blockIdx.x * TILE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long masks[ITEMS];
  for (int r = 0; r < ITEMS; ++r) {
    const long long idx = base + r * BLOCK + threadIdx.x;
    const bool s = idx < n && selected(p, idx, prob[idx], thr);
    masks[r] = __ballot(s);
    if (lane == 0) wcnt[r][wave] = __popcll(masks[r]);
  }
  __syncthreads();
  if (blockIdx.x == nBlocks - 1 && threadIdx.x == 0) {
    int tot = 0;
    for (int r = 0; r < ITEMS; ++r) for (int w = 0; w < 4; ++w) tot += wcnt[r][w];
    *count = blockStart[blockIdx.x] + tot;
  }
  int off = blockStart[blockIdx.x];
  for (int r = 0; r < ITEMS; ++r) {
    for (int w = 0; w < 4; ++w) {
      if (w == wave && !dist) {
        // no distance rows to copy (the sparse head evaluates them later, on the selected rows): every selected lane writes its own entry
        const unsigned long long m = masks[r];
        if ((m >> lane) & 1ull) {
          const int o = off + __popcll(m & ((1ull << lane) - 1));
          if (o < cap) {
            const long long idx = base + r * BLOCK + threadIdx.x;
            oprob[o] = prob[idx];
            long long rem = idx;
            for (int d = p.ndim - 1; d >= 0; --d) { opts[(size_t)o * p.ndim + d] = (int)(rem % p.shape[d]); rem /= p.shape[d]; }
          }
        }
      } else if (w == wave) {
        // this wave's selected pixels of row r start at `off`; copy them cooperatively
        unsigned long long m = masks[r];
        int rank = 0;
        while (m) {
          const int src_lane = __ffsll((long long)m) - 1;
          m &= m - 1;
          const long long idx = base + r * BLOCK + wave * 64 + src_lane;
          const int o = off + rank;
          ++rank;
          if (o < cap) {
            if (dist) for (int k = lane; k < R; k += 64) odist[(size_t)o * R + k] = fmaxf(1e-3f, dist[(size_t)idx * R + k]);
            if (lane == 0) {
              oprob[o] = prob[idx];
              long long rem = idx;
              for (int d = p.ndim - 1; d >= 0; --d) { opts[(size_t)o * p.ndim + d] = (int)(rem % p.shape[d]); rem /= p.shape[d]; }
            }
          }
        }
      }
      off += wcnt[r][w];
    }
  }
}

// candidates in SCORE order: order[k] = index (into the selection's np.where-ordered list) of the k-th best candidate.  One pass writes
// what the stages behind the sort read -- the row of the feature matrix (distance head on the candidate rows, sd_head_rows_device), the
// pixel coordinates as float32 (what the NMS natives take, nms.py:217-218) and as int64 (what the result dict returns, base.py:606) --
// instead of a dozen framework index / multiply / cast launches on (n, ndim) arrays.
struct RowP { int ndim; int full[3], origin[3], grid[3]; };
__global__ void __launch_bounds__(256) k_sorted_rows(const int* __restrict__ pts, const long long* __restrict__ order, int n, RowP p,
                                                     long long* __restrict__ rows, float* __restrict__ pf, long long* __restrict__ pi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const long long src = order ? order[k] : k;
  long long row = 0;
  for (int d = 0; d < p.ndim; ++d) {
    const int c = pts[src * p.ndim + d];
    row = row * p.full[d] + (c + p.origin[d]);
    const long long pix = (long long)c * p.grid[d];
    if (pf) pf[(size_t)k * p.ndim + d] = (float)pix;
    if (pi) pi[(size_t)k * p.ndim + d] = pix;
  }
  if (rows) rows[k] = row;
}

__global__ void k_sort_iota(unsigned int* v, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = (unsigned int)i; }
__global__ void k_sort_reverse(const float* __restrict__ keys, const unsigned int* __restrict__ vals, int n, float* __restrict__ sorted,
                               long long* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sorted[i] = keys[n - 1 - i];
  order[i] = (long long)vals[n - 1 - i];
}

}  // namespace

// np.argsort(scores)[::-1] as the NMS glue states it (stardist/nms.py:114,167; nms.py _argsort_desc): a STABLE ascending sort, reversed --
// best score first, equal scores in descending order of their position.  One radix sort of (score, position) pairs (scores are
// probabilities: finite, the radix order of their bit patterns is their numeric order) and one reversing write, instead of the
// framework's merge sort (a block sort, nine merge passes of two launches each, two transforms, two flips at 4 x 10^5 candidates).
extern "C" int sd_sort_scores_desc_device(const float* d_scores, int n, float* d_sorted, int64_t* d_order, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return 0;
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  float* keys = A.take_n<float>(n);
  unsigned int* vin = A.take_n<unsigned int>(n);
  unsigned int* vout = A.take_n<unsigned int>(n);
  size_t tmpBytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, d_scores, keys, vin, vout, n, 0, 32, s);
  void* tmp = A.take(tmpBytes + 256);
  if (!keys || !vin || !vout || !tmp) return -1;
  hipLaunchKernelGGL(k_sort_iota, dim3(sd::div_up(n, 256)), dim3(256), 0, s, vin, n);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, tmpBytes, d_scores, keys, vin, vout, n, 0, 32, s));
  hipLaunchKernelGGL(k_sort_reverse, dim3(sd::div_up(n, 256)), dim3(256), 0, s, (const float*)keys, (const unsigned int*)vout, n, d_sorted, (long long*)d_order);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_sorted_rows_device(const int32_t* d_points, const int64_t* d_order, int n, int ndim, const int* full_shape, const int* origin,
                                     const int* grid, int64_t* d_rows, float* d_points_f32, int64_t* d_points_i64, void* stream) {
  if (n <= 0) return 0;
  if (ndim < 1 || ndim > 3) { sd::set_error("sd_sorted_rows: ndim must be 1..3"); return -1; }
  RowP p; p.ndim = ndim;
  for (int d = 0; d < 3; ++d) { p.full[d] = d < ndim ? full_shape[d] : 1; p.origin[d] = d < ndim && origin ? origin[d] : 0; p.grid[d] = d < ndim && grid ? grid[d] : 1; }
  hipLaunchKernelGGL(k_sorted_rows, dim3(sd::div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, (const int*)d_points, (const long long*)d_order, n, p,
                     (long long*)d_rows, d_points_f32, (long long*)d_points_i64);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_select_candidates_device(const float* d_prob, const float* d_dist, int ndim, const int* shape, const int* b,
                                           int n_rays, float thresh, int cap, float* d_out_prob, float* d_out_dist,
                                           int32_t* d_out_points, int32_t* d_count, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (ndim < 1 || ndim > 3) { sd::set_error("sd_select_candidates: ndim must be 1..3"); return -1; }
  SelP p; p.ndim = ndim;
  long long n = 1;
  for (int d = 0; d < 3; ++d) {
    p.shape[d] = d < ndim ? shape[d] : 1;
    p.lo[d] = d < ndim && b ? b[2 * d] : 0;
    p.hi[d] = d < ndim && b ? b[2 * d + 1] : 0;
    n *= p.shape[d];
  }
  SD_CHECK(hipMemsetAsync(d_count, 0, sizeof(int), s));
  if (n <= 0) return 0;
  const int nBlocks = (int)((n + TILE - 1) / TILE);
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  int* blockCount = A.take_n<int>(nBlocks + 1);
  int* blockStart = A.take_n<int>(nBlocks + 1);
  size_t tmpBytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, blockCount, blockStart, nBlocks, s);
  void* tmp = A.take(tmpBytes + 256);
  if (!blockCount || !blockStart || !tmp) return -1;
  hipLaunchKernelGGL(k_count, dim3(nBlocks), dim3(BLOCK), 0, s, d_prob, n, p, thresh, blockCount);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp, tmpBytes, blockCount, blockStart, nBlocks, s));
  hipLaunchKernelGGL(k_write, dim3(nBlocks), dim3(BLOCK), 0, s, d_prob, d_dist, n, p, thresh, n_rays, blockStart, cap,
                     d_out_prob, d_out_dist, d_out_points, d_count, nBlocks);
  SD_LAUNCH_CHECK();
  return 0;
}
