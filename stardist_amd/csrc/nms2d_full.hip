// nms2d_full.hip -- exact-join path of the 2D pair intersection + the pair-level probe.
//
// The fast pair kernel (nms2d.hip, k_pairs) is exact whenever Clipper records no joins for the
// pair.  Pairs with joins are re-evaluated here with the full ring bookkeeping
// (clip_sweep_full.h), which also restates JoinCommonEdges.
#include "common.h"
#include "clip_sweep.h"
#include "clip_sweep_full.h"
#include "../../include/stardist_hip.h"

namespace {
using sdclip::i64;

template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
__global__ void __launch_bounds__(64) k_full_pairs(const int2* __restrict__ pairs, unsigned int n, int R,
                                                   const int* __restrict__ vx, const int* __restrict__ vy,
                                                   i64* __restrict__ twice, int* __restrict__ flags) {
  const unsigned int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int2 ij = pairs[p];
  sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ> sw;
  sw.reset_state();
  sw.add_path(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, R, sdclip::kClip, 0);
  sw.add_path(vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, sdclip::kSubject, MAXV);
  twice[p] = sw.execute();
  flags[p] = sw.status;
}

// latency variant of the exact-join kernel: the sweep's core arrays in LDS, the point rings stay private
enum { LDSF_T = 32 };
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
__global__ void __launch_bounds__(LDSF_T) k_full_pairs_lds(const int2* __restrict__ pairs, unsigned int n, int R,
                                                           const int* __restrict__ vx, const int* __restrict__ vy,
                                                           i64* __restrict__ twice, int* __restrict__ flags) {
  typedef sdclip::LdsStorage<LDSF_T> LP;
  for (unsigned int p = blockIdx.x * LDSF_T + threadIdx.x; p < n; p += gridDim.x * LDSF_T) {
    const int2 ij = pairs[p];
    sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ, LP> sw;
    sw.reset_state();
    sw.add_path(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, R, sdclip::kClip, 0);
    sw.add_path(vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, sdclip::kSubject, MAXV);
    twice[p] = sw.execute();
    flags[p] = sw.status;
  }
}
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
size_t lds_full_bytes() {
  typedef sdclip::LdsStorage<LDSF_T> LP;
  return (size_t)sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ, LP>::lds_bytes() + 64;
}

// probe: explicit vertex arrays per pair; fast sweep first, full sweep when joins were recorded
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
__global__ void __launch_bounds__(64) k_probe(const int* __restrict__ xa, const int* __restrict__ ya,
                                              const int* __restrict__ xb, const int* __restrict__ yb, int n, int R,
                                              i64* __restrict__ twice, int* __restrict__ flags) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  i64 t;
  int fl;
  bool need_full;
  {
    sdclip::Sweep<MAXV, MAXIL, MAXREC> sw;
    sw.reset_state();
    sw.add_path(xa + (size_t)p * R, ya + (size_t)p * R, R, sdclip::kClip, 0);
    sw.add_path(xb + (size_t)p * R, yb + (size_t)p * R, R, sdclip::kSubject, MAXV);
    t = sw.execute();
    fl = sw.status;
    need_full = sw.n_joins > 0;
  }
  if (need_full) {
    sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ> sf;
    sf.reset_state();
    sf.add_path(xa + (size_t)p * R, ya + (size_t)p * R, R, sdclip::kClip, 0);
    sf.add_path(xb + (size_t)p * R, yb + (size_t)p * R, R, sdclip::kSubject, MAXV);
    t = sf.execute();
    fl = sf.status | 256;
  }
  twice[p] = t;
  flags[p] = fl;
}
}  // namespace

namespace sd {
int clip_full_pairs(const int2* d_pairs, unsigned int n, int R, const int* d_vx, const int* d_vy, i64* d_twice,
                    int* d_flags, hipStream_t s) {
  if (n == 0) return 0;
  const unsigned int blocks = (n + 63) / 64;
  if (R <= 32 && n <= 8192u) {   // one resident pass: launch time = one pair's latency (~5x lower than scratch)
    static const size_t ldsBytes = lds_full_bytes<32, 64, 32, 192, 64>();
    static bool attr_set = false;
    if (!attr_set) {
      SD_CHECK(hipFuncSetAttribute((const void*)k_full_pairs_lds<32, 64, 32, 192, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
      attr_set = true;
    }
    const unsigned int bl = (n + LDSF_T - 1) / LDSF_T;
    hipLaunchKernelGGL((k_full_pairs_lds<32, 64, 32, 192, 64>), dim3(bl < 2048u ? bl : 2048u), dim3(LDSF_T), ldsBytes, s, d_pairs, n, R, d_vx, d_vy, d_twice, d_flags);
  } else
  if (R <= 32) hipLaunchKernelGGL((k_full_pairs<32, 64, 32, 192, 64>), dim3(blocks), dim3(64), 0, s, d_pairs, n, R, d_vx, d_vy, d_twice, d_flags);
  else if (R <= 64) hipLaunchKernelGGL((k_full_pairs<64, 96, 48, 384, 96>), dim3(blocks), dim3(64), 0, s, d_pairs, n, R, d_vx, d_vy, d_twice, d_flags);
  else if (R <= 128) hipLaunchKernelGGL((k_full_pairs<128, 128, 64, 768, 128>), dim3(blocks), dim3(64), 0, s, d_pairs, n, R, d_vx, d_vy, d_twice, d_flags);
  else hipLaunchKernelGGL((k_full_pairs<256, 192, 96, 1536, 192>), dim3(blocks), dim3(64), 0, s, d_pairs, n, R, d_vx, d_vy, d_twice, d_flags);
  SD_LAUNCH_CHECK();
  return 0;
}
}  // namespace sd

extern "C" int sd_clip_pairs_device(const int32_t* d_xa, const int32_t* d_ya, const int32_t* d_xb, const int32_t* d_yb,
                                    int n_pairs, int n_verts, int64_t* d_out_twice_area, int32_t* d_out_flags,
                                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_pairs <= 0) return 0;
  const int R = n_verts;
  if (R < 1 || R > 256) { sd::set_error("sd_clip_pairs: n_verts=%d unsupported (1..256)", R); return -1; }
  const int blocks = (n_pairs + 63) / 64;
  i64* out = (i64*)d_out_twice_area;
  if (R <= 32) hipLaunchKernelGGL((k_probe<32, 64, 32, 192, 64>), dim3(blocks), dim3(64), 0, s, d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags);
  else if (R <= 64) hipLaunchKernelGGL((k_probe<64, 96, 48, 384, 96>), dim3(blocks), dim3(64), 0, s, d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags);
  else if (R <= 128) hipLaunchKernelGGL((k_probe<128, 128, 64, 768, 128>), dim3(blocks), dim3(64), 0, s, d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags);
  else hipLaunchKernelGGL((k_probe<256, 192, 96, 1536, 192>), dim3(blocks), dim3(64), 0, s, d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags);
  SD_LAUNCH_CHECK();
  return 0;
}
