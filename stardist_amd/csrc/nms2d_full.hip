// nms2d_full.hip -- exact-join path of the 2D pair intersection + the pair-level probe.
//
// The fast pair kernel (nms2d.hip, k_pairs) is exact whenever Clipper records no joins for the
// pair.  Pairs with joins are re-evaluated here with the full ring bookkeeping
// (clip_sweep_full.h), which also restates JoinCommonEdges.
#include <type_traits>

#include "common.h"
#include "clip_sweep.h"
#include "clip_sweep_full.h"
#include "clip_beam.h"
#include <stdlib.h>
#include "../../include/stardist_hip.h"

namespace {
using sdclip::i64;

// One thread per pair; persistent grid, the pair count is read on the device.  The result is applied directly
// (stardist2d.cpp:579-585): no intermediate area array, no host round trip.
enum { ST_SUPPRESSED_ = 2 };
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
__global__ void __launch_bounds__(64) k_full_pairs(const int2* __restrict__ pairs, const unsigned int* __restrict__ idx, const unsigned int* __restrict__ nPtr, unsigned int cap, int R,
                                                   const int* __restrict__ vx, const int* __restrict__ vy,
                                                   const float* __restrict__ area, float thr, unsigned char* __restrict__ state, unsigned char* __restrict__ supp,
                                                   unsigned int* errCount) {
  unsigned int n = *nPtr; if (n > cap) n = cap;
  // pair t -> lane (t / gridDim.x) of workgroup (t % gridDim.x): a short list is spread over the workgroups' FIRST lanes, one sweep per wave
  // -- the sweeps of different pairs share no control flow, and 62 of them packed into two waves take as long as their sum
  for (unsigned int t = threadIdx.x * gridDim.x + blockIdx.x; t < n; t += gridDim.x * blockDim.x) {
    // the call fails as soon as ONE pair exceeds the fixed capacities (sd_nms2d reports it): nobody needs the other pairs then, and with
    // 128 / 256 vertices per polygon and every pair overflowing (rays that vary by +-90 %) they would take minutes
    if (MAXV > 64 && __hip_atomic_load(errCount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    const unsigned int p = idx ? idx[t] : t;
    const int2 ij = pairs[p];
    sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ> sw;
    sw.reset_state();
    sw.add_path(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, R, sdclip::kClip, 0);
    sw.add_path(vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, sdclip::kSubject, MAXV);
    const i64 twice = sw.execute();
    if (sw.status & ~sdclip::ST_FAIL) atomicAdd(errCount, 1u);
    const float area_inter = 0.5f * (float)twice;
    const float overlap = (float)((double)area_inter / fmin((double)area[ij.x] + 1.e-10, (double)area[ij.y] + 1.e-10));
    if (overlap > thr) { if (supp) supp[p] = 1; else state[ij.y] = ST_SUPPRESSED_; }
  }
}

// latency variant of the general kernel: the sweep's core arrays in LDS, the point rings stay private
enum { LDSF_T = 32 };
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
__global__ void __launch_bounds__(LDSF_T) k_full_pairs_lds(const int2* __restrict__ pairs, const unsigned int* __restrict__ idx, const unsigned int* __restrict__ nPtr, unsigned int cap, int R,
                                                           const int* __restrict__ vx, const int* __restrict__ vy,
                                                           const float* __restrict__ area, float thr, unsigned char* __restrict__ state, unsigned char* __restrict__ supp,
                                                           unsigned int* errCount) {
  typedef sdclip::LdsStorage<LDSF_T> LP;
  unsigned int n = *nPtr; if (n > cap) n = cap;
  // (pair t -> lane t / gridDim.x of workgroup t % gridDim.x: see k_full_pairs -- the tail batch's last launch holds a few dozen pairs)
  for (unsigned int t = threadIdx.x * gridDim.x + blockIdx.x; t < n; t += gridDim.x * LDSF_T) {
    const unsigned int p = idx ? idx[t] : t;
    const int2 ij = pairs[p];
    sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ, LP> sw;
    sw.reset_state();
    sw.add_path(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, R, sdclip::kClip, 0);
    sw.add_path(vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, sdclip::kSubject, MAXV);
    const i64 twice = sw.execute();
    if (sw.status & ~sdclip::ST_FAIL) atomicAdd(errCount, 1u);
    const float area_inter = 0.5f * (float)twice;
    const float overlap = (float)((double)area_inter / fmin((double)area[ij.x] + 1.e-10, (double)area[ij.y] + 1.e-10));
    if (overlap > thr) { if (supp) supp[p] = 1; else state[ij.y] = ST_SUPPRESSED_; }
  }
}
template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ>
size_t lds_full_bytes() {
  typedef sdclip::LdsStorage<LDSF_T> LP;
  return (size_t)sdclip::SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ, LP>::lds_bytes() + 64;
}

// probe: explicit vertex arrays per pair, evaluated the way the NMS does: prepared polygons + bound-slot sweep
// (clip_beam.h, tier capacities K/BIL/BREC); the general sweep when that flags a capacity or records joins.
// flags: sweep status | 256 (general path used) | 512 (bound-slot result used)
template <int MAXV, int K, int BIL, int BREC, int S, int MAXIL, int MAXREC, int MAXPT, int MAXJ, bool REL16 = false>
__global__ void __launch_bounds__(S) k_probe(const int* __restrict__ xa, const int* __restrict__ ya,
                                             const int* __restrict__ xb, const int* __restrict__ yb, int n, int R,
                                             sdclip::PolyPrep<MAXV>* __restrict__ prepbuf,
                                             i64* __restrict__ twice, int* __restrict__ flags, int nofull) {
  const int p = blockIdx.x * S + threadIdx.x;
  if (p >= n) return;
  typedef sdclip::LdsStorage<S> LP;
  sdclip::PolyPrep<MAXV>* pa = prepbuf + 2 * (size_t)p;
  sdclip::PolyPrep<MAXV>* pb = pa + 1;
  {
    sdclip::PrepWork<LP, MAXV> w;
    w.prepare(xa + (size_t)p * R, ya + (size_t)p * R, R, pa);
    w.prepare(xb + (size_t)p * R, yb + (size_t)p * R, R, pb);
  }
  i64 t;
  int fl;
  bool need_full;
  {
    typedef typename std::conditional<REL16, sdclip::LdsStorage16<S>, LP>::type BP;      // REL16: the NMS's tier-1 form (16-bit coordinates)
    sdclip::Beam<MAXV, K, BIL, BREC, BP> bm;
    bm.reset_state(pa, pb);
    t = bm.execute();
    fl = bm.status | 512;
    need_full = (bm.status & ~sdclip::ST_FAIL) != 0 || bm.n_joins > 0;
  }
  if (need_full && !nofull) {
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
template <int MAXV, int K, int BIL, int BREC, int S, int MAXIL, int MAXREC, int MAXPT, int MAXJ, bool REL16 = false>
int launch_probe(const int* xa, const int* ya, const int* xb, const int* yb, int n, int R, i64* out, int* flags, hipStream_t s) {
  typedef sdclip::LdsStorage<S> LP;
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  sdclip::PolyPrep<MAXV>* prepbuf = (sdclip::PolyPrep<MAXV>*)A.take((size_t)n * 2 * sizeof(sdclip::PolyPrep<MAXV>));
  if (!prepbuf) return -1;
  size_t lds = sdclip::Beam<MAXV, K, BIL, BREC, LP>::lds_bytes();
  const size_t lds2 = sdclip::PrepWork<LP, MAXV>::lds_bytes();
  if (lds2 > lds) lds = lds2;
  lds += 64;
  hipLaunchKernelGGL((k_probe<MAXV, K, BIL, BREC, S, MAXIL, MAXREC, MAXPT, MAXJ, REL16>), dim3((n + S - 1) / S), dim3(S), lds, s, xa, ya, xb, yb, n, R, prepbuf, out, flags,
                     sd::option(sd::OPT_PROBE_NO_GENERAL) ? 1 : 0);
  SD_LAUNCH_CHECK();
  return 0;
}

// test probe: the prepared-polygon records of n polygons
template <int MAXV, int S>
__global__ void __launch_bounds__(S) k_prepare_probe(const int* __restrict__ x, const int* __restrict__ y, int n, int R, sdclip::PolyPrep<MAXV>* out) {
  const int i = blockIdx.x * S + threadIdx.x;
  if (i >= n) return;
  sdclip::PrepWork<sdclip::LdsStorage<S>, MAXV> w;
  w.prepare(x + (size_t)i * R, y + (size_t)i * R, R, out + i);
}
template <int MAXV, int S>
int launch_prepare_probe(const int* x, const int* y, int n, int R, void* out, hipStream_t s) {
  const size_t lds = sdclip::PrepWork<sdclip::LdsStorage<S>, MAXV>::lds_bytes() + 64;
  hipLaunchKernelGGL((k_prepare_probe<MAXV, S>), dim3((n + S - 1) / S), dim3(S), lds, s, x, y, n, R, (sdclip::PolyPrep<MAXV>*)out);
  SD_LAUNCH_CHECK();
  return 0;
}
}  // namespace

namespace sd {
// general (exact-join) path over a device-side queue of pair indices: evaluates pairs[idx[0 .. min(*d_n, cap))] and either
// applies the suppression to state[j] or records it per pair in supp[] (tail batch)
int clip_full_pairs(const int2* d_pairs, const unsigned int* d_idx, const unsigned int* d_n, unsigned int cap, int R, const int* d_vx, const int* d_vy,
                    const float* d_area, float thr, unsigned char* d_state, unsigned char* d_supp, unsigned int* d_errCount, hipStream_t s) {
  if (R <= 32) {
    static const size_t ldsBytes = lds_full_bytes<32, 64, 32, 192, 64>();
    static bool attr_set = false;
    if (!attr_set) {
      SD_CHECK(hipFuncSetAttribute((const void*)k_full_pairs_lds<32, 64, 32, 192, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
      attr_set = true;
    }
    hipLaunchKernelGGL((k_full_pairs_lds<32, 64, 32, 192, 64>), dim3(1024), dim3(LDSF_T), ldsBytes, s, d_pairs, d_idx, d_n, cap, R, d_vx, d_vy, d_area, thr, d_state, d_supp, d_errCount);
  } else if (R <= 64) hipLaunchKernelGGL((k_full_pairs<64, 96, 48, 384, 96>), dim3(2048), dim3(64), 0, s, d_pairs, d_idx, d_n, cap, R, d_vx, d_vy, d_area, thr, d_state, d_supp, d_errCount);
  // (the private sweep state of these two is 32 / 62 KB per LANE: 2 / 4 MB of scratch per wave.  With 2048 workgroups the 256-vertex form
  // faulted -- "memory aperture violation" -- on its first NMS-level run in round 6; the grid is sized so that the launch's scratch
  // stays near 1 GB.  Pairs are dealt round-robin over whatever grid there is.)
  else if (R <= 128) hipLaunchKernelGGL((k_full_pairs<128, 128, 64, 768, 128>), dim3(512), dim3(64), 0, s, d_pairs, d_idx, d_n, cap, R, d_vx, d_vy, d_area, thr, d_state, d_supp, d_errCount);
  else hipLaunchKernelGGL((k_full_pairs<256, 192, 96, 1536, 192>), dim3(256), dim3(64), 0, s, d_pairs, d_idx, d_n, cap, R, d_vx, d_vy, d_area, thr, d_state, d_supp, d_errCount);
  SD_LAUNCH_CHECK();
  return 0;
}
}  // namespace sd

extern "C" int sd_prepare_polys_device(const int32_t* d_x, const int32_t* d_y, int n_polys, int n_verts, void* d_out, int64_t out_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int R = n_verts;
  if (n_polys <= 0) return 0;
  if (R < 1 || R > 256) { sd::set_error("sd_prepare_polys: n_verts=%d unsupported (1..256)", R); return -1; }
  const size_t rec = R <= 32 ? sizeof(sdclip::PolyPrep<32>) : R <= 64 ? sizeof(sdclip::PolyPrep<64>) : R <= 128 ? sizeof(sdclip::PolyPrep<128>) : sizeof(sdclip::PolyPrep<256>);
  if ((int64_t)(rec * (size_t)n_polys) > out_bytes) { sd::set_error("sd_prepare_polys: output buffer too small (%zu bytes per polygon)", rec); return -1; }
  if (R <= 32) return launch_prepare_probe<32, 64>(d_x, d_y, n_polys, R, d_out, s);
  if (R <= 64) return launch_prepare_probe<64, 64>(d_x, d_y, n_polys, R, d_out, s);
  if (R <= 128) return launch_prepare_probe<128, 32>(d_x, d_y, n_polys, R, d_out, s);
  return launch_prepare_probe<256, 16>(d_x, d_y, n_polys, R, d_out, s);
}

extern "C" int sd_clip_pairs_device(const int32_t* d_xa, const int32_t* d_ya, const int32_t* d_xb, const int32_t* d_yb,
                                    int n_pairs, int n_verts, int64_t* d_out_twice_area, int32_t* d_out_flags,
                                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n_pairs <= 0) return 0;
  const int R = n_verts;
  if (R < 1 || R > 256) { sd::set_error("sd_clip_pairs: n_verts=%d unsupported (1..256)", R); return -1; }
  i64* out = (i64*)d_out_twice_area;
  const int tier = sd::option(sd::OPT_PROBE_TIER);   // 1: K = 8 capacities, 16-bit coordinates (the NMS's first tier), 3: the same with 32-bit coordinates, 2: K = 15
  if (R <= 32 && tier == 1) return launch_probe<32, 8, 6, 4, 64, 64, 32, 192, 64, true>(d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags, s);
  if (R <= 32 && tier == 3) return launch_probe<32, 8, 6, 4, 64, 64, 32, 192, 64>(d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags, s);
  if (R <= 32) return launch_probe<32, 15, 16, 8, 32, 64, 32, 192, 64>(d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags, s);
  if (R <= 64) return launch_probe<64, 15, 16, 8, 32, 96, 48, 384, 96>(d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags, s);
  if (R <= 128) return launch_probe<128, 15, 16, 8, 32, 128, 64, 768, 128>(d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags, s);
  return launch_probe<256, 15, 16, 8, 16, 192, 96, 1536, 192>(d_xa, d_ya, d_xb, d_yb, n_pairs, R, out, d_out_flags, s);
}
