// nms2d.hip -- greedy non-maximum suppression of star-convex polygons on gfx950.
//
// Replaces stardist.lib.stardist2d.c_non_max_suppression_inds
// (stardist/lib/stardist2d.cpp:390-615).  Same inputs (candidates sorted by score
// descending), same survivor set, different schedule:
//
//   K1 build      one wave per candidate: integer vertices (float math + truncation exactly as
//                 stardist2d.cpp:447-471, sin/cos table computed by the HOST libm so the device
//                 never evaluates sinf/cosf), int bbox, outer radius, float area (:128-138).
//   K2 bin        counting sort of candidates into a uniform grid (replaces the nanoflann
//                 kd-tree, stardist2d.cpp:486-513; result-neutral).
//   K3 neighbours CSR lists of candidates whose bounding boxes can touch (symmetric superset
//                 of every pair the reference would test).
//   greedy rounds A: a candidate becomes a survivor once every higher-scored neighbour is
//                    decided and none suppressed it;   (wave ballot over the neighbour list)
//                 B: each new survivor emits the (i, j) pairs the reference would evaluate
//                    for it (stardist2d.cpp:566-577 predicate, exact) via ballot/prefix-sum
//                    compaction into a pair queue;
//                 C: one thread per pair runs the integer scan-beam intersection
//                    (clip_sweep.h) and applies  overlap > threshold  (:579-585).
//   The fixed point of the rounds is the reference's sequential greedy result: j is
//   suppressed iff some survivor i < j has overlap(i,j) > threshold, and a candidate is only
//   promoted to survivor after all of its possible suppressors are final.
#include "common.h"
#include "clip_sweep.h"
#include "../../include/stardist_hip.h"
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdlib.h>
#include <vector>

namespace {

using sdclip::i64;

enum { ST_UNDECIDED = 0, ST_KEPT = 1, ST_SUPPRESSED = 2 };

__device__ __forceinline__ float wave_min(float v) { for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ i64 wave_sum(i64 v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }

// gstats layout (ints): 0 max radius bits, 1 min y, 2 max y, 3 min x, 4 max x
__global__ void __launch_bounds__(256) k_build(const float* __restrict__ dist, const float* __restrict__ pts,
                                               const float2* __restrict__ sincos, int N, int R,
                                               int* __restrict__ vx, int* __restrict__ vy, int4* __restrict__ bbox,
                                               float* __restrict__ radius, float* __restrict__ area, int* gstats) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + wave;
  if (i >= N) return;
  int* sx = lds + wave * 2 * R;
  int* sy = sx + R;
  const float py = pts[2 * i], px = pts[2 * i + 1];
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, rmax = 0.f;
  for (int k = lane; k < R; k += 64) {
    const float d = dist[(size_t)i * R + k];
    const float2 sc = sincos[k];
    const float y = py + d * sc.x;   // stardist2d.cpp:454 (compiled with -ffp-contract=off)
    const float x = px + d * sc.y;   // stardist2d.cpp:455
    xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
    ymin = fminf(ymin, y); ymax = fmaxf(ymax, y);
    const int X = (int)(long long)x, Y = (int)(long long)y;   // IntPoint(cInt(x), cInt(y)) :471
    sx[k] = X; sy[k] = Y;
    vx[(size_t)i * R + k] = X; vy[(size_t)i * R + k] = Y;
    rmax = fmaxf(rmax, d);
  }
  xmin = wave_min(xmin); xmax = wave_max(xmax); ymin = wave_min(ymin); ymax = wave_max(ymax);
  rmax = wave_max(rmax);
  __builtin_amdgcn_wave_barrier();
  // area_from_path :128-138: float accumulation of int64 cross products in path order; this
  // equals the exact integer sum whenever sum|term| < 2^24, else fall back to the serial order.
  i64 s = 0, sa = 0;
  for (int k = lane; k < R; k += 64) {
    const int kn = (k + 1 == R) ? 0 : k + 1;
    const i64 c = (i64)sx[k] * sy[kn] - (i64)sy[k] * sx[kn];
    s += c; sa += (c < 0 ? -c : c);
  }
  s = wave_sum(s); sa = wave_sum(sa);
  if (lane == 0) {
    float a;
    if (sa < (1ll << 24)) a = (float)s;
    else {
      a = 0.f;
      for (int k = 0; k < R; ++k) {
        const int kn = (k + 1 == R) ? 0 : k + 1;
        a += (float)((i64)sx[k] * sy[kn] - (i64)sy[k] * sx[kn]);
      }
    }
    area[i] = (float)(0.5 * (double)fabsf(a));
    radius[i] = rmax;
    bbox[i] = make_int4((int)xmin, (int)xmax, (int)ymin, (int)ymax);   // bbox_intersect takes ints :142-148
    // contended global atomics only when the running extremum actually improves
    const int rb = __float_as_int(rmax);
    const int iy = (int)floorf(py), ix = (int)floorf(px);
    volatile int* gs = gstats;
    if (rb > gs[0]) atomicMax(&gstats[0], rb);
    if (iy < gs[1]) atomicMin(&gstats[1], iy);
    if (iy > gs[2]) atomicMax(&gstats[2], iy);
    if (ix < gs[3]) atomicMin(&gstats[3], ix);
    if (ix > gs[4]) atomicMax(&gstats[4], ix);
  }
}

struct GridP { float y0, x0, inv_cs; int ny, nx; };

__device__ __forceinline__ int cell_of(const GridP g, float py, float px, int& cy, int& cx) {
  cy = (int)((py - g.y0) * g.inv_cs); cx = (int)((px - g.x0) * g.inv_cs);
  cy = min(max(cy, 0), g.ny - 1); cx = min(max(cx, 0), g.nx - 1);
  return cy * g.nx + cx;
}

__global__ void k_cell_count(const float* __restrict__ pts, int N, GridP g, int* __restrict__ cellCount, int* __restrict__ candCell) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int cy, cx;
  const int c = cell_of(g, pts[2 * i], pts[2 * i + 1], cy, cx);
  candCell[i] = c;
  atomicAdd(&cellCount[c], 1);
}
// candidates re-packed in cell order: the broad phase streams these 32-byte records (coalesced) instead of gathering
// bbox / centre / area of every cell item by candidate index
struct __attribute__((aligned(16))) CellRec { int4 bb; float py, px, area; int j; };
__global__ void k_cell_fill(int N, const int* __restrict__ candCell, const int* __restrict__ cellStart,
                            int* __restrict__ cellFill, const float* __restrict__ pts, const int4* __restrict__ bbox,
                            const float* __restrict__ area, CellRec* __restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int c = candCell[i];
  const int pos = atomicAdd(&cellFill[c], 1);
  CellRec r;
  r.bb = bbox[i]; r.py = pts[2 * i]; r.px = pts[2 * i + 1]; r.area = area[i]; r.j = i;
  rec[cellStart[c] + pos] = r;
}

__device__ __forceinline__ bool bbox_intersect(const int4 a, const int4 b) {   // stardist2d.cpp:142-148
  return (b.x <= a.y && a.x <= b.y && b.z <= a.w && a.z <= b.w);
}

struct Flags { int use_kdtree, use_bbox, thr_nonneg; float thr; float max_dist; };

// symmetric "may interact" predicate used for the dependency lists
__device__ __forceinline__ bool may_interact(const Flags f, const int4 bi, const int4 bj, float pyi, float pxi, float pyj, float pxj,
                                             float ai, float aj) {
  if (f.thr_nonneg) {
    // disjoint integer bboxes => area 0 => overlap 0 <= thr; more generally area_inter <= area(bbox_i ∩ bbox_j), so a pair
    // whose bbox-intersection area cannot exceed thr * min(area) can never suppress (same bound as in k_round_emit)
    if (!bbox_intersect(bi, bj)) return false;
    const double w = (double)(min(bi.y, bj.y) - max(bi.x, bj.x)), hgt = (double)(min(bi.w, bj.w) - max(bi.z, bj.z));
    const float ub = (float)((w * hgt) / fmin((double)ai + 1.e-10, (double)aj + 1.e-10));
    return ub > f.thr;
  }
  bool ok = true;
  if (f.use_bbox) ok = ok && bbox_intersect(bi, bj);
  if (f.use_kdtree) {
    const float dy = pyi - pyj, dxx = pxi - pxj;
    const float rr = 2.f * f.max_dist + 1.f;
    ok = ok && (dy * dy + dxx * dxx < rr * rr);
  }
  return ok;
}

// MODE 0: count neighbours, MODE 1: fill CSR
#define WAIT_NONE (-2)
#define WAIT_SCAN (-1)
template <int MODE>
__global__ void __launch_bounds__(256) k_neighbours(int N, GridP g, Flags f, const CellRec* __restrict__ rec, const int* __restrict__ cellStart,
                                                    int* __restrict__ nbrCount, const i64* __restrict__ nbrStart, int* __restrict__ nbr,
                                                    int* __restrict__ waitOn, int W) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave w handles the w-th candidate IN CELL ORDER, and consecutive workgroups of one XCD (blockIdx % 8) get consecutive
  // cells: the 5x5 cell neighbourhoods of successive waves overlap almost completely and stay in that XCD's L2
  const int per = gridDim.x >> 3;                                   // grid is a multiple of 8
  const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int w = blk * (blockDim.x >> 6) + wave;
  if (w >= N) return;
  const CellRec me = rec[w];
  const int i = me.j;
  int cy, cx;
  cell_of(g, me.py, me.px, cy, cx);
  const int4 bi = me.bb;
  const float pyi = me.py, pxi = me.px, ai = me.area;
  int total = 0;
  int minj = INT32_MAX;                      // MODE 1: best-scored neighbour above i (first wait target of the greedy scan)
  i64 base = MODE ? nbrStart[i] : 0;
  const int x_lo = max(cx - W, 0), x_hi = min(cx + W, g.nx - 1);
  for (int yy = max(cy - W, 0); yy <= min(cy + W, g.ny - 1); ++yy) {
    const int beg = cellStart[yy * g.nx + x_lo], end = cellStart[yy * g.nx + x_hi + 1];
    for (int t = beg; t < end; t += 64) {
      const int idx = t + lane;
      bool hit = false;
      int j = -1;
      if (idx < end) {
        const CellRec r = rec[idx];
        j = r.j;
        if (j != i) hit = may_interact(f, bi, r.bb, pyi, pxi, r.py, r.px, ai, r.area);
      }
      const unsigned long long m = __ballot(hit);
      if (MODE) {
        if (hit) { nbr[base + total + __popcll(m & ((1ull << lane) - 1))] = j; if (j < minj) minj = j; }
      }
      total += __popcll(m);
    }
  }
  if (!MODE && lane == 0) nbrCount[i] = total;
  if (MODE) {
    for (int o = 32; o; o >>= 1) minj = min(minj, __shfl_xor(minj, o));
    if (lane == 0) waitOn[i] = (minj < i) ? minj : WAIT_NONE;
  }
}

// Round kernel A1: thread per undecided candidate, O(1): waitOn[i] is the higher-scored neighbour i was last seen waiting
// for (k_neighbours seeds it with the best-scored one; WAIT_NONE = there is none, WAIT_SCAN = unknown).  Most waits persist
// from round to round, so only the candidates whose wait target has just been decided go to the list scan (A2).
__global__ void __launch_bounds__(256) k_round_triage(const int* __restrict__ U, int nU, const unsigned char* __restrict__ state,
                                                      const int* __restrict__ waitOn, int* __restrict__ Unext, int* __restrict__ K,
                                                      int* __restrict__ S, int* counters /*0:nUnext 1:nK 2:nS*/) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int kind = 0, i = -1;                       // 0 drop, 1 still waiting, 2 becomes a survivor, 3 needs the list scan
  if (t < nU) {
    i = U[t];
    if (state[i] != ST_SUPPRESSED) {
      const int wo = waitOn[i];
      if (wo == WAIT_NONE) kind = 2;
      else if (wo >= 0 && state[wo] == ST_UNDECIDED) kind = 1;
      else kind = 3;
    }
  }
#pragma unroll
  for (int q = 1; q <= 3; ++q) {
    const unsigned long long m = __ballot(kind == q);
    if (!m) continue;
    int base = 0;
    if (lane == 0) base = atomicAdd(&counters[q - 1], __popcll(m));
    base = __shfl(base, 0);
    if (kind == q) (q == 1 ? Unext : (q == 2 ? K : S))[base + __popcll(m & ((1ull << lane) - 1))] = i;
  }
}

// Round kernel A2: wave per candidate of the scan list (persistent grid; the list length is read on the device).
__global__ void __launch_bounds__(256) k_round_scan(const int* __restrict__ S, const unsigned char* __restrict__ state,
                                                    const i64* __restrict__ nbrStart, const int* __restrict__ nbr,
                                                    int* __restrict__ waitOn, int* __restrict__ Unext, int* __restrict__ K,
                                                    int* counters /*0:nUnext 1:nK 2:nS*/) {
  const int lane = threadIdx.x & 63;
  const int nS = counters[2];
  const int nWaves = gridDim.x * (blockDim.x >> 6);
  for (int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < nS; w += nWaves) {
    const int i = S[w];
    const i64 beg = nbrStart[i], end = nbrStart[i + 1];
    int found = -1;
    for (i64 t = beg; t < end && found < 0; t += 64) {
      const i64 idx = t + lane;
      int j = -1;
      if (idx < end) { j = nbr[idx]; if (!(j < i && state[j] == ST_UNDECIDED)) j = -1; }
      const unsigned long long m = __ballot(j >= 0);
      if (m) found = __shfl(j, __ffsll((long long)m) - 1);
    }
    if (lane == 0) {
      if (found >= 0) { waitOn[i] = found; Unext[atomicAdd(&counters[0], 1)] = i; }
      else { waitOn[i] = WAIT_NONE; K[atomicAdd(&counters[1], 1)] = i; }
    }
  }
}

// Round kernel B: wave per new survivor: mark it, emit the pairs the reference would evaluate.
__global__ void __launch_bounds__(256) k_round_emit(const int* __restrict__ K, int nK, unsigned char* __restrict__ state,
                                                    const i64* __restrict__ nbrStart, const int* __restrict__ nbr, Flags f,
                                                    const float* __restrict__ pts, const int4* __restrict__ bbox,
                                                    const float* __restrict__ radius, const float* __restrict__ area,
                                                    int2* __restrict__ pairs, unsigned long long* pairCount,
                                                    unsigned long long pairCap) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = blockIdx.x * (blockDim.x >> 6) + wave;
  if (w >= nK) return;
  const int i = K[w];
  if (lane == 0) state[i] = ST_KEPT;
  const i64 beg = nbrStart[i], end = nbrStart[i + 1];
  const int4 bi = bbox[i];
  const float pyi = pts[2 * i], pxi = pts[2 * i + 1];
  const float rad = f.max_dist + radius[i];
  const float rad2 = rad * rad;                       // stardist2d.cpp:549
  for (i64 t = beg; t < end; t += 64) {
    const i64 idx = t + lane;
    bool emit = false;
    int j = -1;
    if (idx < end) {
      j = nbr[idx];
      if (j > i && state[j] == ST_UNDECIDED) {        // :572
        bool ok = true;
        if (f.use_kdtree) {                           // nanoflann L2_Simple, strict '<' (nanoflann.hpp:249-253)
          const float d0 = pyi - pts[2 * j], d1 = pxi - pts[2 * j + 1];
          float d2 = d0 * d0; d2 += d1 * d1;
          ok = d2 < rad2;
        }
        if (ok && (f.use_bbox || f.thr_nonneg)) {
          const int4 bj = bbox[j];
          ok = bbox_intersect(bi, bj);                                               // :576
          if (ok && f.thr_nonneg) {
            // rigorous upper bound: Clipper's output (input vertices + lattice-rounded crossings) stays inside the
            // intersection of the two integer bounding boxes, so area_inter <= w*h; if even that cannot exceed
            // the threshold the reference's  overlap > thr  (:580-581) is false without running the sweep.
            const double w = (double)(min(bi.y, bj.y) - max(bi.x, bj.x)), hgt = (double)(min(bi.w, bj.w) - max(bi.z, bj.z));
            const float ub = (float)((w * hgt) / fmin((double)area[i] + 1.e-10, (double)area[j] + 1.e-10));   // monotone in the area
            if (!(ub > f.thr)) ok = false;
          }
        }
        emit = ok;
      }
    }
    const unsigned long long m = __ballot(emit);
    if (m) {
      unsigned long long base = 0;
      if (lane == 0) base = atomicAdd(pairCount, (unsigned long long)__popcll(m));
      base = __shfl(base, 0);
      if (emit) {
        const unsigned long long pos = base + __popcll(m & ((1ull << lane) - 1));
        if (pos < pairCap) pairs[pos] = make_int2(i, j);
      }
    }
  }
}

// Round kernel C: one thread per pair.
template <int MAXV, int MAXIL, int MAXREC>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) k_pairs(const int2* __restrict__ pairs, unsigned long long nPairs, int R,
                                              const int* __restrict__ vx, const int* __restrict__ vy,
                                              const float* __restrict__ area, float thr, unsigned char* __restrict__ state,
                                              int2* __restrict__ joinPairs, unsigned int* joinCount, unsigned int joinCap,
                                              unsigned int* errCount) {
  const unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nPairs) return;
  const int2 ij = pairs[p];
  sdclip::Sweep<MAXV, MAXIL, MAXREC> sw;
  sw.reset_state();
  sw.add_path(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, R, sdclip::kClip, 0);        // :157
  sw.add_path(vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, sdclip::kSubject, MAXV);  // :158
  const i64 twice = sw.execute();
  if (sw.status & ~sdclip::ST_FAIL) { atomicAdd(errCount, 1u); }
  if (sw.n_joins > 0 || sw.sum_abs_terms >= (1ll << 24)) {
    // shared-edge joins (or float-order sensitivity) can change the reference's area: re-run on the exact path
    const unsigned int q = atomicAdd(joinCount, 1u);
    if (q < joinCap) joinPairs[q] = ij;
    return;
  }
  const float area_inter = 0.5f * (float)twice;
  const float overlap = (float)((double)area_inter / fmin((double)area[ij.x] + 1.e-10, (double)area[ij.y] + 1.e-10));  // :580
  if (overlap > thr) state[ij.y] = ST_SUPPRESSED;                                           // :581-585
}

// Round kernel C', latency variant: same sweep, per-pair state in LDS (LdsStorage) instead of scratch.  Used for the
// small rounds of the greedy scan, where the launch time is one pair's serial latency, not throughput.
enum { LDS_T = 32 };
template <int MAXV, int MAXIL, int MAXREC>
__global__ void __launch_bounds__(LDS_T) k_pairs_lds(const int2* __restrict__ pairs, unsigned long long nPairs, int R,
                                                     const int* __restrict__ vx, const int* __restrict__ vy,
                                                     const float* __restrict__ area, float thr, unsigned char* __restrict__ state,
                                                     int2* __restrict__ joinPairs, unsigned int* joinCount, unsigned int joinCap) {
  typedef sdclip::LdsStorage<LDS_T> LP;
  for (unsigned long long p = (unsigned long long)blockIdx.x * LDS_T + threadIdx.x; p < nPairs; p += (unsigned long long)gridDim.x * LDS_T) {
    const int2 ij = pairs[p];
    sdclip::Sweep<MAXV, MAXIL, MAXREC, LP> sw;       // arrays live in dynamic LDS (stateless, see LdsStorage)
    sw.reset_state();
    sw.add_path(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, R, sdclip::kClip, 0);
    sw.add_path(vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, sdclip::kSubject, MAXV);
    const i64 twice = sw.execute();
    if ((sw.status & ~sdclip::ST_FAIL) || sw.n_joins > 0 || sw.sum_abs_terms >= (1ll << 24)) {
      // capacity of the compact variant exceeded, or joins recorded: the exact path decides
      const unsigned int q = atomicAdd(joinCount, 1u);
      if (q < joinCap) joinPairs[q] = ij;
      continue;
    }
    const float area_inter = 0.5f * (float)twice;
    const float overlap = (float)((double)area_inter / fmin((double)area[ij.x] + 1.e-10, (double)area[ij.y] + 1.e-10));
    if (overlap > thr) state[ij.y] = ST_SUPPRESSED;
  }
}

template <int MAXV, int MAXIL, int MAXREC>
size_t lds_pairs_bytes() {
  typedef sdclip::LdsStorage<LDS_T> LP;
  return (size_t)sdclip::Sweep<MAXV, MAXIL, MAXREC, LP>::lds_bytes() + 64;
}

__global__ void k_iota(int* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = i; }
__global__ void k_keep(const unsigned char* __restrict__ state, unsigned char* __restrict__ keep, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = (state[i] != ST_SUPPRESSED);
}

}  // namespace

namespace sd {
// exact-join path (nms2d_full.hip): evaluates pairs whose result depends on Clipper's
// JoinCommonEdges; returns per pair 2*area as the reference sums it.
int clip_full_pairs(const int2* d_pairs, unsigned int n, int R, const int* d_vx, const int* d_vy, i64* d_twice,
                    int* d_flags, hipStream_t stream);
}

namespace {
__global__ void k_apply_full(const int2* __restrict__ pairs, unsigned int n, const i64* __restrict__ twice,
                             const float* __restrict__ area, float thr, unsigned char* __restrict__ state) {
  const unsigned int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int2 ij = pairs[p];
  const float area_inter = 0.5f * (float)twice[p];
  const float overlap = (float)((double)area_inter / fmin((double)area[ij.x] + 1.e-10, (double)area[ij.y] + 1.e-10));
  if (overlap > thr) state[ij.y] = ST_SUPPRESSED;
}

template <int MAXV, int MAXIL, int MAXREC>
void launch_pairs(const int2* pairs, unsigned long long nPairs, int R, const int* vx, const int* vy, const float* area,
                  float thr, unsigned char* state, int2* joinPairs, unsigned int* joinCount, unsigned int joinCap,
                  unsigned int* errCount, hipStream_t s) {
  const unsigned int blocks = (unsigned int)((nPairs + 63) / 64);
  hipLaunchKernelGGL((k_pairs<MAXV, MAXIL, MAXREC>), dim3(blocks), dim3(64), 0, s, pairs, nPairs, R, vx, vy, area, thr,
                     state, joinPairs, joinCount, joinCap, errCount);
}
}  // namespace

extern "C" int sd_nms2d_device(const float* d_dist, const float* d_points, int n_polys, int n_rays, int use_kdtree,
                               int use_bbox, int verbose, float threshold, uint8_t* d_keep, int64_t* stats,
                               void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  const int N = n_polys, R = n_rays;
  if (stats) memset(stats, 0, 16 * sizeof(int64_t));
  if (N <= 0) return 0;
  if (R < 1 || R > 256) { sd::set_error("sd_nms2d: n_rays=%d unsupported (1..256)", R); return -1; }
  if (verbose) {
    printf("Non Maximum Suppression (2D) ++++ \n");
    printf("NMS: n_polys    = %d \nNMS: n_rays     = %d  \nNMS: thresh     = %.3f \nNMS: use_bbox   = %d\nNMS: use_kdtree = %d\n",
           N, R, threshold, use_bbox, use_kdtree);
    printf("NMS: using HIP (gfx950), uniform-grid broad phase + scan-beam pair kernel\n");
  }
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  // HIP events on the caller's stream: per-kernel durations for the roofline report (bench.py)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (stats) { SD_CHECK(hipEventCreate(&ev0)); SD_CHECK(hipEventCreate(&ev1)); }
  struct EvGuard { hipEvent_t a, b; ~EvGuard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evguard{ev0, ev1};
  double ns_pairs = 0, ns_full = 0, ns_pre = 0;
  long long n_pair_launches = 0, n_lds_launches = 0;
  static const long long lds_max_pairs = getenv("SD_LDS_MAX_PAIRS") ? atoll(getenv("SD_LDS_MAX_PAIRS")) : 8192;   // = pairs resident in one pass (256 CUs x 32 threads)

  // all-pairs configuration with a negative threshold: every pair (0, j) passes the reference's
  // filters and overlap >= 0 > thr, so polygon 0 suppresses everything else.
  if (!use_kdtree && !use_bbox && threshold < 0) {
    SD_CHECK(hipMemsetAsync(d_keep, 0, N, s));
    SD_CHECK(hipMemsetAsync(d_keep, 1, 1, s));
    SD_CHECK(hipStreamSynchronize(s));
    return 0;
  }

  // host-side sin/cos table with the host libm (stardist2d.cpp:419,454-455)
  std::vector<float2> sc(R);
  const float ANGLE_PI = 2 * M_PI / R;
  for (int k = 0; k < R; ++k) { sc[k].x = sinf(ANGLE_PI * k); sc[k].y = cosf(ANGLE_PI * k); }
  float2* d_sc = A.take_n<float2>(R);
  int* vx = A.take_n<int>((size_t)N * R);
  int* vy = A.take_n<int>((size_t)N * R);
  int4* bbox = A.take_n<int4>(N);
  float* radius = A.take_n<float>(N);
  float* area = A.take_n<float>(N);
  int* gstats = A.take_n<int>(8);
  unsigned char* state = A.take_n<unsigned char>(N);
  int* candCell = A.take_n<int>(N);
  if (!d_sc || !vx || !vy || !bbox || !radius || !area || !gstats || !state || !candCell) return -1;
  SD_CHECK(hipMemcpyAsync(d_sc, sc.data(), R * sizeof(float2), hipMemcpyHostToDevice, s));
  const int gs_init[8] = {0, INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN, 0, 0, 0};
  SD_CHECK(hipMemcpyAsync(gstats, gs_init, sizeof(gs_init), hipMemcpyHostToDevice, s));
  SD_CHECK(hipMemsetAsync(state, 0, N, s));
  if (stats) SD_CHECK(hipEventRecord(ev0, s));
  hipLaunchKernelGGL(k_build, dim3(sd::div_up(N, 4)), dim3(256), 4 * 2 * R * sizeof(int), s, d_dist, d_points, d_sc, N, R,
                     vx, vy, bbox, radius, area, gstats);
  SD_LAUNCH_CHECK();
  int gs[8];
  SD_CHECK(hipMemcpyAsync(gs, gstats, sizeof(gs), hipMemcpyDeviceToHost, s));
  SD_CHECK(hipStreamSynchronize(s));
  float max_dist;
  memcpy(&max_dist, &gs[0], 4);

  // ---- uniform grid
  GridP g;
  float cs = (max_dist + 1.f) * 1.0001f + 1e-3f;
  if (!(cs >= 1.f)) cs = 1.f;
  const int W = 2;
  for (;;) {
    g.ny = (int)(((double)gs[2] - gs[1]) / cs) + 1;
    g.nx = (int)(((double)gs[4] - gs[3]) / cs) + 1;
    if ((long long)g.ny * g.nx <= (1ll << 26)) break;
    cs *= 2.f;
  }
  g.y0 = (float)gs[1]; g.x0 = (float)gs[3]; g.inv_cs = 1.f / cs;
  const int nCells = g.ny * g.nx;
  int* cellCount = A.take_n<int>(nCells + 1);
  int* cellStart = A.take_n<int>(nCells + 1);
  int* cellFill = A.take_n<int>(nCells + 1);
  CellRec* cellRec = (CellRec*)A.take((size_t)N * sizeof(CellRec));
  int* nbrCount = A.take_n<int>(N + 1);
  i64* nbrStart = A.take_n<i64>(N + 1);
  if (!cellCount || !cellStart || !cellFill || !cellRec || !nbrCount || !nbrStart) return -1;
  SD_CHECK(hipMemsetAsync(cellCount, 0, (nCells + 1) * sizeof(int), s));
  SD_CHECK(hipMemsetAsync(cellFill, 0, (nCells + 1) * sizeof(int), s));
  hipLaunchKernelGGL(k_cell_count, dim3(sd::div_up(N, 256)), dim3(256), 0, s, d_points, N, g, cellCount, candCell);
  SD_LAUNCH_CHECK();
  size_t tmpBytes = 0, tmpBytes2 = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, cellCount, cellStart, nCells + 1, s);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes2, nbrCount, nbrStart, N + 1, s);
  if (tmpBytes2 > tmpBytes) tmpBytes = tmpBytes2;
  void* scanTmp = A.take(tmpBytes + 256);
  if (!scanTmp) return -1;
  SD_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp, tmpBytes, cellCount, cellStart, nCells + 1, s));
  hipLaunchKernelGGL(k_cell_fill, dim3(sd::div_up(N, 256)), dim3(256), 0, s, N, candCell, cellStart, cellFill, d_points, bbox, area, cellRec);
  SD_LAUNCH_CHECK();

  Flags f;
  f.use_kdtree = use_kdtree; f.use_bbox = use_bbox; f.thr_nonneg = (threshold >= 0.f); f.thr = threshold; f.max_dist = max_dist;

  // ---- neighbour CSR
  SD_CHECK(hipMemsetAsync(nbrCount, 0, (N + 1) * sizeof(int), s));
  const int nbBlocks = (sd::div_up(N, 4) + 7) & ~7;
  hipLaunchKernelGGL((k_neighbours<0>), dim3(nbBlocks), dim3(256), 0, s, N, g, f, cellRec, cellStart, nbrCount, (const i64*)nullptr, (int*)nullptr,
                     (int*)nullptr, W);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp, tmpBytes, nbrCount, nbrStart, N + 1, s));
  i64 totalNbr = 0;
  SD_CHECK(hipMemcpyAsync(&totalNbr, nbrStart + N, sizeof(i64), hipMemcpyDeviceToHost, s));
  SD_CHECK(hipStreamSynchronize(s));
  int* nbr = A.take_n<int>((size_t)totalNbr);
  int* waitOn = A.take_n<int>(N);
  if (!nbr || !waitOn) return -1;
  hipLaunchKernelGGL((k_neighbours<1>), dim3(nbBlocks), dim3(256), 0, s, N, g, f, cellRec, cellStart, nbrCount, (const i64*)nbrStart, nbr, waitOn, W);
  SD_LAUNCH_CHECK();

  if (stats) { SD_CHECK(hipEventRecord(ev1, s)); SD_CHECK(hipEventSynchronize(ev1)); float ms = 0; SD_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); ns_pre = ms * 1e6; }

  // ---- greedy rounds
  const unsigned long long pairCap = (unsigned long long)(totalNbr / 2 + 64);
  int* U0 = A.take_n<int>(N);
  int* U1 = A.take_n<int>(N);
  int* K = A.take_n<int>(N);
  int2* pairs = A.take_n<int2>(pairCap);
  const unsigned int joinCap = (unsigned int)(pairCap < (1ull << 30) ? pairCap : (1ull << 30));
  int2* joinPairs = A.take_n<int2>(joinCap);
  i64* joinTwice = A.take_n<i64>(joinCap);
  int* joinFlags = A.take_n<int>(joinCap);
  struct Counters { int nU, nK, nS, pad; unsigned long long nPairs; unsigned int nJoin, nErr; };
  int* Sl = A.take_n<int>(N);
  if (!Sl) return -1;
  Counters* d_cnt = (Counters*)A.take(sizeof(Counters));
  if (!U0 || !U1 || !K || !pairs || !joinPairs || !joinTwice || !joinFlags || !d_cnt) return -1;
  hipLaunchKernelGGL(k_iota, dim3(sd::div_up(N, 256)), dim3(256), 0, s, U0, N);
  int nU = N, rounds = 0;
  i64 totalPairs = 0, totalJoin = 0;
  int* Ucur = U0; int* Unext = U1;
  Counters h;
  while (nU > 0) {
    ++rounds;
    SD_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), s));
    hipLaunchKernelGGL(k_round_triage, dim3(sd::div_up(nU, 256)), dim3(256), 0, s, Ucur, nU, state, waitOn, Unext, K, Sl, (int*)d_cnt);
    {
      const int scanBlocks = sd::div_up(nU, 4) < 2048 ? sd::div_up(nU, 4) : 2048;
      hipLaunchKernelGGL(k_round_scan, dim3(scanBlocks), dim3(256), 0, s, Sl, state, nbrStart, nbr, waitOn, Unext, K, (int*)d_cnt);
    }
    SD_LAUNCH_CHECK();
    SD_CHECK(hipMemcpyAsync(&h, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    if (h.nK == 0 && h.nU > 0) { sd::set_error("sd_nms2d: greedy scan made no progress (internal error)"); return -1; }
    if (h.nK > 0) {
      hipLaunchKernelGGL(k_round_emit, dim3(sd::div_up(h.nK, 4)), dim3(256), 0, s, K, h.nK, state, nbrStart, nbr, f, d_points, bbox,
                         radius, area, pairs, &d_cnt->nPairs, pairCap);
      SD_LAUNCH_CHECK();
      SD_CHECK(hipMemcpyAsync(&h, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
      SD_CHECK(hipStreamSynchronize(s));
      if (h.nPairs > pairCap) { sd::set_error("sd_nms2d: pair queue overflow (internal error)"); return -1; }
      if (h.nPairs > 0) {
        totalPairs += (i64)h.nPairs;
        if (stats) SD_CHECK(hipEventRecord(ev0, s));
        if (R <= 32 && h.nPairs <= (unsigned long long)lds_max_pairs) {
          // latency-bound round: LDS-resident sweep state, one 32-thread workgroup per CU
          static const size_t ldsBytes = lds_pairs_bytes<32, 48, 16>();
          static bool attr_set = false;
          if (!attr_set) {   // > 64 KiB of dynamic LDS needs an explicit opt-in
            SD_CHECK(hipFuncSetAttribute((const void*)k_pairs_lds<32, 48, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
            attr_set = true;
          }
          const unsigned int blocks = (unsigned int)((h.nPairs + LDS_T - 1) / LDS_T);
          hipLaunchKernelGGL((k_pairs_lds<32, 48, 16>), dim3(blocks < 2048u ? blocks : 2048u), dim3(LDS_T), ldsBytes, s, pairs, h.nPairs, R, vx, vy,
                             area, threshold, state, joinPairs, &d_cnt->nJoin, joinCap);
          ++n_lds_launches;
        } else
        if (R <= 32) launch_pairs<32, 64, 32>(pairs, h.nPairs, R, vx, vy, area, threshold, state, joinPairs, &d_cnt->nJoin, joinCap, &d_cnt->nErr, s);
        else if (R <= 64) launch_pairs<64, 96, 48>(pairs, h.nPairs, R, vx, vy, area, threshold, state, joinPairs, &d_cnt->nJoin, joinCap, &d_cnt->nErr, s);
        else if (R <= 128) launch_pairs<128, 128, 64>(pairs, h.nPairs, R, vx, vy, area, threshold, state, joinPairs, &d_cnt->nJoin, joinCap, &d_cnt->nErr, s);
        else launch_pairs<256, 192, 96>(pairs, h.nPairs, R, vx, vy, area, threshold, state, joinPairs, &d_cnt->nJoin, joinCap, &d_cnt->nErr, s);
        SD_LAUNCH_CHECK();
        if (stats) SD_CHECK(hipEventRecord(ev1, s));
        SD_CHECK(hipMemcpyAsync(&h, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
        SD_CHECK(hipStreamSynchronize(s));
        if (stats) { float ms = 0; SD_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); ns_pairs += ms * 1e6; ++n_pair_launches;
                     if (getenv("SD_TRACE")) printf("round %d: nU=%d nK=%d pairs=%llu joins=%u pair_kernel=%.3f ms\n", rounds, h.nU, h.nK, h.nPairs, h.nJoin, ms); }
        if (h.nErr) { sd::set_error("sd_nms2d: %u pairs exceeded the scan-beam kernel's fixed capacities", h.nErr); return -1; }
        if (h.nJoin > 0) {
          if (h.nJoin > joinCap) { sd::set_error("sd_nms2d: join queue overflow"); return -1; }
          totalJoin += h.nJoin;
          if (stats) SD_CHECK(hipEventRecord(ev0, s));
          if (sd::clip_full_pairs(joinPairs, h.nJoin, R, vx, vy, joinTwice, joinFlags, s)) return -1;
          if (stats) { SD_CHECK(hipEventRecord(ev1, s)); SD_CHECK(hipEventSynchronize(ev1)); float ms = 0; SD_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); ns_full += ms * 1e6; }
          hipLaunchKernelGGL(k_apply_full, dim3(sd::div_up(h.nJoin, 256)), dim3(256), 0, s, joinPairs, h.nJoin, joinTwice, area, threshold, state);
          SD_LAUNCH_CHECK();
        }
      }
    }
    nU = h.nU;
    int* t = Ucur; Ucur = Unext; Unext = t;
  }
  hipLaunchKernelGGL(k_keep, dim3(sd::div_up(N, 256)), dim3(256), 0, s, state, d_keep, N);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipStreamSynchronize(s));
  if (stats) { stats[0] = totalPairs; stats[1] = totalJoin; stats[2] = rounds; stats[3] = totalNbr;
               stats[4] = (int64_t)ns_pairs; stats[5] = n_pair_launches; stats[6] = (int64_t)ns_full; stats[7] = (int64_t)ns_pre;
               stats[8] = n_lds_launches; }
  if (verbose) {
    printf("NMS: %lld pair intersections (%lld on the exact-join path), %d greedy rounds, %lld neighbour entries\n",
           (long long)totalPairs, (long long)totalJoin, rounds, (long long)totalNbr);
    fflush(stdout);
  }
  return 0;
}

extern "C" int sd_nms2d_host(const float* dist, const float* points, int n_polys, int n_rays, int use_kdtree, int use_bbox,
                             int verbose, float threshold, uint8_t* keep, int64_t* stats) {
  if (n_polys <= 0) return 0;
  float *d_dist = nullptr, *d_pts = nullptr;
  uint8_t* d_keep = nullptr;
  SD_CHECK(hipMalloc(&d_dist, (size_t)n_polys * n_rays * sizeof(float)));
  SD_CHECK(hipMalloc(&d_pts, (size_t)n_polys * 2 * sizeof(float)));
  SD_CHECK(hipMalloc(&d_keep, n_polys));
  int rc = -1;
  do {
    if (hipMemcpy(d_dist, dist, (size_t)n_polys * n_rays * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (hipMemcpy(d_pts, points, (size_t)n_polys * 2 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (sd_nms2d_device(d_dist, d_pts, n_polys, n_rays, use_kdtree, use_bbox, verbose, threshold, d_keep, stats, nullptr)) break;
    if (hipMemcpy(keep, d_keep, n_polys, hipMemcpyDeviceToHost) != hipSuccess) { sd::set_error("D2H failed"); break; }
    rc = 0;
  } while (0);
  (void)hipFree(d_dist); (void)hipFree(d_pts); (void)hipFree(d_keep);
  return rc;
}
