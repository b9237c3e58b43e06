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
#include <type_traits>

#include "common.h"
#include "clip_sweep.h"
#include "clip_beam.h"
#include "area_bounds.h"
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

// k_build for n_rays <= 32: HALF a wave per candidate (a 32-ray polygon leaves half of a wave idle in k_build: 0.47 -> 0.25 ms for the
// 418 577 candidates of the 2048^2 bench set, on the critical path of the grid set-up).  Same arithmetic, same order of the float area sum.
__global__ void __launch_bounds__(256) k_build32(const float* __restrict__ dist, const float* __restrict__ pts,
                                                 const float2* __restrict__ sincos, int N, int R,
                                                 int* __restrict__ vx, int* __restrict__ vy, int4* __restrict__ bbox,
                                                 float* __restrict__ radius, float* __restrict__ area, int* gstats) {
  __shared__ int sxy[8][2][32];
  const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const int i = blockIdx.x * 8 + hw;
  const bool cv = i < N, lv = cv && l < R;
  int* sx = sxy[hw][0];
  int* sy = sxy[hw][1];
  float py = 0.f, px = 0.f;
  if (cv) { py = pts[2 * i]; px = pts[2 * i + 1]; }
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, rmax = 0.f;
  if (lv) {
    const float d = dist[(size_t)i * R + l];
    const float2 sc = sincos[l];
    const float y = py + d * sc.x;   // stardist2d.cpp:454 (compiled with -ffp-contract=off)
    const float x = px + d * sc.y;   // stardist2d.cpp:455
    xmin = xmax = x; ymin = ymax = y;
    const int X = (int)(long long)x, Y = (int)(long long)y;   // IntPoint(cInt(x), cInt(y)) :471
    sx[l] = X; sy[l] = Y;
    vx[(size_t)i * R + l] = X; vy[(size_t)i * R + l] = Y;
    rmax = fmaxf(0.f, d);
  }
  for (int o = 16; o; o >>= 1) {
    xmin = fminf(xmin, __shfl_xor(xmin, o)); xmax = fmaxf(xmax, __shfl_xor(xmax, o));
    ymin = fminf(ymin, __shfl_xor(ymin, o)); ymax = fmaxf(ymax, __shfl_xor(ymax, o));
    rmax = fmaxf(rmax, __shfl_xor(rmax, o));
  }
  __builtin_amdgcn_wave_barrier();           // (a wave's LDS accesses are processed in order)
  i64 s = 0, sa = 0;
  if (lv) {
    const int kn = (l + 1 == R) ? 0 : l + 1;
    const i64 c = (i64)sx[l] * sy[kn] - (i64)sy[l] * sx[kn];
    s = c; sa = (c < 0 ? -c : c);
  }
  for (int o = 16; o; o >>= 1) { s += __shfl_xor(s, o); sa += __shfl_xor(sa, o); }
  if (cv && l == 0) {
    // area_from_path :128-138: float accumulation of int64 cross products in path order; equals the exact integer sum whenever
    // sum|term| < 2^24, else the serial order is replayed
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

// The cells a candidate's neighbours can lie in (cells are indexed by the CENTRE of a candidate).  With the bbox test in force (threshold >= 0
// or use_bbox: may_interact requires intersecting integer bounding boxes) a neighbour j of i has bbox_j inside [p_j - (r + 1), p_j + (r + 1)],
// r = the largest distance of ANY candidate (a vertex is centre + d (sin, cos) truncated), so its centre lies within bbox_i grown by r + 1:
// a window per candidate (round 6; before: the 5 x 5 cells of size r + 1 around its own cell, 1.5 x the area), with half a pixel of slack
// for the float cell arithmetic.  Without the bbox test (kd-tree radius only) the window is the centre +- (2 r + 1).
struct Window { int ylo, yhi, xlo, xhi; };
__device__ __forceinline__ Window cell_window(const GridP g, bool by_bbox, float reach, const int4 bb, float py, float px) {
  float y0, y1, x0, x1;
  if (by_bbox) { x0 = (float)bb.x - reach; x1 = (float)bb.y + reach; y0 = (float)bb.z - reach; y1 = (float)bb.w + reach; }
  else { const float rr = 2.f * reach; x0 = px - rr; x1 = px + rr; y0 = py - rr; y1 = py + rr; }
  Window w;
  w.ylo = min(max((int)floorf((y0 - 0.5f - g.y0) * g.inv_cs), 0), g.ny - 1); w.yhi = min(max((int)floorf((y1 + 0.5f - g.y0) * g.inv_cs), 0), g.ny - 1);
  w.xlo = min(max((int)floorf((x0 - 0.5f - g.x0) * g.inv_cs), 0), g.nx - 1); w.xhi = min(max((int)floorf((x1 + 0.5f - g.x0) * g.inv_cs), 0), g.nx - 1);
  return w;
}
constexpr int MAX_WIN_ROWS = 32;      // rows of a window (<= 10 with cells of half the reach; the host falls back to coarser cells otherwise)

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
// slotCap (may be null): upper bound of candidate i's neighbour count = the population of the cells of its window (cell_window) its list is built from,
// minus itself -- the capacity of its slot in the single-pass neighbour lists (k_neighbours<2>)
__global__ void k_cell_fill(int N, const int* __restrict__ candCell, const int* __restrict__ cellStart,
                            int* __restrict__ cellFill, const float* __restrict__ pts, const int4* __restrict__ bbox,
                            const float* __restrict__ area, CellRec* __restrict__ rec, GridP g, int by_bbox, float reach, int* __restrict__ slotCap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int c = candCell[i];
  const int pos = atomicAdd(&cellFill[c], 1);
  CellRec r;
  r.bb = bbox[i]; r.py = pts[2 * i]; r.px = pts[2 * i + 1]; r.area = area[i]; r.j = i;
  rec[cellStart[c] + pos] = r;
  if (slotCap) {
    const Window w = cell_window(g, by_bbox != 0, reach, r.bb, r.py, r.px);
    int u = -1;
    for (int yy = w.ylo; yy <= w.yhi; ++yy) u += cellStart[yy * g.nx + w.xhi + 1] - cellStart[yy * g.nx + w.xlo];
    slotCap[i] = u;
  }
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
    // whose bbox-intersection area cannot exceed thr * min(area) can never suppress.  The lists only have to be a SUPERSET of the pairs
    // the emission kernels accept with the exact form of this bound (k_round_emit / k_tail_emit: the double quotient): here it is the
    // division-free float form with a relative margin of 1e-5 (every rounding of either form is below 2e-7).
    if (!bbox_intersect(bi, bj)) return false;
    const float wh = (float)(min(bi.y, bj.y) - max(bi.x, bj.x)) * (float)(min(bi.w, bj.w) - max(bi.z, bj.z));
    return wh * 1.00001f >= f.thr * fminf(ai, aj) * 0.99999f;
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

// MODE 0: count neighbours, MODE 1: fill CSR (two passes: exact-size lists);
// MODE 2: ONE pass into per-candidate slots whose capacity is the population of the cells scanned (k_cell_fill slotCap; nbrStart = slot
// starts): the better-scored neighbours are written from the slot's front, the others from its back, nbrLow / nbrCount (= the number of
// the others) tell the consumers where each half ends -- the candidate tests of the counting pass are not repeated
#define WAIT_NONE (-2)
#define WAIT_SCAN (-1)
template <int MODE>
__global__ void __launch_bounds__(256) k_neighbours(int N, GridP g, Flags f, const CellRec* __restrict__ rec, const int* __restrict__ cellStart,
                                                    int* __restrict__ nbrCount, int* __restrict__ nbrLow, const i64* __restrict__ nbrStart,
                                                    int* __restrict__ nbr, int* __restrict__ waitOn, int by_bbox, float reach, unsigned long long* __restrict__ total) {
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
  // a candidate's list holds its better-scored (lower-index) neighbours first -- what the greedy scan and the tail batch look at --
  // then the others -- what a new survivor is paired with: each consumer reads its half only
  int nLo = 0, nHi = 0;
  int minj = INT32_MAX;                      // MODE 1: best-scored neighbour above i (first wait target of the greedy scan)
  const i64 baseLo = MODE ? nbrStart[i] : 0;
  const i64 baseHi = MODE == 1 ? baseLo + nbrLow[i] : (MODE == 2 ? nbrStart[i + 1] - 1 : 0);      // MODE 2: the slot's last entry, filled downwards
  // The window's rows are contiguous runs of the cell-ordered records; the runs are walked as ONE list (lane r holds row r's start and its
  // exclusive prefix), 64 records per step whatever the row lengths -- a row of ~35 records no longer costs a step of its own.
  const Window win = cell_window(g, by_bbox != 0, reach, bi, pyi, pxi);
  const int nrows = win.yhi - win.ylo + 1;
  int rbeg = 0, rlen = 0;
  if (lane < nrows) { rbeg = cellStart[(win.ylo + lane) * g.nx + win.xlo]; rlen = cellStart[(win.ylo + lane) * g.nx + win.xhi + 1] - rbeg; }
  int incl = rlen;
  for (int o = 1; o < MAX_WIN_ROWS; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
  const int total_recs = __shfl(incl, nrows - 1);
  const int excl = incl - rlen;
  for (int t = 0; t < total_recs; t += 64) {
    const int v = t + lane;
    bool hit = false;
    int j = -1;
    {
      // the row of virtual index v: the last row whose exclusive prefix is <= v (rows may be empty)
      int rb = 0, re = 0;
      for (int r = 0; r < nrows; ++r) {
        const int e_r = __shfl(excl, r), b_r = __shfl(rbeg, r);
        if (e_r <= v) { rb = b_r; re = e_r; }
      }
      if (v < total_recs) {
        const CellRec r = rec[rb + (v - re)];
        j = r.j;
        if (j != i) hit = may_interact(f, bi, r.bb, pyi, pxi, r.py, r.px, ai, r.area);
      }
    }
    const unsigned long long mLo = __ballot(hit && j < i), mHi = __ballot(hit && j > i);
    if (MODE && hit) {
      const unsigned long long below = (1ull << lane) - 1;
      if (j < i) { nbr[baseLo + nLo + __popcll(mLo & below)] = j; if (j < minj) minj = j; }
      else if (MODE == 1) nbr[baseHi + nHi + __popcll(mHi & below)] = j;
      else nbr[baseHi - (nHi + __popcll(mHi & below))] = j;
    }
    nLo += __popcll(mLo); nHi += __popcll(mHi);
  }
  if (!MODE && lane == 0) { nbrCount[i] = nLo + nHi; nbrLow[i] = nLo; }
  if (MODE == 2 && lane == 0) nbrLow[i] = nLo;          // (the total is summed by k_sum_halves: one atomic per candidate on one word serialises at the L2)
  if (MODE && lane == 0) nbrCount[i] = nHi;            // from here on nbrCount holds the size of the worse-scored half (k_round_emit)
  if (MODE) {
    for (int o = 32; o; o >>= 1) minj = min(minj, __shfl_xor(minj, o));
    if (lane == 0) waitOn[i] = (minj < i) ? minj : WAIT_NONE;
  }
}

// exact number of list entries of the single-pass form: sum of both halves' sizes, one atomic per workgroup
__global__ void __launch_bounds__(256) k_sum_halves(const int* __restrict__ nLow, const int* __restrict__ nHigh, int N, unsigned long long* total) {
  __shared__ unsigned long long ws[4];
  unsigned long long v = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256) v += (unsigned long long)(nLow[i] + nHigh[i]);
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(total, ws[0] + ws[1] + ws[2] + ws[3]);
}

// Round kernel A1: thread per undecided candidate, O(1): waitOn[i] is the higher-scored neighbour i was last seen waiting
// for (k_neighbours seeds it with the best-scored one; WAIT_NONE = there is none, WAIT_SCAN = unknown).  Most waits persist
// from round to round, so only the candidates whose wait target has just been decided go to the list scan (A2).
__global__ void __launch_bounds__(1024) k_round_triage(const int* __restrict__ U, int nU, const unsigned char* __restrict__ state,
                                                       const int* __restrict__ waitOn, int* __restrict__ Unext, int* __restrict__ K,
                                                       int* __restrict__ S, int* counters /*0:nUnext 1:nK 2:nS*/,
                                                       const unsigned char* __restrict__ pend) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int kind = 0, i = -1;                       // 0 drop, 1 still waiting, 2 becomes a survivor, 3 needs the list scan
  if (t < nU) {
    i = U[t];
    if (state[i] != ST_SUPPRESSED) {
      const int wo = waitOn[i];
      if (pend[i]) kind = 1;                  // a survivor's pair with i is deferred to the general path: i cannot be promoted yet
      else if (wo == WAIT_NONE) kind = 2;
      else if (wo >= 0 && state[wo] == ST_UNDECIDED) kind = 1;
      else kind = 3;
    }
  }
  // ONE atomic per list and workgroup of 1024 candidates: an atomic per wave (6 500 waves x 3 lists at 2048^2) serialises at the L2 --
  // measured 138 us for this kernel in round 1, most of it waiting for the three counters
  __shared__ int wcnt[3][16];
  __shared__ int bbase[3];
#pragma unroll
  for (int q = 1; q <= 3; ++q) {
    const unsigned long long m = __ballot(kind == q);
    if (lane == 0) wcnt[q - 1][wave] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    int sum = 0;
    for (int w = 0; w < nw; ++w) { const int c = wcnt[threadIdx.x][w]; wcnt[threadIdx.x][w] = sum; sum += c; }
    bbase[threadIdx.x] = sum ? atomicAdd(&counters[threadIdx.x], sum) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 1; q <= 3; ++q) {
    const unsigned long long m = __ballot(kind == q);
    if (kind == q) (q == 1 ? Unext : (q == 2 ? K : S))[bbase[q - 1] + wcnt[q - 1][wave] + __popcll(m & ((1ull << lane) - 1))] = i;
  }
}

// Round kernel A2: wave per candidate of the scan list (persistent grid; the list length is read on the device).
__global__ void __launch_bounds__(256) k_round_scan(const int* __restrict__ S, const unsigned char* __restrict__ state,
                                                    const i64* __restrict__ nbrStart, const int* __restrict__ nbrLow, const int* __restrict__ nbr,
                                                    int* __restrict__ waitOn, int* __restrict__ Unext, int* __restrict__ K,
                                                    int* counters /*0:nUnext 1:nK 2:nS*/, const unsigned char* __restrict__ pend) {
  const int lane = threadIdx.x & 63;
  const int nS = counters[2];
  const int nWaves = gridDim.x * (blockDim.x >> 6);
  // a wave visits its candidates one after the other; the outcomes are collected (lane k keeps the k-th) and appended to the two
  // lists with ONE atomic per list and 64 candidates -- an atomicAdd per candidate on a single counter serialises at the L2
  // (10^5 candidates in the second round = 1 ms)
  int myI = -1, myKind = 0, nbuf = 0;
  auto flush = [&]() {
#pragma unroll
    for (int q = 1; q <= 2; ++q) {
      const unsigned long long m = __ballot(lane < nbuf && myKind == q);
      if (!m) continue;
      int base = 0;
      if (lane == 0) base = atomicAdd(&counters[q - 1], __popcll(m));
      base = __shfl(base, 0);
      if (lane < nbuf && myKind == q) (q == 1 ? Unext : K)[base + __popcll(m & ((1ull << lane) - 1))] = myI;
    }
    nbuf = 0;
  };
  // FOUR candidates per wave at a time, 16 lanes each (a list of better-scored neighbours holds ~40 entries): the kernel is a chain of
  // dependent gathers (list bounds -> neighbour -> its state), so candidates in flight are what counts
  const int sub = lane >> 4, sl = lane & 15;
  for (int w0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4; w0 < nS; w0 += nWaves * 4) {
    const int w = w0 + sub;
    const bool valid = w < nS;
    const int i = valid ? S[w] : -1;
    i64 t = 0, end = 0;
    if (valid) { t = nbrStart[i]; end = t + nbrLow[i]; }        // the better-scored neighbours
    int found = -1;
    while (__any(found < 0 && t < end)) {
      const i64 idx = t + sl;
      int j = -1;
      if (found < 0 && idx < end) { j = nbr[idx]; if (!(j < i && state[j] == ST_UNDECIDED)) j = -1; }
      const unsigned long long m = __ballot(j >= 0);
      const unsigned int m16 = (unsigned int)(m >> (sub << 4)) & 0xffffu;
      const int src = (sub << 4) + (m16 ? __ffs((int)m16) - 1 : 0);
      const int jf = __shfl(j, src);
      if (found < 0 && m16) found = jf;
      t += 16;
    }
    if (valid && sl == 0) waitOn[i] = found >= 0 ? found : WAIT_NONE;
    const int kind = (found >= 0 || (valid && pend[i])) ? 1 : 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int vi = __shfl(i, q << 4), vk = __shfl(kind, q << 4);
      if (vi >= 0) { if (lane == nbuf) { myI = vi; myKind = vk; } ++nbuf; }      // (vi is wave-uniform)
    }
    if (nbuf > 60) flush();
  }
  flush();
}

// Round kernel B: wave per new survivor: mark it, emit the pairs the reference would evaluate.
constexpr int EMIT_STAGE = 512;
__global__ void __launch_bounds__(256) k_round_emit(const int* __restrict__ K, const int* __restrict__ nKPtr, unsigned char* __restrict__ state,
                                                    const i64* __restrict__ nbrStart, const int* __restrict__ nbrHigh, const int* __restrict__ nbr, Flags f,
                                                    const float* __restrict__ pts, const int4* __restrict__ bbox,
                                                    const float* __restrict__ radius, const float* __restrict__ area,
                                                    int2* __restrict__ pairs, unsigned long long* pairCount,
                                                    unsigned long long pairCap) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nK = *nKPtr;                               // persistent grid: the survivor count of this round is read on the device
  // pairs are staged per wave in LDS and appended with ONE atomic per EMIT_STAGE pairs: an atomic per 64-entry chunk of a neighbour list
  // (1.5 x 10^5 of them on one counter in round 1 at 2048^2) serialises at the L2
  __shared__ int2 stage[4][EMIT_STAGE];
  int nst = 0;                                         // (wave-uniform)
  auto flush = [&]() {
    if (!nst) return;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(pairCount, (unsigned long long)nst);
    base = __shfl(base, 0);
    __builtin_amdgcn_wave_barrier();                   // (a wave's LDS accesses are processed in order)
    for (int k = lane; k < nst; k += 64) { const unsigned long long pos = base + k; if (pos < pairCap) pairs[pos] = stage[wave][k]; }
    __builtin_amdgcn_wave_barrier();
    nst = 0;
  };
  for (int w = blockIdx.x * (blockDim.x >> 6) + wave; w < nK; w += gridDim.x * (blockDim.x >> 6)) {
  const int i = K[w];
  if (lane == 0) state[i] = ST_KEPT;
  const i64 end = nbrStart[i + 1], beg = end - nbrHigh[i];             // the neighbours scored below i (the back of i's slot)
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
      const int c = __popcll(m);
      if (nst + c > EMIT_STAGE) flush();
      if (emit) stage[wave][nst + __popcll(m & ((1ull << lane) - 1))] = make_int2(i, j);
      nst += c;
    }
  }
  }
  flush();
}

// ---- prepared polygons: Clipper::AddPath once per candidate (clip_beam.h), one thread per candidate, working arrays
// lane-interleaved in LDS
template <int MAXV, int S>
__global__ void __launch_bounds__(S) k_prepare(const int* __restrict__ vx, const int* __restrict__ vy, int N, int R,
                                               sdclip::PolyPrep<MAXV>* __restrict__ prep) {
  const int i = blockIdx.x * S + threadIdx.x;
  if (i >= N) return;
  sdclip::PrepWork<sdclip::LdsStorage<S>, MAXV> w;
  w.prepare(vx + (size_t)i * R, vy + (size_t)i * R, R, prep + i);
}

// Round kernel C: one thread per pair, bound-slot sweep with the whole per-pair state lane-interleaved in LDS.
// Persistent grid: the pair count is read on the device (no host round trip between emit and this kernel).
//   result 'capacity exceeded'  -> spill queue (next tier with larger K / list capacities, finally the general path)
//   joins recorded / float-order risk / any other flag -> exact queue (clip_sweep_full.h restates JoinCommonEdges)
// queues hold PAIR INDICES (into the round's pair list), so a result can be written per pair (tail batch) as well as applied
struct PairQueues { unsigned int* spill; unsigned int* spillCount; unsigned int* exact; unsigned int* exactCount; unsigned int cap; };
enum { BEAM_CAPFLAGS = sdclip::ST_OVERFLOW_IL | sdclip::ST_OVERFLOW_REC | sdclip::ST_OVERFLOW_AEL | sdclip::ST_OVERFLOW_LM | sdclip::ST_OVERFLOW_GJ };
// idx == nullptr: pairs[0 .. *nPtr);  else pairs[idx[0 .. *nPtr)].   supp == nullptr: a suppressing pair marks state[j]
// (greedy round, i is a survivor); else supp[pair index] = 1 (tail batch: i is still undecided, the result is kept per edge).
// REL16: the sweep state with 16-bit coordinates relative to the pair's origin and recomputed slopes (clip_sweep.h LdsStorage16): 420
// instead of 636 bytes per pair; a pair whose coordinates do not fit raises a capacity flag and spills to the next tier like any other
template <int MAXV, int K, int MAXIL, int MAXREC, int S, typename CNT, bool REL16 = false>
__global__ void __launch_bounds__(S) k_pairs_beam(const int2* __restrict__ pairs, const unsigned int* __restrict__ idx, const CNT* __restrict__ nPtr,
                                                  const unsigned int* __restrict__ firstPtr, const sdclip::PolyPrep<MAXV>* __restrict__ prep,
                                                  const float* __restrict__ area, float thr, unsigned char* __restrict__ state,
                                                  unsigned char* __restrict__ supp, PairQueues q) {
  typedef typename std::conditional<REL16, sdclip::LdsStorage16<S>, sdclip::LdsStorage<S>>::type Storage;
  typedef sdclip::Beam<MAXV, K, MAXIL, MAXREC, Storage> BeamT;
  const unsigned long long n = (unsigned long long)*nPtr;
  const unsigned long long first = firstPtr ? (unsigned long long)*firstPtr : 0ull;      // entries before `first` are already queued for the general path
  // Latency regime (fewer pairs than the launch has lanes: the late rounds, the tail batch): every workgroup takes ONE contiguous chunk of the
  // (offset-ordered) list with its first `chunk` lanes, the others stay idle -- the sweeps of a wave's lanes share no control flow, so a
  // launch lasts as long as the busiest wave's lanes take one after the other; spreading 25 000 pairs over all 1 536 waves at 16 lanes each
  // instead of filling 390 waves halves that.  Otherwise the grid-stride loop over full waves.
  const unsigned long long total = n > first ? n - first : 0ull;
  const unsigned long long chunk = (total + gridDim.x - 1) / gridDim.x;
  const bool spread = chunk <= (unsigned long long)S;
  unsigned long long t = spread ? first + (unsigned long long)blockIdx.x * chunk + threadIdx.x : first + (unsigned long long)blockIdx.x * S + threadIdx.x;
  if (spread && (unsigned long long)threadIdx.x >= chunk) t = n;
  const unsigned long long step = spread ? (1ull << 62) : (unsigned long long)gridDim.x * S;
  for (; t < n; t += step) {
    const unsigned int p = idx ? idx[t] : (unsigned int)t;
    const int2 ij = pairs[p];
    BeamT bm;
    bm.reset_state(prep + ij.x, prep + ij.y);                                  // clip = i (:157), subject = j (:158)
    const i64 twice = bm.execute();
    const int st = bm.status;
    if (st & BEAM_CAPFLAGS) {
      const unsigned int k = atomicAdd(q.spillCount, 1u);
      if (k < q.cap) q.spill[k] = p;
      continue;
    }
    if ((st & ~sdclip::ST_FAIL) || bm.n_joins > 0 || bm.sum_abs_terms >= (1ll << 24)) {
      const unsigned int k = atomicAdd(q.exactCount, 1u);
      if (k < q.cap) q.exact[k] = p;
      continue;
    }
    const float area_inter = 0.5f * (float)twice;
    const float overlap = (float)((double)area_inter / fmin((double)area[ij.x] + 1.e-10, (double)area[ij.y] + 1.e-10));  // :580
    if (overlap > thr) { if (supp) supp[p] = 1; else state[ij.y] = ST_SUPPRESSED; }                // :581-585
  }
}
template <int MAXV, int K, int MAXIL, int MAXREC, int S, bool REL16 = false>
size_t beam_lds_bytes() {
  typedef typename std::conditional<REL16, sdclip::LdsStorage16<S>, sdclip::LdsStorage<S>>::type Storage;
  return (size_t)sdclip::Beam<MAXV, K, MAXIL, MAXREC, Storage>::lds_bytes() + 64;
}

// Round kernel C0 (area_bounds.h): the pairs whose decision follows from an enclosure of the intersection area -- regular arithmetic,
// 32 lanes per pair -- are decided here; decided[t] != 0 removes a pair from the sweep kernels' work list (k_pair_bucket_*).
__global__ void __launch_bounds__(256) k_pairs_decide(const int2* __restrict__ pairs, const unsigned long long* __restrict__ nPtr,
                                                      const unsigned int* __restrict__ firstPtr, const int* __restrict__ vx, const int* __restrict__ vy, int R,
                                                      const sdarea::PolyProps* __restrict__ props, const float* __restrict__ area, float thr,
                                                      unsigned char* __restrict__ state, unsigned char* __restrict__ supp,
                                                      unsigned char* __restrict__ decided, unsigned int* __restrict__ nDecided) {
  __shared__ float2 sq[4][2][32];
  const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31, wv = threadIdx.x >> 6;
  const unsigned long long n = *nPtr, first = firstPtr ? (unsigned long long)*firstPtr : 0ull;
  const unsigned long long nw = (unsigned long long)gridDim.x * 4;
  unsigned int mine = 0;
  for (unsigned long long base = first + 2ull * ((unsigned long long)blockIdx.x * 4 + wv); base < n; base += 2ull * nw) {
    const unsigned long long t = base + half;
    const bool active = t < n;
    const int2 ij = active ? pairs[t] : make_int2(0, 0);
    const sdarea::PolyProps pp = props[ij.x], pq = props[ij.y];
    const sdarea::Enclosure E = sdarea::pair_enclosure(vx + (size_t)ij.x * R, vy + (size_t)ij.x * R, vx + (size_t)ij.y * R, vy + (size_t)ij.y * R, R, pp, pq,
                                                       active, sq[wv][half], l, half);             // clip = i, subject = j (:157-158)
    if (active && l == 0) {
      const int dec = sdarea::decide(E, area[ij.x], area[ij.y], thr);
      decided[t] = (unsigned char)dec;
      if (dec == 2) { if (supp) supp[t] = 1; else state[ij.y] = ST_SUPPRESSED; }                  // :581-585
      if (dec) ++mine;
    }
    __builtin_amdgcn_wave_barrier();
  }
  mine += __shfl_xor(mine, 32);
  if (lane == 0 && mine) atomicAdd(nDecided, mine);
}

// ---- tail batch.  Late greedy rounds hold few pairs but each costs one sweep's serial latency; once few candidates are
// undecided, ALL pairs (i < j, both undecided) the reference could still evaluate are emitted at once, their overlap decisions
// are computed speculatively (supp[edge]), and one workgroup then replays the remaining greedy rounds on the device:
// j is suppressed iff some KEPT i < j has supp(i, j); it is kept once every such i is decided and none suppresses it.
__global__ void __launch_bounds__(256) k_tail_emit(const int* __restrict__ U, int nU, const unsigned char* __restrict__ state,
                                                   const i64* __restrict__ nbrStart, const int* __restrict__ nbrLow, const int* __restrict__ nbr, Flags f,
                                                   const float* __restrict__ pts, const int4* __restrict__ bbox,
                                                   const float* __restrict__ radius, const float* __restrict__ area,
                                                   int2* __restrict__ pairs, unsigned long long* pairCount, unsigned long long pairCap,
                                                   unsigned int* __restrict__ segStart, int* __restrict__ segCnt) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int w = blockIdx.x * (blockDim.x >> 6) + wave; w < nU; w += gridDim.x * (blockDim.x >> 6)) {
    const int j = U[w];
    int cnt = 0;
    unsigned long long base = 0;
    if (state[j] == ST_UNDECIDED) {
      const i64 beg = nbrStart[j], end = beg + nbrLow[j];      // the better-scored neighbours
      const int4 bj = bbox[j];
      const float pyj = pts[2 * j], pxj = pts[2 * j + 1];
      const float aj = area[j];
      for (int pass = 0; pass < 2; ++pass) {
        int k = 0;
        for (i64 t = beg; t < end; t += 64) {
          const i64 id = t + lane;
          bool emit = false;
          int i = -1;
          if (id < end) {
            i = nbr[id];
            if (i < j && state[i] == ST_UNDECIDED) {          // i may become a survivor before j is decided (:572 seen from i)
              bool ok = true;
              if (f.use_kdtree) {                             // nanoflann radius search around i (:546-550)
                const float rad = f.max_dist + radius[i];
                const float d0 = pts[2 * i] - pyj, d1 = pts[2 * i + 1] - pxj;
                float d2 = d0 * d0; d2 += d1 * d1;
                ok = d2 < rad * rad;
              }
              if (ok && (f.use_bbox || f.thr_nonneg)) {
                const int4 bi = bbox[i];
                ok = bbox_intersect(bi, bj);                                             // :576
                if (ok && f.thr_nonneg) {                     // same rigorous bbox-area bound as k_round_emit
                  const double w2 = (double)(min(bi.y, bj.y) - max(bi.x, bj.x)), hgt = (double)(min(bi.w, bj.w) - max(bi.z, bj.z));
                  const float ub = (float)((w2 * hgt) / fmin((double)area[i] + 1.e-10, (double)aj + 1.e-10));
                  if (!(ub > f.thr)) ok = false;
                }
              }
              emit = ok;
            }
          }
          const unsigned long long m = __ballot(emit);
          if (pass == 1 && emit) {
            const unsigned long long pos = base + k + __popcll(m & ((1ull << lane) - 1));
            if (pos < pairCap) pairs[pos] = make_int2(i, j);
          }
          k += __popcll(m);
        }
        if (pass == 0) {
          cnt = k;
          if (cnt == 0) break;
          if (lane == 0) base = atomicAdd(pairCount, (unsigned long long)cnt);
          base = __shfl(base, 0);
        }
      }
    }
    if (lane == 0) { segStart[w] = (unsigned int)base; segCnt[w] = cnt; }
  }
}
// ---- deferral of the general path.  A pair of a normal round that needs the general path (joins / capacities) is NOT evaluated
// in that round: it is appended to the deferred list, j is marked pending (it can still be suppressed by another survivor's pair,
// but cannot be promoted), and all deferred pairs are evaluated by ONE launch of the general kernel in the tail batch -- instead
// of one latency-bound launch per round.  Per candidate the deferred pairs form a linked list (defHead / defNext).
struct Deferred { int2* pairs; int* next; int* head; unsigned char* pend; unsigned int* count; unsigned int cap; };
// (a deferred list that is full cannot take the pair: counted in *nErr, the call then fails loudly instead of dropping a decision)
__global__ void k_defer(const int2* __restrict__ pairs, const unsigned int* __restrict__ exact, const unsigned int* __restrict__ nExact, unsigned int qcap,
                        Deferred d, unsigned int* __restrict__ nErr) {
  unsigned int n = *nExact; if (n > qcap) n = qcap;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    const int2 ij = pairs[exact[t]];
    const unsigned int k = atomicAdd(d.count, 1u);
    if (k < d.cap) { d.pairs[k] = ij; d.pend[ij.y] = 1; d.next[k] = atomicExch(&d.head[ij.y], (int)k); }
    else atomicAdd(nErr, 1u);
  }
}
// tail batch: the deferred pairs become entries 0 .. nDef-1 of the tail's pair list and of its general-path queue
__global__ void k_tail_init(Deferred d, int2* __restrict__ pairs, unsigned int* __restrict__ exact, unsigned long long* nPairs,
                            unsigned int* nExact, unsigned int* firstNew) {
  unsigned int n = *d.count; if (n > d.cap) n = d.cap;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) { pairs[t] = d.pairs[t]; exact[t] = t; }
  if (blockIdx.x == 0 && threadIdx.x == 0) { *nPairs = n; *nExact = n; *firstNew = n; }
}
// Deferral of the pairs the area enclosure leaves undecided (option "nms2d_defer_undecided" = first round that defers): a sweep launch
// costs one sweep's serial latency (~0.65 ms) however few pairs it holds, so from that round on the undecided pairs are not swept in
// their round but carried to the tail batch like the general-path pairs -- kind[k] = 1 marks them: there they are swept (tier 1 / 2),
// while the kind-0 pairs go to the general kernel as before.  j is pending until then (can be suppressed, cannot be promoted).
__global__ void k_defer_undecided(const int2* __restrict__ pairs, const unsigned long long* __restrict__ nPtr, const unsigned int* __restrict__ nDecided,
                                  unsigned int limit, unsigned char* __restrict__ decided, const unsigned char* __restrict__ state, Deferred d,
                                  unsigned char* __restrict__ kind, unsigned int* __restrict__ nErr) {
  // only a round whose undecided pairs would make a latency-bound sweep launch defers them (many undecided pairs -- a threshold inside
  // the overlap range of one object's candidates -- are swept at once more cheaply than they are carried along); a deferred pair gets
  // decided[t] = 4, so that the bucketing and the sweeps that follow in the stream find nothing to do
  const unsigned long long n = *nPtr;
  if (n - (unsigned long long)*nDecided > (unsigned long long)limit) return;
  for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * blockDim.x) {
    if (decided[t]) continue;
    decided[t] = 4;
    const int2 ij = pairs[t];
    if (state[ij.y] == ST_SUPPRESSED) continue;            // a decided pair of this round already suppressed j
    const unsigned int k = atomicAdd(d.count, 1u);
    if (k < d.cap) { d.pairs[k] = ij; d.pend[ij.y] = 1; d.next[k] = atomicExch(&d.head[ij.y], (int)k); kind[k] = 1; }
    else atomicAdd(nErr, 1u);
  }
}
// tail batch with both kinds of deferred pairs: all become entries 0 .. nDef-1 of the pair list; kind 0 -> general-path queue and
// decided[t] = 3 (the sweeps skip them), kind 1 -> decided[t] = 0 (bucketed and swept with the tail's own undecided pairs).
// *nExact must be zero on entry; *nJoin receives the number of kind-0 pairs (the queue's prefix).
__global__ void k_tail_init2(Deferred d, const unsigned char* __restrict__ kind, int2* __restrict__ pairs, unsigned int* __restrict__ exact,
                             unsigned long long* nPairs, unsigned int* nExact, unsigned int* firstNew, unsigned char* __restrict__ decided) {
  unsigned int n = *d.count; if (n > d.cap) n = d.cap;
  for (unsigned int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
    pairs[t] = d.pairs[t];
    if (kind[t] == 0) { exact[atomicAdd(nExact, 1u)] = t; decided[t] = 3; } else decided[t] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { *nPairs = n; *firstNew = n; }
}
__global__ void k_copy_u32(unsigned int* dst, const unsigned int* src) { *dst = *src; }
// a sweep launch over fewer pairs than this is pure latency (98 304 pairs are in flight at once): option "nms2d_defer_max", default 16 384
// deferred pairs of candidate j: true if one of them suppresses it (their survivors i are KEPT by construction)
__device__ __forceinline__ bool deferred_suppresses(int j, const int* __restrict__ defHead, const int* __restrict__ defNext,
                                                    const unsigned char* __restrict__ supp) {
  for (int k = defHead[j]; k >= 0; k = defNext[k]) if (supp[k]) return true;
  return false;
}

// One replayed greedy round for the whole chip (thread per candidate of the tail list).  Decisions only ever go
// UNDECIDED -> final, and a kernel boundary makes the previous step's decisions visible to every CU, so running this a fixed
// number of times needs no host round trip; whatever is still undecided afterwards (dependency chains longer than that) is
// finished by the single-workgroup loop below.
__global__ void __launch_bounds__(256) k_tail_step(const int* __restrict__ U, int nU, unsigned char* state, const int2* __restrict__ pairs,
                                                   const unsigned char* __restrict__ supp, const unsigned int* __restrict__ segStart,
                                                   const int* __restrict__ segCnt, unsigned long long pairCap,
                                                   const int* __restrict__ defHead, const int* __restrict__ defNext) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nU) return;
  const int j = U[t];
  if (state[j] != ST_UNDECIDED) return;
  bool wait = false, sup = deferred_suppresses(j, defHead, defNext, supp);
  const unsigned int e0 = segStart[t];
  const int c = segCnt[t];
  for (int e = 0; e < c && !sup; ++e) {
    if ((unsigned long long)e0 + e >= pairCap) { wait = true; break; }
    const unsigned char si = state[pairs[e0 + e].x];
    if (si == ST_UNDECIDED) wait = true;
    else if (si == ST_KEPT && supp[e0 + e]) sup = true;
  }
  if (sup) state[j] = ST_SUPPRESSED;
  else if (!wait) state[j] = ST_KEPT;
}
// one workgroup; 'left' receives the number of candidates still undecided (0 unless something is inconsistent)
__global__ void __launch_bounds__(1024) k_tail_resolve(const int* __restrict__ U, int nU, volatile unsigned char* state,
                                                       const int2* __restrict__ pairs, const unsigned char* __restrict__ supp,
                                                       const unsigned int* __restrict__ segStart, const int* __restrict__ segCnt,
                                                       unsigned long long pairCap, int* left, const int* __restrict__ defHead,
                                                       const int* __restrict__ defNext) {
  __shared__ int changed, undecided;
  for (int iter = 0; iter <= nU; ++iter) {
    if (threadIdx.x == 0) { changed = 0; undecided = 0; }
    __syncthreads();
    for (int t = threadIdx.x; t < nU; t += blockDim.x) {
      const int j = U[t];
      if (state[j] != ST_UNDECIDED) continue;
      bool wait = false, sup = deferred_suppresses(j, defHead, defNext, supp);
      const unsigned int e0 = segStart[t];
      const int c = segCnt[t];
      for (int e = 0; e < c && !sup; ++e) {
        if ((unsigned long long)e0 + e >= pairCap) { wait = true; break; }
        const unsigned char si = state[pairs[e0 + e].x];
        if (si == ST_UNDECIDED) wait = true;
        else if (si == ST_KEPT && supp[e0 + e]) sup = true;
      }
      if (sup) { state[j] = ST_SUPPRESSED; changed = 1; }
      else if (!wait) { state[j] = ST_KEPT; changed = 1; }
      else undecided = 1;
    }
    __threadfence_block();
    __syncthreads();
    const int ch = changed, un = undecided;
    __syncthreads();
    if (!un || !ch) { if (threadIdx.x == 0) *left = un; break; }
  }
}

// ---- pair order.  The pair kernel runs one sweep per lane; lanes of a wave that sweep geometrically similar pairs take similar
// control paths (the number of scan-beams and where the second polygon enters are functions of the centre offset).  The pair
// list [first, n) is therefore bucketed by the quantised centre offset (dy major, dx minor) before tier 1: histogram, scan,
// scatter of PAIR INDICES -- the kernels behind it already work on index lists.  Decisions are per pair, so the order (and the
// arbitrary order inside a bucket) cannot change any result.
constexpr int PAIR_BUCKETS = 4096;
// key = (local-minima class of the two polygons [nl], dy bin [ny], dx bin [nx]); nl * ny * nx <= PAIR_BUCKETS
struct PairKey { const char* prep; size_t prepStride; float inv; int nl, ny, nx; };
__device__ __forceinline__ int pair_bucket(const float* __restrict__ pts, int2 ij, const PairKey& k) {
  const float dy = pts[2 * ij.y] - pts[2 * ij.x], dx = pts[2 * ij.y + 1] - pts[2 * ij.x + 1];
  int by = (int)((dy * k.inv + 0.5f) * (float)k.ny), bx = (int)((dx * k.inv + 0.5f) * (float)k.nx);
  by = by < 0 ? 0 : (by >= k.ny ? k.ny - 1 : by); bx = bx < 0 ? 0 : (bx >= k.nx ? k.nx - 1 : bx);
  int lm = 0;
  if (k.nl > 1) {                                    // n_lm is the second int of a PolyPrep record (clip_beam.h)
    int a = *(const int*)(k.prep + (size_t)ij.x * k.prepStride + 4), b = *(const int*)(k.prep + (size_t)ij.y * k.prepStride + 4);
    a = a < 1 ? 0 : (a > 4 ? 3 : a - 1); b = b < 1 ? 0 : (b > 4 ? 3 : b - 1);
    lm = (a * 4 + b) % k.nl;
  }
  return (lm * k.ny + by) * k.nx + bx;
}
__global__ void __launch_bounds__(256) k_pair_bucket_count(const int2* __restrict__ pairs, const unsigned long long* __restrict__ nPtr,
                                                           const unsigned int* __restrict__ firstPtr, const float* __restrict__ pts, PairKey key,
                                                           unsigned int* __restrict__ hist, const unsigned char* __restrict__ decided) {
  __shared__ unsigned int h[PAIR_BUCKETS];
  for (int b = threadIdx.x; b < PAIR_BUCKETS; b += 256) h[b] = 0;
  __syncthreads();
  const unsigned long long n = *nPtr, first = firstPtr ? *firstPtr : 0u;
  for (unsigned long long t = first + (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * 256)
    if (!decided || !decided[t]) atomicAdd(&h[pair_bucket(pts, pairs[t], key)], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < PAIR_BUCKETS; b += 256) if (h[b]) atomicAdd(&hist[b], h[b]);
}
__global__ void __launch_bounds__(1024) k_pair_bucket_scan(const unsigned int* __restrict__ hist, unsigned int* __restrict__ cursor,
                                                           unsigned long long* __restrict__ nOrdered) {
  __shared__ unsigned int sh[1024];
  const int t = threadIdx.x;
  unsigned int v[PAIR_BUCKETS / 1024], tot = 0;
#pragma unroll
  for (int k = 0; k < PAIR_BUCKETS / 1024; ++k) { v[k] = hist[t * (PAIR_BUCKETS / 1024) + k]; tot += v[k]; }
  sh[t] = tot;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned int add = t >= o ? sh[t - o] : 0u;
    __syncthreads();
    sh[t] += add;
    __syncthreads();
  }
  unsigned int run = sh[t] - tot;
#pragma unroll
  for (int k = 0; k < PAIR_BUCKETS / 1024; ++k) { cursor[t * (PAIR_BUCKETS / 1024) + k] = run; run += v[k]; }
  if (t == 1023) *nOrdered = sh[t];
}
__global__ void __launch_bounds__(256) k_pair_bucket_scatter(const int2* __restrict__ pairs, const unsigned long long* __restrict__ nPtr,
                                                             const unsigned int* __restrict__ firstPtr, const float* __restrict__ pts, PairKey key,
                                                             unsigned int* __restrict__ cursor, unsigned int* __restrict__ order, unsigned int cap,
                                                             const unsigned char* __restrict__ decided) {
  const unsigned long long n = *nPtr, first = firstPtr ? *firstPtr : 0u;
  for (unsigned long long t = first + (unsigned long long)blockIdx.x * 256 + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * 256) {
    if (decided && decided[t]) continue;
    const unsigned int k = atomicAdd(&cursor[pair_bucket(pts, pairs[t], key)], 1u);
    if (k < cap) order[k] = (unsigned int)t;
  }
}

__global__ void k_count_after(unsigned int* out, const unsigned int* total, const unsigned int* first) { *out = *total > *first ? *total - *first : 0u; }
__global__ void k_iota(int* a, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = i; }
__global__ void k_keep(const unsigned char* __restrict__ state, unsigned char* __restrict__ keep, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = (state[i] != ST_SUPPRESSED);
}

}  // namespace

namespace sd {
// general path (nms2d_full.hip): pairs whose result depends on Clipper's JoinCommonEdges, or that exceed the
// capacities of the bound-slot kernels; device-side queue, result applied directly.
int clip_full_pairs(const int2* d_pairs, const unsigned int* d_idx, const unsigned int* d_n, unsigned int cap, int R, const int* d_vx, const int* d_vy,
                    const float* d_area, float thr, unsigned char* d_state, unsigned char* d_supp, unsigned int* d_errCount, hipStream_t stream);
}

namespace {
struct Counters { int nU, nK, nS, left; unsigned long long nPairs; unsigned int nSpill, nExact, nErr, nDecided; };

// prepared polygons + the two tiers of the bound-slot pair kernel for one vertex capacity
template <int MAXV, int SPREP>
struct BeamPath {
  typedef sdclip::PolyPrep<MAXV> Prep;
  static int prepare(const int* vx, const int* vy, int N, int R, void* prep, hipStream_t s) {
    const size_t lds = sdclip::PrepWork<sdclip::LdsStorage<SPREP>, MAXV>::lds_bytes() + 64;
    hipLaunchKernelGGL((k_prepare<MAXV, SPREP>), dim3(sd::div_up(N, SPREP)), dim3(SPREP), lds, s, vx, vy, N, R, (Prep*)prep);
    SD_LAUNCH_CHECK();
    return 0;
  }
  // tier 1 (K = 8): all pairs of the round; capacity spills -> q.spill
  static int tier1(const int2* pairs, const unsigned int* idx, const unsigned long long* nPairs, const unsigned int* first, const void* prep, const float* area,
                   float thr, unsigned char* state, unsigned char* supp, PairQueues q, hipStream_t s) {
    // One pair per lane, the pair's whole sweep state in LDS.  The kernel is latency-bound under lane divergence (VALU issue 12.5 % at
    // one wave per SIMD, profiles/r03_pmc_mfma.md) and LDS capacity sets how many pairs a CU has in flight: with 32-bit coordinates and
    // stored slopes a pair takes 636 bytes -> 40.7 KB per wave -> FOUR waves per CU; with 16-bit coordinates relative to the pair's origin
    // and recomputed slopes (LdsStorage16, the default since round 4) 420 bytes -> 26.9 KB -> SIX waves per CU, 1.5x the pairs in flight.
    // Same arithmetic (tests/host/beam_check.cpp: 0 mismatches against Clipper in either form); pairs whose coordinates do not fit 16 bits
    // spill to tier 2.  Option "nms2d_pair_lanes": 64 (default) | 32 (half-filled waves: measured no gain) | 6464 (the 32-bit form).
    const int lanes = sd::option(sd::OPT_NMS2D_PAIR_LANES);
    if (lanes == 32) {
      static const size_t lds32 = beam_lds_bytes<MAXV, 8, 6, 4, 32>();
      hipLaunchKernelGGL((k_pairs_beam<MAXV, 8, 6, 4, 32, unsigned long long>), dim3(256 * 8), dim3(32), lds32, s, pairs, idx, nPairs, first, (const Prep*)prep, area, thr, state, supp, q);
    } else if (lanes == 6464) {
      static const size_t lds = beam_lds_bytes<MAXV, 8, 6, 4, 64>();
      hipLaunchKernelGGL((k_pairs_beam<MAXV, 8, 6, 4, 64, unsigned long long>), dim3(256 * 4), dim3(64), lds, s, pairs, idx, nPairs, first, (const Prep*)prep, area, thr, state, supp, q);
    } else {
      static const size_t lds16 = beam_lds_bytes<MAXV, 8, 6, 4, 64, true>();
      static const int per_cu = (int)((160 * 1024) / lds16);
      hipLaunchKernelGGL((k_pairs_beam<MAXV, 8, 6, 4, 64, unsigned long long, true>), dim3(256 * per_cu), dim3(64), lds16, s, pairs, idx, nPairs, first, (const Prep*)prep, area, thr, state, supp, q);
    }
    SD_LAUNCH_CHECK();
    return 0;
  }
  // tier 2 (K = 15, larger lists): the spills of tier 1 (or, for n_rays > 32, all pairs); its spills -> general path
  template <typename CNT>
  static int tier2(const int2* pairs, const unsigned int* idx, const CNT* nPairs, const unsigned int* first, const void* prep, const float* area, float thr,
                   unsigned char* state, unsigned char* supp, PairQueues q, hipStream_t s) {
    static const size_t lds = beam_lds_bytes<MAXV, 15, 16, 8, 32>();
    hipLaunchKernelGGL((k_pairs_beam<MAXV, 15, 16, 8, 32, CNT>), dim3(256 * 3), dim3(32), lds, s, pairs, idx, nPairs, first, (const Prep*)prep, area, thr, state, supp, q);
    SD_LAUNCH_CHECK();
    return 0;
  }
};
}  // namespace

// one non-blocking helper stream per device for work that is independent of the caller's stream for a while (fork / join by events)
static hipStream_t side_stream() {
  static hipStream_t st[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) { sd::set_error("sd_nms2d: no current device"); return nullptr; }
  if (!st[d] && hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking) != hipSuccess) { sd::set_error("sd_nms2d: cannot create a stream"); return nullptr; }
  return st[d];
}

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
  long long n_pair_launches = 0;

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
  if (R <= 32)
    hipLaunchKernelGGL(k_build32, dim3(sd::div_up(N, 8)), dim3(256), 0, s, d_dist, d_points, d_sc, N, R, vx, vy, bbox, radius, area, gstats);
  else
    hipLaunchKernelGGL(k_build, dim3(sd::div_up(N, 4)), dim3(256), 4 * 2 * R * sizeof(int), s, d_dist, d_points, d_sc, N, R,
                       vx, vy, bbox, radius, area, gstats);
  SD_LAUNCH_CHECK();
  // ---- prepared polygons (Clipper::AddPath once per candidate), on a second stream: they depend on the integer vertices only, the
  // grid and the neighbour lists that follow on the caller's stream do not need them (0.84 ms of independent work at 2048^2)
  size_t prepStride;
  if (R <= 32) prepStride = sizeof(sdclip::PolyPrep<32>); else if (R <= 64) prepStride = sizeof(sdclip::PolyPrep<64>);
  else if (R <= 128) prepStride = sizeof(sdclip::PolyPrep<128>); else prepStride = sizeof(sdclip::PolyPrep<256>);
  void* prep = A.take((size_t)N * prepStride);
  if (!prep) return -1;
  hipStream_t side = side_stream();
  if (!side) return -1;
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  SD_CHECK(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
  SD_CHECK(hipEventCreateWithFlags(&evJoin, hipEventDisableTiming));
  EvGuard evguardFJ{evFork, evJoin};
  // polygon properties of the decision shortcut (area_bounds.h) first: they also depend on the integer vertices only, and the
  // decision kernel of the first round needs them before any sweep needs a prepared polygon (evProps / evPrep)
  const bool areaBounds = R <= 32 && R >= 3 && sd::option(sd::OPT_NMS2D_AREA_BOUNDS) != 0 && sd::option(sd::OPT_NMS2D_STRICT) == 0;
  sdarea::PolyProps* props = nullptr;
  hipEvent_t evProps = nullptr, evPrep = nullptr;
  SD_CHECK(hipEventCreateWithFlags(&evProps, hipEventDisableTiming));
  SD_CHECK(hipEventCreateWithFlags(&evPrep, hipEventDisableTiming));
  EvGuard evguardP{evProps, evPrep};
  if (areaBounds) {
    props = (sdarea::PolyProps*)A.take((size_t)N * sizeof(sdarea::PolyProps));
    if (!props) return -1;
  }
  // The helper stream's two kernels are enqueued once the LAST read-back of the grid set-up is behind us (launch_side below): k_poly_props
  // fills every wave slot of the chip, and an 8-byte device -> host copy of the caller's stream issued while it runs waits for a slot
  // until it drains (measured: 0.69 ms for that copy, the neighbour-list kernel started 0.7 ms late; profiles/r05_step_timeline_2d.txt).
  auto launch_side = [&]() -> int {
    SD_CHECK(hipEventRecord(evFork, s));
    SD_CHECK(hipStreamWaitEvent(side, evFork, 0));
    if (areaBounds) {
      hipLaunchKernelGGL(sdarea::k_poly_props, dim3(sd::div_up(N, 8)), dim3(256), 0, side, vx, vy, N, R, props);
      SD_LAUNCH_CHECK();
    }
    SD_CHECK(hipEventRecord(evProps, side));
    int rc;
    if (R <= 32) rc = BeamPath<32, 64>::prepare(vx, vy, N, R, prep, side);
    else if (R <= 64) rc = BeamPath<64, 64>::prepare(vx, vy, N, R, prep, side);
    else if (R <= 128) rc = BeamPath<128, 32>::prepare(vx, vy, N, R, prep, side);
    else rc = BeamPath<256, 16>::prepare(vx, vy, N, R, prep, side);
    if (rc) return -1;
    SD_CHECK(hipEventRecord(evPrep, side));
    SD_CHECK(hipEventRecord(evJoin, side));
    return 0;
  };
  int gs[8];
  SD_CHECK(hipMemcpyAsync(gs, gstats, sizeof(gs), hipMemcpyDeviceToHost, s));
  SD_CHECK(hipStreamSynchronize(s));
  float max_dist;
  memcpy(&max_dist, &gs[0], 4);

  // ---- uniform grid
  GridP g;
  // cells of HALF the reach (max_dist + 1): a candidate's window (cell_window: its bounding box grown by the reach) then spans at most
  // (2 (max_dist + 1) + 2 (max_dist + 1)) / cs + 2 = 10 rows; a grid that would exceed 2^26 cells gets coarser cells (fewer rows still)
  const float reach = (max_dist + 1.f) * 1.0001f + 1e-3f;
  const int by_bbox = (threshold >= 0.f || use_bbox) ? 1 : 0;
  float cs = 0.5f * reach;
  if (!(cs >= 1.f)) cs = 1.f;
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
  int* nbrLow = A.take_n<int>(N + 1);
  if (!cellCount || !cellStart || !cellFill || !cellRec || !nbrCount || !nbrStart || !nbrLow) return -1;
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
  // Neighbour lists in ONE pass (option "nms2d_neighbours_single_pass", default 1): every candidate gets a slot as large as the population
  // of the cells its list is built from (known from the cell table: no candidate test needed), the lists are written into the slots --
  // better-scored neighbours from the front, the others from the back -- and the exact total is counted on the way.  The two-pass form
  // (count, scan, fill: every candidate test done twice, 1.2 + 0.8 ms at 2048^2) remains for inputs whose slots would exceed 32-bit indices.
  const bool singlePass = sd::option(sd::OPT_NMS2D_NBR_SINGLE) != 0;
  SD_CHECK(hipMemsetAsync(nbrCount, 0, (N + 1) * sizeof(int), s));
  hipLaunchKernelGGL(k_cell_fill, dim3(sd::div_up(N, 256)), dim3(256), 0, s, N, candCell, cellStart, cellFill, d_points, bbox, area, cellRec, g, by_bbox, reach,
                     singlePass ? nbrCount : (int*)nullptr);
  SD_LAUNCH_CHECK();
  // The N-sized work lists of the greedy rounds are set up HERE, in front of the neighbour lists: behind the read-back of the list total
  // their four small launches sat on the critical path in front of round 1 (~60 us of 5-us kernels and gaps)
  int* U0 = A.take_n<int>(N);
  unsigned char* pend0 = A.take_n<unsigned char>(N);
  int* head0 = A.take_n<int>(N);
  unsigned int* dcount0 = A.take_n<unsigned int>(1);
  if (!U0 || !pend0 || !head0 || !dcount0) return -1;
  hipLaunchKernelGGL(k_iota, dim3(sd::div_up(N, 256)), dim3(256), 0, s, U0, N);
  SD_CHECK(hipMemsetAsync(pend0, 0, N, s));
  SD_CHECK(hipMemsetAsync(head0, 0xFF, (size_t)N * sizeof(int), s));
  SD_CHECK(hipMemsetAsync(dcount0, 0, sizeof(unsigned int), s));

  Flags f;
  f.use_kdtree = use_kdtree; f.use_bbox = use_bbox; f.thr_nonneg = (threshold >= 0.f); f.thr = threshold; f.max_dist = max_dist;

  // ---- neighbour CSR
  const int nbBlocks = (sd::div_up(N, 4) + 7) & ~7;
  i64 totalNbr = 0, slotTotal = 0;
  bool slots = false;
  unsigned long long* d_total = A.take_n<unsigned long long>(1);
  if (!d_total) return -1;
  if (singlePass) {
    SD_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp, tmpBytes, nbrCount, nbrStart, N + 1, s));
    SD_CHECK(hipMemcpyAsync(&slotTotal, nbrStart + N, sizeof(i64), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    slots = slotTotal >= 0 && slotTotal < (i64)0x7fffffff;
  }
  // the slots can be several times the exact list size (every candidate of the window counts): when they do not fit the workspace the
  // call falls back to the exact-size two-pass form instead of failing
  int* nbr = nullptr;
  if (slots) {
    nbr = A.take_n<int>((size_t)slotTotal);
    if (!nbr) slots = false;
  }
  if (slots && launch_side()) return -1;
  if (!slots) {
    SD_CHECK(hipMemsetAsync(nbrCount, 0, (N + 1) * sizeof(int), s));
    hipLaunchKernelGGL((k_neighbours<0>), dim3(nbBlocks), dim3(256), 0, s, N, g, f, cellRec, cellStart, nbrCount, nbrLow, (const i64*)nullptr, (int*)nullptr,
                       (int*)nullptr, by_bbox, reach, (unsigned long long*)nullptr);
    SD_LAUNCH_CHECK();
    SD_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp, tmpBytes, nbrCount, nbrStart, N + 1, s));
    SD_CHECK(hipMemcpyAsync(&totalNbr, nbrStart + N, sizeof(i64), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    if (launch_side()) return -1;
  }
  // capacity of one call: neighbour lists and pair queues are indexed with 32 bits.  Beyond it (about 13 M candidates at the density
  // of the 2048^2 bench set) the input has to be sharded -- predict_instances_sharded / predict_instances_big do exactly that.
  if (totalNbr < 0 || totalNbr >= (i64)0x7fffffff) {
    sd::set_error("sd_nms2d: %lld neighbour entries for %d candidates exceed the capacity of one call (2^31 - 1): shard the input "
                  "(predict_instances_sharded / predict_instances_big)", (long long)totalNbr, N);
    return -1;
  }
  if (!slots) nbr = A.take_n<int>((size_t)totalNbr);
  int* waitOn = A.take_n<int>(N);
  if (!nbr || !waitOn) return -1;
  if (slots) {
    SD_CHECK(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL((k_neighbours<2>), dim3(nbBlocks), dim3(256), 0, s, N, g, f, cellRec, cellStart, nbrCount, nbrLow, (const i64*)nbrStart, nbr, waitOn, by_bbox, reach, d_total);
    hipLaunchKernelGGL(k_sum_halves, dim3(sd::div_up(N, 256) < 1024 ? sd::div_up(N, 256) : 1024), dim3(256), 0, s, nbrLow, nbrCount, N, d_total);
    SD_LAUNCH_CHECK();
    unsigned long long tot = 0;
    SD_CHECK(hipMemcpyAsync(&tot, d_total, sizeof(tot), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    totalNbr = (i64)tot;
  } else {
    hipLaunchKernelGGL((k_neighbours<1>), dim3(nbBlocks), dim3(256), 0, s, N, g, f, cellRec, cellStart, nbrCount, nbrLow, (const i64*)nbrStart, nbr, waitOn, by_bbox, reach,
                       (unsigned long long*)nullptr);
    SD_LAUNCH_CHECK();
  }

  // (the prepared polygons are being written on the side stream meanwhile; the sweep kernels are their first readers and wait for
  // evPrep in run_pairs -- with the shortcut on, the decision kernel of round 1 runs before that and only needs the properties)
  SD_CHECK(hipStreamWaitEvent(s, evProps, 0));
  if (stats) { SD_CHECK(hipEventRecord(ev1, s)); SD_CHECK(hipEventSynchronize(ev1)); float ms = 0; SD_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); ns_pre = ms * 1e6; }

  // ---- greedy rounds: every kernel of a round takes its work-list length from device memory; ONE host round trip per
  // round (the undecided / survivor counts that end the loop)
  const unsigned long long pairCap = (unsigned long long)(totalNbr / 2 + 64);
  int* U1 = A.take_n<int>(N);
  int* K = A.take_n<int>(N);
  int2* pairs = A.take_n<int2>(pairCap);
  const unsigned int qCap = (unsigned int)(pairCap < (1ull << 30) ? pairCap : (1ull << 30));
  unsigned int* spillPairs = A.take_n<unsigned int>(qCap);
  unsigned int* exactPairs = A.take_n<unsigned int>(qCap);
  int* Sl = A.take_n<int>(N);
  Counters* d_cnt = (Counters*)A.take(sizeof(Counters));
  // pair order (see k_pair_bucket_*): SD_NMS_PAIR_SORT=0 keeps emission order
  static const bool pairSort = sd::tuning_env("SD_NMS_PAIR_SORT", 1) != 0;
  unsigned int* pairOrder = pairSort && R <= 32 ? A.take_n<unsigned int>(qCap) : nullptr;
  unsigned int* bucketHist = A.take_n<unsigned int>(2 * PAIR_BUCKETS);
  unsigned long long* nOrdered = A.take_n<unsigned long long>(1);
  if (!U0 || !U1 || !K || !pairs || !spillPairs || !exactPairs || !Sl || !d_cnt || !bucketHist || !nOrdered || (pairSort && R <= 32 && !pairOrder)) return -1;
  unsigned char* decided = (areaBounds && pairOrder) ? A.take_n<unsigned char>(pairCap) : nullptr;       // (the shortcut filters through the ordered index list)
  if (areaBounds && pairOrder && !decided) return -1;
  i64 totalDecided = 0;
  int nU = N, rounds = 0;
  i64 totalPairs = 0, totalExact = 0, totalSpill = 0;
  int* Ucur = U0; int* Unext = U1;
  Counters h;
  hipEvent_t ev2 = nullptr, ev3 = nullptr;
  if (stats) { SD_CHECK(hipEventCreate(&ev2)); SD_CHECK(hipEventCreate(&ev3)); }
  EvGuard evguard2{ev2, ev3};
  // tail batch threshold: undecided candidates at or below which the remaining rounds are replayed on the device
  static const int tailDiv = sd::tuning_env("SD_NMS_TAIL_DIV", 6);
  static const int tailMax = sd::tuning_env("SD_NMS_TAIL_MAX", 65536);
  const int tailT = tailDiv > 0 ? ((N / tailDiv) < tailMax ? (N / tailDiv) : tailMax) : -1;
  unsigned char* supp = nullptr; unsigned int* segStart = nullptr; int* segCnt = nullptr;
  // deferral of the general path to the tail batch (only with a tail batch to run it in)
  static const bool deferEnv = sd::tuning_env("SD_NMS_DEFER", 1) != 0;
  const bool deferOn = tailT >= 0 && deferEnv;
  Deferred dfr{nullptr, nullptr, nullptr, nullptr, nullptr, qCap};
  unsigned int* firstNew = A.take_n<unsigned int>(1);
  dfr.pend = pend0; dfr.head = head0; dfr.count = dcount0;          // (allocated and cleared in front of the neighbour lists)
  if (!firstNew) return -1;
  if (deferOn) {
    dfr.pairs = A.take_n<int2>(qCap);
    dfr.next = A.take_n<int>(qCap);
    if (!dfr.pairs || !dfr.next) return -1;
  }
  i64 nDeferred = 0;
  bool sideGeneral = false;                   // tail batch: the deferred pairs' general-path launch runs on the helper stream
  unsigned int* nNewExact = A.take_n<unsigned int>(1);
  if (!nNewExact) return -1;
  // deferral of the enclosure's undecided pairs (see k_defer_undecided): from round 2 on by default (option nms2d_defer_undecided)
  const int deferFrom = (deferOn && decided) ? sd::option(sd::OPT_NMS2D_DEFER_UNDECIDED) : 0;
  const unsigned int deferMax = (unsigned int)(sd::option(sd::OPT_NMS2D_DEFER_MAX) > 0 ? sd::option(sd::OPT_NMS2D_DEFER_MAX) : 0);
  unsigned char* defKind = nullptr; unsigned int* nJoinDef = nullptr;
  if (deferFrom > 0) {
    defKind = A.take_n<unsigned char>(qCap); nJoinDef = A.take_n<unsigned int>(1);
    if (!defKind || !nJoinDef) return -1;
    SD_CHECK(hipMemsetAsync(defKind, 0, qCap, s));
  }
  i64 totalUndecDeferred = 0;
  i64 nUndecDeferredUpper = 0;               // upper bound of the kind-1 deferred pairs so far (those whose j was suppressed meanwhile are skipped)
  // one beam-path pass over the current pair list (tier 1, tier 2, then the general path -- or its deferral); suppOut == nullptr
  // applies decisions to state (normal round), else records them per pair (tail batch, whose first *firstNew entries are the
  // deferred pairs, already queued for the general path)
  auto run_pairs = [&](unsigned char* suppOut) -> int {
    const unsigned int* first = suppOut ? firstNew : nullptr;
    if (stats) SD_CHECK(hipEventRecord(ev0, s));
    PairQueues q1{spillPairs, &d_cnt->nSpill, exactPairs, &d_cnt->nExact, qCap};
    PairQueues q2{exactPairs, &d_cnt->nExact, exactPairs, &d_cnt->nExact, qCap};   // what tier 2 cannot hold goes to the general path
    int rc;
    if (R <= 32) {
      if (pairOrder) {
        // measured on the 2048^2 bench set (pair kernels incl. the bucketing, ms): emission order 9.9 | 32x32 7.85 | 64x16 7.38 | 64x64 6.70 |
        // 16 local-minima classes x 16x16 8.53: resolution of the offset is what counts (64x64 = whole pixels at radius 10)
        static const int keyMode = sd::tuning_env("SD_NMS_PAIR_KEY", 2);
        static const int modes[6][3] = {{1, 32, 32}, {1, 64, 16}, {1, 64, 64}, {16, 16, 16}, {16, 32, 8}, {4, 32, 32}};
        const int* md = modes[keyMode >= 0 && keyMode < 6 ? keyMode : 2];
        const PairKey key{(const char*)prep, prepStride, 1.f / (4.f * (max_dist + 1.f)), md[0], md[1], md[2]};   // offsets lie in (-2 max_dist, 2 max_dist)
        SD_CHECK(hipMemsetAsync(bucketHist, 0, PAIR_BUCKETS * sizeof(unsigned int), s));
        if (decided)
          hipLaunchKernelGGL(k_pairs_decide, dim3(256 * 8), dim3(256), 0, s, pairs, &d_cnt->nPairs, first, vx, vy, R, props, area, threshold, state, suppOut,
                             decided, &d_cnt->nDecided);
        if (!suppOut && deferFrom > 0 && rounds >= deferFrom)      // few undecided pairs: they wait for the tail batch's sweep launch
          hipLaunchKernelGGL(k_defer_undecided, dim3(256), dim3(256), 0, s, pairs, &d_cnt->nPairs, &d_cnt->nDecided, deferMax, decided, state, dfr, defKind, &d_cnt->nErr);
        // tail batch with deferred undecided pairs: they sit in the list's prefix with decided[] = 0 and are bucketed with the rest
        const unsigned int* bfirst = (suppOut && deferFrom > 0) ? nullptr : first;
        SD_CHECK(hipStreamWaitEvent(s, evPrep, 0));          // the prepared polygons (side stream; complete long before, except in round 1)
        hipLaunchKernelGGL(k_pair_bucket_count, dim3(512), dim3(256), 0, s, pairs, &d_cnt->nPairs, bfirst, d_points, key, bucketHist, decided);
        hipLaunchKernelGGL(k_pair_bucket_scan, dim3(1), dim3(1024), 0, s, bucketHist, bucketHist + PAIR_BUCKETS, nOrdered);
        hipLaunchKernelGGL(k_pair_bucket_scatter, dim3(512), dim3(256), 0, s, pairs, &d_cnt->nPairs, bfirst, d_points, key, bucketHist + PAIR_BUCKETS,
                           pairOrder, qCap, decided);
        SD_LAUNCH_CHECK();
        rc = BeamPath<32, 64>::tier1(pairs, pairOrder, nOrdered, (const unsigned int*)nullptr, prep, area, threshold, state, suppOut, q1, s);
      } else {
        SD_CHECK(hipStreamWaitEvent(s, evPrep, 0));
        rc = BeamPath<32, 64>::tier1(pairs, (const unsigned int*)nullptr, &d_cnt->nPairs, first, prep, area, threshold, state, suppOut, q1, s);
      }
      if (stats) SD_CHECK(hipEventRecord(ev1, s));
      if (!rc) rc = BeamPath<32, 64>::tier2(pairs, spillPairs, &d_cnt->nSpill, (const unsigned int*)nullptr, prep, area, threshold, state, suppOut, q2, s);
    } else {
      SD_CHECK(hipStreamWaitEvent(s, evPrep, 0));
      if (R <= 64) rc = BeamPath<64, 64>::tier2(pairs, (const unsigned int*)nullptr, &d_cnt->nPairs, first, prep, area, threshold, state, suppOut, q2, s);
      else if (R <= 128) rc = BeamPath<128, 32>::tier2(pairs, (const unsigned int*)nullptr, &d_cnt->nPairs, first, prep, area, threshold, state, suppOut, q2, s);
      else rc = BeamPath<256, 16>::tier2(pairs, (const unsigned int*)nullptr, &d_cnt->nPairs, first, prep, area, threshold, state, suppOut, q2, s);
      if (stats) SD_CHECK(hipEventRecord(ev1, s));
    }
    if (rc) return -1;
    if (stats) SD_CHECK(hipEventRecord(ev2, s));
    if (!suppOut && deferOn) {
      hipLaunchKernelGGL(k_defer, dim3(64), dim3(256), 0, s, pairs, exactPairs, &d_cnt->nExact, qCap, dfr, &d_cnt->nErr);
      SD_LAUNCH_CHECK();
    } else if (suppOut && sideGeneral) {
      // tail batch: the deferred pairs (the first nDeferred queue entries) are being evaluated on the helper stream since the batch
      // began; here only what the two tiers added behind them, then join
      hipLaunchKernelGGL(k_count_after, dim3(1), dim3(1), 0, s, nNewExact, &d_cnt->nExact, deferFrom > 0 ? nJoinDef : firstNew);
      SD_LAUNCH_CHECK();
      if (sd::clip_full_pairs(pairs, exactPairs + nDeferred, nNewExact, qCap - (unsigned int)nDeferred, R, vx, vy, area, threshold, state, suppOut, &d_cnt->nErr, s)) return -1;
      SD_CHECK(hipStreamWaitEvent(s, evJoin, 0));
    } else if (sd::clip_full_pairs(pairs, exactPairs, &d_cnt->nExact, qCap, R, vx, vy, area, threshold, state, suppOut, &d_cnt->nErr, s)) return -1;
    if (stats) SD_CHECK(hipEventRecord(ev3, s));
    return 0;
  };
  auto account = [&](const char* what) -> int {
    if (h.nPairs > pairCap || h.nSpill > qCap || h.nExact > qCap) { sd::set_error("sd_nms2d: pair queue overflow (internal error)"); return -1; }
    if (h.nErr) { sd::set_error("sd_nms2d: %u pairs exceeded the general path's fixed capacities or the deferred-pair list (with more than 64 rays: at least that many; the launch stops at the first)", h.nErr); return -1; }
    totalPairs += (i64)h.nPairs; totalExact += h.nExact; totalSpill += h.nSpill; totalDecided += h.nDecided;
    if (stats) {
      float ms = 0, ms2 = 0;
      SD_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); SD_CHECK(hipEventElapsedTime(&ms2, ev2, ev3));
      if (h.nPairs) { ns_pairs += ms * 1e6; ++n_pair_launches; }
      ns_full += ms2 * 1e6;
      if (sd::option(sd::OPT_TRACE)) printf("%s %d: nU=%d nK=%d pairs=%llu decided by the area enclosure=%u spill=%u exact=%u pair_kernel=%.3f ms general_path=%.3f ms\n", what, rounds, h.nU, h.nK, h.nPairs, h.nDecided, h.nSpill, h.nExact, ms, ms2);
    }
    return 0;
  };
  bool forceTail = false;
  while (nU > 0) {
    ++rounds;
    if (tailT >= 0 && ((nU <= tailT && rounds > 1) || forceTail)) {
      // ---- tail batch: every remaining (undecided, undecided) pair at once, then the greedy rounds replayed on the device
      if (!supp) { supp = A.take_n<unsigned char>(pairCap); segStart = A.take_n<unsigned int>(N); segCnt = A.take_n<int>(N); }
      if (!supp || !segStart || !segCnt) return -1;
      SD_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), s));
      SD_CHECK(hipMemsetAsync(supp, 0, pairCap, s));
      const int wg = sd::div_up(nU, 4) < 2048 ? sd::div_up(nU, 4) : 2048;
      if (deferFrom > 0) {
        hipLaunchKernelGGL(k_tail_init2, dim3(64), dim3(256), 0, s, dfr, defKind, pairs, exactPairs, &d_cnt->nPairs, &d_cnt->nExact, firstNew, decided);
        hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(1), 0, s, nJoinDef, &d_cnt->nExact);
      } else
        hipLaunchKernelGGL(k_tail_init, dim3(64), dim3(256), 0, s, dfr, pairs, exactPairs, &d_cnt->nPairs, &d_cnt->nExact, firstNew);
      // The deferred pairs all need the general path (a latency-bound launch of ~1 ms over a few thousand pairs, a fraction of the
      // chip): it starts NOW on the helper stream, next to the emission of the remaining pairs and the two bound-slot tiers; decisions
      // are recorded per pair (supp[]), so the two streams write disjoint bytes.  Joined in run_pairs.
      sideGeneral = deferOn && nDeferred > 0 && nDeferred < (i64)qCap && side != nullptr;
      if (sideGeneral) {
        SD_CHECK(hipEventRecord(evFork, s));
        SD_CHECK(hipStreamWaitEvent(side, evFork, 0));
        if (sd::clip_full_pairs(pairs, exactPairs, deferFrom > 0 ? nJoinDef : firstNew, qCap, R, vx, vy, area, threshold, state, supp, &d_cnt->nErr, side)) return -1;
        SD_CHECK(hipEventRecord(evJoin, side));
      }
      hipLaunchKernelGGL(k_tail_emit, dim3(wg), dim3(256), 0, s, Ucur, nU, state, nbrStart, nbrLow, nbr, f, d_points, bbox, radius, area, pairs,
                         &d_cnt->nPairs, pairCap, segStart, segCnt);
      SD_LAUNCH_CHECK();
      if (run_pairs(supp)) return -1;
      hipEvent_t evr0 = nullptr, evr1 = nullptr;
      const bool tr = stats && sd::option(sd::OPT_TRACE);
      if (tr) { SD_CHECK(hipEventCreate(&evr0)); SD_CHECK(hipEventCreate(&evr1)); SD_CHECK(hipEventRecord(evr0, s)); }
      for (int it = 0; it < 10; ++it)
        hipLaunchKernelGGL(k_tail_step, dim3(sd::div_up(nU, 256)), dim3(256), 0, s, Ucur, nU, state, pairs, supp, segStart, segCnt, pairCap, dfr.head, dfr.next);
      hipLaunchKernelGGL(k_tail_resolve, dim3(1), dim3(1024), 0, s, Ucur, nU, state, pairs, supp, segStart, segCnt, pairCap, &d_cnt->left, dfr.head, dfr.next);
      SD_LAUNCH_CHECK();
      if (tr) { SD_CHECK(hipEventRecord(evr1, s)); SD_CHECK(hipEventSynchronize(evr1)); float ms = 0; SD_CHECK(hipEventElapsedTime(&ms, evr0, evr1));
                printf("tail replay (10 steps + resolve loop): %.3f ms\n", ms); (void)hipEventDestroy(evr0); (void)hipEventDestroy(evr1); }
      unsigned int hDefTotal = (unsigned int)nDeferred;
      SD_CHECK(hipMemcpyAsync(&h, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
      if (deferFrom > 0) SD_CHECK(hipMemcpyAsync(&hDefTotal, dfr.count, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
      SD_CHECK(hipStreamSynchronize(s));
      h.nU = nU; h.nK = 0;
      if (deferFrom > 0) { totalUndecDeferred = (i64)hDefTotal - nDeferred; if ((unsigned long long)hDefTotal <= h.nPairs) h.nPairs -= hDefTotal; }
      else
      if ((i64)h.nPairs >= nDeferred) h.nPairs -= (unsigned long long)nDeferred;      // the deferred pairs were counted in their rounds
      if ((i64)h.nExact >= nDeferred) h.nExact -= (unsigned int)nDeferred;
      if (account("tail batch after round")) return -1;
      if (h.left) { sd::set_error("sd_nms2d: tail batch left candidates undecided (internal error)"); return -1; }
      nU = 0;
      break;
    }
    SD_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), s));
    hipLaunchKernelGGL(k_round_triage, dim3(sd::div_up(nU, 1024)), dim3(1024), 0, s, Ucur, nU, state, waitOn, Unext, K, Sl, (int*)d_cnt, dfr.pend);
    const int wgrid = sd::div_up(nU, 4) < 2048 ? sd::div_up(nU, 4) : 2048;
    hipLaunchKernelGGL(k_round_scan, dim3(wgrid), dim3(256), 0, s, Sl, state, nbrStart, nbrLow, nbr, waitOn, Unext, K, (int*)d_cnt, dfr.pend);
    hipLaunchKernelGGL(k_round_emit, dim3(wgrid), dim3(256), 0, s, K, &d_cnt->nK, state, nbrStart, nbrCount, nbr, f, d_points, bbox,
                       radius, area, pairs, &d_cnt->nPairs, pairCap);
    SD_LAUNCH_CHECK();
    if (run_pairs(nullptr)) return -1;
    SD_CHECK(hipMemcpyAsync(&h, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    if (h.nK == 0 && h.nU > 0) {
      if (deferOn && (nDeferred > 0 || nUndecDeferredUpper > 0) && !forceTail) forceTail = true;      // everything left waits on deferred pairs: run the tail batch now
      else { sd::set_error("sd_nms2d: greedy scan made no progress (internal error)"); return -1; }
    }
    if (deferOn) nDeferred += h.nExact;
    if (deferFrom > 0 && rounds >= deferFrom && h.nPairs > h.nDecided && h.nPairs - h.nDecided <= deferMax) nUndecDeferredUpper += (i64)(h.nPairs - h.nDecided);
    if (account("round")) return -1;
    nU = h.nU;
    int* t = Ucur; Ucur = Unext; Unext = t;
  }
  hipLaunchKernelGGL(k_keep, dim3(sd::div_up(N, 256)), dim3(256), 0, s, state, d_keep, N);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipStreamSynchronize(s));
  if (stats) { stats[0] = totalPairs; stats[1] = totalExact; stats[2] = rounds; stats[3] = totalNbr;
               stats[4] = (int64_t)ns_pairs; stats[5] = n_pair_launches; stats[6] = (int64_t)ns_full; stats[7] = (int64_t)ns_pre;
               stats[8] = totalSpill; stats[9] = totalDecided; stats[10] = totalUndecDeferred; }
  if (verbose) {
    printf("NMS: %lld pair intersections (%lld on the exact-join path), %d greedy rounds, %lld neighbour entries\n",
           (long long)totalPairs, (long long)totalExact, rounds, (long long)totalNbr);
    fflush(stdout);
  }
  return 0;
}

// ---- the legacy variant: c_non_max_suppression_inds_old (stardist2d.cpp:173-386; caller stardist/nms.py:20-74, reference test
// tests/test_nms2D.py:78-110 "old == new").  Input = integer polygons (n, 2, R) (row 0 = y, row 1 = x) sorted by score descending and,
// with max_bbox_search, the map pixel -> polygon id: polygon i is only compared with the polygons j > i found in a window of the map
// around i's bounding box (:285-301); without it, with every j > i (:337).  Not on predict_instances(): kept as the reference keeps it,
// a second statement of the same NMS.  Every pair the reference could evaluate is evaluated once (Clipper-exact sweep, decisions per
// pair), then the greedy order is replayed by one workgroup in index order -- exactly the reference's outer loop.
namespace {
__global__ void __launch_bounds__(256) k_old_build(const int* __restrict__ polys, int N, int R, int* __restrict__ vx, int* __restrict__ vy,
                                                   int4* __restrict__ bbox, float* __restrict__ area, int* __restrict__ gmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int* py = polys + (size_t)i * 2 * R;
  const int* px = py + R;
  int x1 = 0, x2 = 0, y1 = 0, y2 = 0;
  float a = 0.f;
  for (int k = 0; k < R; ++k) {
    const int y = py[k], x = px[k];
    if (k == 0) { x1 = x2 = x; y1 = y2 = y; }
    else { x1 = x < x1 ? x : x1; x2 = x > x2 ? x : x2; y1 = y < y1 ? y : y1; y2 = y > y2 ? y : y2; }   // :229-239
    vx[(size_t)i * R + k] = x; vy[(size_t)i * R + k] = y;
    const int kn = (k + 1 == R) ? 0 : k + 1;
    a += (float)((i64)x * py[kn] - (i64)y * px[kn]);                       // area_from_path :128-138
  }
  area[i] = (float)(0.5 * (double)fabsf(a));
  bbox[i] = make_int4(x1, x2, y1, y2);
  atomicMax(&gmax[0], x2 - x1); atomicMax(&gmax[1], y2 - y1);            // :242-253
}
struct OldSearch { int max_bbox_search, grid_y, grid_x, height, width; };
// wave per polygon i; MODE 0 counts the pairs (i, j) the reference would test, MODE 1 writes them
template <int MODE>
__global__ void __launch_bounds__(256) k_old_pairs(int N, OldSearch o, const int* __restrict__ mapping, const int4* __restrict__ bbox,
                                                   const int* __restrict__ gmax, int* __restrict__ cnt, const i64* __restrict__ start,
                                                   int2* __restrict__ pairs) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= N) return;
  const int4 bi = bbox[i];
  const i64 base = MODE ? start[i] : 0;
  int n = 0;
  auto visit = [&](bool in, int j) {
    const bool hit = in && j > i && j < N && bbox_intersect(bi, bbox[j]);   // :305-311 (j <= i: higher score or no candidate)
    const unsigned long long m = __ballot(hit);
    if (MODE && hit) pairs[base + n + __popcll(m & ((1ull << lane) - 1))] = make_int2(i, j);
    n += __popcll(m);
  };
  if (o.max_bbox_search) {
    const int mx = gmax[0], my = gmax[1];
    const int xs = max((bi.x - mx) / o.grid_x, 0), xe = min((bi.y + mx) / o.grid_x, o.width);       // :285-288 (C division, exclusive end)
    const int ys = max((bi.z - my) / o.grid_y, 0), ye = min((bi.w + my) / o.grid_y, o.height);
    const int w = xe - xs, h = ye - ys;
    if (w > 0 && h > 0) {
      const i64 tot = (i64)w * h;
      for (i64 t = 0; t < tot; t += 64) {
        const i64 q = t + lane;
        const bool in = q < tot;
        int j = -1;
        if (in) { const int jj = ys + (int)(q / w), ii = xs + (int)(q % w); j = mapping[(size_t)jj * o.width + ii]; }
        visit(in, j);
      }
    }
  } else {
    for (int t = i + 1; t < N; t += 64) visit(t + lane < N, t + lane);   // :337
  }
  if (!MODE && lane == 0) cnt[i] = n;
}
// the reference's outer loop (:273-323): i ascending; an unsuppressed i suppresses the j of its pairs with overlap > threshold
__global__ void __launch_bounds__(1024) k_old_replay(int N, const i64* __restrict__ start, const int2* __restrict__ pairs,
                                                     const unsigned char* __restrict__ supp, volatile unsigned char* state) {
  for (int i = 0; i < N; ++i) {
    const i64 beg = start[i], end = start[i + 1];
    if (beg == end) continue;
    if (state[i] != ST_SUPPRESSED)
      for (i64 t = beg + threadIdx.x; t < end; t += blockDim.x) if (supp[t]) state[pairs[t].y] = ST_SUPPRESSED;
    __threadfence_block();
    __syncthreads();
  }
}
}  // namespace

extern "C" int sd_nms2d_old_device(const int32_t* d_polys, int n_polys, int n_rays, const int32_t* d_mapping, int height, int width,
                                   float threshold, int max_bbox_search, int grid_y, int grid_x, int verbose, uint8_t* d_keep,
                                   void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  const int N = n_polys, R = n_rays;
  if (N <= 0) return 0;
  if (R < 1 || R > 256) { sd::set_error("sd_nms2d_old: n_rays=%d unsupported (1..256)", R); return -1; }
  if (max_bbox_search && (!d_mapping || height <= 0 || width <= 0 || grid_y <= 0 || grid_x <= 0)) {
    sd::set_error("sd_nms2d_old: max_bbox_search needs a (height, width) mapping and positive grid steps"); return -1;
  }
  if (verbose) {
    printf("Non Maximum Suppression (2D) ++++ \n");
    printf("NMS: n_polys  = %d \nNMS: n_rays   = %d  \nNMS: thresh   = %.3f \nNMS: max_bbox_search = %d \n", N, R, threshold, max_bbox_search);
    printf("NMS: using HIP (gfx950), every candidate pair by the scan-beam pair kernel, greedy order replayed on the device\n");
  }
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  int* vx = A.take_n<int>((size_t)N * R);
  int* vy = A.take_n<int>((size_t)N * R);
  int4* bbox = A.take_n<int4>(N);
  float* area = A.take_n<float>(N);
  int* gmax = A.take_n<int>(2);
  unsigned char* state = A.take_n<unsigned char>(N);
  int* cnt = A.take_n<int>(N + 1);
  i64* start = A.take_n<i64>(N + 1);
  Counters* d_cnt = (Counters*)A.take(sizeof(Counters));
  if (!vx || !vy || !bbox || !area || !gmax || !state || !cnt || !start || !d_cnt) return -1;
  SD_CHECK(hipMemsetAsync(gmax, 0, 2 * sizeof(int), s));
  SD_CHECK(hipMemsetAsync(state, 0, N, s));
  SD_CHECK(hipMemsetAsync(cnt, 0, (N + 1) * sizeof(int), s));
  SD_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(Counters), s));
  hipLaunchKernelGGL(k_old_build, dim3(sd::div_up(N, 256)), dim3(256), 0, s, d_polys, N, R, vx, vy, bbox, area, gmax);
  SD_LAUNCH_CHECK();
  const OldSearch o{max_bbox_search, grid_y, grid_x, height, width};
  hipLaunchKernelGGL((k_old_pairs<0>), dim3(sd::div_up(N, 4)), dim3(256), 0, s, N, o, d_mapping, bbox, gmax, cnt, (const i64*)nullptr, (int2*)nullptr);
  SD_LAUNCH_CHECK();
  size_t tmpBytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmpBytes, cnt, start, N + 1, s);
  void* scanTmp = A.take(tmpBytes + 256);
  if (!scanTmp) return -1;
  SD_CHECK(hipcub::DeviceScan::ExclusiveSum(scanTmp, tmpBytes, cnt, start, N + 1, s));
  i64 total = 0;
  SD_CHECK(hipMemcpyAsync(&total, start + N, sizeof(i64), hipMemcpyDeviceToHost, s));
  SD_CHECK(hipStreamSynchronize(s));
  if (total < 0 || total >= (i64)(1ll << 30)) {
    sd::set_error("sd_nms2d_old: %lld candidate pairs exceed the capacity of one call (2^30)", (long long)total); return -1;
  }
  if (total > 0) {
    const unsigned int qCap = (unsigned int)total + 64u;
    int2* pairs = A.take_n<int2>((size_t)total + 64);
    unsigned char* supp = A.take_n<unsigned char>((size_t)total + 64);
    unsigned int* spillPairs = A.take_n<unsigned int>(qCap);
    unsigned int* exactPairs = A.take_n<unsigned int>(qCap);
    size_t prepStride;
    if (R <= 32) prepStride = sizeof(sdclip::PolyPrep<32>); else if (R <= 64) prepStride = sizeof(sdclip::PolyPrep<64>);
    else if (R <= 128) prepStride = sizeof(sdclip::PolyPrep<128>); else prepStride = sizeof(sdclip::PolyPrep<256>);
    void* prep = A.take((size_t)N * prepStride);
    if (!pairs || !supp || !spillPairs || !exactPairs || !prep) return -1;
    SD_CHECK(hipMemsetAsync(supp, 0, (size_t)total + 64, s));
    hipLaunchKernelGGL((k_old_pairs<1>), dim3(sd::div_up(N, 4)), dim3(256), 0, s, N, o, d_mapping, bbox, gmax, cnt, (const i64*)start, pairs);
    SD_LAUNCH_CHECK();
    const unsigned long long nP = (unsigned long long)total;
    SD_CHECK(hipMemcpyAsync(&d_cnt->nPairs, &nP, sizeof(nP), hipMemcpyHostToDevice, s));
    PairQueues q1{spillPairs, &d_cnt->nSpill, exactPairs, &d_cnt->nExact, qCap};
    PairQueues q2{exactPairs, &d_cnt->nExact, exactPairs, &d_cnt->nExact, qCap};
    int rc;
    const unsigned int* none = nullptr;
    if (R <= 32) {
      rc = BeamPath<32, 64>::prepare(vx, vy, N, R, prep, s);
      if (!rc) rc = BeamPath<32, 64>::tier1(pairs, none, &d_cnt->nPairs, none, prep, area, threshold, state, supp, q1, s);
      if (!rc) rc = BeamPath<32, 64>::tier2(pairs, spillPairs, &d_cnt->nSpill, none, prep, area, threshold, state, supp, q2, s);
    } else if (R <= 64) {
      rc = BeamPath<64, 64>::prepare(vx, vy, N, R, prep, s);
      if (!rc) rc = BeamPath<64, 64>::tier2(pairs, none, &d_cnt->nPairs, none, prep, area, threshold, state, supp, q2, s);
    } else if (R <= 128) {
      rc = BeamPath<128, 32>::prepare(vx, vy, N, R, prep, s);
      if (!rc) rc = BeamPath<128, 32>::tier2(pairs, none, &d_cnt->nPairs, none, prep, area, threshold, state, supp, q2, s);
    } else {
      rc = BeamPath<256, 16>::prepare(vx, vy, N, R, prep, s);
      if (!rc) rc = BeamPath<256, 16>::tier2(pairs, none, &d_cnt->nPairs, none, prep, area, threshold, state, supp, q2, s);
    }
    if (rc) return -1;
    if (sd::clip_full_pairs(pairs, exactPairs, &d_cnt->nExact, qCap, R, vx, vy, area, threshold, state, supp, &d_cnt->nErr, s)) return -1;
    hipLaunchKernelGGL(k_old_replay, dim3(1), dim3(1024), 0, s, N, (const i64*)start, (const int2*)pairs, (const unsigned char*)supp, state);
    SD_LAUNCH_CHECK();
    Counters h;
    SD_CHECK(hipMemcpyAsync(&h, d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, s));
    SD_CHECK(hipStreamSynchronize(s));
    if (h.nSpill > qCap || h.nExact > qCap) { sd::set_error("sd_nms2d_old: pair queue overflow (internal error)"); return -1; }
    if (h.nErr) { sd::set_error("sd_nms2d_old: %u pairs exceeded the general path's fixed capacities", h.nErr); return -1; }
  }
  hipLaunchKernelGGL(k_keep, dim3(sd::div_up(N, 256)), dim3(256), 0, s, state, d_keep, N);
  SD_LAUNCH_CHECK();
  SD_CHECK(hipStreamSynchronize(s));
  if (verbose) { printf("NMS: %lld candidate pairs evaluated\n", (long long)total); fflush(stdout); }
  return 0;
}

extern "C" int sd_nms2d_old_host(const int32_t* polys, int n_polys, int n_rays, const int32_t* mapping, int height, int width,
                                 float threshold, int max_bbox_search, int grid_y, int grid_x, int verbose, uint8_t* keep) {
  if (n_polys <= 0) return 0;
  int32_t *d_polys = nullptr, *d_map = nullptr;
  uint8_t* d_keep = nullptr;
  const size_t pb = (size_t)n_polys * 2 * n_rays * sizeof(int32_t), mb = max_bbox_search ? (size_t)height * width * sizeof(int32_t) : 0;
  SD_CHECK(hipMalloc(&d_polys, pb));
  if (mb) SD_CHECK(hipMalloc(&d_map, mb));
  SD_CHECK(hipMalloc(&d_keep, n_polys));
  int rc = -1;
  do {
    if (hipMemcpy(d_polys, polys, pb, hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (mb && hipMemcpy(d_map, mapping, mb, hipMemcpyHostToDevice) != hipSuccess) { sd::set_error("H2D failed"); break; }
    if (sd_nms2d_old_device(d_polys, n_polys, n_rays, d_map, height, width, threshold, max_bbox_search, grid_y, grid_x, verbose, d_keep, nullptr)) break;
    if (hipMemcpy(keep, d_keep, n_polys, hipMemcpyDeviceToHost) != hipSuccess) { sd::set_error("D2H failed"); break; }
    rc = 0;
  } while (0);
  (void)hipFree(d_polys); if (d_map) (void)hipFree(d_map); (void)hipFree(d_keep);
  return rc;
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
