// clip_sweep.h -- integer scan-beam polygon intersection, one pair per thread.
//
// What this is
// ------------
// The reference's 2D NMS overlap is   area(A ∩ B) / min(area A, area B)   where the
// intersection is computed by the vendored Clipper 6.4.2 on *integer* vertices
// (stardist/lib/stardist2d.cpp:152-165: clip = A, subject = B, ctIntersection,
// pftNonZero / pftNonZero) and the area is the shoelace sum over Clipper's output paths
// (stardist2d.cpp:128-138).  Clipper snaps every edge-edge crossing to the integer lattice
// with scan-beam dependent clamping (external/clipper/clipper.cpp:615-689), so its area
// differs from the exact-arithmetic area by up to O(perimeter * 0.5 px) -- percent level
// for radius-10 polygons -- and survivor parity with the reference requires reproducing
// that arithmetic, not merely "an" intersection area.
//
// This header restates the published Vatti/Clipper scan-beam sweep for exactly that one
// use (two closed paths, intersection, non-zero fill, no poly-tree, no strictly-simple,
// no preserve-collinear) in a form that runs as ONE GPU THREAD PER PAIR:
//   * fixed-capacity index arrays instead of heap nodes and pointers (private/scratch
//     memory, no allocation),
//   * no output point lists at all: every output ring is kept as {front point, back
//     point, running shoelace sum}; points are only ever appended at either end and rings
//     are concatenated/reversed (clipper.cpp:2367-2460, 2463-2499), so the shoelace sum is
//     maintained incrementally in int64, exactly,
//   * libstdc++'s std::sort is restated (introsort + final insertion sort) because the
//     order of equal-Y local minima / intersections is decided by it
//     (clipper.cpp:1251, 2940).
// Host and device share this code (plain C++, no STL) so the CPU test harness can compare
// it against the compiled reference on millions of pairs without a GPU.
//
// Not restated (documented deviations):
//   * JoinCommonEdges / JoinPoints (clipper.cpp:3458-3783).  Joins only merge or split
//     output rings along shared edges; the reference sums |area| per ring, so a join can
//     change the result only if it merges rings of opposite orientation (a hole touching
//     its outer ring), which needs a self-overlapping input polygon.  The side effects of
//     the join *detection* code that do matter (extra AddOutPt calls) are kept.
//   * float accumulation order of stardist2d.cpp:128-138: the reference sums int64 cross
//     products into a float; this is exact (order independent) while every partial sum is
//     < 2^24.  `sum_abs_terms` is returned so callers can detect pairs beyond that bound.
#pragma once
#include <stdint.h>

// SD_HD : small helpers, always inlined.  SD_HDN: the large building blocks of the sweep are compiled ONCE per kernel
// (real calls): fully inlined the pair kernel is > 230 KB of code and thrashes the 64 KB instruction cache that two
// CUs share, which -- not memory latency -- dominated the per-pair latency (rocprof + code-size analysis).
#if defined(__HIPCC__)
#define SD_HD __host__ __device__ __forceinline__
#if defined(SD_SWEEP_INLINE)
#define SD_HDN __host__ __device__ __forceinline__
#else
#define SD_HDN __host__ __device__ __noinline__
#endif
#else
#define SD_HD inline
#define SD_HDN inline
#endif
// SD_BLK: a large block of the sweep with ONE call site -- part of its caller (no code growth, no call, and no scratch round trip for the
// return address a calling function has to keep: clip_beam.h, note at add_out_pt_t).  -DSD_BEAM_BLOCKS_AS_CALLS: every block a real call.
#if defined(SD_BEAM_BLOCKS_AS_CALLS)
#define SD_BLK SD_HDN
#else
#define SD_BLK SD_HD
#endif

namespace sdclip {

typedef long long i64;

enum { kUnassigned = -1 };
enum { kLeft = 1, kRight = 2 };
enum { kClip = 1, kSubject = 0 };   // PolyType: ptSubject = 0, ptClip = 1
enum {
  ST_OK = 0,
  ST_OVERFLOW_IL = 1,    // more intersections in one scan-beam than capacity
  ST_OVERFLOW_REC = 2,   // more output rings than capacity
  ST_SORT_DEPTH = 4,     // introsort depth limit hit (heap-sort fallback not restated)
  ST_FAIL = 8,           // Clipper itself would have failed (FixupIntersectionOrder / DoMaxima)
  ST_ITER = 16           // safety iteration bound tripped
};

#define SD_HORIZONTAL (-1.0E+40)

SD_HD i64 sd_round(double v) { return (v < 0) ? (i64)(v - 0.5) : (i64)(v + 0.5); }  // clipper.cpp:136-140
SD_HD i64 sd_cross(int ax, int ay, int bx, int by) { return (i64)ax * by - (i64)ay * bx; }
SD_HD i64 sd_abs64(i64 v) { return v < 0 ? -v : v; }

// ---- libstdc++ std::sort restated (bits/stl_algo.h: __sort, __introsort_loop,
// __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort).
// Key = int (the Y); payload moves with it. Comparator: comp(a,b) := key(b) < key(a)
// (LocMinSorter clipper.cpp:125-131, IntersectListSort clipper.cpp:2921-2924).
template <typename T>
struct StdSort {
  static SD_HD bool comp(const T& a, const T& b) { return b.y < a.y; }
  static SD_HD void swp(T& a, T& b) { T t = a; a = b; b = t; }
  static SD_HD void unguarded_linear_insert(T* a, int last) {
    T val = a[last];
    int next = last - 1;
    while (comp(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
  }
  static SD_HD void insertion_sort(T* a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
      if (comp(a[i], a[first])) {
        T val = a[i];
        for (int k = i; k > first; --k) a[k] = a[k - 1];
        a[first] = val;
      } else unguarded_linear_insert(a, i);
    }
  }
  static SD_HD void move_median_to_first(T* arr, int result, int a, int b, int c) {
    if (comp(arr[a], arr[b])) {
      if (comp(arr[b], arr[c])) swp(arr[result], arr[b]);
      else if (comp(arr[a], arr[c])) swp(arr[result], arr[c]);
      else swp(arr[result], arr[a]);
    } else if (comp(arr[a], arr[c])) swp(arr[result], arr[a]);
    else if (comp(arr[b], arr[c])) swp(arr[result], arr[c]);
    else swp(arr[result], arr[b]);
  }
  static SD_HD int unguarded_partition(T* a, int first, int last, int pivot) {
    while (true) {
      while (comp(a[first], a[pivot])) ++first;
      --last;
      while (comp(a[pivot], a[last])) --last;
      if (!(first < last)) return first;
      swp(a[first], a[last]);
      ++first;
    }
  }
  static SD_HD int lg(int n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }
  // returns false if the depth limit was hit (heap-sort fallback not restated)
  static SD_HDN bool sort(T* a, int n) {
    if (n <= 1) return true;
    bool ok = true;
    if (n > 16) {
      // explicit stack replaces the recursion of __introsort_loop (recursion on the right
      // part, loop on the left part -- same order of operations on disjoint ranges).
      int stk_first[32], stk_last[32], stk_depth[32];
      int sp = 0;
      stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = lg(n) * 2; sp = 1;
      while (sp > 0) {
        --sp;
        int first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
        while (last - first > 16) {
          if (depth == 0) { ok = false; break; }
          --depth;
          int mid = first + (last - first) / 2;
          move_median_to_first(a, first, first + 1, mid, last - 1);
          int cut = unguarded_partition(a, first + 1, last, first);
          if (sp < 32) { stk_first[sp] = cut; stk_last[sp] = last; stk_depth[sp] = depth; ++sp; }
          else ok = false;
          last = cut;
        }
      }
      insertion_sort(a, 0, 16);
      for (int i = 16; i != n; ++i) unguarded_linear_insert(a, i);
    } else insertion_sort(a, 0, n);
    return ok;
  }
};


// ---------------------------------------------------------------------------------------------
// Storage policies.  The sweep's per-pair state lives in small index arrays; WHERE they live decides the
// latency of the (inherently serial) sweep:
//   PlainStorage  : ordinary member arrays -> GPU private (scratch) memory; unlimited threads, ~L2 latency
//   LdsStorage<S> : arrays interleaved across the S threads of a workgroup in LDS (element i of thread t at
//                   base[i*S + t], conflict-free), edge slopes recomputed instead of stored; few threads
//                   per CU but ~5x lower latency per dependent access -- used for the small, latency-bound
//                   rounds of the greedy NMS scan.
struct PlainStorage {
  typedef short idx_t;
  template <class T, int N> static constexpr unsigned region() { return 0; }
  template <class T, int N, unsigned OFF> struct Arr {
    T v[N];
    SD_HD T& operator[](int i) { return v[i]; }
    SD_HD const T& operator[](int i) const { return v[i]; }
  };
  template <int N, unsigned OBX, unsigned OBY, unsigned OTX, unsigned OTY> struct DxArr : Arr<double, N, 0> {};
  template <class T, unsigned OFF> using Scalar = T;
  // clip_beam.h: coordinate arrays (may be stored relative to a per-pair origin, see LdsStorage16) and the slot slopes
  template <int N> static constexpr unsigned coord_region() { return 0; }
  template <int N> static constexpr unsigned slope_region() { return 0; }
  template <int N, unsigned OFF, unsigned OORG, unsigned OST> using CoordArr = Arr<int, N, OFF>;
  template <int N, unsigned OFF, unsigned OBX, unsigned OBY, unsigned OTX, unsigned OTY> using SlopeArr = Arr<double, N, OFF>;
};

// LDS-interleaved storage: element i of thread t of array A lives at  lds_base + OFF_A + (i*STRIDE + t)*sizeof(T).
// The arrays are STATELESS (offset = template constant, t = threadIdx.x, base = start of dynamic LDS), so an access
// never needs a pointer fetched from the (scratch-resident) sweep object.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ char* sd_lds_base() { extern __shared__ __attribute__((aligned(16))) char sd_lds_dyn[]; return sd_lds_dyn; }
__device__ __forceinline__ int sd_lds_tid() { return (int)threadIdx.x; }
#else
struct HostLds { static char*& base() { static thread_local char* b = nullptr; return b; } static int& tid() { static thread_local int t = 0; return t; } };
inline char* sd_lds_base() { return HostLds::base(); }
inline int sd_lds_tid() { return HostLds::tid(); }
#endif
template <int STRIDE>
struct LdsStorage {
  typedef signed char idx_t;
  template <class T, int N> static constexpr unsigned region() { return (unsigned)((N * STRIDE * sizeof(T) + 15u) & ~15u); }
  template <class T, int N, unsigned OFF> struct Arr {
    SD_HD T& operator[](int i) const { return *(T*)(sd_lds_base() + OFF + (unsigned)(i * STRIDE + sd_lds_tid()) * (unsigned)sizeof(T)); }
  };
  template <int N, unsigned OBX, unsigned OBY, unsigned OTX, unsigned OTY> struct DxArr {   // slope = f(bot, top): recomputed, never stored
    struct Ref {
      int e;
      SD_HD operator double() const {
        const Arr<int, N, OBX> bx; const Arr<int, N, OBY> by; const Arr<int, N, OTX> tx; const Arr<int, N, OTY> ty;
        const long long dy = (long long)ty[e] - by[e];
        if (dy == 0) return SD_HORIZONTAL;
        return (double)((long long)tx[e] - bx[e]) / (double)dy;
      }
      SD_HD void operator=(double) const {}
    };
    SD_HD Ref operator[](int e) const { Ref r; r.e = e; return r; }
  };
  // scalar sweep state in LDS as well: inside the (non-inlined) building blocks a plain member would be a load
  // from the scratch-resident object on every use
  template <class T, unsigned OFF> struct Scalar {
    SD_HD T& ref() const { return *(T*)(sd_lds_base() + OFF + (unsigned)sd_lds_tid() * (unsigned)sizeof(T)); }
    SD_HD operator T() const { return ref(); }
    SD_HD T operator=(T v) const { ref() = v; return v; }
    SD_HD T operator=(const Scalar& o) const { const T v = o.ref(); ref() = v; return v; }
    SD_HD T operator+=(T v) const { return ref() += v; }
    SD_HD T operator-=(T v) const { return ref() -= v; }
    SD_HD T operator|=(T v) const { return ref() |= v; }
    SD_HD T operator++() const { return ++ref(); }
    SD_HD T operator++(int) const { return ref()++; }
    SD_HD T operator--() const { return --ref(); }
    SD_HD T operator--(int) const { return ref()--; }
  };
  template <int N> static constexpr unsigned coord_region() { return region<int, N>(); }
  template <int N> static constexpr unsigned slope_region() { return region<double, N>(); }
  template <int N, unsigned OFF, unsigned OORG, unsigned OST> using CoordArr = Arr<int, N, OFF>;
  template <int N, unsigned OFF, unsigned OBX, unsigned OBY, unsigned OTX, unsigned OTY> using SlopeArr = Arr<double, N, OFF>;
};

// LdsStorage with the bound-slot sweep's COORDINATES held as 16-bit offsets from a per-pair origin and its slopes recomputed instead of
// stored (clip_beam.h).  The pair kernel is latency-bound and LDS capacity decides how many pairs a CU has in flight: 636 bytes per
// pair = four waves per CU; with 16-bit coordinates (bot / top / cur of the K slots, intersection points, ring end points: 152 bytes
// less) and no stored slopes (64 bytes less) it is 420 bytes = six waves.  The arithmetic is unchanged: a coordinate reads as
// origin + offset (exact), a slope as the same double quotient of the same integer differences that the stored value was computed
// from (es_dx).  A value that does not fit 16 bits raises `ST_OVERFLOW_AEL` through the status word at OST: the pair is then
// re-run by the next tier, which keeps 32-bit coordinates.
enum { ST_REL16_OVERFLOW = 32 };      // = ST_OVERFLOW_AEL of clip_beam.h (a capacity flag: the pair spills to tier 2)
template <int STRIDE>
struct LdsStorage16 : LdsStorage<STRIDE> {
  typedef LdsStorage<STRIDE> Base;
  template <int N> static constexpr unsigned coord_region() { return Base::template region<short, N>(); }
  template <int N> static constexpr unsigned slope_region() { return 0; }
  template <int N, unsigned OFF, unsigned OORG, unsigned OST> struct CoordArr {
    struct Ref {
      int i;
      SD_HD short& raw() const { return *(short*)(sd_lds_base() + OFF + (unsigned)(i * STRIDE + sd_lds_tid()) * 2u); }
      SD_HD int org() const { return *(const int*)(sd_lds_base() + OORG + (unsigned)sd_lds_tid() * 4u); }
      SD_HD operator int() const { return org() + (int)raw(); }
      SD_HD int operator=(int v) const {
        const int r = v - org();
        if (r != (int)(short)r) *(int*)(sd_lds_base() + OST + (unsigned)sd_lds_tid() * 4u) |= ST_REL16_OVERFLOW;
        raw() = (short)r;
        return v;
      }
      SD_HD int operator=(const Ref& o) const { return *this = (int)o; }
    };
    SD_HD Ref operator[](int i) const { Ref r; r.i = i; return r; }
    SD_HD static int rel(int i) { return (int)*(const short*)(sd_lds_base() + OFF + (unsigned)(i * STRIDE + sd_lds_tid()) * 2u); }
  };
  template <int N, unsigned OFF, unsigned OBX, unsigned OBY, unsigned OTX, unsigned OTY> struct SlopeArr {   // slope = f(bot, top): recomputed, never stored
    struct Ref {
      int e;
      SD_HD operator double() const {
        // differences of the stored 16-bit offsets = differences of the coordinates (same origin)
        const long long dy = (long long)CoordArr<N, OTY, 0, 0>::rel(e) - CoordArr<N, OBY, 0, 0>::rel(e);
        if (dy == 0) return SD_HORIZONTAL;
        return (double)((long long)CoordArr<N, OTX, 0, 0>::rel(e) - CoordArr<N, OBX, 0, 0>::rel(e)) / (double)dy;
      }
      SD_HD void operator=(double) const {}
    };
    SD_HD Ref operator[](int e) const { Ref r; r.e = e; return r; }
  };
};

struct LocMin { int y; short left, right; };
struct INode { int y; int x; short e1, e2; };

// SweepCore: everything of the sweep that does not depend on how output rings are stored.
// D (CRTP) supplies the output side:
//   int  out_add_pt(e, x, y)          AddOutPt :2463-2499, returns a point handle (or -1)
//   void out_append(e1, e2)           AppendPolygon :2367-2460
//   void out_ring_closed(r)           both edges of a local maximum carry ring r (:1888-1892)
//   void out_add_join(op1, op2, x, y) AddJoin :1942-1949
//   int  out_last_pt(e), out_pt_x(op) GetLastOutPt :2502-2509
// MAXV: max vertices per input polygon; MAXIL: intersection-node capacity per scan-beam.
template <class D, class P, int MAXV, int MAXIL>
struct SweepCore {
  enum { NE = 2 * MAXV };
  SD_HD D& self() { return *static_cast<D*>(this); }
  // ---- edges (index = vertex slot; polygon A uses [0,MAXV), polygon B [MAXV,2*MAXV))
  typedef typename P::idx_t idx_t;
  // storage offsets (only meaningful for LdsStorage; all zero-sized for PlainStorage)
  static constexpr unsigned RI = P::template region<int, NE>(), RX = P::template region<idx_t, NE>(),
                            RS = P::template region<short, NE>(), RB = P::template region<signed char, NE>();
  static constexpr unsigned O_BOTX = 0, O_BOTY = O_BOTX + RI, O_TOPX = O_BOTY + RI, O_TOPY = O_TOPX + RI, O_CURX = O_TOPY + RI,
                            O_CURY = O_CURX + RI, O_NXT = O_CURY + RI, O_PRV = O_NXT + RX, O_LML = O_PRV + RX, O_ANEXT = O_LML + RX,
                            O_APREV = O_ANEXT + RX, O_SNEXT = O_APREV + RX, O_SPREV = O_SNEXT + RX, O_WCNT = O_SPREV + RX,
                            O_WCNT2 = O_WCNT + RS, O_OUTIDX = O_WCNT2 + RS, O_PTYP = O_OUTIDX + RX, O_SIDE = O_PTYP + RB,
                            O_WDELTA = O_SIDE + RB, O_SB = O_WDELTA + RB, O_SC = O_SB + P::template region<int, NE + 4>();
  static constexpr unsigned R1 = P::template region<int, 1>();
  static constexpr unsigned O_NLM = O_SC, O_CURLM = O_NLM + R1, O_NSB = O_CURLM + R1, O_NIL = O_NSB + R1, O_AEL = O_NIL + R1, O_SEL = O_AEL + R1,
                            O_STATUS = O_SEL + R1, O_NJOINS = O_STATUS + R1, O_NGJ = O_NJOINS + R1, O_CORE_END = O_NGJ + R1;
  typename P::template Arr<int, NE, O_BOTX> botx; typename P::template Arr<int, NE, O_BOTY> boty;
  typename P::template Arr<int, NE, O_TOPX> topx; typename P::template Arr<int, NE, O_TOPY> topy;
  typename P::template Arr<int, NE, O_CURX> curx; typename P::template Arr<int, NE, O_CURY> cury;
  typename P::template DxArr<NE, O_BOTX, O_BOTY, O_TOPX, O_TOPY> dx;
  typename P::template Arr<idx_t, NE, O_NXT> nxt; typename P::template Arr<idx_t, NE, O_PRV> prv; typename P::template Arr<idx_t, NE, O_LML> lml;
  typename P::template Arr<idx_t, NE, O_ANEXT> anext; typename P::template Arr<idx_t, NE, O_APREV> aprev;
  typename P::template Arr<idx_t, NE, O_SNEXT> snext; typename P::template Arr<idx_t, NE, O_SPREV> sprev;
  typename P::template Arr<short, NE, O_WCNT> wcnt; typename P::template Arr<short, NE, O_WCNT2> wcnt2;
  typename P::template Arr<idx_t, NE, O_OUTIDX> outidx;
  typename P::template Arr<signed char, NE, O_PTYP> ptyp; typename P::template Arr<signed char, NE, O_SIDE> side;
  typename P::template Arr<signed char, NE, O_WDELTA> wdelta;
  // ---- local minima, scan-beam, intersections
  LocMin lm[NE];
  typename P::template Scalar<int, O_NLM> n_lm; typename P::template Scalar<int, O_CURLM> cur_lm;
  typename P::template Arr<int, NE + 4, O_SB> sb;
  typename P::template Scalar<int, O_NSB> n_sb;
  INode il[MAXIL];
  typename P::template Scalar<int, O_NIL> n_il;
  typename P::template Scalar<int, O_AEL> ael; typename P::template Scalar<int, O_SEL> sel;   // heads (-1 = empty)
  typename P::template Scalar<int, O_STATUS> status;
  typename P::template Scalar<int, O_NJOINS> n_joins;   // number of AddJoin calls the reference would have made
  int gjop[16], gjx1[16], gjx2[16], gjy2[16];   // ghost joins of the current scan-line: OutPt1, OutPt1.X, OffPt  :1968-1975
  typename P::template Scalar<int, O_NGJ> n_gj;

  // ------------------------------------------------------------------ helpers
  SD_HD bool is_horz(int e) const { return dx[e] == SD_HORIZONTAL; }
  SD_HD i64 top_x(int e, i64 y) const {                                    // clipper.cpp:615-619
    return (y == topy[e]) ? (i64)topx[e] : (i64)botx[e] + sd_round(dx[e] * (double)(y - boty[e]));
  }
  SD_HD void set_dx(int e) {                                               // clipper.cpp:591-596
    i64 dy = (i64)topy[e] - boty[e];
    if (dy == 0) dx[e] = SD_HORIZONTAL;
    else dx[e] = (double)((i64)topx[e] - botx[e]) / (double)dy;
  }
  SD_HD void reverse_horizontal(int e) { int t = topx[e]; topx[e] = botx[e]; botx[e] = t; }
  static SD_HD bool slopes_equal3(i64 x1, i64 y1, i64 x2, i64 y2, i64 x3, i64 y3) {  // :554-563
    return (y1 - y2) * (x2 - x3) == (x1 - x2) * (y2 - y3);
  }
  static SD_HD bool slopes_equal4(i64 x1, i64 y1, i64 x2, i64 y2, i64 x3, i64 y3, i64 x4, i64 y4) {  // :566-575
    return (y1 - y2) * (x3 - x4) == (x1 - x2) * (y3 - y4);
  }
  SD_HD bool slopes_equal_e(int e1, int e2) const {                        // :541-551
    return ((i64)topy[e1] - boty[e1]) * ((i64)topx[e2] - botx[e2]) ==
           ((i64)topx[e1] - botx[e1]) * ((i64)topy[e2] - boty[e2]);
  }
  static SD_HD bool horz_segments_overlap(i64 a1, i64 a2, i64 b1, i64 b2) {  // :872-877
    if (a1 > a2) { i64 t = a1; a1 = a2; a2 = t; }
    if (b1 > b2) { i64 t = b1; b1 = b2; b2 = t; }
    return (a1 < b2) && (b1 < a2);
  }
  SD_HD void add_join(int op1, int op2, int offx, int offy) { ++n_joins; self().out_add_join(op1, op2, offx, offy); }
  SD_HD void add_ghost_join(int op, int x1, int x2, int y2) {
    if (n_gj < 16) { gjop[n_gj] = op; gjx1[n_gj] = x1; gjx2[n_gj] = x2; gjy2[n_gj] = y2; ++n_gj; }
    else { ++n_joins; status |= ST_OVERFLOW_REC; }
  }
  SD_HD void horz_joins(int horz, int op1) {                                // :2721-2732, 2774-2785
    for (int h = sel; h >= 0; h = snext[h])
      if (outidx[h] >= 0 && horz_segments_overlap(botx[horz], topx[horz], botx[h], topx[h]))
        add_join(self().out_last_pt(h), op1, topx[h], topy[h]);
  }
  SD_HD void insert_scanbeam(int y) {
    for (int i = 0; i < n_sb; ++i) if (sb[i] == y) return;   // duplicates are popped together (:1341-1348)
    if (n_sb < NE + 4) sb[n_sb++] = y; else status |= ST_ITER;
  }
  SD_HD bool pop_scanbeam(int& y) {
    if (n_sb == 0) return false;
    int k = 0;
    for (int i = 1; i < n_sb; ++i) if (sb[i] > sb[k]) k = i;
    y = sb[k];
    sb[k] = sb[--n_sb];
    return true;
  }

  // ------------------------------------------------------------------ AddPath (closed)  :1045-1221
  SD_HD int find_next_loc_min(int E) const {                                // :911-925
    for (;;) {
      while (botx[E] != botx[prv[E]] || boty[E] != boty[prv[E]] ||
             (curx[E] == topx[E] && cury[E] == topy[E])) E = nxt[E];
      if (!is_horz(E) && !is_horz(prv[E])) break;
      while (is_horz(prv[E])) E = prv[E];
      int E2 = E;
      while (is_horz(E)) E = nxt[E];
      if (topy[E] == boty[prv[E]]) continue;   // just an intermediate horizontal
      if (botx[prv[E2]] < botx[E]) E = E2;
      break;
    }
    return E;
  }

  SD_HDN int process_bound(int E, bool fwd) {                                // :928-1042 (no skip edges)
    int Result = E, Horz;
    if (is_horz(E)) {
      int EStart = fwd ? prv[E] : nxt[E];
      if (is_horz(EStart)) {
        if (botx[EStart] != botx[E] && topx[EStart] != botx[E]) reverse_horizontal(E);
      } else if (botx[EStart] != botx[E]) reverse_horizontal(E);
    }
    int EStart = E;
    if (fwd) {
      while (topy[Result] == boty[nxt[Result]]) Result = nxt[Result];
      if (is_horz(Result)) {
        Horz = Result;
        while (is_horz(prv[Horz])) Horz = prv[Horz];
        if (topx[prv[Horz]] > topx[nxt[Result]]) Result = prv[Horz];
      }
      while (E != Result) {
        lml[E] = nxt[E];
        if (is_horz(E) && E != EStart && botx[E] != topx[prv[E]]) reverse_horizontal(E);
        E = nxt[E];
      }
      if (is_horz(E) && E != EStart && botx[E] != topx[prv[E]]) reverse_horizontal(E);
      Result = nxt[Result];
    } else {
      while (topy[Result] == boty[prv[Result]]) Result = prv[Result];
      if (is_horz(Result)) {
        Horz = Result;
        while (is_horz(nxt[Horz])) Horz = nxt[Horz];
        if (topx[nxt[Horz]] == topx[prv[Result]] || topx[nxt[Horz]] > topx[prv[Result]]) Result = nxt[Horz];
      }
      while (E != Result) {
        lml[E] = prv[E];
        if (is_horz(E) && E != EStart && botx[E] != topx[nxt[E]]) reverse_horizontal(E);
        E = prv[E];
      }
      if (is_horz(E) && E != EStart && botx[E] != topx[nxt[E]]) reverse_horizontal(E);
      Result = prv[Result];
    }
    return Result;
  }

  // xs/ys: n integer vertices. base: first edge slot. Returns false if the path is rejected.
  template <typename XT>
  SD_HDN bool add_path(const XT* xs, const XT* ys, int n, int polytype, int base) {
    int highI = n - 1;
    while (highI > 0 && xs[highI] == xs[0] && ys[highI] == ys[0]) --highI;
    while (highI > 0 && xs[highI] == xs[highI - 1] && ys[highI] == ys[highI - 1]) --highI;
    if (highI < 2) return false;
    for (int i = 0; i <= highI; ++i) {
      int e = base + i;
      curx[e] = (int)xs[i]; cury[e] = (int)ys[i];
      nxt[e] = (short)(base + (i == highI ? 0 : i + 1));
      prv[e] = (short)(base + (i == 0 ? highI : i - 1));
      lml[e] = -1; anext[e] = aprev[e] = snext[e] = sprev[e] = -1;
      outidx[e] = kUnassigned; wcnt[e] = wcnt2[e] = 0; wdelta[e] = 0; side[e] = 0; ptyp[e] = 0;
      botx[e] = boty[e] = topx[e] = topy[e] = 0; dx[e] = 0;
    }
    int eStart = base, E = base, eLoopStop = base;
    for (;;) {   // remove duplicate vertices and collinear edges
      if (curx[E] == curx[nxt[E]] && cury[E] == cury[nxt[E]]) {
        if (E == nxt[E]) break;
        if (E == eStart) eStart = nxt[E];
        int en = nxt[E]; nxt[prv[E]] = (short)en; prv[en] = prv[E]; E = en;   // RemoveEdge
        eLoopStop = E;
        continue;
      }
      if (prv[E] == nxt[E]) break;
      else if (slopes_equal3(curx[prv[E]], cury[prv[E]], curx[E], cury[E], curx[nxt[E]], cury[nxt[E]])) {
        if (E == eStart) eStart = nxt[E];
        int en = nxt[E], ep = prv[E]; nxt[ep] = (short)en; prv[en] = (short)ep;  // RemoveEdge
        E = ep;
        eLoopStop = E;
        continue;
      }
      E = nxt[E];
      if (E == eLoopStop) break;
    }
    if (prv[E] == nxt[E]) return false;

    bool isFlat = true;
    E = eStart;
    do {   // InitEdge2 :729-742
      int en = nxt[E];
      if (cury[E] >= cury[en]) { botx[E] = curx[E]; boty[E] = cury[E]; topx[E] = curx[en]; topy[E] = cury[en]; }
      else { topx[E] = curx[E]; topy[E] = cury[E]; botx[E] = curx[en]; boty[E] = cury[en]; }
      set_dx(E);
      ptyp[E] = (signed char)polytype;
      E = nxt[E];
      if (isFlat && cury[E] != cury[eStart]) isFlat = false;
    } while (E != eStart);
    if (isFlat) return false;

    if (botx[prv[E]] == topx[prv[E]] && boty[prv[E]] == topy[prv[E]]) E = nxt[E];
    int EMin = -1;
    int guard = 0;
    for (;;) {
      E = find_next_loc_min(E);
      if (E == EMin) break;
      else if (EMin < 0) EMin = E;
      if (++guard > 2 * MAXV + 2) { status |= ST_ITER; break; }
      LocMin m;
      m.y = boty[E];
      bool leftFwd;
      if (dx[E] < dx[prv[E]]) { m.left = prv[E]; m.right = (short)E; leftFwd = false; }
      else { m.left = (short)E; m.right = prv[E]; leftFwd = true; }
      wdelta[m.left] = (nxt[m.left] == m.right) ? -1 : 1;
      wdelta[m.right] = (signed char)(-wdelta[m.left]);
      E = process_bound(m.left, leftFwd);
      int E2 = process_bound(m.right, !leftFwd);
      if (n_lm < NE) lm[n_lm++] = m; else status |= ST_ITER;
      if (!leftFwd) E = E2;
    }
    return true;
  }

  // ------------------------------------------------------------------ output (delegated to D)
  SD_HD int add_out_pt(int e, int px, int py) { return self().out_add_pt(e, px, py); }
  SD_HDN void add_local_max_poly(int e1, int e2, int px, int py) {           // :1884-1897
    add_out_pt(e1, px, py);
    if (outidx[e1] == outidx[e2]) {
      if (outidx[e1] >= 0) self().out_ring_closed(outidx[e1]);
      outidx[e1] = kUnassigned; outidx[e2] = kUnassigned;
    } else if (outidx[e1] < outidx[e2]) self().out_append(e1, e2);
    else self().out_append(e2, e1);
  }
  SD_HDN int add_local_min_poly(int e1, int e2, int px, int py) {            // :1841-1881
    int e, prevE, result;
    if (is_horz(e2) || dx[e1] > dx[e2]) {
      result = add_out_pt(e1, px, py);
      outidx[e2] = outidx[e1];
      side[e1] = kLeft; side[e2] = kRight;
      e = e1;
      prevE = (aprev[e] == e2) ? aprev[e2] : aprev[e];
    } else {
      result = add_out_pt(e2, px, py);
      outidx[e1] = outidx[e2];
      side[e1] = kRight; side[e2] = kLeft;
      e = e2;
      prevE = (aprev[e] == e1) ? aprev[e1] : aprev[e];
    }
    if (prevE >= 0 && outidx[prevE] >= 0 && topy[prevE] < py && topy[e] < py) {
      i64 xPrev = top_x(prevE, py), xE = top_x(e, py);
      if (xPrev == xE && wdelta[e] != 0 && wdelta[prevE] != 0 &&
          slopes_equal4(xPrev, py, topx[prevE], topy[prevE], xE, py, topx[e], topy[e])) {
        const int outPt = add_out_pt(prevE, px, py);
        add_join(result, outPt, topx[e], topy[e]);
      }
    }
    return result;
  }

  // ------------------------------------------------------------------ AEL / SEL lists
  SD_HD bool e2_inserts_before_e1(int e1, int e2) const {                   // :3278-3287
    if (curx[e2] == curx[e1]) {
      if (topy[e2] > topy[e1]) return (i64)topx[e2] < top_x(e1, topy[e2]);
      else return (i64)topx[e1] > top_x(e2, topy[e1]);
    } else return curx[e2] < curx[e1];
  }
  SD_HDN void insert_edge_into_ael(int edge, int startEdge) {                // :3319-3345
    if (ael < 0) { aprev[edge] = -1; anext[edge] = -1; ael = (short)edge; }
    else if (startEdge < 0 && e2_inserts_before_e1(ael, edge)) {
      aprev[edge] = -1; anext[edge] = ael; aprev[ael] = (short)edge; ael = (short)edge;
    } else {
      if (startEdge < 0) startEdge = ael;
      while (anext[startEdge] >= 0 && !e2_inserts_before_e1(anext[startEdge], edge)) startEdge = anext[startEdge];
      anext[edge] = anext[startEdge];
      if (anext[startEdge] >= 0) aprev[anext[startEdge]] = (short)edge;
      aprev[edge] = (short)startEdge;
      anext[startEdge] = (short)edge;
    }
  }
  SD_HD void delete_from_ael(int e) {                                       // :1367-1377
    int p = aprev[e], n = anext[e];
    if (p < 0 && n < 0 && e != ael) return;
    if (p >= 0) anext[p] = (short)n; else ael = (short)n;
    if (n >= 0) aprev[n] = (short)p;
    anext[e] = -1; aprev[e] = -1;
  }
  SD_HDN void swap_positions_in_ael(int e1, int e2) {                        // :1395-1439
    if (anext[e1] == aprev[e1] || anext[e2] == aprev[e2]) return;
    if (anext[e1] == e2) {
      int n = anext[e2]; if (n >= 0) aprev[n] = (short)e1;
      int p = aprev[e1]; if (p >= 0) anext[p] = (short)e2;
      aprev[e2] = (short)p; anext[e2] = (short)e1; aprev[e1] = (short)e2; anext[e1] = (short)n;
    } else if (anext[e2] == e1) {
      int n = anext[e1]; if (n >= 0) aprev[n] = (short)e2;
      int p = aprev[e2]; if (p >= 0) anext[p] = (short)e1;
      aprev[e1] = (short)p; anext[e1] = (short)e2; aprev[e2] = (short)e1; anext[e2] = (short)n;
    } else {
      int n = anext[e1], p = aprev[e1];
      anext[e1] = anext[e2]; if (anext[e1] >= 0) aprev[anext[e1]] = (short)e1;
      aprev[e1] = aprev[e2]; if (aprev[e1] >= 0) anext[aprev[e1]] = (short)e1;
      anext[e2] = (short)n; if (n >= 0) aprev[n] = (short)e2;
      aprev[e2] = (short)p; if (p >= 0) anext[p] = (short)e2;
    }
    if (aprev[e1] < 0) ael = (short)e1; else if (aprev[e2] < 0) ael = (short)e2;
  }
  SD_HDN void swap_positions_in_sel(int e1, int e2) {                        // :2558-2601
    if (snext[e1] < 0 && sprev[e1] < 0) return;
    if (snext[e2] < 0 && sprev[e2] < 0) return;
    if (snext[e1] == e2) {
      int n = snext[e2]; if (n >= 0) sprev[n] = (short)e1;
      int p = sprev[e1]; if (p >= 0) snext[p] = (short)e2;
      sprev[e2] = (short)p; snext[e2] = (short)e1; sprev[e1] = (short)e2; snext[e1] = (short)n;
    } else if (snext[e2] == e1) {
      int n = snext[e1]; if (n >= 0) sprev[n] = (short)e2;
      int p = sprev[e2]; if (p >= 0) snext[p] = (short)e1;
      sprev[e1] = (short)p; snext[e1] = (short)e2; sprev[e2] = (short)e1; snext[e2] = (short)n;
    } else {
      int n = snext[e1], p = sprev[e1];
      snext[e1] = snext[e2]; if (snext[e1] >= 0) sprev[snext[e1]] = (short)e1;
      sprev[e1] = sprev[e2]; if (sprev[e1] >= 0) snext[sprev[e1]] = (short)e1;
      snext[e2] = (short)n; if (n >= 0) sprev[n] = (short)e2;
      sprev[e2] = (short)p; if (p >= 0) snext[p] = (short)e2;
    }
    if (sprev[e1] < 0) sel = (short)e1; else if (sprev[e2] < 0) sel = (short)e2;
  }
  SD_HD void add_edge_to_sel(int edge) {                                    // :1900-1917
    if (sel < 0) { sel = (short)edge; sprev[edge] = -1; snext[edge] = -1; }
    else { snext[edge] = sel; sprev[edge] = -1; sprev[sel] = (short)edge; sel = (short)edge; }
  }
  SD_HD void delete_from_sel(int e) {                                       // :2080-2090
    int p = sprev[e], n = snext[e];
    if (p < 0 && n < 0 && e != sel) return;
    if (p >= 0) snext[p] = (short)n; else sel = (short)n;
    if (n >= 0) sprev[n] = (short)p;
    snext[e] = -1; sprev[e] = -1;
  }
  // UpdateEdgeIntoAEL :1442-1462 ; returns the new edge
  SD_HDN int update_edge_into_ael(int e) {
    int n = lml[e];
    if (n < 0) { status |= ST_FAIL; return e; }
    outidx[n] = outidx[e];
    int p = aprev[e], q = anext[e];
    if (p >= 0) anext[p] = (short)n; else ael = (short)n;
    if (q >= 0) aprev[q] = (short)n;
    side[n] = side[e]; wdelta[n] = wdelta[e]; wcnt[n] = wcnt[e]; wcnt2[n] = wcnt2[e];
    curx[n] = botx[n]; cury[n] = boty[n];
    aprev[n] = (short)p; anext[n] = (short)q;
    if (!is_horz(n)) insert_scanbeam(topy[n]);
    return n;
  }

  // ------------------------------------------------------------------ winding  (NonZero both, ctIntersection)
  SD_HDN void set_winding_count(int edge) {                                  // :1624-1722
    int e = aprev[edge];
    while (e >= 0 && (ptyp[e] != ptyp[edge] || wdelta[e] == 0)) e = aprev[e];
    if (e < 0) {
      wcnt[edge] = wdelta[edge];
      wcnt2[edge] = 0;
      e = ael;
    } else {
      if (wcnt[e] * wdelta[e] < 0) {
        int a = wcnt[e] < 0 ? -wcnt[e] : wcnt[e];
        if (a > 1) {
          if (wdelta[e] * wdelta[edge] < 0) wcnt[edge] = wcnt[e];
          else wcnt[edge] = (short)(wcnt[e] + wdelta[edge]);
        } else wcnt[edge] = (wdelta[edge] == 0 ? 1 : wdelta[edge]);
      } else {
        if (wdelta[edge] == 0) wcnt[edge] = (short)(wcnt[e] < 0 ? wcnt[e] - 1 : wcnt[e] + 1);
        else if (wdelta[e] * wdelta[edge] < 0) wcnt[edge] = wcnt[e];
        else wcnt[edge] = (short)(wcnt[e] + wdelta[edge]);
      }
      wcnt2[edge] = wcnt2[e];
      e = anext[e];
    }
    while (e != edge) { wcnt2[edge] = (short)(wcnt2[edge] + wdelta[e]); e = anext[e]; }
  }
  SD_HD bool is_contributing(int e) const {                                 // :1741-1838
    int a = wcnt[e] < 0 ? -wcnt[e] : wcnt[e];
    if (a != 1) return false;
    return wcnt2[e] != 0;
  }

  // ------------------------------------------------------------------ IntersectEdges  :2106-2298
  SD_HDN void intersect_edges(int e1, int e2, int px, int py) {
    bool c1 = outidx[e1] >= 0, c2 = outidx[e2] >= 0;
    if (ptyp[e1] == ptyp[e2]) {
      if (wcnt[e1] + wdelta[e2] == 0) wcnt[e1] = (short)-wcnt[e1]; else wcnt[e1] = (short)(wcnt[e1] + wdelta[e2]);
      if (wcnt[e2] - wdelta[e1] == 0) wcnt[e2] = (short)-wcnt[e2]; else wcnt[e2] = (short)(wcnt[e2] - wdelta[e1]);
    } else {
      wcnt2[e1] = (short)(wcnt2[e1] + wdelta[e2]);
      wcnt2[e2] = (short)(wcnt2[e2] - wdelta[e1]);
    }
    int e1Wc = wcnt[e1] < 0 ? -wcnt[e1] : wcnt[e1];
    int e2Wc = wcnt[e2] < 0 ? -wcnt[e2] : wcnt[e2];
    if (c1 && c2) {
      if ((e1Wc != 0 && e1Wc != 1) || (e2Wc != 0 && e2Wc != 1) || (ptyp[e1] != ptyp[e2])) {
        add_local_max_poly(e1, e2, px, py);
      } else {
        add_out_pt(e1, px, py);
        add_out_pt(e2, px, py);
        signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s;
        short o = outidx[e1]; outidx[e1] = outidx[e2]; outidx[e2] = o;
      }
    } else if (c1) {
      if (e2Wc == 0 || e2Wc == 1) {
        add_out_pt(e1, px, py);
        signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s;
        short o = outidx[e1]; outidx[e1] = outidx[e2]; outidx[e2] = o;
      }
    } else if (c2) {
      if (e1Wc == 0 || e1Wc == 1) {
        add_out_pt(e2, px, py);
        signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s;
        short o = outidx[e1]; outidx[e1] = outidx[e2]; outidx[e2] = o;
      }
    } else if ((e1Wc == 0 || e1Wc == 1) && (e2Wc == 0 || e2Wc == 1)) {
      int e1Wc2 = wcnt2[e1] < 0 ? -wcnt2[e1] : wcnt2[e1];
      int e2Wc2 = wcnt2[e2] < 0 ? -wcnt2[e2] : wcnt2[e2];
      if (ptyp[e1] != ptyp[e2]) add_local_min_poly(e1, e2, px, py);
      else if (e1Wc == 1 && e2Wc == 1) {
        if (e1Wc2 > 0 && e2Wc2 > 0) add_local_min_poly(e1, e2, px, py);
      } else { signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s; }
    }
  }

  // ------------------------------------------------------------------ InsertLocalMinimaIntoAEL  :1978-2077
  SD_BLK void insert_local_minima_into_ael(int botY) {
    while (cur_lm < n_lm && lm[cur_lm].y == botY) {
      int lb = lm[cur_lm].left, rb = lm[cur_lm].right;
      ++cur_lm;
      int op1 = -1; bool have_op1 = false;
      insert_edge_into_ael(lb, -1);
      insert_edge_into_ael(rb, lb);
      set_winding_count(lb);
      wcnt[rb] = wcnt[lb]; wcnt2[rb] = wcnt2[lb];
      if (is_contributing(lb)) { op1 = add_local_min_poly(lb, rb, botx[lb], boty[lb]); have_op1 = true; }
      insert_scanbeam(topy[lb]);
      if (is_horz(rb)) {
        add_edge_to_sel(rb);
        if (lml[rb] >= 0) insert_scanbeam(topy[lml[rb]]);
      } else insert_scanbeam(topy[rb]);

      if (have_op1 && is_horz(rb) && n_gj > 0 && wdelta[rb] != 0) {       // :2029-2040
        for (int g = 0; g < n_gj; ++g)
          if (horz_segments_overlap(gjx1[g], gjx2[g], botx[rb], topx[rb])) add_join(gjop[g], op1, gjx2[g], gjy2[g]);
      }
      int lp = aprev[lb];
      if (outidx[lb] >= 0 && lp >= 0 && curx[lp] == botx[lb] && outidx[lp] >= 0 &&
          slopes_equal4(botx[lp], boty[lp], topx[lp], topy[lp], curx[lb], cury[lb], topx[lb], topy[lb]) &&
          wdelta[lb] != 0 && wdelta[lp] != 0) {
        const int op2 = add_out_pt(lp, botx[lb], boty[lb]);
        add_join(op1, op2, topx[lb], topy[lb]);
      }
      if (anext[lb] != rb) {
        int rp = aprev[rb];
        if (outidx[rb] >= 0 && rp >= 0 && outidx[rp] >= 0 &&
            slopes_equal4(curx[rp], cury[rp], topx[rp], topy[rp], curx[rb], cury[rb], topx[rb], topy[rb]) &&
            wdelta[rb] != 0 && wdelta[rp] != 0) {
          const int op2 = add_out_pt(rp, botx[rb], boty[rb]);
          add_join(op1, op2, topx[rb], topy[rb]);
        }
        int e = anext[lb];
        int guard = 0;
        while (e >= 0 && e != rb) {
          intersect_edges(rb, e, curx[lb], cury[lb]);
          e = anext[e];
          if (++guard > NE) { status |= ST_ITER; break; }
        }
      }
    }
  }

  // ------------------------------------------------------------------ horizontals  :2512-2824
  SD_HD int get_maxima_pair(int e) const {                                  // :2538-2545
    int n = nxt[e], p = prv[e];
    if (topx[n] == topx[e] && topy[n] == topy[e] && lml[n] < 0) return n;
    else if (topx[p] == topx[e] && topy[p] == topy[e] && lml[p] < 0) return p;
    return -1;
  }
  SD_HD int get_maxima_pair_ex(int e) const {                               // :2548-2555
    int r = get_maxima_pair(e);
    if (r >= 0 && (anext[r] == aprev[r] && !is_horz(r))) return -1;
    return r;
  }
  SD_BLK void process_horizontal(int horz) {
    bool l2r; i64 hl, hr;
    if (botx[horz] < topx[horz]) { hl = botx[horz]; hr = topx[horz]; l2r = true; }
    else { hl = topx[horz]; hr = botx[horz]; l2r = false; }
    int eLast = horz, eMaxPair = -1;
    while (lml[eLast] >= 0 && is_horz(lml[eLast])) eLast = lml[eLast];
    if (lml[eLast] < 0) eMaxPair = get_maxima_pair(eLast);
    int op1 = -1; bool have_op1 = false;
    int guard = 0;
    for (;;) {
      bool isLast = (horz == eLast);
      int e = l2r ? anext[horz] : aprev[horz];
      while (e >= 0) {
        if (++guard > 4 * NE * NE) { status |= ST_ITER; return; }
        if ((l2r && curx[e] > hr) || (!l2r && curx[e] < hl)) break;
        if (curx[e] == topx[horz] && lml[horz] >= 0 && dx[e] < dx[lml[horz]]) break;
        if (outidx[horz] >= 0) {
          op1 = add_out_pt(horz, curx[e], cury[e]);
          have_op1 = true;
          horz_joins(horz, op1);
          add_ghost_join(op1, curx[e], botx[horz], boty[horz]);
        }
        if (e == eMaxPair && isLast) {
          if (outidx[horz] >= 0) add_local_max_poly(horz, eMaxPair, topx[horz], topy[horz]);
          delete_from_ael(horz);
          delete_from_ael(eMaxPair);
          return;
        }
        if (l2r) intersect_edges(horz, e, curx[e], cury[horz]);
        else intersect_edges(e, horz, curx[e], cury[horz]);
        int eNext = l2r ? anext[e] : aprev[e];
        swap_positions_in_ael(horz, e);
        e = eNext;
      }
      if (lml[horz] < 0 || !is_horz(lml[horz])) break;
      horz = update_edge_into_ael(horz);
      if (outidx[horz] >= 0) add_out_pt(horz, botx[horz], boty[horz]);
      if (botx[horz] < topx[horz]) { hl = botx[horz]; hr = topx[horz]; l2r = true; }
      else { hl = topx[horz]; hr = botx[horz]; l2r = false; }
    }
    if (outidx[horz] >= 0 && !have_op1) {                                   // :2771-2787
      op1 = self().out_last_pt(horz);
      horz_joins(horz, op1);
      add_ghost_join(op1, self().out_last_pt_x(horz), topx[horz], topy[horz]);
    }
    if (lml[horz] >= 0) {
      if (outidx[horz] >= 0) {
        op1 = add_out_pt(horz, topx[horz], topy[horz]);
        horz = update_edge_into_ael(horz);
        if (wdelta[horz] == 0) return;
        int ePrev = aprev[horz], eNext = anext[horz];
        if (ePrev >= 0 && curx[ePrev] == botx[horz] && cury[ePrev] == boty[horz] && wdelta[ePrev] != 0 &&
            (outidx[ePrev] >= 0 && cury[ePrev] > topy[ePrev] && slopes_equal_e(horz, ePrev))) {
          const int op2 = add_out_pt(ePrev, botx[horz], boty[horz]);
          add_join(op1, op2, topx[horz], topy[horz]);
        } else if (eNext >= 0 && curx[eNext] == botx[horz] && cury[eNext] == boty[horz] && wdelta[eNext] != 0 &&
                   outidx[eNext] >= 0 && cury[eNext] > topy[eNext] && slopes_equal_e(horz, eNext)) {
          const int op2 = add_out_pt(eNext, botx[horz], boty[horz]);
          add_join(op1, op2, topx[horz], topy[horz]);
        }
      } else update_edge_into_ael(horz);
    } else {
      if (outidx[horz] >= 0) add_out_pt(horz, topx[horz], topy[horz]);
      delete_from_ael(horz);
    }
  }
  SD_HD void process_horizontals() {
    int guard = 0;
    while (sel >= 0) {
      int h = sel;
      delete_from_sel(h);
      process_horizontal(h);
      if (++guard > 4 * NE) { status |= ST_ITER; break; }
    }
  }

  // ------------------------------------------------------------------ intersections  :2827-2954, 622-689
  SD_HDN void intersect_point(int e1, int e2, i64& ipx, i64& ipy) const {
    double b1, b2;
    if (dx[e1] == dx[e2]) { ipy = cury[e1]; ipx = top_x(e1, ipy); return; }
    else if (dx[e1] == 0) {
      ipx = botx[e1];
      if (is_horz(e2)) ipy = boty[e2];
      else { b2 = (double)boty[e2] - ((double)botx[e2] / dx[e2]); ipy = sd_round((double)ipx / dx[e2] + b2); }
    } else if (dx[e2] == 0) {
      ipx = botx[e2];
      if (is_horz(e1)) ipy = boty[e1];
      else { b1 = (double)boty[e1] - ((double)botx[e1] / dx[e1]); ipy = sd_round((double)ipx / dx[e1] + b1); }
    } else {
      b1 = (double)botx[e1] - (double)boty[e1] * dx[e1];
      b2 = (double)botx[e2] - (double)boty[e2] * dx[e2];
      double q = (b2 - b1) / (dx[e1] - dx[e2]);
      ipy = sd_round(q);
      double a1 = dx[e1] < 0 ? -dx[e1] : dx[e1], a2 = dx[e2] < 0 ? -dx[e2] : dx[e2];
      if (a1 < a2) ipx = sd_round(dx[e1] * q + b1);
      else ipx = sd_round(dx[e2] * q + b2);
    }
    if (ipy < topy[e1] || ipy < topy[e2]) {
      if (topy[e1] > topy[e2]) ipy = topy[e1]; else ipy = topy[e2];
      double a1 = dx[e1] < 0 ? -dx[e1] : dx[e1], a2 = dx[e2] < 0 ? -dx[e2] : dx[e2];
      if (a1 < a2) ipx = top_x(e1, ipy); else ipx = top_x(e2, ipy);
    }
    if (ipy > cury[e1]) {
      ipy = cury[e1];
      double a1 = dx[e1] < 0 ? -dx[e1] : dx[e1], a2 = dx[e2] < 0 ? -dx[e2] : dx[e2];
      if (a1 > a2) ipx = top_x(e2, ipy); else ipx = top_x(e1, ipy);
    }
  }
  SD_BLK void build_intersect_list(int topY) {
    if (ael < 0) return;
    int e = ael;
    sel = (short)e;
    while (e >= 0) {
      sprev[e] = aprev[e]; snext[e] = anext[e];
      curx[e] = (int)top_x(e, topY);
      e = anext[e];
    }
    bool isModified;
    int guard = 0;
    do {
      isModified = false;
      e = sel;
      while (snext[e] >= 0) {
        int eNext = snext[e];
        if (curx[e] > curx[eNext]) {
          i64 px, py;
          intersect_point(e, eNext, px, py);
          if (py < topY) { px = top_x(e, topY); py = topY; }
          if (n_il < MAXIL) { il[n_il].e1 = (short)e; il[n_il].e2 = (short)eNext; il[n_il].x = (int)px; il[n_il].y = (int)py; ++n_il; }
          else status |= ST_OVERFLOW_IL;
          swap_positions_in_sel(e, eNext);
          isModified = true;
        } else e = eNext;
        if (++guard > NE * NE * 2) { status |= ST_ITER; return; }
      }
      if (sprev[e] >= 0) snext[sprev[e]] = -1; else break;
    } while (isModified);
    sel = -1;
  }
  SD_HD bool edges_adjacent(const INode& n) const { return snext[n.e1] == n.e2 || sprev[n.e1] == n.e2; }
  SD_HDN bool fixup_intersection_order() {
    // CopyAELToSEL :1929-1939
    int e = ael; sel = (short)e;
    while (e >= 0) { sprev[e] = aprev[e]; snext[e] = anext[e]; e = anext[e]; }
    if (!StdSort<INode>::sort(il, n_il)) status |= ST_SORT_DEPTH;
    for (int i = 0; i < n_il; ++i) {
      if (!edges_adjacent(il[i])) {
        int j = i + 1;
        while (j < n_il && !edges_adjacent(il[j])) j++;
        if (j == n_il) return false;
        INode t = il[i]; il[i] = il[j]; il[j] = t;
      }
      swap_positions_in_sel(il[i].e1, il[i].e2);
    }
    return true;
  }
  SD_HD bool process_intersections(int topY) {
    if (ael < 0) return true;
    n_il = 0;
    build_intersect_list(topY);
    if (n_il == 0) return true;
    if (n_il == 1 || fixup_intersection_order()) {
      for (int i = 0; i < n_il; ++i) {
        intersect_edges(il[i].e1, il[i].e2, il[i].x, il[i].y);
        swap_positions_in_ael(il[i].e1, il[i].e2);
      }
      n_il = 0;
    } else return false;
    sel = -1;
    return true;
  }

  // ------------------------------------------------------------------ top of scan-beam  :2957-3113
  SD_BLK void do_maxima(int e) {
    int eMaxPair = get_maxima_pair_ex(e);
    if (eMaxPair < 0) {
      if (outidx[e] >= 0) add_out_pt(e, topx[e], topy[e]);
      delete_from_ael(e);
      return;
    }
    int eNext = anext[e];
    int guard = 0;
    while (eNext >= 0 && eNext != eMaxPair) {
      intersect_edges(e, eNext, topx[e], topy[e]);
      swap_positions_in_ael(e, eNext);
      eNext = anext[e];
      if (++guard > NE) { status |= ST_ITER; break; }
    }
    if (outidx[e] == kUnassigned && outidx[eMaxPair] == kUnassigned) {
      delete_from_ael(e); delete_from_ael(eMaxPair);
    } else if (outidx[e] >= 0 && outidx[eMaxPair] >= 0) {
      add_local_max_poly(e, eMaxPair, topx[e], topy[e]);
      delete_from_ael(e); delete_from_ael(eMaxPair);
    } else status |= ST_FAIL;   // "DoMaxima error" -> Execute fails, empty solution
  }
  SD_BLK void process_edges_at_top_of_scanbeam(int topY) {
    int e = ael;
    int guard = 0;
    while (e >= 0) {
      if (++guard > 4 * NE) { status |= ST_ITER; break; }
      bool isMax = (topy[e] == topY && lml[e] < 0);
      if (isMax) {
        int mp = get_maxima_pair_ex(e);
        isMax = (mp < 0 || !is_horz(mp));
      }
      if (isMax) {
        int ePrev = aprev[e];
        do_maxima(e);
        if (status & ST_FAIL) return;
        e = (ePrev < 0) ? (int)ael : (int)anext[ePrev];
      } else {
        if (topy[e] == topY && lml[e] >= 0 && is_horz(lml[e])) {
          e = update_edge_into_ael(e);
          if (outidx[e] >= 0) add_out_pt(e, botx[e], boty[e]);
          add_edge_to_sel(e);
        } else {
          curx[e] = (int)top_x(e, topY);
          cury[e] = topY;
        }
        e = anext[e];
      }
    }
    process_horizontals();
    e = ael;
    guard = 0;
    while (e >= 0) {
      if (++guard > 4 * NE) { status |= ST_ITER; break; }
      if (topy[e] == topY && lml[e] >= 0) {
        bool op = false; int oph = -1;
        if (outidx[e] >= 0) { oph = add_out_pt(e, topx[e], topy[e]); op = true; }
        e = update_edge_into_ael(e);
        int ePrev = aprev[e], eNext = anext[e];
        if (ePrev >= 0 && curx[ePrev] == botx[e] && cury[ePrev] == boty[e] && op &&
            outidx[ePrev] >= 0 && cury[ePrev] > topy[ePrev] &&
            slopes_equal4(curx[e], cury[e], topx[e], topy[e], curx[ePrev], cury[ePrev], topx[ePrev], topy[ePrev]) &&
            wdelta[e] != 0 && wdelta[ePrev] != 0) {
          const int op2 = add_out_pt(ePrev, botx[e], boty[e]);
          add_join(oph, op2, topx[e], topy[e]);
        } else if (eNext >= 0 && curx[eNext] == botx[e] && cury[eNext] == boty[e] && op &&
                   outidx[eNext] >= 0 && cury[eNext] > topy[eNext] &&
                   slopes_equal4(curx[e], cury[e], topx[e], topy[e], curx[eNext], cury[eNext], topx[eNext], topy[eNext]) &&
                   wdelta[e] != 0 && wdelta[eNext] != 0) {
          const int op2 = add_out_pt(eNext, botx[e], boty[e]);
          add_join(oph, op2, topx[e], topy[e]);
        }
      }
      e = anext[e];
    }
  }

  // ------------------------------------------------------------------ Execute  :1560-1621, 1247-1276
  SD_HD void reset_core() {
    n_lm = 0; cur_lm = 0; n_sb = 0; n_il = 0; ael = -1; sel = -1;
    status = ST_OK; n_joins = 0; n_gj = 0;
  }
  // Runs the sweep (Clipper::ExecuteInternal up to the end of the scan-beam loop). Returns false if
  // Clipper's Execute would fail (empty solution).
  SD_BLK bool run_sweep() {
    if (n_lm == 0) return true;
    if (!StdSort<LocMin>::sort(lm, n_lm)) status |= ST_SORT_DEPTH;
    for (int i = 0; i < n_lm; ++i) {
      insert_scanbeam(lm[i].y);
      int e = lm[i].left;  curx[e] = botx[e]; cury[e] = boty[e]; side[e] = kLeft;  outidx[e] = kUnassigned;
      e = lm[i].right;     curx[e] = botx[e]; cury[e] = boty[e]; side[e] = kRight; outidx[e] = kUnassigned;
    }
    ael = -1; cur_lm = 0;
    int botY, topY = 0;
    if (!pop_scanbeam(botY)) return false;
    int guard = 0;
    bool ok = true;
    for (;;) {
      insert_local_minima_into_ael(botY);              // (in front of the loop and at its end in Clipper: ONE call site here, same sequence)
      bool popped = pop_scanbeam(topY);
      if (!popped && !(cur_lm < n_lm)) break;
      if (++guard > 4 * NE) { status |= ST_ITER; break; }
      process_horizontals();
      n_gj = 0;                                                             // ClearGhostJoins :1575
      if (!process_intersections(topY)) { ok = false; break; }
      process_edges_at_top_of_scanbeam(topY);
      if (status & ST_FAIL) { ok = false; break; }
      botY = topY;
    }
    if (!ok || (status & ST_FAIL)) { status |= ST_FAIL; return false; }
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// Sweep: the fast variant.  Output rings are kept as {front point, back point, running shoelace
// sum}; exact whenever the reference records no joins for the pair (n_joins == 0).
// MAXREC: output-ring capacity.
template <int MAXV, int MAXIL, int MAXREC, class P = PlainStorage>
struct Sweep : SweepCore<Sweep<MAXV, MAXIL, MAXREC, P>, P, MAXV, MAXIL> {
  typedef SweepCore<Sweep<MAXV, MAXIL, MAXREC, P>, P, MAXV, MAXIL> B;
  using B::outidx; using B::side; using B::status; using B::ael; using B::anext;
  static constexpr unsigned RR = P::template region<int, MAXREC>();
  static constexpr unsigned O_RFX = B::O_CORE_END, O_RFY = O_RFX + RR, O_RLX = O_RFY + RR, O_RLY = O_RLX + RR, O_RSUM = O_RLY + RR,
                            O_NREC = O_RSUM + P::template region<i64, MAXREC>(), O_TWICE = O_NREC + P::template region<i64, 1>(),
                            O_SABS = O_TWICE + P::template region<i64, 1>(), O_END = O_SABS + P::template region<i64, 1>();
  typename P::template Arr<int, MAXREC, O_RFX> rfx; typename P::template Arr<int, MAXREC, O_RFY> rfy;
  typename P::template Arr<int, MAXREC, O_RLX> rlx; typename P::template Arr<int, MAXREC, O_RLY> rly;
  typename P::template Arr<i64, MAXREC, O_RSUM> rsum;
  static constexpr unsigned lds_bytes() { return O_END; }
  typename P::template Scalar<int, O_NREC> n_rec;
  typename P::template Scalar<i64, O_TWICE> twice_area;      // sum over closed rings of |2*area|
  typename P::template Scalar<i64, O_SABS> sum_abs_terms;    // sum of |cross| terms (exactness bound for the float path)
  SD_HD void term(i64 c) { sum_abs_terms += sd_abs64(c); }
  SD_HDN int out_add_pt(int e, int px, int py) {                             // :2463-2499
    int r = outidx[e];
    if (r < 0) {
      if (n_rec >= MAXREC) { status |= ST_OVERFLOW_REC; return -1; }
      r = n_rec++;
      rfx[r] = rlx[r] = px; rfy[r] = rly[r] = py; rsum[r] = 0;
      outidx[e] = (short)r;
    } else {
      if (side[e] == kLeft) {           // to front
        if (px == rfx[r] && py == rfy[r]) return -1;
        i64 c = sd_cross(px, py, rfx[r], rfy[r]); term(c);
        rsum[r] += c; rfx[r] = px; rfy[r] = py;
      } else {
        if (px == rlx[r] && py == rly[r]) return -1;
        i64 c = sd_cross(rlx[r], rly[r], px, py); term(c);
        rsum[r] += c; rlx[r] = px; rly[r] = py;
      }
    }
    return -1;
  }
  SD_HD void out_ring_closed(int r) {
    i64 c = sd_cross(rlx[r], rly[r], rfx[r], rfy[r]); term(c);
    twice_area += sd_abs64(rsum[r] + c);
  }
  SD_HDN void out_append(int e1, int e2) {                                   // :2367-2460
    int r1 = outidx[e1], r2 = outidx[e2];
    i64 c;
    if (side[e1] == kLeft) {
      if (side[e2] == kLeft) {        // reverse(2) + 1
        c = sd_cross(rfx[r2], rfy[r2], rfx[r1], rfy[r1]);
        rsum[r1] = -rsum[r2] + c + rsum[r1];
        rfx[r1] = rlx[r2]; rfy[r1] = rly[r2];
      } else {                        // 2 + 1
        c = sd_cross(rlx[r2], rly[r2], rfx[r1], rfy[r1]);
        rsum[r1] = rsum[r2] + c + rsum[r1];
        rfx[r1] = rfx[r2]; rfy[r1] = rfy[r2];
      }
    } else {
      if (side[e2] == kRight) {       // 1 + reverse(2)
        c = sd_cross(rlx[r1], rly[r1], rlx[r2], rly[r2]);
        rsum[r1] = rsum[r1] + c - rsum[r2];
        rlx[r1] = rfx[r2]; rly[r1] = rfy[r2];
      } else {                        // 1 + 2
        c = sd_cross(rlx[r1], rly[r1], rfx[r2], rfy[r2]);
        rsum[r1] = rsum[r1] + c + rsum[r2];
        rlx[r1] = rlx[r2]; rly[r1] = rly[r2];
      }
    }
    term(c);
    int okIdx = r1, obsolete = r2;
    outidx[e1] = kUnassigned; outidx[e2] = kUnassigned;
    for (int e = ael; e >= 0; e = anext[e]) {
      if (outidx[e] == obsolete) { outidx[e] = (short)okIdx; side[e] = side[e1]; break; }
    }
  }

  SD_HD void out_add_join(int, int, int, int) {}
  SD_HD int out_last_pt(int) { return -1; }
  SD_HD int out_last_pt_x(int e) { const int r = outidx[e]; return (side[e] == kLeft) ? rfx[r] : rlx[r]; }
  SD_HD void reset_state() { B::reset_core(); n_rec = 0; twice_area = 0; sum_abs_terms = 0; }
  // Returns 2*area of (A ∩ B) as the reference would sum it (0 if Clipper's Execute fails).
  SD_HD i64 execute() { return B::run_sweep() ? twice_area : 0; }
};

}  // namespace sdclip
