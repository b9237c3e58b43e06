// clip_beam.h -- the scan-beam polygon intersection re-laid-out for the GPU ("bound slots").
//
// Same arithmetic as clip_sweep.h (the restated Vatti/Clipper 6.4.2 sweep behind
// stardist/lib/stardist2d.cpp:152-165, reference clipper.cpp line numbers are cited there and
// repeated here per function), different data layout -- chosen so that one pair's whole dynamic
// state is a few hundred bytes and can live in LDS at real occupancy:
//
//   * PREPARED POLYGONS.  Everything Clipper::AddPath does (clipper.cpp:1045-1221: duplicate /
//     collinear vertex removal, InitEdge2, local minima, ProcessBound incl. horizontal reversal) depends
//     on ONE polygon only.  It is done once per candidate (prepare_polygon) and stored as a compact
//     record: cleaned vertex ring + one code byte per edge {successor in its bound, bot/top swap} +
//     the local-minima list, stable-sorted by Y.  The pair sweep never touches per-edge arrays of size
//     2*n_rays again.
//   * BOUND SLOTS.  An edge in the active edge list is always "the current edge of a bound": when it
//     ends, UpdateEdgeIntoAEL (:1442-1462) replaces it IN PLACE by its successor, which inherits
//     position, winding counts, output ring and side.  So the dynamic per-edge state is kept per
//     bound in K slots (K = 8 for 32 rays); a slot caches the static data of its current edge.
//   * AEL = ONE 64-BIT WORD.  The active edge list is the sequence of slot numbers, one nibble per
//     position (unused nibbles = 0xF).  next/prev/insert/delete/swap are shifts and masks in
//     registers instead of dependent memory loads along a linked list; the same for the sorted copy
//     used by BuildIntersectList / FixupIntersectionOrder and for the stack of pending horizontals.
//   * NO SCAN-BEAM QUEUE.  Clipper's queue holds exactly {Y of pending local minima} + {top Y of every
//     non-horizontal edge currently in the AEL} + {top Y of the successor of a horizontal right bound
//     inserted at a local minimum (:2018-2021)}; the next scan-beam is the maximum of those, computed
//     from the slots.
//   * std::sort of local minima / intersections is a stable sort for n <= 16 (libstdc++ insertion
//     sort); more than 16 of either is flagged (ST_OVERFLOW_*) and the pair is re-run on the general
//     path (clip_sweep_full.h).
//
// The CPU harness tests/host/beam_check.cpp compares this against clip_sweep.h (itself checked against
// the compiled reference Clipper on millions of pairs) and against the reference Clipper directly.
#pragma once
#include "clip_sweep.h"
#include <limits.h>

#if defined(BEAM_COUNT) && !defined(__HIP_DEVICE_COMPILE__)
struct BeamCounters { long beams, topx, isect_pt, isect_edges, outpt, update, maxima, lm, horz, ael_sum, il_beams, es_loads; };
inline BeamCounters& beam_counters() { static BeamCounters c = {}; return c; }
#define BEAM_CNT(f, v) (beam_counters().f += (v))
#else
#define BEAM_CNT(f, v) ((void)0)
#endif

namespace sdclip {

typedef unsigned long long u64;

enum {
  ST_OVERFLOW_AEL = 32,   // more than K bounds active at once
  ST_OVERFLOW_LM = 64,    // more than 8 local minima in one polygon / 16 in the pair
  ST_OVERFLOW_GJ = 128    // ghost-join / extra scan-beam capacity
};
enum { BEAM_MAXLM = 8 };

// ---------------------------------------------------------------------------------------------
// Prepared polygon (one per candidate).  n == 0: the path is rejected by AddPath (fewer than 3
// distinct vertices, or flat) and contributes no edges.
struct PrepPt { int x, y; };
template <bool WIDE> struct PrepIdx { typedef unsigned char type; enum { NONE = 255 }; };
template <> struct PrepIdx<true> { typedef unsigned short type; enum { NONE = 65535 }; };
template <int MAXV>
struct PolyPrep {
  typedef typename PrepIdx<(MAXV > 128)>::type pidx;
  enum { NONE = PrepIdx<(MAXV > 128)>::NONE };
  int n;                              // edges (= vertices) of the cleaned ring
  int n_lm;                           // local minima, stable-sorted by Y descending
  int status;                         // ST_* flags raised while preparing
  int pad;
  PrepPt v[MAXV + 1];                 // cleaned ring, v[n] = v[0]; edge e runs from vertex e to vertex e+1
  unsigned char ecode[MAXV];          // bits 0-1: NextInLML (0 none, 1 ring-next, 2 ring-prev); bit 2: Bot is vertex e+1;
                                      // bit 3: NextInLML exists and is horizontal
  pidx mpair[MAXV];                   // GetMaximaPair(e) (:2538-2545) as a ring index, NONE = no pair
  pidx hlast[MAXV];                   // horizontal e: last edge of the run of horizontals that starts at e in its bound (:2519-2521)
  pidx lm_left[BEAM_MAXLM], lm_right[BEAM_MAXLM];
};

// Working storage + routine of the preparation.  P = storage policy (PlainStorage / LdsStorage<S>).
#ifndef BEAM_PREP_IDX
#define BEAM_PREP_IDX short
#endif
template <class P, int MAXV>
struct PrepWork {
  typedef BEAM_PREP_IDX ridx;
  static constexpr unsigned RI = P::template region<int, MAXV>(), RS = P::template region<ridx, MAXV>(),
                            RB = P::template region<unsigned char, MAXV>();
  static constexpr unsigned O_CX = 0, O_CY = O_CX + RI, O_NXT = O_CY + RI, O_PRV = O_NXT + RS, O_CODE = O_PRV + RS,
                            O_N = O_CODE + RB, O_LMY = O_N + P::template region<int, 1>(), O_LML = O_LMY + P::template region<int, BEAM_MAXLM>(),
                            O_LMR = O_LML + P::template region<short, BEAM_MAXLM>(), O_END = O_LMR + P::template region<short, BEAM_MAXLM>();
  static constexpr unsigned lds_bytes() { return O_END; }
  typename P::template Arr<int, MAXV, O_CX> cx; typename P::template Arr<int, MAXV, O_CY> cy;
  typename P::template Arr<ridx, MAXV, O_NXT> nxt; typename P::template Arr<ridx, MAXV, O_PRV> prv;
  typename P::template Arr<unsigned char, MAXV, O_CODE> code;
  typename P::template Scalar<int, O_N> n;
  typename P::template Arr<int, BEAM_MAXLM, O_LMY> lmy;
  typename P::template Arr<short, BEAM_MAXLM, O_LML> lml_; typename P::template Arr<short, BEAM_MAXLM, O_LMR> lmr_;

  // ---- accessors on the COMPACT ring (after cleaning): edge e = vertex e -> vertex e+1
  SD_HD int nx(int e) const { return e + 1 == n ? 0 : e + 1; }
  SD_HD int pv(int e) const { return e == 0 ? n - 1 : e - 1; }
  SD_HD bool sw(int e) const { return (code[e] & 4) != 0; }
  SD_HD int botx(int e) const { return sw(e) ? cx[nx(e)] : cx[e]; }
  SD_HD int boty(int e) const { return sw(e) ? cy[nx(e)] : cy[e]; }
  SD_HD int topx(int e) const { return sw(e) ? cx[e] : cx[nx(e)]; }
  SD_HD int topy(int e) const { return sw(e) ? cy[e] : cy[nx(e)]; }
  SD_HD bool is_horz(int e) const { return cy[e] == cy[nx(e)]; }
  SD_HD double dx(int e) const {                                              // clipper.cpp:591-596
    const i64 dy = (i64)topy(e) - boty(e);
    if (dy == 0) return SD_HORIZONTAL;
    return (double)((i64)topx(e) - botx(e)) / (double)dy;
  }
  SD_HD void reverse_horizontal(int e) { code[e] = (unsigned char)(code[e] ^ 4); }
  SD_HD void set_lml(int e, int c) { code[e] = (unsigned char)((code[e] & ~3) | c); }

#ifndef BEAM_PREP_FN
#define BEAM_PREP_FN SD_HD
#endif
  BEAM_PREP_FN int find_next_loc_min(int E) const {                                  // :911-925
    // (the reference's for(;;) / continue / break nest, written with one exit flag)
    bool done = false;
    int guard = 0;
    while (!done) {
      while (botx(E) != botx(pv(E)) || boty(E) != boty(pv(E)) || (cx[E] == topx(E) && cy[E] == topy(E))) E = nx(E);
      if (!is_horz(E) && !is_horz(pv(E))) done = true;
      else {
        while (is_horz(pv(E))) E = pv(E);
        const int E2 = E;
        while (is_horz(E)) E = nx(E);
        if (topy(E) != boty(pv(E))) {            // otherwise: just an intermediate horizontal, keep looking
          if (botx(pv(E2)) < botx(E)) E = E2;
          done = true;
        }
      }
      if (++guard > 4 * MAXV) done = true;
    }
    return E;
  }
  BEAM_PREP_FN int process_bound(int E, bool fwd) {                                  // :928-1042 (no skip edges)
    int Result = E, Horz;
    if (is_horz(E)) {
      int EStart = fwd ? pv(E) : nx(E);
      if (is_horz(EStart)) {
        if (botx(EStart) != botx(E) && topx(EStart) != botx(E)) reverse_horizontal(E);
      } else if (botx(EStart) != botx(E)) reverse_horizontal(E);
    }
    int EStart = E;
    if (fwd) {
      while (topy(Result) == boty(nx(Result))) Result = nx(Result);
      if (is_horz(Result)) {
        Horz = Result;
        while (is_horz(pv(Horz))) Horz = pv(Horz);
        if (topx(pv(Horz)) > topx(nx(Result))) Result = pv(Horz);
      }
      while (E != Result) {
        set_lml(E, 1);
        if (is_horz(E) && E != EStart && botx(E) != topx(pv(E))) reverse_horizontal(E);
        E = nx(E);
      }
      if (is_horz(E) && E != EStart && botx(E) != topx(pv(E))) reverse_horizontal(E);
      Result = nx(Result);
    } else {
      while (topy(Result) == boty(pv(Result))) Result = pv(Result);
      if (is_horz(Result)) {
        Horz = Result;
        while (is_horz(nx(Horz))) Horz = nx(Horz);
        if (topx(nx(Horz)) == topx(pv(Result)) || topx(nx(Horz)) > topx(pv(Result))) Result = nx(Horz);
      }
      while (E != Result) {
        set_lml(E, 2);
        if (is_horz(E) && E != EStart && botx(E) != topx(nx(E))) reverse_horizontal(E);
        E = pv(E);
      }
      if (is_horz(E) && E != EStart && botx(E) != topx(nx(E))) reverse_horizontal(E);
      Result = pv(Result);
    }
    return Result;
  }

  // Clipper::AddPath for one closed path (:1045-1221).  xs/ys: the n_in integer vertices.
  template <typename XT>
  SD_HDN void prepare(const XT* xs, const XT* ys, int n_in, PolyPrep<MAXV>* out) {
    out->n = 0; out->n_lm = 0; out->status = 0; out->pad = 0;
    int highI = n_in - 1;
    while (highI > 0 && xs[highI] == xs[0] && ys[highI] == ys[0]) --highI;
    while (highI > 0 && xs[highI] == xs[highI - 1] && ys[highI] == ys[highI - 1]) --highI;
    if (highI < 2) return;
    for (int i = 0; i <= highI; ++i) {
      cx[i] = (int)xs[i]; cy[i] = (int)ys[i];
      nxt[i] = (ridx)(i == highI ? 0 : i + 1);
      prv[i] = (ridx)(i == 0 ? highI : i - 1);
    }
    int eStart = 0, E = 0, eLoopStop = 0;
    for (;;) {   // remove duplicate vertices and collinear edges (:1098-1122)
      if (cx[E] == cx[nxt[E]] && cy[E] == cy[nxt[E]]) {
        if (E == nxt[E]) break;
        if (E == eStart) eStart = nxt[E];
        const int en = nxt[E]; nxt[prv[E]] = (ridx)en; prv[en] = prv[E]; E = en;
        eLoopStop = E;
        continue;
      }
      if (prv[E] == nxt[E]) break;
      const int ep0 = prv[E], en0 = nxt[E];
      if (((i64)cy[ep0] - cy[E]) * ((i64)cx[E] - cx[en0]) == ((i64)cx[ep0] - cx[E]) * ((i64)cy[E] - cy[en0])) {   // SlopesEqual :554-563
        if (E == eStart) eStart = nxt[E];
        nxt[ep0] = (ridx)en0; prv[en0] = (ridx)ep0;
        E = ep0;
        eLoopStop = E;
        continue;
      }
      E = nxt[E];
      if (E == eLoopStop) break;
    }
    if (prv[E] == nxt[E]) return;
    // compact the surviving ring in place (ring order = increasing original index, cyclically; the traversal below starts
    // at the image of eStart, so every later step visits the edges in the order the reference does)
    int m = 0, E0 = 0;
    {
      for (int i = 0; i <= highI; ++i) code[i] = 0;
      int e = eStart, g0 = 0;
      do { code[e] = 1; e = nxt[e]; } while (e != eStart && ++g0 <= MAXV);
      for (int i = 0; i <= highI; ++i) {
        if (!code[i]) continue;
        if (i == eStart) E0 = m;
        if (m != i) { cx[m] = cx[i]; cy[m] = cy[i]; }
        ++m;
      }
    }
    n = m;
    bool isFlat = true;
    for (int e = 0; e < m; ++e) {   // InitEdge2 :729-742
      const int en = nx(e);
      code[e] = (unsigned char)((cy[e] >= cy[en]) ? 0 : 4);
      if (cy[en] != cy[0]) isFlat = false;
    }
    if (isFlat) return;
    E = E0;
    if (botx(pv(E)) == topx(pv(E)) && boty(pv(E)) == topy(pv(E))) E = nx(E);
    int EMin = -1, guard = 0, n_lm = 0;
    int st = 0;
    for (;;) {
      E = find_next_loc_min(E);
      if (E == EMin) break;
      else if (EMin < 0) EMin = E;
      if (++guard > 2 * MAXV + 2) { st |= ST_ITER; break; }
      int left, right; bool leftFwd;
      if (dx(E) < dx(pv(E))) { left = pv(E); right = E; leftFwd = false; }
      else { left = E; right = pv(E); leftFwd = true; }
      const int y = boty(E);
      E = process_bound(left, leftFwd);
      const int E2 = process_bound(right, !leftFwd);
      if (n_lm < BEAM_MAXLM) {
        // stable insertion by Y descending (== libstdc++ std::sort for n <= 16, LocMinSorter :125-131)
        int k = n_lm;
        while (k > 0 && lmy[k - 1] < y) { lmy[k] = lmy[k - 1]; lml_[k] = lml_[k - 1]; lmr_[k] = lmr_[k - 1]; --k; }
        lmy[k] = y; lml_[k] = (short)left; lmr_[k] = (short)right;
        ++n_lm;
      } else st |= ST_OVERFLOW_LM;
      if (!leftFwd) E = E2;
    }
    for (int i = 0; i < m; ++i) { out->v[i].x = cx[i]; out->v[i].y = cy[i]; }
    out->v[m].x = cx[0]; out->v[m].y = cy[0];
    for (int i = 0; i < m; ++i) {
      const int c = code[i] & 3;
      const int nl = (c == 1) ? nx(i) : (c == 2 ? pv(i) : -1);
      int cc = code[i] & 7;
      if (nl >= 0 && is_horz(nl)) cc |= 8;
      out->ecode[i] = (unsigned char)cc;
      // GetMaximaPair :2538-2545
      const int a = nx(i), b = pv(i);
      int mp = PolyPrep<MAXV>::NONE;
      if (topx(a) == topx(i) && topy(a) == topy(i) && (code[a] & 3) == 0) mp = a;
      else if (topx(b) == topx(i) && topy(b) == topy(i) && (code[b] & 3) == 0) mp = b;
      out->mpair[i] = (typename PolyPrep<MAXV>::pidx)mp;
      // last horizontal of the run (:2519-2521)
      int last = i, g2 = 0;
      if (is_horz(i)) {
        for (;;) {
          const int c2 = code[last] & 3;
          const int n2 = (c2 == 1) ? nx(last) : (c2 == 2 ? pv(last) : -1);
          if (n2 < 0 || !is_horz(n2)) break;
          last = n2;
          if (++g2 > MAXV) { st |= ST_ITER; break; }
        }
      }
      out->hlast[i] = (typename PolyPrep<MAXV>::pidx)last;
    }
    for (int i = 0; i < n_lm; ++i) { out->lm_left[i] = (typename PolyPrep<MAXV>::pidx)lml_[i]; out->lm_right[i] = (typename PolyPrep<MAXV>::pidx)lmr_[i]; }
    out->n = m; out->n_lm = n_lm; out->status = st;
  }
};

// ---------------------------------------------------------------------------------------------
// BeamCore: the sweep over two prepared polygons.  D (CRTP) supplies the output side exactly as for
// SweepCore (out_add_pt, out_append, out_ring_closed, out_add_join, out_last_pt, out_last_pt_x).
// Handles are slot numbers (0..K-1), -1 = none.  K <= 15.
template <class D, class P, int MAXV, int K, int MAXIL, bool FULLGJ = false>
struct BeamCore {
  typedef PolyPrep<MAXV> Prep;
  // nibble lists: one 32-bit word holds 8 positions (K <= 8), a 64-bit word 15
  template <bool W, int = 0> struct OrdT { typedef unsigned int type; };
  template <int D0> struct OrdT<true, D0> { typedef u64 type; };
  typedef typename OrdT<(K > 8)>::type ord_t;
  enum { ORD_BITS = (K > 8) ? 64 : 32 };
  enum { GJ = (K <= 8 ? 3 : 4), NX = (K <= 8 ? 2 : 4), EID_POLY = 4096 };
  SD_HD D& self() { return *static_cast<D*>(this); }
  static constexpr unsigned RI = P::template coord_region<K>(), RD = P::template slope_region<K>(),
                            RS = P::template region<short, K>(), RB = P::template region<signed char, K>(),
                            RIL = P::template coord_region<MAXIL>(), RILB = P::template region<signed char, MAXIL>(),
                            RG = P::template coord_region<GJ>(), RGF = P::template region<int, (FULLGJ ? GJ : 1)>(), R1 = P::template region<int, 1>(), R8 = P::template region<u64, 1>(), RO = P::template region<ord_t, 1>();
  static constexpr unsigned O_BOTX = 0, O_BOTY = O_BOTX + RI, O_TOPX = O_BOTY + RI, O_TOPY = O_TOPX + RI, O_CURX = O_TOPY + RI,
                            O_CURY = O_CURX + RI, O_DX = O_CURY + RI, O_EID = O_DX + RD, O_WCNT = O_EID + RS, O_WCNT2 = O_WCNT + RB,
                            O_OUTIDX = O_WCNT2 + RB, O_LMLC = O_OUTIDX + RB, O_WDELTA = O_LMLC + RB, O_PTYP = O_WDELTA + RB,
                            O_SIDE = O_PTYP + RB, O_ILX = O_SIDE + RB, O_ILY = O_ILX + RIL, O_ILE1 = O_ILY + RIL, O_ILE2 = O_ILE1 + RILB,
                            O_GJOP = O_ILE2 + RILB, O_GJX1 = O_GJOP + RGF, O_GJX2 = O_GJX1 + RG, O_GJY2 = O_GJX2 + RG,
                            O_XTRA = O_GJY2 + RGF, O_MLM = O_XTRA + P::template region<int, NX>(),
                            O_PA = O_MLM + P::template region<unsigned char, 16>(),
                            O_ORD = O_PA + 2 * R8, O_HSEL = O_ORD + RO, O_AFTER_ORD = O_HSEL + RO, O_PB = O_PA + R8, O_NAEL = O_AFTER_ORD, O_FREE = O_NAEL + R1, O_NLM = O_FREE + R1,
                            O_CURLM = O_NLM + R1, O_NIL = O_CURLM + R1, O_STATUS = O_NIL + R1, O_NJOINS = O_STATUS + R1,
                            O_NGJ = O_NJOINS + R1, O_NXTRA = O_NGJ + R1, O_LMY = O_NXTRA + R1, O_ORGX = O_LMY + R1, O_ORGY = O_ORGX + R1, O_CORE_END = O_ORGY + R1;
  // ---- bound slots
  // (coordinates: plain ints, or 16-bit offsets from the pair's origin (orgx, orgy) with LdsStorage16; slopes: stored, or recomputed)
  typename P::template CoordArr<K, O_BOTX, O_ORGX, O_STATUS> botx; typename P::template CoordArr<K, O_BOTY, O_ORGY, O_STATUS> boty;
  typename P::template CoordArr<K, O_TOPX, O_ORGX, O_STATUS> topx; typename P::template CoordArr<K, O_TOPY, O_ORGY, O_STATUS> topy;
  typename P::template CoordArr<K, O_CURX, O_ORGX, O_STATUS> curx; typename P::template CoordArr<K, O_CURY, O_ORGY, O_STATUS> cury;
  typename P::template SlopeArr<K, O_DX, O_BOTX, O_BOTY, O_TOPX, O_TOPY> dx;
  typename P::template Arr<short, K, O_EID> eid;                 // current edge: poly * EID_POLY + index in that polygon's ring
  typename P::template Arr<signed char, K, O_WCNT> wcnt; typename P::template Arr<signed char, K, O_WCNT2> wcnt2;   // |winding| <= n_lm <= 16
  typename P::template Arr<signed char, K, O_OUTIDX> outidx;
  typename P::template Arr<signed char, K, O_LMLC> lmlc;        // current edge: bits 0-1 NextInLML (0 none, 1 next, 2 prev), bit 3 NextInLML is horizontal
  typename P::template Arr<signed char, K, O_WDELTA> wdelta; typename P::template Arr<signed char, K, O_PTYP> ptyp;
  typename P::template Arr<signed char, K, O_SIDE> side;
  // ---- intersections of the current scan-beam
  typename P::template CoordArr<MAXIL, O_ILX, O_ORGX, O_STATUS> ilx; typename P::template CoordArr<MAXIL, O_ILY, O_ORGY, O_STATUS> ily;
  typename P::template Arr<signed char, MAXIL, O_ILE1> ile1; typename P::template Arr<signed char, MAXIL, O_ILE2> ile2;
  // ---- ghost joins of the current scan-line (:1968-1975), extra scan-beam Ys, merged local minima
  typename P::template Arr<int, (FULLGJ ? GJ : 1), O_GJOP> gjop; typename P::template CoordArr<GJ, O_GJX1, O_ORGX, O_STATUS> gjx1;
  typename P::template CoordArr<GJ, O_GJX2, O_ORGX, O_STATUS> gjx2; typename P::template Arr<int, (FULLGJ ? GJ : 1), O_GJY2> gjy2;
  typename P::template Arr<int, NX, O_XTRA> xtra;
  typename P::template Arr<unsigned char, 16, O_MLM> mlm;       // poly * 128 + index into that polygon's lm list
  // ---- scalars
  typename P::template Scalar<ord_t, O_ORD> ord;                 // AEL: nibble p = slot at position p; unused = 0xF
  typename P::template Scalar<ord_t, O_HSEL> hsel;                // SEL as the stack of pending horizontals (nibble 0 = head)
  typename P::template Scalar<const Prep*, O_PA> prepA; typename P::template Scalar<const Prep*, O_PB> prepB;
  typename P::template Scalar<int, O_NAEL> n_ael; typename P::template Scalar<int, O_FREE> freemask;
  typename P::template Scalar<int, O_NLM> n_lm; typename P::template Scalar<int, O_CURLM> cur_lm;
  typename P::template Scalar<int, O_NIL> n_il; typename P::template Scalar<int, O_STATUS> status;
  typename P::template Scalar<int, O_NJOINS> n_joins; typename P::template Scalar<int, O_NGJ> n_gj;
  typename P::template Scalar<int, O_NXTRA> n_xtra;
  typename P::template Scalar<int, O_LMY> next_lm_y;             // Y of local minimum cur_lm (valid while cur_lm < n_lm)
  typename P::template Scalar<int, O_ORGX> orgx; typename P::template Scalar<int, O_ORGY> orgy;   // origin of the 16-bit coordinate arrays

  static constexpr ord_t ALLF = (ord_t)~(ord_t)0;
  static constexpr ord_t ONES = (ord_t)0x1111111111111111ull, HIGHS = (ord_t)0x8888888888888888ull;
  // ------------------------------------------------------------------ nibble lists
  static SD_HD int nib(ord_t w, int p) { return (int)((w >> (4 * p)) & (ord_t)15); }
  static SD_HD int nib_find(ord_t w, int h) {                       // position of slot h (unused nibbles are 0xF, h < 15), or -1
    const ord_t x = w ^ ((ord_t)h * ONES);
    const ord_t t = (ord_t)(x - ONES) & ~x & HIGHS;                  // lowest set bit marks the first zero nibble (exact)
    if (!t) return -1;
#if defined(__HIP_DEVICE_COMPILE__)
    return (ORD_BITS == 64) ? ((__ffsll((long long)t) - 1) >> 2) : ((__ffs((int)t) - 1) >> 2);
#else
    return (ORD_BITS == 64) ? (__builtin_ctzll((u64)t) >> 2) : (__builtin_ctz((unsigned)t) >> 2);
#endif
  }
  static SD_HD ord_t nib_insert(ord_t w, int p, int h) {
    const ord_t lowmask = (p == 0) ? (ord_t)0 : (ord_t)(ALLF >> (ORD_BITS - 4 * p));
    return (ord_t)((w & lowmask) | ((ord_t)h << (4 * p)) | ((ord_t)(w & ~lowmask) << 4));
  }
  static SD_HD ord_t nib_remove(ord_t w, int p) {
    const ord_t lowmask = (p == 0) ? (ord_t)0 : (ord_t)(ALLF >> (ORD_BITS - 4 * p));
    return (ord_t)((w & lowmask) | (((w >> 4) & ~lowmask)) | ((ord_t)0xF << (ORD_BITS - 4)));
  }
  static SD_HD ord_t nib_swap(ord_t w, int p, int q) {
    const ord_t d = (ord_t)(nib(w, p) ^ nib(w, q));
    return (ord_t)(w ^ (d << (4 * p)) ^ (d << (4 * q)));
  }
  SD_HD int ael_head() const { return n_ael > 0 ? nib(ord, 0) : -1; }
  SD_HD int anext(int h) const { const ord_t w = ord; const int p = nib_find(w, h); if (p < 0 || p + 1 >= n_ael) return -1; return nib(w, p + 1); }
  SD_HD int aprev(int h) const { const ord_t w = ord; const int p = nib_find(w, h); if (p <= 0) return -1; return nib(w, p - 1); }
  SD_HD bool in_ael(int h) const { return nib_find(ord, h) >= 0; }
  SD_HD int alloc_slot() {
    const int f = freemask;
    if (!f) { status |= ST_OVERFLOW_AEL; return 0; }
#if defined(__HIP_DEVICE_COMPILE__)
    const int h = __ffs(f) - 1;
#else
    const int h = __builtin_ctz((unsigned)f);
#endif
    freemask = f & (f - 1);
    return h;
  }

  // ------------------------------------------------------------------ static edge data
  SD_HD const Prep* prep_of(int id) const { return (id >= EID_POLY) ? (const Prep*)prepB : (const Prep*)prepA; }
  static SD_HD int ring_nx(const Prep* q, int i) { return i + 1 == q->n ? 0 : i + 1; }
  static SD_HD int ring_pv(const Prep* q, int i) { return i == 0 ? q->n - 1 : i - 1; }
  struct ES { int bx, by, tx, ty, code; };
  static SD_HD ES edge_static(const Prep* q, int i) {
    BEAM_CNT(es_loads, 1);
    ES s;
    s.code = q->ecode[i];
    const PrepPt p0 = q->v[i], p1 = q->v[i + 1];
    const int x0 = p0.x, y0 = p0.y, x1 = p1.x, y1 = p1.y;
    if (s.code & 4) { s.bx = x1; s.by = y1; s.tx = x0; s.ty = y0; }
    else { s.bx = x0; s.by = y0; s.tx = x1; s.ty = y1; }
    return s;
  }
  // NextInLML of edge id (or -1)
  SD_HD int lml_id(int id, int code) const {
    const int c = code & 3;
    if (!c) return -1;
    const Prep* q = prep_of(id);
    const int base = id & ~(EID_POLY - 1), i = id & (EID_POLY - 1);
    return base + (c == 1 ? ring_nx(q, i) : ring_pv(q, i));
  }
  SD_HD ES es_of(int id) const { return edge_static(prep_of(id), id & (EID_POLY - 1)); }
  static SD_HD double es_dx(const ES& s) {
    const i64 dy = (i64)s.ty - s.by;
    if (dy == 0) return SD_HORIZONTAL;
    return (double)((i64)s.tx - s.bx) / (double)dy;
  }
  SD_HD void load_slot(int h, int id) {
    const ES s = es_of(id);
    botx[h] = s.bx; boty[h] = s.by; topx[h] = s.tx; topy[h] = s.ty;
    dx[h] = es_dx(s);
    eid[h] = (short)id; lmlc[h] = (signed char)(s.code & 11);
  }

  // ------------------------------------------------------------------ helpers (as SweepCore)
  SD_HD bool is_horz(int h) const { return topy[h] == boty[h]; }
  SD_HD i64 top_x(int h, i64 y) const {                                      // clipper.cpp:615-619
    BEAM_CNT(topx, 1);
    return (y == topy[h]) ? (i64)topx[h] : (i64)botx[h] + sd_round(dx[h] * (double)(y - boty[h]));
  }
  static SD_HD bool slopes_equal4(i64 x1, i64 y1, i64 x2, i64 y2, i64 x3, i64 y3, i64 x4, i64 y4) {  // :566-575
    return (y1 - y2) * (x3 - x4) == (x1 - x2) * (y3 - y4);
  }
  SD_HD bool slopes_equal_e(int e1, int e2) const {                          // :541-551
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
    const int g = n_gj;
    if (g < GJ) { gjop[FULLGJ ? g : 0] = op; gjx1[g] = x1; gjx2[g] = x2; gjy2[FULLGJ ? g : 0] = y2; n_gj = g + 1; }
    else { ++n_joins; status |= ST_OVERFLOW_GJ; }
  }
  SD_HD void horz_joins(int horz, int op1) {                                  // :2721-2732, 2774-2785
    ord_t w = hsel;
    for (int h = (int)(w & 15u); h != 15; w = (ord_t)((w >> 4) | ((ord_t)0xF << (ORD_BITS - 4))), h = (int)(w & 15u))
      if (outidx[h] >= 0 && horz_segments_overlap(botx[horz], topx[horz], botx[h], topx[h]))
        add_join(self().out_last_pt(h), op1, topx[h], topy[h]);
  }
  SD_HD void push_xtra(int y) {
    const int k = n_xtra;
    for (int i = 0; i < k; ++i) if (xtra[i] == y) return;
    if (k < NX) { xtra[k] = y; n_xtra = k + 1; } else status |= ST_OVERFLOW_GJ;
  }
  SD_HD int lm_y(int i) const {
    const int m = mlm[i];
    const Prep* q = (m & 128) ? (const Prep*)prepB : (const Prep*)prepA;
    const int l = q->lm_left[m & 127];
    const ES s = edge_static(q, l);
    return s.by;
  }
  // PopScanbeam :1341-1348 on the implicit queue (see header).  curY: the scan-line just processed.
  SD_HD bool pop_scanbeam(int curY, int& y) {
    bool have = false; int best = 0;
    if (cur_lm < n_lm) { best = next_lm_y; have = true; }
    const int k = n_xtra;
    for (int i = 0; i < k; ++i) { const int v = xtra[i]; if (!have || v > best) { best = v; have = true; } }
    const ord_t w = ord; const int n = n_ael;
    for (int p = 0; p < n; ++p) {
      const int h = nib(w, p);
      const int t = topy[h];
      if (t != boty[h] && t < curY && (!have || t > best)) { best = t; have = true; }
    }
    if (!have) return false;
    int kk = 0;
    for (int i = 0; i < k; ++i) { const int v = xtra[i]; if (v != best) { xtra[kk] = v; ++kk; } }
    n_xtra = kk;
    y = best;
    return true;
  }

  // ------------------------------------------------------------------ output (delegated to D)
  // Calls and the call graph (round 6).  A device function that calls another one has to keep its return address over that call: the
  // compiler parks it in a lane of a callee-saved VGPR, and that VGPR is stored to SCRATCH in the prologue and re-loaded -- with a full
  // wait on memory -- in front of every return.  The sweep of one pair returned from ~250 such functions (several per scan-beam), a
  // round trip to memory each: a large part of the ~0.5 ms a single sweep lasts, which is what bounds the late launches of the NMS.
  // So: the big blocks with ONE call site (run_sweep, process_horizontal, build_intersect_list, process_edges_at_top_of_scanbeam,
  // do_maxima, insert_local_minima_into_ael) are part of the kernel body (no code growth), and every real call is a LEAF -- the
  // blocks with several call sites (intersect_edges, add_local_min_poly, add_local_max_poly, out_add_pt, out_append, ...) have their
  // own callees inlined (the <true> forms below) and touch no scratch.  The code stays near the size of the instruction cache.
  template <bool IN> SD_HD int add_out_pt_t(int e, int px, int py) {
    BEAM_CNT(outpt, 1);
    if (IN) return self().out_add_pt_i(e, px, py);
    return self().out_add_pt(e, px, py);
  }
  SD_HD int add_out_pt(int e, int px, int py) { return add_out_pt_t<false>(e, px, py); }
  template <bool IN> SD_HD void add_local_max_poly_t(int e1, int e2, int px, int py) {           // :1884-1897
    add_out_pt_t<IN>(e1, px, py);
    if (outidx[e1] == outidx[e2]) {
      if (outidx[e1] >= 0) self().out_ring_closed(outidx[e1]);
      outidx[e1] = kUnassigned; outidx[e2] = kUnassigned;
    } else if (self().out_ring_before(outidx[e1], outidx[e2])) { if (IN) self().out_append_i(e1, e2); else self().out_append(e1, e2); }   // OutRec index order (:1893)
    else { if (IN) self().out_append_i(e2, e1); else self().out_append(e2, e1); }
  }
  SD_HDN void add_local_max_poly(int e1, int e2, int px, int py) { add_local_max_poly_t<true>(e1, e2, px, py); }
  SD_HDN int add_local_min_poly(int e1, int e2, int px, int py) { return add_local_min_poly_t<true>(e1, e2, px, py); }
  template <bool IN> SD_HD int add_local_min_poly_t(int e1, int e2, int px, int py) {            // :1841-1881
    int e, prevE, result;
    if (is_horz(e2) || dx[e1] > dx[e2]) {
      result = add_out_pt_t<IN>(e1, px, py);
      outidx[e2] = outidx[e1];
      side[e1] = kLeft; side[e2] = kRight;
      e = e1;
      prevE = (aprev(e) == e2) ? aprev(e2) : aprev(e);
    } else {
      result = add_out_pt_t<IN>(e2, px, py);
      outidx[e1] = outidx[e2];
      side[e1] = kRight; side[e2] = kLeft;
      e = e2;
      prevE = (aprev(e) == e1) ? aprev(e1) : aprev(e);
    }
    if (prevE >= 0 && outidx[prevE] >= 0 && topy[prevE] < py && topy[e] < py) {
      i64 xPrev = top_x(prevE, py), xE = top_x(e, py);
      if (xPrev == xE && wdelta[e] != 0 && wdelta[prevE] != 0 &&
          slopes_equal4(xPrev, py, topx[prevE], topy[prevE], xE, py, topx[e], topy[e])) {
        const int outPt = add_out_pt_t<IN>(prevE, px, py);
        add_join(result, outPt, topx[e], topy[e]);
      }
    }
    return result;
  }

  // ------------------------------------------------------------------ AEL
  SD_HD bool e2_inserts_before_e1(int e1, int e2) const {                   // :3278-3287
    if (curx[e2] == curx[e1]) {
      if (topy[e2] > topy[e1]) return (i64)topx[e2] < top_x(e1, topy[e2]);
      else return (i64)topx[e1] > top_x(e2, topy[e1]);
    } else return curx[e2] < curx[e1];
  }
  SD_HDN void insert_edge_into_ael(int edge, int startEdge) {                // :3319-3345
    const int n = n_ael;
    ord_t w = ord;
    int at;
    if (n == 0) at = 0;
    else if (startEdge < 0 && e2_inserts_before_e1(nib(w, 0), edge)) at = 0;
    else {
      int p = (startEdge < 0) ? 0 : nib_find(w, startEdge);
      if (p < 0) p = 0;
      while (p + 1 < n && !e2_inserts_before_e1(nib(w, p + 1), edge)) ++p;
      at = p + 1;
    }
    ord = nib_insert(w, at, edge);
    n_ael = n + 1;
  }
  SD_HD void delete_from_ael(int e) {                                       // :1367-1377
    const ord_t w = ord;
    const int p = nib_find(w, e);
    if (p < 0) return;
    ord = nib_remove(w, p);
    n_ael = n_ael - 1;
    freemask = freemask | (1 << e);
  }
  SD_HD void swap_positions_in_ael(int e1, int e2) {                        // :1395-1439
    const ord_t w = ord;
    const int p = nib_find(w, e1), q = nib_find(w, e2);
    if (p < 0 || q < 0) return;
    ord = nib_swap(w, p, q);
  }
  SD_HD void add_edge_to_sel(int edge) { hsel = (ord_t)(((ord_t)hsel << 4) | (ord_t)edge); }   // :1900-1917 (push at the head)
  // UpdateEdgeIntoAEL :1442-1462 ; the successor takes over the slot
  SD_HDN int update_edge_into_ael(int e) {
    BEAM_CNT(update, 1);
    const int id = eid[e];
    const int n = lml_id(id, lmlc[e]);
    if (n < 0) { status |= ST_FAIL; return e; }
    load_slot(e, n);
    curx[e] = botx[e]; cury[e] = boty[e];
    return e;
  }

  // ------------------------------------------------------------------ winding  (NonZero both, ctIntersection)
  SD_HDN void set_winding_count(int edge) {                                  // :1624-1722
    const ord_t w = ord;
    const int pe = nib_find(w, edge);
    int p = pe - 1;
    while (p >= 0 && (ptyp[nib(w, p)] != ptyp[edge] || wdelta[nib(w, p)] == 0)) --p;
    if (p < 0) {
      wcnt[edge] = wdelta[edge];
      wcnt2[edge] = 0;
      p = 0;
    } else {
      const int e = nib(w, p);
      if (wcnt[e] * wdelta[e] < 0) {
        int a = wcnt[e] < 0 ? -wcnt[e] : wcnt[e];
        if (a > 1) {
          if (wdelta[e] * wdelta[edge] < 0) wcnt[edge] = wcnt[e];
          else wcnt[edge] = (signed char)(wcnt[e] + wdelta[edge]);
        } else wcnt[edge] = (wdelta[edge] == 0 ? 1 : wdelta[edge]);
      } else {
        if (wdelta[edge] == 0) wcnt[edge] = (signed char)(wcnt[e] < 0 ? wcnt[e] - 1 : wcnt[e] + 1);
        else if (wdelta[e] * wdelta[edge] < 0) wcnt[edge] = wcnt[e];
        else wcnt[edge] = (signed char)(wcnt[e] + wdelta[edge]);
      }
      wcnt2[edge] = wcnt2[e];
      ++p;
    }
    for (; p < pe; ++p) wcnt2[edge] = (signed char)(wcnt2[edge] + wdelta[nib(w, p)]);
  }
  SD_HD bool is_contributing(int e) const {                                 // :1741-1838
    int a = wcnt[e] < 0 ? -wcnt[e] : wcnt[e];
    if (a != 1) return false;
    return wcnt2[e] != 0;
  }

  // ------------------------------------------------------------------ IntersectEdges  :2106-2298
  SD_HDN void intersect_edges(int e1, int e2, int px, int py) {
    BEAM_CNT(isect_edges, 1);
    bool c1 = outidx[e1] >= 0, c2 = outidx[e2] >= 0;
    if (ptyp[e1] == ptyp[e2]) {
      if (wcnt[e1] + wdelta[e2] == 0) wcnt[e1] = (signed char)-wcnt[e1]; else wcnt[e1] = (signed char)(wcnt[e1] + wdelta[e2]);
      if (wcnt[e2] - wdelta[e1] == 0) wcnt[e2] = (signed char)-wcnt[e2]; else wcnt[e2] = (signed char)(wcnt[e2] - wdelta[e1]);
    } else {
      wcnt2[e1] = (signed char)(wcnt2[e1] + wdelta[e2]);
      wcnt2[e2] = (signed char)(wcnt2[e2] - wdelta[e1]);
    }
    int e1Wc = wcnt[e1] < 0 ? -wcnt[e1] : wcnt[e1];
    int e2Wc = wcnt[e2] < 0 ? -wcnt[e2] : wcnt[e2];
    if (c1 && c2) {
      if ((e1Wc != 0 && e1Wc != 1) || (e2Wc != 0 && e2Wc != 1) || (ptyp[e1] != ptyp[e2])) {
        add_local_max_poly_t<true>(e1, e2, px, py);
      } else {
        add_out_pt_t<true>(e1, px, py);
        add_out_pt_t<true>(e2, px, py);
        signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s;
        signed char o = outidx[e1]; outidx[e1] = outidx[e2]; outidx[e2] = o;
      }
    } else if (c1) {
      if (e2Wc == 0 || e2Wc == 1) {
        add_out_pt_t<true>(e1, px, py);
        signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s;
        signed char o = outidx[e1]; outidx[e1] = outidx[e2]; outidx[e2] = o;
      }
    } else if (c2) {
      if (e1Wc == 0 || e1Wc == 1) {
        add_out_pt_t<true>(e2, px, py);
        signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s;
        signed char o = outidx[e1]; outidx[e1] = outidx[e2]; outidx[e2] = o;
      }
    } else if ((e1Wc == 0 || e1Wc == 1) && (e2Wc == 0 || e2Wc == 1)) {
      int e1Wc2 = wcnt2[e1] < 0 ? -wcnt2[e1] : wcnt2[e1];
      int e2Wc2 = wcnt2[e2] < 0 ? -wcnt2[e2] : wcnt2[e2];
      if (ptyp[e1] != ptyp[e2]) add_local_min_poly_t<true>(e1, e2, px, py);
      else if (e1Wc == 1 && e2Wc == 1) {
        if (e1Wc2 > 0 && e2Wc2 > 0) add_local_min_poly_t<true>(e1, e2, px, py);
      } else { signed char s = side[e1]; side[e1] = side[e2]; side[e2] = s; }
    }
  }

  // ------------------------------------------------------------------ InsertLocalMinimaIntoAEL  :1978-2077
  SD_BLK void insert_local_minima_into_ael(int botY) {
    while (cur_lm < n_lm && next_lm_y == botY) {
      const int m = mlm[cur_lm];
      ++cur_lm;
      if (cur_lm < n_lm) next_lm_y = lm_y(cur_lm);
      BEAM_CNT(lm, 1);
      const int poly = (m & 128) ? 1 : 0;
      const Prep* q = poly ? (const Prep*)prepB : (const Prep*)prepA;
      const int li = q->lm_left[m & 127], ri = q->lm_right[m & 127];
      const int lb = alloc_slot();
      const int rb = alloc_slot();
      if (status & ST_OVERFLOW_AEL) return;
      load_slot(lb, poly * EID_POLY + li);
      load_slot(rb, poly * EID_POLY + ri);
      // AddPath :1185-1189 + Reset :1260-1273
      const signed char wl = (ring_nx(q, li) == ri) ? -1 : 1;
      wdelta[lb] = wl; wdelta[rb] = (signed char)-wl;
      const signed char pt = poly ? (signed char)kSubject : (signed char)kClip;
      ptyp[lb] = pt; ptyp[rb] = pt;
      curx[lb] = botx[lb]; cury[lb] = boty[lb]; side[lb] = kLeft; outidx[lb] = kUnassigned; wcnt[lb] = 0; wcnt2[lb] = 0;
      curx[rb] = botx[rb]; cury[rb] = boty[rb]; side[rb] = kRight; outidx[rb] = kUnassigned; wcnt[rb] = 0; wcnt2[rb] = 0;

      int op1 = -1; bool have_op1 = false;
      insert_edge_into_ael(lb, -1);
      insert_edge_into_ael(rb, lb);
      set_winding_count(lb);
      wcnt[rb] = wcnt[lb]; wcnt2[rb] = wcnt2[lb];
      if (is_contributing(lb)) { op1 = add_local_min_poly(lb, rb, botx[lb], boty[lb]); have_op1 = true; }
      // InsertScanbeam(lb->Top.Y) is implicit (lb is never horizontal: AddPath puts a horizontal at a minimum on the right bound)
      if (is_horz(rb)) {
        add_edge_to_sel(rb);
        const int nid = lml_id(eid[rb], lmlc[rb]);
        if (nid >= 0) push_xtra(es_of(nid).ty);                              // :2020
      }
      if (is_horz(lb)) status |= ST_ITER;                                    // cannot happen (see above); flagged, never silently wrong

      if (have_op1 && is_horz(rb) && n_gj > 0 && wdelta[rb] != 0) {       // :2029-2040
        const int ng = n_gj;
        for (int g = 0; g < ng; ++g)
          if (horz_segments_overlap(gjx1[g], gjx2[g], botx[rb], topx[rb])) add_join(gjop[FULLGJ ? g : 0], op1, gjx2[g], gjy2[FULLGJ ? g : 0]);
      }
      const int lp = aprev(lb);
      if (outidx[lb] >= 0 && lp >= 0 && curx[lp] == botx[lb] && outidx[lp] >= 0 &&
          slopes_equal4(botx[lp], boty[lp], topx[lp], topy[lp], curx[lb], cury[lb], topx[lb], topy[lb]) &&
          wdelta[lb] != 0 && wdelta[lp] != 0) {
        const int op2 = add_out_pt(lp, botx[lb], boty[lb]);
        add_join(op1, op2, topx[lb], topy[lb]);
      }
      if (anext(lb) != rb) {
        const int rp = aprev(rb);
        if (outidx[rb] >= 0 && rp >= 0 && outidx[rp] >= 0 &&
            slopes_equal4(curx[rp], cury[rp], topx[rp], topy[rp], curx[rb], cury[rb], topx[rb], topy[rb]) &&
            wdelta[rb] != 0 && wdelta[rp] != 0) {
          const int op2 = add_out_pt(rp, botx[rb], boty[rb]);
          add_join(op1, op2, topx[rb], topy[rb]);
        }
        int e = anext(lb);
        int guard = 0;
        while (e >= 0 && e != rb) {
          intersect_edges(rb, e, curx[lb], cury[lb]);
          e = anext(e);
          if (++guard > K) { status |= ST_ITER; break; }
        }
      }
    }
  }

  // ------------------------------------------------------------------ horizontals  :2512-2824
  // GetMaximaPair :2538-2545 for edge id `id` (prepared per edge); returns an edge id or -1
  SD_HD int get_maxima_pair_id(int id) const {
    const Prep* q = prep_of(id);
    const int mp = q->mpair[id & (EID_POLY - 1)];
    return (mp == Prep::NONE) ? -1 : (id & ~(EID_POLY - 1)) + mp;
  }
  // slot holding edge id (or -1)
  SD_HD int slot_of(int id) const {
    const ord_t w = ord; const int n = n_ael;
    for (int p = 0; p < n; ++p) { const int h = nib(w, p); if (eid[h] == id) return h; }
    return -1;
  }
  // GetMaximaPairEx :2548-2555 for the edge in slot e.  Returns: -1 none; otherwise the pair's edge id, with
  // `slot` = its slot if it is in the AEL (else -1: the pair is a horizontal that has not entered the AEL yet).
  SD_HD int get_maxima_pair_ex(int e, int& slot) const {
    slot = -1;
    const int r = get_maxima_pair_id(eid[e]);
    if (r < 0) return -1;
    slot = slot_of(r);
    if (slot < 0) {
      const ES s = es_of(r);
      if (s.ty != s.by) return -1;            // not in the AEL and not horizontal
    }
    return r;
  }
  SD_BLK void process_horizontal(int horz) {
    BEAM_CNT(horz, 1);
    bool l2r; i64 hl, hr;
    if (botx[horz] < topx[horz]) { hl = botx[horz]; hr = topx[horz]; l2r = true; }
    else { hl = topx[horz]; hr = botx[horz]; l2r = false; }
    // eLast: last horizontal of the run in this bound; eMaxPair only if the run ends the bound (both prepared per edge)
    int eLast, eMaxPair = -1;
    {
      const int id = eid[horz];
      const Prep* q = prep_of(id);
      const int base = id & ~(EID_POLY - 1);
      const int li = q->hlast[id & (EID_POLY - 1)];
      eLast = base + li;
      if ((q->ecode[li] & 3) == 0) { const int mp = q->mpair[li]; if (mp != Prep::NONE) eMaxPair = base + mp; }
    }
    int op1 = -1; bool have_op1 = false;
    int guard = 0;
    for (;;) {
      const bool isLast = (eid[horz] == eLast);
      int e = l2r ? anext(horz) : aprev(horz);
      while (e >= 0) {
        if (++guard > 4 * K * K + 4 * MAXV) { status |= ST_ITER; return; }
        if ((l2r && curx[e] > hr) || (!l2r && curx[e] < hl)) break;
        if (curx[e] == topx[horz] && (lmlc[horz] & 3) != 0) {
          const ES s = es_of(lml_id(eid[horz], lmlc[horz]));
          if (dx[e] < es_dx(s)) break;
        }
        if (outidx[horz] >= 0) {
          op1 = add_out_pt(horz, curx[e], cury[e]);
          have_op1 = true;
          horz_joins(horz, op1);
          add_ghost_join(op1, curx[e], botx[horz], boty[horz]);
        }
        if (eid[e] == eMaxPair && isLast) {
          if (outidx[horz] >= 0) add_local_max_poly(horz, e, topx[horz], topy[horz]);
          delete_from_ael(horz);
          delete_from_ael(e);
          return;
        }
        if (l2r) intersect_edges(horz, e, curx[e], cury[horz]);
        else intersect_edges(e, horz, curx[e], cury[horz]);
        const int eNext = l2r ? anext(e) : aprev(e);
        swap_positions_in_ael(horz, e);
        e = eNext;
      }
      if ((lmlc[horz] & 8) == 0) break;                                      // no successor, or it is not horizontal
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
    if ((lmlc[horz] & 3) != 0) {
      if (outidx[horz] >= 0) {
        op1 = add_out_pt(horz, topx[horz], topy[horz]);
        horz = update_edge_into_ael(horz);
        if (wdelta[horz] == 0) return;
        const int ePrev = aprev(horz), eNext = anext(horz);
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
    for (;;) {
      const ord_t w = hsel;
      const int h = (int)(w & 15u);
      if (h == 15) break;
      hsel = (ord_t)((w >> 4) | ((ord_t)0xF << (ORD_BITS - 4)));                                   // DeleteFromSEL (head)
      process_horizontal(h);
      if (++guard > 4 * MAXV) { status |= ST_ITER; break; }
    }
  }

  // ------------------------------------------------------------------ intersections  :2827-2954, 622-689
  SD_HDN void intersect_point(int e1, int e2, i64& ipx, i64& ipy) const {
    double b1, b2;
    const double d1 = dx[e1], d2 = dx[e2];
    if (d1 == d2) { ipy = cury[e1]; ipx = top_x(e1, ipy); return; }
    else if (d1 == 0) {
      ipx = botx[e1];
      if (is_horz(e2)) ipy = boty[e2];
      else { b2 = (double)boty[e2] - ((double)botx[e2] / d2); ipy = sd_round((double)ipx / d2 + b2); }
    } else if (d2 == 0) {
      ipx = botx[e2];
      if (is_horz(e1)) ipy = boty[e1];
      else { b1 = (double)boty[e1] - ((double)botx[e1] / d1); ipy = sd_round((double)ipx / d1 + b1); }
    } else {
      b1 = (double)botx[e1] - (double)boty[e1] * d1;
      b2 = (double)botx[e2] - (double)boty[e2] * d2;
      double q = (b2 - b1) / (d1 - d2);
      ipy = sd_round(q);
      double a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
      if (a1 < a2) ipx = sd_round(d1 * q + b1);
      else ipx = sd_round(d2 * q + b2);
    }
    if (ipy < topy[e1] || ipy < topy[e2]) {
      if (topy[e1] > topy[e2]) ipy = topy[e1]; else ipy = topy[e2];
      double a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
      if (a1 < a2) ipx = top_x(e1, ipy); else ipx = top_x(e2, ipy);
    }
    if (ipy > cury[e1]) {
      ipy = cury[e1];
      double a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
      if (a1 > a2) ipx = top_x(e2, ipy); else ipx = top_x(e1, ipy);
    }
  }
  SD_BLK void build_intersect_list(int topY) {
    const int n = n_ael;
    if (n == 0) return;
    ord_t w = ord;
    for (int p = 0; p < n; ++p) { const int h = nib(w, p); curx[h] = (int)top_x(h, topY); }
    // bubble sort of the SEL copy (:2837-2862): a pass carries the largest element to the end, which is then cut off
    int m = n, guard = 0;
    bool isModified;
    do {
      isModified = false;
      int p = 0;
      while (p + 1 < m) {
        const int e = nib(w, p), eNext = nib(w, p + 1);
        if (curx[e] > curx[eNext]) {
          i64 px, py;
          intersect_point(e, eNext, px, py);
          if (py < topY) { px = top_x(e, topY); py = topY; }
          const int k = n_il;
          if (k < MAXIL) { ile1[k] = (signed char)e; ile2[k] = (signed char)eNext; ilx[k] = (int)px; ily[k] = (int)py; n_il = k + 1; }
          else status |= ST_OVERFLOW_IL;
          w = nib_swap(w, p, p + 1);
          isModified = true;
        }
        ++p;
        if (++guard > K * K * 2 + 8) { status |= ST_ITER; return; }
      }
      if (m > 1) --m; else break;
    } while (isModified);
  }
  SD_HDN bool fixup_intersection_order() {
    // CopyAELToSEL :1929-1939
    ord_t w = ord;
    const int n = n_il;
    // std::sort(IntersectListSort :2921-2924): n <= 16 -> insertion sort == stable sort by Y descending
    for (int i = 1; i < n; ++i) {
      const int y = ily[i], x = ilx[i]; const signed char a = ile1[i], b = ile2[i];
      int k = i;
      while (k > 0 && ily[k - 1] < y) { ily[k] = ily[k - 1]; ilx[k] = ilx[k - 1]; ile1[k] = ile1[k - 1]; ile2[k] = ile2[k - 1]; --k; }
      if (k != i) { ily[k] = y; ilx[k] = x; ile1[k] = a; ile2[k] = b; }
    }
    for (int i = 0; i < n; ++i) {
      int p = nib_find(w, ile1[i]), q = nib_find(w, ile2[i]);
      if (!(p - q == 1 || q - p == 1)) {
        int j = i + 1;
        while (j < n) {
          const int pj = nib_find(w, ile1[j]), qj = nib_find(w, ile2[j]);
          if (pj - qj == 1 || qj - pj == 1) break;
          j++;
        }
        if (j == n) return false;
        const int y = ily[i], x = ilx[i]; const signed char a = ile1[i], b = ile2[i];
        ily[i] = ily[j]; ilx[i] = ilx[j]; ile1[i] = ile1[j]; ile2[i] = ile2[j];
        ily[j] = y; ilx[j] = x; ile1[j] = a; ile2[j] = b;
        p = nib_find(w, ile1[i]); q = nib_find(w, ile2[i]);
      }
      w = nib_swap(w, p, q);
    }
    return true;
  }
  SD_HD bool process_intersections(int topY) {
    if (n_ael == 0) return true;
    n_il = 0;
    build_intersect_list(topY);
    const int n = n_il;
    if (n == 0) return true;
    BEAM_CNT(il_beams, 1); BEAM_CNT(isect_pt, n);
    if (n == 1 || fixup_intersection_order()) {
      for (int i = 0; i < n; ++i) {
        const int e1 = ile1[i], e2 = ile2[i];
        intersect_edges(e1, e2, ilx[i], ily[i]);
        swap_positions_in_ael(e1, e2);
      }
      n_il = 0;
    } else return false;
    return true;
  }

  // ------------------------------------------------------------------ top of scan-beam  :2957-3113
  SD_BLK void do_maxima(int e) {
    BEAM_CNT(maxima, 1);
    int mp;
    const int mpid = get_maxima_pair_ex(e, mp);
    if (mpid < 0) {
      if (outidx[e] >= 0) add_out_pt(e, topx[e], topy[e]);
      delete_from_ael(e);
      return;
    }
    if (mp < 0) { status |= ST_FAIL; return; }      // unreachable: do_maxima is only entered with the pair in the AEL
    int eNext = anext(e);
    int guard = 0;
    while (eNext >= 0 && eNext != mp) {
      intersect_edges(e, eNext, topx[e], topy[e]);
      swap_positions_in_ael(e, eNext);
      eNext = anext(e);
      if (++guard > K) { status |= ST_ITER; break; }
    }
    if (outidx[e] == kUnassigned && outidx[mp] == kUnassigned) {
      delete_from_ael(e); delete_from_ael(mp);
    } else if (outidx[e] >= 0 && outidx[mp] >= 0) {
      add_local_max_poly(e, mp, topx[e], topy[e]);
      delete_from_ael(e); delete_from_ael(mp);
    } else status |= ST_FAIL;   // "DoMaxima error" -> Execute fails, empty solution
  }
  SD_BLK void process_edges_at_top_of_scanbeam(int topY) {
    int e = ael_head();
    int guard = 0;
    while (e >= 0) {
      if (++guard > 4 * MAXV) { status |= ST_ITER; break; }
      bool isMax = (topy[e] == topY && (lmlc[e] & 3) == 0);
      if (isMax) {
        int mps;
        const int mp = get_maxima_pair_ex(e, mps);
        bool mpHorz = false;
        if (mp >= 0) { if (mps >= 0) mpHorz = is_horz(mps); else mpHorz = true; }
        isMax = (mp < 0 || !mpHorz);
      }
      if (isMax) {
        const int ePrev = aprev(e);
        do_maxima(e);
        if (status & ST_FAIL) return;
        e = (ePrev < 0) ? ael_head() : anext(ePrev);
      } else {
        if (topy[e] == topY && (lmlc[e] & 8) != 0) {
          e = update_edge_into_ael(e);
          if (outidx[e] >= 0) add_out_pt(e, botx[e], boty[e]);
          add_edge_to_sel(e);
        } else {
          curx[e] = (int)top_x(e, topY);
          cury[e] = topY;
        }
        e = anext(e);
      }
    }
    process_horizontals();
    e = ael_head();
    guard = 0;
    while (e >= 0) {
      if (++guard > 4 * MAXV) { status |= ST_ITER; break; }
      if (topy[e] == topY && (lmlc[e] & 3) != 0) {
        bool op = false; int oph = -1;
        if (outidx[e] >= 0) { oph = add_out_pt(e, topx[e], topy[e]); op = true; }
        e = update_edge_into_ael(e);
        const int ePrev = aprev(e), eNext = anext(e);
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
      e = anext(e);
    }
  }

  // ------------------------------------------------------------------ Execute  :1560-1621, 1247-1276
  SD_HD void reset_core(const Prep* a, const Prep* b) {
    prepA = a; prepB = b;
    orgx = a->n ? a->v[0].x : (b->n ? b->v[0].x : 0); orgy = a->n ? a->v[0].y : (b->n ? b->v[0].y : 0);      // (before any coordinate is stored)
    ord = ALLF; hsel = ALLF; n_ael = 0; freemask = (1 << K) - 1;
    n_lm = 0; cur_lm = 0; n_il = 0; status = ST_OK; n_joins = 0; n_gj = 0; n_xtra = 0;
  }
  // Runs the sweep.  Returns false if Clipper's Execute would fail (empty solution).
  SD_BLK bool run_sweep() {
    const Prep* a = prepA; const Prep* b = prepB;
    status |= (a->status | b->status);
    // merged local-minima list: stable by Y descending, polygon A (added first, :157) before B on ties
    {
      const int na = a->n ? a->n_lm : 0, nb = b->n ? b->n_lm : 0;
      if (na + nb > 16) { status |= ST_OVERFLOW_LM; return false; }
      int ia = 0, ib = 0, k = 0;
      while (ia < na || ib < nb) {
        bool takeA;
        if (ia >= na) takeA = false;
        else if (ib >= nb) takeA = true;
        else takeA = edge_static(a, a->lm_left[ia]).by >= edge_static(b, b->lm_left[ib]).by;
        if (takeA) { mlm[k] = (unsigned char)ia; ++ia; } else { mlm[k] = (unsigned char)(128 + ib); ++ib; }
        ++k;
      }
      n_lm = k;
    }
    if (n_lm == 0) return true;
    cur_lm = 0;
    next_lm_y = lm_y(0);
    int botY = next_lm_y, topY = 0;
    int guard = 0;
    bool ok = true;
    for (;;) {
      insert_local_minima_into_ael(botY);              // (in front of the loop and at its end in Clipper: ONE call site here, same sequence)
      if (status & (ST_OVERFLOW_AEL | ST_OVERFLOW_REC | ST_OVERFLOW_IL | ST_ITER)) { ok = false; break; }
      const bool popped = pop_scanbeam(botY, topY);
      if (!popped) break;
      if (++guard > 4 * MAXV + 8) { status |= ST_ITER; break; }
      BEAM_CNT(beams, 1); BEAM_CNT(ael_sum, n_ael);
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
// Beam: the fast variant (rings as {front, back, running shoelace sum}), exact whenever the reference
// records no joins for the pair (n_joins == 0) -- same contract as Sweep in clip_sweep.h.
template <int MAXV, int K, int MAXIL, int MAXREC, class P = PlainStorage>
struct Beam : BeamCore<Beam<MAXV, K, MAXIL, MAXREC, P>, P, MAXV, K, MAXIL> {
  typedef BeamCore<Beam<MAXV, K, MAXIL, MAXREC, P>, P, MAXV, K, MAXIL> B;
  using B::outidx; using B::side; using B::status; using B::ord; using B::n_ael;
  static constexpr unsigned RR = P::template coord_region<MAXREC>();
  static constexpr unsigned O_RFX = B::O_CORE_END, O_RFY = O_RFX + RR, O_RLX = O_RFY + RR, O_RLY = O_RLX + RR, O_RSUM = O_RLY + RR,
                            O_RSER = O_RSUM + P::template region<i64, MAXREC>(), O_NREC = O_RSER + P::template region<unsigned char, MAXREC>(),
                            O_RFREE = O_NREC + P::template region<int, 1>(), O_TWICE = O_RFREE + P::template region<int, 1>(),
                            O_SABS = O_TWICE + P::template region<i64, 1>(), O_END = O_SABS + P::template region<i64, 1>();
  typename P::template CoordArr<MAXREC, O_RFX, B::O_ORGX, B::O_STATUS> rfx; typename P::template CoordArr<MAXREC, O_RFY, B::O_ORGY, B::O_STATUS> rfy;
  typename P::template CoordArr<MAXREC, O_RLX, B::O_ORGX, B::O_STATUS> rlx; typename P::template CoordArr<MAXREC, O_RLY, B::O_ORGY, B::O_STATUS> rly;
  typename P::template Arr<i64, MAXREC, O_RSUM> rsum;
  static constexpr unsigned lds_bytes() { return O_END; }
  typename P::template Arr<unsigned char, MAXREC, O_RSER> rser;   // creation order of the ring held in a slot (Clipper's OutRec index)
  typename P::template Scalar<int, O_NREC> n_rec;            // rings created so far
  typename P::template Scalar<int, O_RFREE> rfree;           // free ring slots (closed / appended rings are never referenced again)
  typename P::template Scalar<i64, O_TWICE> twice_area;      // sum over closed rings of |2*area|
  typename P::template Scalar<i64, O_SABS> sum_abs_terms;    // sum of |cross| terms (exactness bound for the float path)
  SD_HD void term(i64 c) { sum_abs_terms += sd_abs64(c); }
  SD_HDN int out_add_pt(int e, int px, int py) { return out_add_pt_i(e, px, py); }
  SD_HDN void out_append(int e1, int e2) { out_append_i(e1, e2); }
  SD_HD int out_add_pt_i(int e, int px, int py) {                            // :2463-2499
    int r = outidx[e];
    if (r < 0) {
      const int f = rfree;
      if (!f || n_rec >= 255) { status |= ST_OVERFLOW_REC; return -1; }
#if defined(__HIP_DEVICE_COMPILE__)
      r = __ffs(f) - 1;
#else
      r = __builtin_ctz((unsigned)f);
#endif
      rfree = f & (f - 1);
      rser[r] = (unsigned char)n_rec; n_rec = n_rec + 1;
      rfx[r] = rlx[r] = px; rfy[r] = rly[r] = py; rsum[r] = 0;
      outidx[e] = (signed char)r;
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
    if (r < 0 || r >= MAXREC) { status |= ST_OVERFLOW_REC; return; }
    i64 c = sd_cross(rlx[r], rly[r], rfx[r], rfy[r]); term(c);
    twice_area += sd_abs64(rsum[r] + c);
    rfree = rfree | (1 << r);
  }
  SD_HD bool out_ring_before(int r1, int r2) const {
    if (r1 < 0 || r2 < 0) return r1 < r2;
    return rser[r1] < rser[r2];
  }
  SD_HD void out_append_i(int e1, int e2) {                                  // :2367-2460
    int r1 = outidx[e1], r2 = outidx[e2];
    if (r1 < 0 || r2 < 0) { status |= ST_OVERFLOW_REC; return; }          // only after a ring-capacity overflow
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
    const int okIdx = r1, obsolete = r2;
    const signed char s1 = side[e1];
    outidx[e1] = kUnassigned; outidx[e2] = kUnassigned;
    rfree = rfree | (1 << obsolete);
    const typename B::ord_t w = ord; const int n = n_ael;
    for (int p = 0; p < n; ++p) {
      const int e = B::nib(w, p);
      if (outidx[e] == obsolete) { outidx[e] = (signed char)okIdx; side[e] = s1; break; }
    }
  }
  SD_HD void out_add_join(int, int, int, int) {}
  SD_HD int out_last_pt(int) { return -1; }
  SD_HD int out_last_pt_x(int e) { const int r = outidx[e]; if (r < 0) return 0; return (side[e] == kLeft) ? rfx[r] : rlx[r]; }
  SD_HD void reset_state(const PolyPrep<MAXV>* a, const PolyPrep<MAXV>* b) { B::reset_core(a, b); n_rec = 0; rfree = (1 << MAXREC) - 1; twice_area = 0; sum_abs_terms = 0; }
  // Returns 2*area of (A ∩ B) as the reference would sum it (0 if Clipper's Execute fails).
  SD_HD i64 execute() { return B::run_sweep() ? (i64)twice_area : 0; }
};

}  // namespace sdclip
