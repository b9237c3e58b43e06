// clip_sweep_full.h -- exact-join variant of the scan-beam sweep.
//
// The fast variant (clip_sweep.h, Sweep) keeps every output ring as {front, back, shoelace sum}
// and is exact while the reference's Clipper records no joins for the pair.  When it does
// (two output rings share an edge, ~1-4 % of pairs on the integer lattice), Clipper's
// JoinCommonEdges (external/clipper/clipper.cpp:3679-3783) may merge rings or split one ring
// in two, and because stardist2d.cpp:161-164 sums |area| per output path that changes the
// result when the pieces have opposite orientation.  This variant therefore keeps the real
// point rings (doubly linked, index based), the hole-state bookkeeping that decides ring
// orientation before the joins (SetHoleState :2301-2324, GetLowermostRec :2327-2344,
// GetBottomPt :822-857, FirstIsBottomPt :798-819), and restates JoinPoints / JoinHorz /
// JoinCommonEdges.  FixupOutPolygon (:3143-3181) only removes duplicate / collinear points and
// never changes a ring's area, so it is not needed for the area result.
#pragma once
#include "clip_sweep.h"

namespace sdclip {

template <int MAXV, int MAXIL, int MAXREC, int MAXPT, int MAXJ, class P = PlainStorage>
struct SweepFull : SweepCore<SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ, P>, P, MAXV, MAXIL> {
  typedef SweepCore<SweepFull<MAXV, MAXIL, MAXREC, MAXPT, MAXJ, P>, P, MAXV, MAXIL> B;
  static constexpr unsigned lds_bytes() { return B::O_CORE_END; }   // only the core arrays follow the policy; rings stay private
  using B::outidx; using B::side; using B::status; using B::ael; using B::anext; using B::aprev; using B::wdelta;
  // ---- output points (rings)
  int px[MAXPT], py[MAXPT];
  short pn[MAXPT], pp[MAXPT], pidx[MAXPT];
  int n_pt;
  // ---- output records
  short r_idx[MAXREC], r_first[MAXREC], r_pts[MAXREC], r_bot[MAXREC];
  bool r_hole[MAXREC];
  int n_rec;
  // ---- joins
  short j1[MAXJ], j2[MAXJ];
  int jx[MAXJ], jy[MAXJ];
  int n_j;
  i64 sum_abs_terms;

  SD_HD void reset_state() { B::reset_core(); n_pt = 0; n_rec = 0; n_j = 0; sum_abs_terms = 0; }

  // ------------------------------------------------------------------ small helpers
  SD_HD int alloc_pt(int x, int y, int idx) {
    if (n_pt >= MAXPT) { status |= ST_OVERFLOW_REC; return MAXPT - 1; }
    const int p = n_pt++;
    px[p] = x; py[p] = y; pidx[p] = (short)idx; pn[p] = (short)p; pp[p] = (short)p;
    return p;
  }
  SD_HD int create_outrec() {                                                // :1380-1392
    if (n_rec >= MAXREC) { status |= ST_OVERFLOW_REC; return MAXREC - 1; }
    const int r = n_rec++;
    r_hole[r] = false; r_first[r] = -1; r_pts[r] = -1; r_bot[r] = -1; r_idx[r] = (short)r;
    return r;
  }
  SD_HD bool pt_eq(int a, int b) const { return px[a] == px[b] && py[a] == py[b]; }
  SD_HD void reverse_links(int p) {                                          // ReversePolyPtLinks :692-703
    if (p < 0) return;
    int p1 = p;
    int guard = 0;
    do {
      const int p2 = pn[p1];
      pn[p1] = pp[p1]; pp[p1] = (short)p2;
      p1 = p2;
      if (++guard > MAXPT) { status |= ST_ITER; break; }
    } while (p1 != p);
  }
  SD_HD double ring_area(int op) const {                                     // Area(OutPt*) :406-416
    if (op < 0) return 0;
    const int start = op;
    double a = 0;
    int guard = 0;
    do {
      a += (double)((i64)px[pp[op]] + px[op]) * (double)((i64)py[pp[op]] - py[op]);
      op = pn[op];
      if (++guard > MAXPT) break;
    } while (op != start);
    return a * 0.5;
  }
  static SD_HD double get_dx(int x1, int y1, int x2, int y2) {               // GetDx :584-588
    return (y1 == y2) ? SD_HORIZONTAL : (double)((i64)x2 - x1) / (double)((i64)y2 - y1);
  }
  static SD_HD double dabs(double v) { return v < 0 ? -v : v; }

  // ------------------------------------------------------------------ hole state   :2301-2324
  SD_HD void set_hole_state(int e, int r) {
    int e2 = aprev[e];
    int eTmp = -1;
    while (e2 >= 0) {
      if (outidx[e2] >= 0 && wdelta[e2] != 0) {
        if (eTmp < 0) eTmp = e2;
        else if (outidx[eTmp] == outidx[e2]) eTmp = -1;
      }
      e2 = aprev[e2];
    }
    if (eTmp < 0) { r_first[r] = -1; r_hole[r] = false; }
    else { r_first[r] = outidx[eTmp]; r_hole[r] = !r_hole[r_first[r]]; }
  }

  // ------------------------------------------------------------------ AddOutPt   :2463-2499
  SD_HDN int out_add_pt(int e, int x, int y) {
    if (outidx[e] < 0) {
      const int r = create_outrec();
      const int op = alloc_pt(x, y, r);
      r_pts[r] = (short)op;
      set_hole_state(e, r);
      outidx[e] = (short)r;
      return op;
    }
    const int r = outidx[e];
    const int op = r_pts[r];
    const bool toFront = (side[e] == kLeft);
    if (toFront && x == px[op] && y == py[op]) return op;
    else if (!toFront && x == px[pp[op]] && y == py[pp[op]]) return pp[op];
    const int nw = alloc_pt(x, y, r_idx[r]);
    pn[nw] = (short)op; pp[nw] = pp[op];
    pn[pp[nw]] = (short)nw; pp[op] = (short)nw;
    if (toFront) r_pts[r] = (short)nw;
    return nw;
  }
  SD_HD int out_last_pt(int e) { const int r = outidx[e]; return (side[e] == kLeft) ? (int)r_pts[r] : (int)pp[r_pts[r]]; }  // :2502-2509
  SD_HD int out_last_pt_x(int e) { return px[out_last_pt(e)]; }
  SD_HD void out_ring_closed(int) {}
  SD_HD void out_add_join(int op1, int op2, int offx, int offy) {            // :1942-1949
    if (n_j >= MAXJ) { status |= ST_OVERFLOW_REC; return; }
    j1[n_j] = (short)op1; j2[n_j] = (short)op2; jx[n_j] = offx; jy[n_j] = offy; ++n_j;
  }

  // ------------------------------------------------------------------ bottom point   :798-857
  SD_HDN bool first_is_bottom_pt(int b1, int b2) const {
    int p = pp[b1];
    while (pt_eq(p, b1) && p != b1) p = pp[p];
    const double dx1p = dabs(get_dx(px[b1], py[b1], px[p], py[p]));
    p = pn[b1];
    while (pt_eq(p, b1) && p != b1) p = pn[p];
    const double dx1n = dabs(get_dx(px[b1], py[b1], px[p], py[p]));
    p = pp[b2];
    while (pt_eq(p, b2) && p != b2) p = pp[p];
    const double dx2p = dabs(get_dx(px[b2], py[b2], px[p], py[p]));
    p = pn[b2];
    while (pt_eq(p, b2) && p != b2) p = pn[p];
    const double dx2n = dabs(get_dx(px[b2], py[b2], px[p], py[p]));
    const double mx1 = dx1p > dx1n ? dx1p : dx1n, mn1 = dx1p < dx1n ? dx1p : dx1n;
    const double mx2 = dx2p > dx2n ? dx2p : dx2n, mn2 = dx2p < dx2n ? dx2p : dx2n;
    // std::max(a,b) returns a when equal; value-wise identical
    if (mx1 == mx2 && mn1 == mn2) return ring_area(b1) > 0;
    else return (dx1p >= dx2p && dx1p >= dx2n) || (dx1n >= dx2p && dx1n >= dx2n);
  }
  SD_HDN int get_bottom_pt(int ppt) {
    int dups = -1;
    int p = pn[ppt];
    int guard = 0;
    while (p != ppt) {
      if (py[p] > py[ppt]) { ppt = p; dups = -1; }
      else if (py[p] == py[ppt] && px[p] <= px[ppt]) {
        if (px[p] < px[ppt]) { dups = -1; ppt = p; }
        else { if (pn[p] != ppt && pp[p] != ppt) dups = p; }
      }
      p = pn[p];
      if (++guard > 2 * MAXPT) { status |= ST_ITER; break; }
    }
    if (dups >= 0) {
      guard = 0;
      while (dups != p) {
        if (!first_is_bottom_pt(p, dups)) ppt = dups;
        dups = pn[dups];
        while (!pt_eq(dups, ppt)) { dups = pn[dups]; if (++guard > 4 * MAXPT) { status |= ST_ITER; return ppt; } }
        if (++guard > 4 * MAXPT) { status |= ST_ITER; break; }
      }
    }
    return ppt;
  }
  SD_HD int get_lowermost_rec(int r1, int r2) {                              // :2327-2344
    if (r_bot[r1] < 0) r_bot[r1] = (short)get_bottom_pt(r_pts[r1]);
    if (r_bot[r2] < 0) r_bot[r2] = (short)get_bottom_pt(r_pts[r2]);
    const int o1 = r_bot[r1], o2 = r_bot[r2];
    if (py[o1] > py[o2]) return r1;
    else if (py[o1] < py[o2]) return r2;
    else if (px[o1] < px[o2]) return r1;
    else if (px[o1] > px[o2]) return r2;
    else if (pn[o1] == o1) return r2;
    else if (pn[o2] == o2) return r1;
    else if (first_is_bottom_pt(o1, o2)) return r1;
    else return r2;
  }
  SD_HD bool rec1_right_of_rec2(int r1, int r2) const {                      // :2347-2355
    int guard = 0;
    do {
      r1 = r_first[r1];
      if (r1 == r2) return true;
      if (++guard > MAXREC) return false;
    } while (r1 >= 0);
    return false;
  }
  SD_HD int get_outrec(int idx) const {                                      // :2358-2364
    int r = idx, guard = 0;
    while (r != r_idx[r]) { r = r_idx[r]; if (++guard > MAXREC) break; }
    return r;
  }

  // ------------------------------------------------------------------ AppendPolygon   :2367-2460
  SD_HDN void out_append(int e1, int e2) {
    const int r1 = outidx[e1], r2 = outidx[e2];
    int holeStateRec;
    if (rec1_right_of_rec2(r1, r2)) holeStateRec = r2;
    else if (rec1_right_of_rec2(r2, r1)) holeStateRec = r1;
    else holeStateRec = get_lowermost_rec(r1, r2);
    const int p1_lft = r_pts[r1], p1_rt = pp[p1_lft];
    const int p2_lft = r_pts[r2], p2_rt = pp[p2_lft];
    if (side[e1] == kLeft) {
      if (side[e2] == kLeft) {
        reverse_links(p2_lft);
        pn[p2_lft] = (short)p1_lft; pp[p1_lft] = (short)p2_lft;
        pn[p1_rt] = (short)p2_rt; pp[p2_rt] = (short)p1_rt;
        r_pts[r1] = (short)p2_rt;
      } else {
        pn[p2_rt] = (short)p1_lft; pp[p1_lft] = (short)p2_rt;
        pp[p2_lft] = (short)p1_rt; pn[p1_rt] = (short)p2_lft;
        r_pts[r1] = (short)p2_lft;
      }
    } else {
      if (side[e2] == kRight) {
        reverse_links(p2_lft);
        pn[p1_rt] = (short)p2_rt; pp[p2_rt] = (short)p1_rt;
        pn[p2_lft] = (short)p1_lft; pp[p1_lft] = (short)p2_lft;
      } else {
        pn[p1_rt] = (short)p2_lft; pp[p2_lft] = (short)p1_rt;
        pp[p1_lft] = (short)p2_rt; pn[p2_rt] = (short)p1_lft;
      }
    }
    r_bot[r1] = -1;
    if (holeStateRec == r2) {
      if (r_first[r2] != r1) r_first[r1] = r_first[r2];
      r_hole[r1] = r_hole[r2];
    }
    r_pts[r2] = -1; r_bot[r2] = -1; r_first[r2] = (short)r1;
    const int okIdx = r1, obsolete = r2;
    outidx[e1] = kUnassigned; outidx[e2] = kUnassigned;
    for (int e = ael; e >= 0; e = anext[e]) {
      if (outidx[e] == obsolete) { outidx[e] = (short)okIdx; side[e] = side[e1]; break; }
    }
    r_idx[r2] = r_idx[r1];
  }

  // ------------------------------------------------------------------ joins   :3348-3783
  SD_HDN int dup_out_pt(int o, bool insertAfter) {                            // :3348-3368
    const int r = alloc_pt(px[o], py[o], pidx[o]);
    if (insertAfter) { pn[r] = pn[o]; pp[r] = (short)o; pp[pn[o]] = (short)r; pn[o] = (short)r; }
    else { pp[r] = pp[o]; pn[r] = (short)o; pn[pp[o]] = (short)r; pp[o] = (short)r; }
    return r;
  }
  static SD_HD bool get_overlap(i64 a1, i64 a2, i64 b1, i64 b2, i64& L, i64& R) {   // :3290-3304
    if (a1 < a2) {
      if (b1 < b2) { L = a1 > b1 ? a1 : b1; R = a2 < b2 ? a2 : b2; }
      else { L = a1 > b2 ? a1 : b2; R = a2 < b1 ? a2 : b1; }
    } else {
      if (b1 < b2) { L = a2 > b1 ? a2 : b1; R = a1 < b2 ? a1 : b2; }
      else { L = a2 > b2 ? a2 : b2; R = a1 < b1 ? a1 : b1; }
    }
    return L < R;
  }
  SD_HDN bool join_horz(int op1, int op1b, int op2, int op2b, int ptx, int pty, bool discardLeft) {   // :3371-3456
    const bool d1_l2r = !(px[op1] > px[op1b]);
    const bool d2_l2r = !(px[op2] > px[op2b]);
    if (d1_l2r == d2_l2r) return false;
    int guard = 0;
    if (d1_l2r) {
      while (px[pn[op1]] <= ptx && px[pn[op1]] >= px[op1] && py[pn[op1]] == pty) { op1 = pn[op1]; if (++guard > MAXPT) break; }
      if (discardLeft && px[op1] != ptx) op1 = pn[op1];
      op1b = dup_out_pt(op1, !discardLeft);
      if (px[op1b] != ptx || py[op1b] != pty) { op1 = op1b; px[op1] = ptx; py[op1] = pty; op1b = dup_out_pt(op1, !discardLeft); }
    } else {
      while (px[pn[op1]] >= ptx && px[pn[op1]] <= px[op1] && py[pn[op1]] == pty) { op1 = pn[op1]; if (++guard > MAXPT) break; }
      if (!discardLeft && px[op1] != ptx) op1 = pn[op1];
      op1b = dup_out_pt(op1, discardLeft);
      if (px[op1b] != ptx || py[op1b] != pty) { op1 = op1b; px[op1] = ptx; py[op1] = pty; op1b = dup_out_pt(op1, discardLeft); }
    }
    guard = 0;
    if (d2_l2r) {
      while (px[pn[op2]] <= ptx && px[pn[op2]] >= px[op2] && py[pn[op2]] == pty) { op2 = pn[op2]; if (++guard > MAXPT) break; }
      if (discardLeft && px[op2] != ptx) op2 = pn[op2];
      op2b = dup_out_pt(op2, !discardLeft);
      if (px[op2b] != ptx || py[op2b] != pty) { op2 = op2b; px[op2] = ptx; py[op2] = pty; op2b = dup_out_pt(op2, !discardLeft); }
    } else {
      while (px[pn[op2]] >= ptx && px[pn[op2]] <= px[op2] && py[pn[op2]] == pty) { op2 = pn[op2]; if (++guard > MAXPT) break; }
      if (!discardLeft && px[op2] != ptx) op2 = pn[op2];
      op2b = dup_out_pt(op2, discardLeft);
      if (px[op2b] != ptx || py[op2b] != pty) { op2 = op2b; px[op2] = ptx; py[op2] = pty; op2b = dup_out_pt(op2, discardLeft); }
    }
    if (d1_l2r == discardLeft) {
      pp[op1] = (short)op2; pn[op2] = (short)op1; pn[op1b] = (short)op2b; pp[op2b] = (short)op1b;
    } else {
      pn[op1] = (short)op2; pp[op2] = (short)op1; pp[op1b] = (short)op2b; pn[op2b] = (short)op1b;
    }
    return true;
  }
  SD_HD bool slopes_eq_pts(int a, int b, int offx, int offy) const {         // SlopesEqual(pt1, pt2, pt3) :554-563
    return ((i64)py[a] - py[b]) * ((i64)px[b] - offx) == ((i64)px[a] - px[b]) * ((i64)py[b] - offy);
  }
  SD_HDN bool join_points(int j, int outRec1, int outRec2) {                  // :3458-3615
    int op1 = j1[j], op1b;
    int op2 = j2[j], op2b;
    const int offx = jx[j], offy = jy[j];
    const bool isHorizontal = (py[op1] == offy);
    if (isHorizontal && offx == px[op1] && offy == py[op1] && offx == px[op2] && offy == py[op2]) {
      // strictly-simple join
      if (outRec1 != outRec2) return false;
      op1b = pn[op1];
      while (op1b != op1 && px[op1b] == offx && py[op1b] == offy) op1b = pn[op1b];
      const bool reverse1 = (py[op1b] > offy);
      op2b = pn[op2];
      while (op2b != op2 && px[op2b] == offx && py[op2b] == offy) op2b = pn[op2b];
      const bool reverse2 = (py[op2b] > offy);
      if (reverse1 == reverse2) return false;
      if (reverse1) {
        op1b = dup_out_pt(op1, false); op2b = dup_out_pt(op2, true);
        pp[op1] = (short)op2; pn[op2] = (short)op1; pn[op1b] = (short)op2b; pp[op2b] = (short)op1b;
      } else {
        op1b = dup_out_pt(op1, true); op2b = dup_out_pt(op2, false);
        pn[op1] = (short)op2; pp[op2] = (short)op1; pp[op1b] = (short)op2b; pn[op2b] = (short)op1b;
      }
      j1[j] = (short)op1; j2[j] = (short)op1b;
      return true;
    } else if (isHorizontal) {
      op1b = op1;
      int guard = 0;
      while (py[pp[op1]] == py[op1] && pp[op1] != op1b && pp[op1] != op2) { op1 = pp[op1]; if (++guard > MAXPT) break; }
      while (py[pn[op1b]] == py[op1b] && pn[op1b] != op1 && pn[op1b] != op2) { op1b = pn[op1b]; if (++guard > 2 * MAXPT) break; }
      if (pn[op1b] == op1 || pn[op1b] == op2) return false;
      op2b = op2;
      guard = 0;
      while (py[pp[op2]] == py[op2] && pp[op2] != op2b && pp[op2] != op1b) { op2 = pp[op2]; if (++guard > MAXPT) break; }
      while (py[pn[op2b]] == py[op2b] && pn[op2b] != op2 && pn[op2b] != op1) { op2b = pn[op2b]; if (++guard > 2 * MAXPT) break; }
      if (pn[op2b] == op2 || pn[op2b] == op1) return false;
      i64 L, R;
      if (!get_overlap(px[op1], px[op1b], px[op2], px[op2b], L, R)) return false;
      int ptx, pty; bool discardLeft;
      if (px[op1] >= L && px[op1] <= R) { ptx = px[op1]; pty = py[op1]; discardLeft = (px[op1] > px[op1b]); }
      else if (px[op2] >= L && px[op2] <= R) { ptx = px[op2]; pty = py[op2]; discardLeft = (px[op2] > px[op2b]); }
      else if (px[op1b] >= L && px[op1b] <= R) { ptx = px[op1b]; pty = py[op1b]; discardLeft = px[op1b] > px[op1]; }
      else { ptx = px[op2b]; pty = py[op2b]; discardLeft = (px[op2b] > px[op2]); }
      j1[j] = (short)op1; j2[j] = (short)op2;
      return join_horz(op1, op1b, op2, op2b, ptx, pty, discardLeft);
    } else {
      op1b = pn[op1];
      while (pt_eq(op1b, op1) && op1b != op1) op1b = pn[op1b];
      const bool Reverse1 = (py[op1b] > py[op1]) || !slopes_eq_pts(op1, op1b, offx, offy);
      if (Reverse1) {
        op1b = pp[op1];
        while (pt_eq(op1b, op1) && op1b != op1) op1b = pp[op1b];
        if ((py[op1b] > py[op1]) || !slopes_eq_pts(op1, op1b, offx, offy)) return false;
      }
      op2b = pn[op2];
      while (pt_eq(op2b, op2) && op2b != op2) op2b = pn[op2b];
      const bool Reverse2 = (py[op2b] > py[op2]) || !slopes_eq_pts(op2, op2b, offx, offy);
      if (Reverse2) {
        op2b = pp[op2];
        while (pt_eq(op2b, op2) && op2b != op2) op2b = pp[op2b];
        if ((py[op2b] > py[op2]) || !slopes_eq_pts(op2, op2b, offx, offy)) return false;
      }
      if (op1b == op1 || op2b == op2 || op1b == op2b || (outRec1 == outRec2 && Reverse1 == Reverse2)) return false;
      if (Reverse1) {
        op1b = dup_out_pt(op1, false); op2b = dup_out_pt(op2, true);
        pp[op1] = (short)op2; pn[op2] = (short)op1; pn[op1b] = (short)op2b; pp[op2b] = (short)op1b;
      } else {
        op1b = dup_out_pt(op1, true); op2b = dup_out_pt(op2, false);
        pn[op1] = (short)op2; pp[op2] = (short)op1; pp[op1b] = (short)op2b; pn[op2b] = (short)op1b;
      }
      j1[j] = (short)op1; j2[j] = (short)op1b;
      return true;
    }
  }
  SD_HDN int point_in_polygon(int ptx_, int pty_, int op) const {             // PointInPolygon(pt, OutPt*) :484-523
    const i64 X = ptx_, Y = pty_;
    int result = 0;
    const int startOp = op;
    int guard = 0;
    for (;;) {
      const int nx = pn[op];
      if (py[nx] == Y) {
        if ((px[nx] == X) || (py[op] == Y && ((px[nx] > X) == (px[op] < X)))) return -1;
      }
      if ((py[op] < Y) != (py[nx] < Y)) {
        if (px[op] >= X) {
          if (px[nx] > X) result = 1 - result;
          else {
            const double d = (double)(px[op] - X) * (double)(py[nx] - Y) - (double)(px[nx] - X) * (double)(py[op] - Y);
            if (!d) return -1;
            if ((d > 0) == (py[nx] > py[op])) result = 1 - result;
          }
        } else {
          if (px[nx] > X) {
            const double d = (double)(px[op] - X) * (double)(py[nx] - Y) - (double)(px[nx] - X) * (double)(py[op] - Y);
            if (!d) return -1;
            if ((d > 0) == (py[nx] > py[op])) result = 1 - result;
          }
        }
      }
      op = pn[op];
      if (startOp == op) break;
      if (++guard > MAXPT) break;
    }
    return result;
  }
  SD_HD bool poly2_contains_poly1(int o1, int o2) const {                    // :526-538
    int op = o1, guard = 0;
    do {
      const int res = point_in_polygon(px[op], py[op], o2);
      if (res >= 0) return res > 0;
      op = pn[op];
      if (++guard > MAXPT) break;
    } while (op != o1);
    return true;
  }
  SD_HD void update_out_pt_idxs(int r) {                                     // :3307-3316
    int op = r_pts[r], guard = 0;
    do { pidx[op] = r_idx[r]; op = pp[op]; if (++guard > MAXPT) break; } while (op != r_pts[r]);
  }
  SD_HDN void join_common_edges() {                                           // :3679-3783
    for (int i = 0; i < n_j; ++i) {
      int outRec1 = get_outrec(pidx[j1[i]]);
      int outRec2 = get_outrec(pidx[j2[i]]);
      if (r_pts[outRec1] < 0 || r_pts[outRec2] < 0) continue;
      int holeStateRec;
      if (outRec1 == outRec2) holeStateRec = outRec1;
      else if (rec1_right_of_rec2(outRec1, outRec2)) holeStateRec = outRec2;
      else if (rec1_right_of_rec2(outRec2, outRec1)) holeStateRec = outRec1;
      else holeStateRec = get_lowermost_rec(outRec1, outRec2);
      if (!join_points(i, outRec1, outRec2)) continue;
      if (outRec1 == outRec2) {
        r_pts[outRec1] = j1[i];
        r_bot[outRec1] = -1;
        outRec2 = create_outrec();
        r_pts[outRec2] = j2[i];
        update_out_pt_idxs(outRec2);
        if (poly2_contains_poly1(r_pts[outRec2], r_pts[outRec1])) {
          r_hole[outRec2] = !r_hole[outRec1];
          r_first[outRec2] = (short)outRec1;
          if (r_hole[outRec2] == (ring_area(r_pts[outRec2]) > 0)) reverse_links(r_pts[outRec2]);
        } else if (poly2_contains_poly1(r_pts[outRec1], r_pts[outRec2])) {
          r_hole[outRec2] = r_hole[outRec1];
          r_hole[outRec1] = !r_hole[outRec2];
          r_first[outRec2] = r_first[outRec1];
          r_first[outRec1] = (short)outRec2;
          if (r_hole[outRec1] == (ring_area(r_pts[outRec1]) > 0)) reverse_links(r_pts[outRec1]);
        } else {
          r_hole[outRec2] = r_hole[outRec1];
          r_first[outRec2] = r_first[outRec1];
        }
      } else {
        r_pts[outRec2] = -1; r_bot[outRec2] = -1; r_idx[outRec2] = r_idx[outRec1];
        r_hole[outRec1] = r_hole[holeStateRec];
        if (holeStateRec == outRec2) r_first[outRec1] = r_first[outRec2];
        r_first[outRec2] = (short)outRec1;
      }
    }
  }

  // ------------------------------------------------------------------ Execute   :1560-1621 + stardist2d.cpp:161-164
  SD_HDN i64 execute() {
    if (!B::run_sweep()) return 0;
    for (int i = 0; i < n_rec; ++i) {                                        // fix orientations :1594-1600
      if (r_pts[i] < 0) continue;
      if (r_hole[i] == (ring_area(r_pts[i]) > 0)) reverse_links(r_pts[i]);
    }
    if (n_j > 0) join_common_edges();
    i64 twice = 0;
    for (int i = 0; i < n_rec; ++i) {
      const int start = r_pts[i];
      if (start < 0) continue;
      i64 s = 0;
      int op = start, guard = 0;
      do {
        const int nx = pn[op];
        const i64 c = sd_cross(px[op], py[op], px[nx], py[nx]);
        s += c; sum_abs_terms += sd_abs64(c);
        op = nx;
        if (++guard > MAXPT) { status |= ST_ITER; break; }
      } while (op != start);
      twice += sd_abs64(s);
    }
    return twice;
  }
};

}  // namespace sdclip
