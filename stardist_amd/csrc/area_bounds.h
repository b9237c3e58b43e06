// Decision shortcut of the 2D NMS: the exact area of P n Q plus / minus a BAND for what the reference's Clipper call can return instead
// ("area enclosure" in the names below; the band is validated empirically and adversarially, it is NOT a theorem about Clipper -- see the
// end of this comment and DESIGN.md 3.4; sd_set_option("nms2d_strict", 1) sends every pair to the Clipper-exact sweep instead).
//
// The reference decides  area_inter / min(area_i, area_j) > thr  (stardist2d.cpp:580-581) with area_inter from ClipperLib
// (poly_intersection_area :152-165).  Clipper's result is the intersection of the two integer polygons with every CROSSING
// POINT of their boundaries rounded to the lattice (clipper.cpp IntersectPoint: Round()), all other output vertices are input
// vertices.  The scan-beam sweep that reproduces it bit for bit (clip_beam.h) is a divergent per-lane state machine; most
// pairs of an NMS, however, are far from the threshold (candidates of one object overlap almost completely, those of
// neighbouring objects hardly at all).  For those this header computes
//   A  = the exact area of P n Q (up to float rounding), by integrating x dy - y dx along the boundary of the intersection:
//        dP inside Q plus dQ inside P.  An edge e = (a -> b) of P contributes cross(a, b) * lambda_e, lambda_e = the part of e
//        inside Q = [a inside Q] + sum over the crossings of e with dQ of  +-(1 - t)  (t: parameter of the crossing on e, + when
//        e enters Q).  All predicates (which edges cross, which vertex is inside) are evaluated EXACTLY on small integers
//        (coordinates relative to the pair, products below 2^24 held in floats), with a symbolic perturbation of Q by
//        (eps, eps^2) so that touching vertices, collinear edges and shared boundaries need no special cases: the perturbed
//        configuration is generic, and its area differs from the given one by O(eps).  n^2 edge pairs, no data-dependent control
//        flow: 32 lanes per pair, lane = edge of P, loop over the edges of Q.
//   K  = the number of boundary crossings, T = the number of edge pairs (e of P, f of Q) whose bounding boxes come within one lattice
//        step of each other, and the band
//        B = (0.5 K + max(NEAR_W T, STRIP_W S)) (lmax_P + lmax_Q) + 0.75 + (float error term),   NEAR_W = 0.15, STRIP_W = 0.45 (round 5:
//        0.125 T alone), S = the number of STRIPS = edges with at least one near partner (the larger of the two polygons' counts).
//        Three mechanisms separate Clipper's area from A.  (1) It rounds each of the K crossing points to the lattice: moving one vertex
//        of a polygon by delta changes its area by |delta x (v_next - v_prev)| / 2 <= 0.71 (|e| + |f|) / 2 -- at most 0.36 (lmax_P +
//        lmax_Q) per crossing (PROVEN for a crossing that is not clamped to its scan beam; the band carries 0.5).  (2) It orders the
//        active edges by their lattice-ROUNDED abscissae at the scan lines: two edges that run closer than one step without crossing can
//        tie, be inserted in the wrong order and later be "uncrossed", which moves the strip between them -- at most one step wide and
//        as long as the shorter edge -- to the wrong side (K = 0 pairs with a deviation of 0.5 exist: tests/test_cpu_area_enclosure.py);
//        along nearly coincident boundaries every edge is near about three edges of the other polygon, so T counts each such strip
//        about three times (measured on bench-like pairs: T / S = 2.7 .. 3.7): NEAR_W T charges a strip about 0.45 there, but charged the
//        ISOLATED near pairs of polygons with few long edges a third of that -- exactly the regime in which the round-5 adversary found its
//        smallest margins (0.53 of the band on K = 0, T = 3 pairs).  Round 6 charges every strip STRIP_W = 0.45 whatever the count splits
//        into, and raised NEAR_W from 0.125 to 0.15: both changes are monotone (the band only grows: pairs only LEAVE the shortcut for
//        the exact sweep, earlier evidence stays valid); the weights are EMPIRICAL.  (3) A polygon whose OWN vertex lies within half a step of one of its own edges
//        is re-ordered by Clipper on its own (found by the adversarial search of round 5): such polygons are not "robustly simple"
//        (k_poly_props) and are never decided here.
//        Evidence for the band: 2.0 x 10^9 GPU pairs of eleven families against the exact sweep (round 6; worst 0.25 B), 18 M CPU pairs against the
//        vendored Clipper (0.27 B), and an annealing ADVERSARY linked to the vendored Clipper (test infrastructure, DESIGN.md 3.4:
//        4.8 x 10^9 evaluations over NMS-realisable (worst 0.42 B) and free integer polygons (worst 0.53 B); profiles/r05_area_band_adversary.txt).
// A pair is decided when (A -+ B) / min(area) clears the threshold by the margins below; everything else -- and every pair with
// a polygon that is not ROBUSTLY SIMPLE (the boundary integral weights regions by winding number, Clipper's NonZero rule does not), with
// polygons of opposite orientation, too large for exact float predicates, or whose reference result could be rounded by the
// float accumulation of area_from_path (:128-138) -- goes to the exact sweep as before.  Decisions, not areas, leave this header.
#pragma once
#include <hip/hip_runtime.h>

namespace sdarea {

enum { PP_PLAIN = 1, PP_POS = 2, PP_NEG = 4 };
struct PolyProps { float lmax, perim; int flags; int xmin, xmax, ymin, ymax; int pad; };   // integer bounding box of the vertices

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float half_sum(float v) { for (int o = 16; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float half_max(float v) { for (int o = 16; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int half_sum_i(int v) { for (int o = 16; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int half_min_i(int v) { for (int o = 16; o; o >>= 1) { const int t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
__device__ __forceinline__ int half_max_i(int v) { for (int o = 16; o; o >>= 1) { const int t = __shfl_xor(v, o); v = t > v ? t : v; } return v; }

constexpr float NEAR_W = 0.15f, STRIP_W = 0.45f;   // band weights of an edge pair within one lattice step / of a strip (mechanism 2); the numpy statement under tests/ and the adversarial search tool of the test infrastructure carry the same values
constexpr int WINDOW = 2047;        // largest |relative coordinate| for which every predicate's products stay below 2^24

// Per polygon (two polygons per wave, lane & 31 = edge): longest edge, L1 perimeter, orientation, integer bounding box and whether the
// polygon is ROBUSTLY SIMPLE once zero-length edges are dropped: no two edges share a point except cyclic neighbours at their common
// vertex, no fold-back between neighbours, at least three edges, and no vertex within half a lattice step (along its scan line) of an edge
// it does not end.  vx / vy: [n][R] (R <= 32).
static __global__ void __launch_bounds__(256) k_poly_props(const int* __restrict__ vx, const int* __restrict__ vy, int n, int R, PolyProps* __restrict__ out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l = lane & 31, hb = half << 5;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int cand = 2 * wave + half;
  const bool cv = cand < n;                     // (uniform within a half)
  const bool valid = cv && l < R;
  int X = 0, Y = 0;
  if (valid) { X = vx[(size_t)cand * R + l]; Y = vy[(size_t)cand * R + l]; }
  const int x0 = __shfl(X, hb), y0 = __shfl(Y, hb);
  const int xmin = half_min_i(valid ? X : 0x7fffffff), xmax = half_max_i(valid ? X : (int)0x80000000);
  const int ymin = half_min_i(valid ? Y : 0x7fffffff), ymax = half_max_i(valid ? Y : (int)0x80000000);
  const bool small = cv && (long long)xmax - xmin <= WINDOW && (long long)ymax - ymin <= WINDOW;
  const int ln = (l + 1 >= R) ? 0 : l + 1;
  const int rx = valid && small ? X - x0 : 0, ry = valid && small ? Y - y0 : 0;           // |.| <= WINDOW
  const int rbx = __shfl(rx, hb + ln), rby = __shfl(ry, hb + ln);
  const float ax = (float)rx, ay = (float)ry, bx = (float)rbx, by = (float)rby;
  const float ex = bx - ax, ey = by - ay;
  const bool deg = !valid || (ex == 0.f && ey == 0.f);
  const unsigned long long bal = __ballot(!deg);
  const unsigned int m32 = (unsigned int)(half ? (bal >> 32) : bal);
  const int count = __popc(m32);
  int nxt = -1;                                   // the next edge of non-zero length
  if (m32) { const unsigned int above = (l >= 31) ? 0u : (m32 & ~((2u << l) - 1u)); nxt = above ? __ffs((int)above) - 1 : __ffs((int)m32) - 1; }
  const int area2 = half_sum_i(valid ? rx * rby - ry * rbx : 0);                            // exact: |terms| < 2^23, 32 of them
  bool bad = false;
  // the vertices and the successor table of the half's polygon through LDS: the loop reads them with half-uniform addresses (broadcast
  // reads) instead of five cross-lane shuffles per iteration
  __shared__ float2 sv[4][2][32];
  __shared__ int sn[4][2][32];
  const int wv = threadIdx.x >> 6;
  sv[wv][half][l] = make_float2(ax, ay); sn[wv][half][l] = nxt;
  __builtin_amdgcn_wave_barrier();                  // (a wave's LDS accesses are processed in order)
  // every UNORDERED pair of edges {l, k} once: lane l meets k = l + dd (cyclically) for dd = 1 .. R / 2 (the pairs at distance R / 2 of an even
  // R twice) and evaluates the symmetric edge-against-edge test once and the vertex-against-edge rule in both directions -- half the
  // iterations of a loop over every k.  A lane's findings are OR-ed over the polygon's lanes below.
  for (int dd = 1; dd <= (R >> 1); ++dd) {
    int k = l + dd; if (k >= R) k -= R;
    if (!valid) k = 0;
    const int kn = (k + 1 >= R) ? 0 : k + 1;
    const float2 c2 = sv[wv][half][k], d2 = sv[wv][half][kn];
    const float cx = c2.x, cy = c2.y, dx = d2.x, dy = d2.y;
    const int nxt_k = sn[wv][half][k];
    const bool degk = ((m32 >> k) & 1u) == 0u;
    const float fx = dx - cx, fy = dy - cy;
    // ROBUSTLY simple (round 5): a vertex must not lie, along its scan line, within HALF a lattice step of an edge it is not an end point
    // of -- there Clipper's rounded abscissae tie and the polygon's OWN edges can be re-ordered, which changes the area it returns by
    // more than any strip between the two polygons (found by the adversarial search of DESIGN.md 3.4: a polygon of area 80 whose
    // spike comes within half a step of a vertex is returned with 68.5).  Exact in float: relative coordinates <= WINDOW.
    //   my vertex a against edge k = (c -> d) ...
    if (valid && small && !degk && !((cx == ax && cy == ay) || (dx == ax && dy == ay)) && ay >= fminf(cy, dy) && ay <= fmaxf(cy, dy)) {
      if (fy == 0.f) { if (ax >= fminf(cx, dx) && ax <= fmaxf(cx, dx)) bad = true; }
      else if (2.f * fabsf((cx - ax) * fy + (ay - cy) * fx) <= fabsf(fy)) bad = true;            // |x_edge(ay) - ax| <= 1/2
    }
    //   ... and vertex c (the start of edge k) against my edge (a -> b)
    if (valid && small && !deg && !((ax == cx && ay == cy) || (bx == cx && by == cy)) && cy >= fminf(ay, by) && cy <= fmaxf(ay, by)) {
      if (ey == 0.f) { if (cx >= fminf(ax, bx) && cx <= fmaxf(ax, bx)) bad = true; }
      else if (2.f * fabsf((ax - cx) * ey + (cy - ay) * ex) <= fabsf(ey)) bad = true;
    }
    if (deg || degk || !valid) continue;
    if (k == nxt || nxt_k == l) {
      // cyclic neighbours: they share one vertex; anything more is a fold-back (when BOTH hold there are only two edges: count < 3)
      const float cr = ex * fy - ey * fx, dt = ex * fx + ey * fy;
      if (cr == 0.f && dt < 0.f) bad = true;
      continue;
    }
    const float o1 = ex * (cy - ay) - ey * (cx - ax), o2 = ex * (dy - ay) - ey * (dx - ax);
    const float o3 = fx * (ay - cy) - fy * (ax - cx), o4 = fx * (by - cy) - fy * (bx - cx);
    bool inter = (sgnf(o1) * sgnf(o2) <= 0.f) && (sgnf(o3) * sgnf(o4) <= 0.f);
    if (o1 == 0.f && o2 == 0.f)                     // collinear: overlap of the two intervals
      inter = fmaxf(fminf(ax, bx), fminf(cx, dx)) <= fminf(fmaxf(ax, bx), fmaxf(cx, dx)) &&
              fmaxf(fminf(ay, by), fminf(cy, dy)) <= fminf(fmaxf(ay, by), fmaxf(cy, dy));
    if (inter) bad = true;
  }
  const unsigned long long badm = __ballot(bad);
  const bool anybad = (unsigned int)(half ? (badm >> 32) : badm) != 0u;
  const float lmax = half_max(deg ? 0.f : sqrtf(ex * ex + ey * ey));
  const float perim = half_sum(deg ? 0.f : fabsf(ex) + fabsf(ey));
  if (cv && l == 0) {
    PolyProps p;
    p.lmax = lmax * (1.f + 1e-6f); p.perim = perim;
    p.flags = ((small && !anybad && count >= 3 && area2 != 0) ? PP_PLAIN : 0) | (area2 > 0 ? PP_POS : 0) | (area2 < 0 ? PP_NEG : 0);
    p.xmin = xmin; p.xmax = xmax; p.ymin = ymin; p.ymax = ymax; p.pad = 0;
    out[cand] = p;
  }
}

struct Enclosure { float area, band; int crossings, near; bool usable; };

// The 32 lanes of one half-wave evaluate one pair: P = (px, py)[R] with props pp, Q = (qx, qy)[R] with props pq (Q is the perturbed
// one).  sq: 64 floats of LDS private to this half-wave.  `active`: uniform within the half (an idle half still takes part in the
// wave-wide operations).  Every lane of the half returns the same values.
__device__ __forceinline__ Enclosure pair_enclosure(const int* __restrict__ px, const int* __restrict__ py, const int* __restrict__ qx,
                                                    const int* __restrict__ qy, int R, const PolyProps& pp, const PolyProps& pq, bool active,
                                                    float2* sq, int l, int half) {
  Enclosure E; E.area = 0.f; E.band = 0.f; E.crossings = 0; E.near = 0; E.usable = false;
  const int hb = half << 5;
  bool use = active && (pp.flags & PP_PLAIN) && (pq.flags & PP_PLAIN) && ((pp.flags & (PP_POS | PP_NEG)) == (pq.flags & (PP_POS | PP_NEG)));
  // origin: the centre of P's box; both polygons within the window
  const int ox = use ? (int)(((long long)pp.xmin + pp.xmax) >> 1) : 0, oy = use ? (int)(((long long)pp.ymin + pp.ymax) >> 1) : 0;
  long long ext = 0;
  if (use) {
    const long long e0 = (long long)pp.xmax - ox, e1 = (long long)ox - pp.xmin, e2 = (long long)pp.ymax - oy, e3 = (long long)oy - pp.ymin;
    const long long e4 = (long long)pq.xmax - ox, e5 = (long long)ox - pq.xmin, e6 = (long long)pq.ymax - oy, e7 = (long long)oy - pq.ymin;
    ext = e0; ext = e1 > ext ? e1 : ext; ext = e2 > ext ? e2 : ext; ext = e3 > ext ? e3 : ext;
    ext = e4 > ext ? e4 : ext; ext = e5 > ext ? e5 : ext; ext = e6 > ext ? e6 : ext; ext = e7 > ext ? e7 : ext;
    if (ext > WINDOW / 2) use = false;              // differences of relative coordinates <= WINDOW: products < 2^22, sums of two < 2^24
  }
  const bool lv = use && l < R;
  float ax = 0.f, ay = 0.f, cqx = 0.f, cqy = 0.f;
  if (lv) { ax = (float)(px[l] - ox); ay = (float)(py[l] - oy); cqx = (float)(qx[l] - ox); cqy = (float)(qy[l] - oy); }
  sq[l] = make_float2(cqx, cqy);
  const int ln = (l + 1 >= R) ? 0 : l + 1;
  const float bx = __shfl(ax, hb + ln), by = __shfl(ay, hb + ln);
  __builtin_amdgcn_wave_barrier();                  // (a wave's LDS accesses are processed in order)
  const float ex = bx - ax, ey = by - ay;
  const bool oke = lv && (ex != 0.f || ey != 0.f);
  // sides as booleans ("on the positive side"); a zero orientation takes the sign of the perturbation term:
  //   Q's vertex against my edge e:      cross(e, (eps, eps^2)) > 0  <=>  ey != 0 ? ey < 0 : ex > 0
  //   my vertex against Q's edge f:     -cross(f, (eps, eps^2)) > 0  <=>  fy != 0 ? fy > 0 : fx < 0
  const bool tie_e_pos = ey != 0.f ? ey < 0.f : ex > 0.f;
  const bool e_up = by > ay;
  const bool sPpos = (pp.flags & PP_POS) != 0, sQpos = (pq.flags & PP_POS) != 0;
  const float exlo = fminf(ax, bx) - 1.f, exhi = fmaxf(ax, bx) + 1.f, eylo = fminf(ay, by) - 1.f, eyhi = fmaxf(ay, by) + 1.f;
  float accP = 0.f, accQ = 0.f;
  int K = 0, T = 0, parA = 0, SQ = 0;
  bool nearP = false;
  float2 c = sq[0];
  float o_ec = ex * (c.y - ay) - ey * (c.x - ax);
  const int Rw = __any(use) ? R : 0;                // (`use` is uniform within a half; the ballot in the loop is wave-wide)
  for (int k = 0; k < Rw; ++k) {
    const int kn = (k + 1 >= R) ? 0 : k + 1;
    const float2 d = sq[kn];
    const float fx = d.x - c.x, fy = d.y - c.y;
    const bool okf = use & ((fx != 0.f) | (fy != 0.f));
    const bool tie_f_pos = fy != 0.f ? fy > 0.f : fx < 0.f;
    const float o_ed = ex * (d.y - ay) - ey * (d.x - ax);
    const float o_fa = fx * (ay - c.y) - fy * (ax - c.x), o_fb = fx * (by - c.y) - fy * (bx - c.x);
    const bool pos_c = (o_ec > 0.f) | ((o_ec == 0.f) & tie_e_pos), pos_d = (o_ed > 0.f) | ((o_ed == 0.f) & tie_e_pos);
    const bool pos_a = (o_fa > 0.f) | ((o_fa == 0.f) & tie_f_pos), pos_b = (o_fb > 0.f) | ((o_fb == 0.f) & tie_f_pos);
    const float ccd = c.x * d.y - c.y * d.x;
    const bool both = oke & okf;
    // edge pairs closer than one lattice step (bounding boxes)
    const bool nearb = both & ((c.x <= exhi) | (d.x <= exhi)) & ((c.x >= exlo) | (d.x >= exlo)) & ((c.y <= eyhi) | (d.y <= eyhi)) & ((c.y >= eylo) | (d.y >= eylo));
    T += nearb ? 1 : 0;
    nearP |= nearb;
    {                                               // Q's edge f has a near partner among P's edges (the same value in every lane of the half)
      const unsigned long long nb64 = __ballot(nearb);
      SQ += ((unsigned int)(half ? (nb64 >> 32) : nb64) != 0u) ? 1 : 0;
    }
    if (both & (pos_c != pos_d) & (pos_a != pos_b)) {                                         // e and f cross
      const float t = o_fa * __builtin_amdgcn_rcpf(o_fa - o_fb), u = o_ec * __builtin_amdgcn_rcpf(o_ec - o_ed);   // (1 ulp: far inside the band)
      const float wt = 1.f - t, wu = 1.f - u;
      accP += (pos_b == sQpos) ? wt : -wt;                                                    // e enters Q: + (1 - t)
      accQ += ccd * ((pos_d == sPpos) ? wu : -wu);
      ++K;
    }
    parA ^= (okf & ((c.y < ay) != (d.y < ay)) & (pos_a == (fy > 0.f))) ? 1 : 0;               // a inside Q: ray towards +x
    const bool hitC = oke & use & ((ay <= c.y) != (by <= c.y)) & (pos_c == e_up);              // c inside P
    const unsigned long long hb64 = __ballot(hitC);
    const unsigned int hm = (unsigned int)(half ? (hb64 >> 32) : hb64);
    if ((__popc(hm) & 1) && l == 0) accQ += ccd;
    c = d; o_ec = o_ed;
  }
  const float cab = ax * by - ay * bx;
  const float contrib = lv ? cab * ((float)parA + accP) + accQ : 0.f;
  const float tot = half_sum(contrib);
  const int Kt = half_sum_i(lv ? K : 0), Tt = half_sum_i(lv ? T : 0), SPt = half_sum_i((lv && nearP) ? 1 : 0);
  const int St = SPt > SQ ? SPt : SQ;              // strips: edges with at least one near partner, the larger of the two polygons' counts
  E.area = 0.5f * fabsf(tot);
  E.crossings = Kt; E.near = Tt;
  // float error of the sum: <= 64 terms of magnitude <= ext * edge length, each with a few ulps
  E.band = (0.5f * (float)Kt + fmaxf(NEAR_W * (float)Tt, STRIP_W * (float)St)) * (pp.lmax + pq.lmax) + 0.75f + 2e-6f * (float)ext * (pp.perim + pq.perim);
  // area_from_path adds integer cross products in float: exact while the sum of their magnitudes stays below 2^24
  // (|p_i x p_{i+1}| <= |p_i| |p_{i+1} - p_i|; the output's edges are parts of the inputs' edges, crossing points moved by < 1.5)
  if (use) {
    long long M = 0;
    const long long a0 = pp.xmin < 0 ? -(long long)pp.xmin : pp.xmin, a1 = pp.xmax < 0 ? -(long long)pp.xmax : pp.xmax;
    const long long a2 = pp.ymin < 0 ? -(long long)pp.ymin : pp.ymin, a3 = pp.ymax < 0 ? -(long long)pp.ymax : pp.ymax;
    const long long b0 = pq.xmin < 0 ? -(long long)pq.xmin : pq.xmin, b1 = pq.xmax < 0 ? -(long long)pq.xmax : pq.xmax;
    const long long b2 = pq.ymin < 0 ? -(long long)pq.ymin : pq.ymin, b3 = pq.ymax < 0 ? -(long long)pq.ymax : pq.ymax;
    M = (a0 > a1 ? a0 : a1); M = b0 > M ? b0 : M; M = b1 > M ? b1 : M;
    long long My = (a2 > a3 ? a2 : a3); My = b2 > My ? b2 : My; My = b3 > My ? b3 : My;
    const double bound = (double)(M + My + 2) * ((double)pp.perim + (double)pq.perim + 3.0 * Kt + 4.0);
    if (!(bound < 16777216.0)) use = false;
  }
  E.usable = use;
  return E;
}

// thresholds of a decision: 1 = certainly not above thr (pair kept), 2 = certainly above (j suppressed), 0 = undecided
__device__ __forceinline__ int decide(const Enclosure& E, float area_i, float area_j, float thr) {
  if (!E.usable) return 0;
  const double amin = fmin((double)area_i + 1.e-10, (double)area_j + 1.e-10);                // :580
  if (!(amin > 0.25)) return 0;
  const double lo = ((double)E.area - (double)E.band) / amin, hi = ((double)E.area + (double)E.band) / amin;
  const double m = 4e-6 * fabs((double)thr) + 1e-6;                                            // float rounding of the quotient and of the comparison
  if (lo > (double)thr + m) return 2;
  if (hi < (double)thr - m) return 1;
  return 0;
}

}  // namespace sdarea
