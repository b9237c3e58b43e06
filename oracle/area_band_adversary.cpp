// oracle/area_band_adversary.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Adversarial search against the decision band of the 2D NMS (stardist_amd/csrc/area_bounds.h): the NMS decides a pair without the
// Clipper-exact sweep when  (A -+ B) / min(area)  clears the threshold, where A is the exact intersection area and B the band that has
// to contain what the reference's Clipper call (stardist/lib/stardist2d.cpp:152-165, vendored Clipper 6.4.2: IntersectPoint + Round,
// clipper.cpp:622-688; edge order by rounded abscissae, InsertEdgeIntoAEL / FixupIntersectionOrder) returns.  This program MAXIMISES
//        ratio = |A_clipper - A| / B
// over integer polygon pairs by simulated annealing from structured starts (random star polygons, nearly coincident boundaries, lattice
// half-steps, thin spikes, nested K = 0 pairs, long nearly parallel edges), linked against the vendored Clipper where it lies
// (oracle/Makefile: _ref/area_band_adversary).  The enclosure below is a host restatement of area_bounds.h (same predicates, same K, T,
// band; predicates in exact integer arithmetic, area in double -- the device's float rounding is a term of B).
//   usage: area_band_adversary <seed> <restarts> <iterations per restart> [mode: 0 star (NMS-realisable polygons), 1 free integer polygons, 2 both] [self rule 1|0] [near-pair weight, default 0.15] [strip weight, default 0.45]
// Prints one line per new overall worst (with the vertices, so a counter-example can be replayed) and a summary per start family.
#include "clipper.hpp"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

typedef long long i64;
static const int MAXR = 32;
static const int WINDOW = 2047;

struct Poly { int n; i64 x[MAXR], y[MAXR]; };
struct Props { double lmax, perim; bool plain; int orient; i64 xmin, xmax, ymin, ymax; };

static int g_self_rule = 1;
static double g_near_w = 0.15, g_strip_w = 0.45;      // band weights of a near edge pair / of a strip (area_bounds.h NEAR_W, STRIP_W; round 5: 0.125 and no strip term)
static int sgn(i64 v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// area_bounds.h k_poly_props
static Props props(const Poly& p) {
  Props r; r.xmin = r.ymin = INT64_MAX; r.xmax = r.ymax = INT64_MIN;
  const int n = p.n;
  for (int k = 0; k < n; ++k) { r.xmin = std::min(r.xmin, p.x[k]); r.xmax = std::max(r.xmax, p.x[k]); r.ymin = std::min(r.ymin, p.y[k]); r.ymax = std::max(r.ymax, p.y[k]); }
  const bool small = r.xmax - r.xmin <= WINDOW && r.ymax - r.ymin <= WINDOW;
  i64 area2 = 0; r.lmax = 0; r.perim = 0;
  bool deg[MAXR]; int nxt[MAXR]; int count = 0;
  for (int k = 0; k < n; ++k) {
    const int kn = (k + 1) % n;
    const i64 ex = p.x[kn] - p.x[k], ey = p.y[kn] - p.y[k];
    deg[k] = (ex == 0 && ey == 0);
    if (!deg[k]) { ++count; r.lmax = std::max(r.lmax, std::sqrt((double)(ex * ex + ey * ey))); r.perim += (double)(std::llabs(ex) + std::llabs(ey)); }
    area2 += (p.x[k] - p.x[0]) * (p.y[kn] - p.y[0]) - (p.y[k] - p.y[0]) * (p.x[kn] - p.x[0]);
  }
  for (int k = 0; k < n; ++k) { nxt[k] = -1; for (int s = 1; s <= n; ++s) { const int q = (k + s) % n; if (!deg[q]) { nxt[k] = q; break; } } }
  bool bad = false;
  for (int l = 0; l < n && !bad; ++l) {
    if (deg[l]) continue;
    const int ln = (l + 1) % n;
    const i64 ax = p.x[l], ay = p.y[l], bx = p.x[ln], by = p.y[ln], ex = bx - ax, ey = by - ay;
    for (int k = 0; k < n; ++k) {
      if (deg[k] || k == l) continue;
      const int kn = (k + 1) % n;
      const i64 cx = p.x[k], cy = p.y[k], dx = p.x[kn], dy = p.y[kn], fx = dx - cx, fy = dy - cy;
      if (k == nxt[l] || nxt[k] == l) { if (ex * fy - ey * fx == 0 && ex * fx + ey * fy < 0) bad = true; continue; }
      const i64 o1 = ex * (cy - ay) - ey * (cx - ax), o2 = ex * (dy - ay) - ey * (dx - ax);
      const i64 o3 = fx * (ay - cy) - fy * (ax - cx), o4 = fx * (by - cy) - fy * (bx - cx);
      bool inter = (sgn(o1) * sgn(o2) <= 0) && (sgn(o3) * sgn(o4) <= 0);
      if (o1 == 0 && o2 == 0)
        inter = std::max(std::min(ax, bx), std::min(cx, dx)) <= std::min(std::max(ax, bx), std::max(cx, dx)) &&
                std::max(std::min(ay, by), std::min(cy, dy)) <= std::min(std::max(ay, by), std::max(cy, dy));
      if (inter) { bad = true; break; }
    }
  }
  // ROBUSTLY simple (rule added in round 5, see area_bounds.h): no vertex within half a lattice step, along its scan line, of an edge of the
  // same polygon it is not an end point of -- where Clipper's rounded abscissae can tie and re-order the polygon's own edges
  if (g_self_rule && !bad) {
    for (int v = 0; v < n && !bad; ++v) {
      const i64 vx = p.x[v], vy = p.y[v];
      for (int k = 0; k < n; ++k) {
        if (deg[k]) continue;
        const int kn = (k + 1) % n;
        const i64 ax = p.x[k], ay = p.y[k], bx = p.x[kn], by = p.y[kn];
        if ((ax == vx && ay == vy) || (bx == vx && by == vy)) continue;          // an end point (or a coincident vertex: caught by `inter` above unless incident)
        if (vy < std::min(ay, by) || vy > std::max(ay, by)) continue;
        if (ay == by) { if (vx >= std::min(ax, bx) && vx <= std::max(ax, bx)) { bad = true; break; } continue; }
        const i64 num = (ax - vx) * (by - ay) + (vy - ay) * (bx - ax);
        if (2 * std::llabs(num) <= g_self_rule * std::llabs(by - ay)) { bad = true; break; }       // |x_edge(vy) - vx| <= g_self_rule / 2
      }
    }
  }
  r.lmax *= (1.0 + 1e-6);
  r.plain = small && !bad && count >= 3 && area2 != 0;
  r.orient = sgn(area2);
  return r;
}

struct Encl { double area, band; int K, T; bool usable; };

// area_bounds.h pair_enclosure (P = clip, Q = subject, Q perturbed by (eps, eps^2))
static Encl enclosure(const Poly& P, const Poly& Q, const Props& pp, const Props& pq) {
  Encl E; E.area = 0; E.band = 0; E.K = 0; E.T = 0; E.usable = false;
  bool use = pp.plain && pq.plain && pp.orient == pq.orient;
  if (!use) return E;
  const i64 ox = (pp.xmin + pp.xmax) >> 1, oy = (pp.ymin + pp.ymax) >> 1;
  i64 ext = 0;
  const i64 es[8] = {pp.xmax - ox, ox - pp.xmin, pp.ymax - oy, oy - pp.ymin, pq.xmax - ox, ox - pq.xmin, pq.ymax - oy, oy - pq.ymin};
  for (int k = 0; k < 8; ++k) ext = std::max(ext, es[k]);
  if (ext > WINDOW / 2) return E;
  const bool sPpos = pp.orient > 0, sQpos = pq.orient > 0;
  double tot = 0;
  int K = 0, T = 0;
  bool nearP[MAXR] = {}, nearQ[MAXR] = {};      // edges with at least one near partner: the strips
  const int n = P.n, m = Q.n;
  // c inside P, per vertex of Q: parity over the edges of P
  for (int l = 0; l < n; ++l) {
    const int ln = (l + 1) % n;
    const i64 ax = P.x[l] - ox, ay = P.y[l] - oy, bx = P.x[ln] - ox, by = P.y[ln] - oy, ex = bx - ax, ey = by - ay;
    const bool oke = ex != 0 || ey != 0;
    const bool tie_e_pos = ey != 0 ? ey < 0 : ex > 0;
    const i64 exlo = std::min(ax, bx) - 1, exhi = std::max(ax, bx) + 1, eylo = std::min(ay, by) - 1, eyhi = std::max(ay, by) + 1;
    double accP = 0; int parA = 0;
    for (int k = 0; k < m; ++k) {
      const int kn = (k + 1) % m;
      const i64 cx = Q.x[k] - ox, cy = Q.y[k] - oy, dx = Q.x[kn] - ox, dy = Q.y[kn] - oy, fx = dx - cx, fy = dy - cy;
      const bool okf = fx != 0 || fy != 0;
      const bool tie_f_pos = fy != 0 ? fy > 0 : fx < 0;
      const i64 o_ec = ex * (cy - ay) - ey * (cx - ax), o_ed = ex * (dy - ay) - ey * (dx - ax);
      const i64 o_fa = fx * (ay - cy) - fy * (ax - cx), o_fb = fx * (by - cy) - fy * (bx - cx);
      const bool pos_c = o_ec > 0 || (o_ec == 0 && tie_e_pos), pos_d = o_ed > 0 || (o_ed == 0 && tie_e_pos);
      const bool pos_a = o_fa > 0 || (o_fa == 0 && tie_f_pos), pos_b = o_fb > 0 || (o_fb == 0 && tie_f_pos);
      const bool both = oke && okf;
      if (both && (cx <= exhi || dx <= exhi) && (cx >= exlo || dx >= exlo) && (cy <= eyhi || dy <= eyhi) && (cy >= eylo || dy >= eylo)) { ++T; nearP[l] = nearQ[k] = true; }
      if (both && pos_c != pos_d && pos_a != pos_b) {
        const double t = (double)o_fa / (double)(o_fa - o_fb), u = (double)o_ec / (double)(o_ec - o_ed);
        accP += (pos_b == sQpos) ? (1.0 - t) : -(1.0 - t);
        tot += (double)(cx * dy - cy * dx) * ((pos_d == sPpos) ? (1.0 - u) : -(1.0 - u));
        ++K;
      }
      if (okf && ((cy < ay) != (dy < ay)) && (pos_a == (fy > 0))) parA ^= 1;
    }
    if (oke || true) tot += (double)(ax * by - ay * bx) * ((double)parA + accP);
  }
  for (int k = 0; k < m; ++k) {
    const int kn = (k + 1) % m;
    const i64 cx = Q.x[k] - ox, cy = Q.y[k] - oy, dx = Q.x[kn] - ox, dy = Q.y[kn] - oy;
    int par = 0;
    for (int l = 0; l < n; ++l) {
      const int ln = (l + 1) % n;
      const i64 ax = P.x[l] - ox, ay = P.y[l] - oy, bx = P.x[ln] - ox, by = P.y[ln] - oy, ex = bx - ax, ey = by - ay;
      const bool oke = ex != 0 || ey != 0;
      const bool tie_e_pos = ey != 0 ? ey < 0 : ex > 0;
      const i64 o_ec = ex * (cy - ay) - ey * (cx - ax);
      const bool pos_c = o_ec > 0 || (o_ec == 0 && tie_e_pos);
      if (oke && ((ay <= cy) != (by <= cy)) && (pos_c == (by > ay))) par ^= 1;
    }
    if (par) tot += (double)(cx * dy - cy * dx);
  }
  E.area = 0.5 * std::fabs(tot); E.K = K; E.T = T;
  int SP = 0, SQ = 0;
  for (int l = 0; l < n; ++l) SP += nearP[l];
  for (int k = 0; k < m; ++k) SQ += nearQ[k];
  const int S = std::max(SP, SQ);
  E.band = (0.5 * K + std::max(g_near_w * T, g_strip_w * S)) * (pp.lmax + pq.lmax) + 0.75 + 2e-6 * (double)ext * (pp.perim + pq.perim);
  i64 M = 0, My = 0;
  const i64 xs[4] = {pp.xmin, pp.xmax, pq.xmin, pq.xmax}, ys[4] = {pp.ymin, pp.ymax, pq.ymin, pq.ymax};
  for (int k = 0; k < 4; ++k) { M = std::max(M, std::llabs(xs[k])); My = std::max(My, std::llabs(ys[k])); }
  const double bound = (double)(M + My + 2) * (pp.perim + pq.perim + 3.0 * K + 4.0);
  E.usable = bound < 16777216.0;
  return E;
}

// the reference's call: stardist2d.cpp:128-138 + :152-165
static float clipper_area(const Poly& A, const Poly& B) {
  ClipperLib::Path a, b;
  for (int i = 0; i < A.n; i++) a << ClipperLib::IntPoint(A.x[i], A.y[i]);
  for (int i = 0; i < B.n; i++) b << ClipperLib::IntPoint(B.x[i], B.y[i]);
  ClipperLib::Clipper c;
  ClipperLib::Paths res;
  c.AddPath(a, ClipperLib::ptClip, true);
  c.AddPath(b, ClipperLib::ptSubject, true);
  c.Execute(ClipperLib::ctIntersection, res, ClipperLib::pftNonZero, ClipperLib::pftNonZero);
  float area_inter = 0;
  for (size_t r = 0; r < res.size(); r++) {
    const ClipperLib::Path& p = res[r];
    float area = 0; const int n = (int)p.size();
    for (int i = 0; i < n; i++) area += p[i].X * p[(i + 1) % n].Y - p[i].Y * p[(i + 1) % n].X;
    area = 0.5 * std::abs(area);
    area_inter += area;
  }
  return area_inter;
}

static unsigned long long g_evals = 0, g_usable = 0;
struct Score { double ratio, dev, band; int K, T; bool usable; };
static Score evaluate(const Poly& P, const Poly& Q) {
  ++g_evals;
  Score s; s.ratio = -1; s.dev = 0; s.band = 0; s.K = s.T = 0; s.usable = false;
  const Props pp = props(P), pq = props(Q);
  const Encl E = enclosure(P, Q, pp, pq);
  if (!E.usable) return s;
  ++g_usable;
  const double C = (double)clipper_area(P, Q);
  s.usable = true; s.dev = std::fabs(C - E.area); s.band = E.band; s.K = E.K; s.T = E.T; s.ratio = s.dev / E.band;
  return s;
}

// ---- polygon models
struct Star { int n; float py, px; float d[MAXR]; };       // what the NMS can produce: p + d (sin, cos), truncated (stardist2d.cpp:447-471)
static float g_sin[MAXR + 1][MAXR], g_cos[MAXR + 1][MAXR];
static void init_tables() {
  for (int n = 3; n <= MAXR; ++n) { const float ang = (float)(2 * M_PI / n); for (int k = 0; k < n; ++k) { g_sin[n][k] = sinf(ang * k); g_cos[n][k] = cosf(ang * k); } }
}
static void star_to_poly(const Star& s, Poly& p) {
  p.n = s.n;
  for (int k = 0; k < s.n; ++k) {
    volatile float ty = s.d[k] * g_sin[s.n][k]; volatile float tx = s.d[k] * g_cos[s.n][k];   // no fused multiply-add (reference: baseline x86-64)
    const float y = s.py + ty, x = s.px + tx;
    p.x[k] = (i64)x; p.y[k] = (i64)y;
  }
}

typedef std::mt19937_64 Rng;
static double U(Rng& r, double a, double b) { return a + (b - a) * (double)(r() >> 11) * (1.0 / 9007199254740992.0); }
static int UI(Rng& r, int a, int b) { return a + (int)(r() % (unsigned long long)(b - a + 1)); }
static double Nrm(Rng& r) { const double u1 = U(r, 1e-12, 1), u2 = U(r, 0, 1); return std::sqrt(-2 * std::log(u1)) * std::cos(2 * M_PI * u2); }

static const char* FAM[] = {"random stars", "nearly coincident", "half-step shift", "spikes", "nested K=0", "long parallel edges", "few rays large", "tiny"};
static const int NFAM = 8;

static void start_pair(Rng& r, int fam, Star& a, Star& b) {
  const int nchoices[6] = {32, 32, 32, 16, 8, 24};
  int n = nchoices[UI(r, 0, 5)];
  double radius = std::exp(U(r, std::log(4.0), std::log(120.0)));
  double noise = U(r, 0.01, 0.4);
  const float off = (float)UI(r, 20, 3000);
  if (fam == 6) { n = UI(r, 3, 8); radius = U(r, 20, 300); }
  if (fam == 7) { radius = U(r, 2.5, 7); }
  a.n = b.n = n;
  a.py = off + (float)UI(r, 0, 40); a.px = off + (float)UI(r, 0, 40);
  for (int k = 0; k < n; ++k) a.d[k] = (float)std::max(1e-3, radius * (1 + noise * U(r, -1, 1)));
  b = a;
  switch (fam) {
    case 0: case 6: case 7:
      b.py = a.py + (float)UI(r, -(int)radius, (int)radius); b.px = a.px + (float)UI(r, -(int)radius, (int)radius);
      for (int k = 0; k < n; ++k) b.d[k] = (float)std::max(1e-3, radius * U(r, 0.7, 1.1) * (1 + noise * U(r, -1, 1)));
      break;
    case 1:   // the same object seen from a neighbouring pixel: the same boundary up to a fraction of a step
      b.py = a.py + (float)UI(r, -2, 2); b.px = a.px + (float)UI(r, -2, 2);
      for (int k = 0; k < n; ++k) {
        // distance from the shifted centre to (roughly) the same boundary point
        const double vy = a.d[k] * g_sin[n][k] - (b.py - a.py), vx = a.d[k] * g_cos[n][k] - (b.px - a.px);
        b.d[k] = (float)std::max(1e-3, std::sqrt(vy * vy + vx * vx) + U(r, -0.6, 0.6));
      }
      break;
    case 2:   // identical shapes, centres one step apart, radii offset by half a step
      b.py = a.py + (float)UI(r, -1, 1); b.px = a.px + (float)UI(r, -1, 1);
      for (int k = 0; k < n; ++k) b.d[k] = a.d[k] + 0.5f + (float)U(r, -0.05, 0.05);
      break;
    case 3:   // thin spikes: lmax much larger than the median edge
      for (int q = 0; q < UI(r, 1, 4); ++q) { a.d[UI(r, 0, n - 1)] *= (float)U(r, 2, 6); }
      b.py = a.py + (float)UI(r, -3, 3); b.px = a.px + (float)UI(r, -3, 3);
      for (int k = 0; k < n; ++k) b.d[k] = a.d[k] * (float)U(r, 0.9, 1.1);
      break;
    case 4:   // nested, boundaries about one step apart
      b.py = a.py + (float)UI(r, -1, 1); b.px = a.px + (float)UI(r, -1, 1);
      for (int k = 0; k < n; ++k) b.d[k] = (float)std::max(1e-3, a.d[k] - U(r, 0.8, 2.2));
      break;
    case 5: { // smooth large shapes: long edges of nearly the same slope
      const double rr = U(r, 60, 300); const double ecc = U(r, 0.6, 1.0); const double ph = U(r, 0, M_PI);
      for (int k = 0; k < n; ++k) { const double th = 2 * M_PI * k / n - ph; a.d[k] = (float)(rr * ecc / std::sqrt((ecc * std::cos(th)) * (ecc * std::cos(th)) + std::sin(th) * std::sin(th))); }
      b = a; b.py = a.py + (float)UI(r, -2, 2); b.px = a.px + (float)UI(r, -2, 2);
      for (int k = 0; k < n; ++k) b.d[k] = a.d[k] + (float)U(r, -1.2, 1.2);
      break; }
  }
}

static void mutate_star(Rng& r, Star& a, Star& b) {
  Star& s = (r() & 1) ? a : b;
  const int n = s.n;
  switch (UI(r, 0, 6)) {
    case 0: s.d[UI(r, 0, n - 1)] += (float)(Nrm(r) * 0.3); break;
    case 1: s.d[UI(r, 0, n - 1)] += (float)(Nrm(r) * 1.5); break;
    case 2: { const int k0 = UI(r, 0, n - 1), len = UI(r, 2, std::max(2, n / 3)); const float dl = (float)(Nrm(r) * 0.7); for (int q = 0; q < len; ++q) s.d[(k0 + q) % n] += dl; break; }
    case 3: if (r() & 1) s.py += (float)UI(r, -1, 1); else s.px += (float)UI(r, -1, 1); break;
    case 4: { const float f = (float)(1 + Nrm(r) * 0.01); for (int k = 0; k < n; ++k) s.d[k] *= f; break; }
    case 5: { const int k = UI(r, 0, n - 1); Star& o = (&s == &a) ? b : a; s.d[k] = o.d[k] + (float)U(r, -0.7, 0.7); break; }     // pull towards the other boundary
    case 6: { const int k = UI(r, 0, n - 1); s.d[k] = std::floor(s.d[k]) + (float)(0.5 + U(r, -0.02, 0.02)); break; }            // half-step radii
  }
  for (int k = 0; k < n; ++k) if (!(s.d[k] > 1e-3f)) s.d[k] = 1e-3f;
}
static void mutate_free(Rng& r, Poly& a, Poly& b) {
  Poly& p = (r() & 1) ? a : b;
  const int k = UI(r, 0, p.n - 1);
  switch (UI(r, 0, 3)) {
    case 0: p.x[k] += UI(r, -1, 1); break;
    case 1: p.y[k] += UI(r, -1, 1); break;
    case 2: p.x[k] += UI(r, -2, 2); p.y[k] += UI(r, -2, 2); break;
    case 3: { const int dx = UI(r, -1, 1), dy = UI(r, -1, 1); for (int q = 0; q < p.n; ++q) { p.x[q] += dx; p.y[q] += dy; } break; }
  }
}

static void print_pair(const Poly& P, const Poly& Q) {
  printf("    P:"); for (int k = 0; k < P.n; ++k) printf(" %lld,%lld", P.x[k], P.y[k]);
  printf("\n    Q:"); for (int k = 0; k < Q.n; ++k) printf(" %lld,%lld", Q.x[k], Q.y[k]);
  printf("\n");
}

// --eval: pairs from stdin ("n  x y x y ... (P)  x y ... (Q)" per line) -> "area band K T usable clipper_area" per line: pins this
// restatement to the numpy statement tests/_area_exact.py (tests/test_cpu_area_enclosure.py)
static int eval_stdin() {
  int n;
  while (scanf("%d", &n) == 1) {
    if (n < 3 || n > MAXR) return 1;
    Poly P, Q; P.n = Q.n = n;
    for (int k = 0; k < n; ++k) if (scanf("%lld %lld", &P.x[k], &P.y[k]) != 2) return 1;
    for (int k = 0; k < n; ++k) if (scanf("%lld %lld", &Q.x[k], &Q.y[k]) != 2) return 1;
    const Props pp = props(P), pq = props(Q);
    const Encl E = enclosure(P, Q, pp, pq);
    printf("%.9g %.9g %d %d %d %.9g\n", E.area, E.band, E.K, E.T, E.usable ? 1 : 0, (double)clipper_area(P, Q));
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--eval")) return eval_stdin();
  const unsigned long long seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  const long restarts = argc > 2 ? atol(argv[2]) : 100;
  const int iters = argc > 3 ? atoi(argv[3]) : 2000;
  const int mode = argc > 4 ? atoi(argv[4]) : 2;
  if (argc > 5) g_self_rule = atoi(argv[5]);
  if (argc > 6) g_near_w = atof(argv[6]);
  if (argc > 7) g_strip_w = atof(argv[7]);
  init_tables();
  Rng r(seed * 0x9E3779B97F4A7C15ull + 12345);
  double worst = 0, famWorst[NFAM][2]; memset(famWorst, 0, sizeof(famWorst));
  unsigned long long famEvals[NFAM][2]; memset(famEvals, 0, sizeof(famEvals));
  for (long rs = 0; rs < restarts; ++rs) {
    const int fam = (int)(rs % NFAM);
    const int free_mode = mode == 2 ? (int)((rs / NFAM) & 1) : mode;
    Star sa, sb; start_pair(r, fam, sa, sb);
    Poly P, Q; star_to_poly(sa, P); star_to_poly(sb, Q);
    Score cur = evaluate(P, Q);
    const unsigned long long e0 = g_evals;
    double temp = 0.02;
    for (int it = 0; it < iters; ++it) {
      Star ta = sa, tb = sb; Poly tP = P, tQ = Q;
      if (free_mode) mutate_free(r, tP, tQ); else { mutate_star(r, ta, tb); star_to_poly(ta, tP); star_to_poly(tb, tQ); }
      const Score s = evaluate(tP, tQ);
      if (!s.usable) continue;
      const bool accept = !cur.usable || s.ratio >= cur.ratio || U(r, 0, 1) < std::exp((s.ratio - cur.ratio) / temp);
      if (accept) { sa = ta; sb = tb; P = tP; Q = tQ; cur = s; }
      if (s.ratio > famWorst[fam][free_mode]) famWorst[fam][free_mode] = s.ratio;
      if (s.ratio > worst) {
        worst = s.ratio;
        printf("new worst %.4f  (|A_clipper - A| = %.3f, band %.3f, K = %d, T = %d; family '%s', %s, restart %ld, iteration %d, %llu evaluations)\n",
               s.ratio, s.dev, s.band, s.K, s.T, FAM[fam], free_mode ? "free" : "star", rs, it, g_evals);
        print_pair(tP, tQ); fflush(stdout);
      }
      temp = 0.02 * (1.0 - (double)it / iters) + 0.002;
    }
    famEvals[fam][free_mode] += g_evals - e0;
    if ((rs + 1) % 2000 == 0) {          // progress (a run that is cut short still states how far it came)
      double ws = 0, wf = 0;
      for (int f = 0; f < NFAM; ++f) { ws = famWorst[f][0] > ws ? famWorst[f][0] : ws; wf = famWorst[f][1] > wf ? famWorst[f][1] : wf; }
      printf("progress: seed %llu, %ld restarts, %llu evaluations (%llu usable), worst %.4f (star %.4f, free %.4f)\n", seed, rs + 1, g_evals, g_usable, worst, ws, wf);
      fflush(stdout);
    }
  }
  printf("seed %llu: %llu evaluations (%llu usable), worst |A_clipper - A| / band = %.4f\n", seed, g_evals, g_usable, worst);
  for (int f = 0; f < NFAM; ++f)
    printf("  %-22s star %.4f (%llu)   free %.4f (%llu)\n", FAM[f], famWorst[f][0], famEvals[f][0], famWorst[f][1], famEvals[f][1]);
  return 0;
}
