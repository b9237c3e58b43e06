// oracle/clipper_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Pair-level probe of the reference's 2D overlap arithmetic: calls the vendored
// Clipper 6.4.2 (compiled from /root/reference/stardist/lib/external/clipper where it
// lies) exactly the way stardist/lib/stardist2d.cpp:152-165 does (clip=A, subject=B,
// ctIntersection, pftNonZero/pftNonZero) and returns, per pair, the raw output
// paths so the GPU restatement can be compared vertex for vertex.
#include "clipper.hpp"
#include <cstdint>
#include <cstdlib>

extern "C" {

// xa,ya / xb,yb: integer vertices. out_xy receives concatenated (x,y) of all result
// paths, out_len the per-path vertex counts. Returns number of paths (or -1 if the
// caller's buffers are too small).
int clipper_ref_intersect(const int64_t* xa, const int64_t* ya, int na,
                          const int64_t* xb, const int64_t* yb, int nb,
                          int64_t* out_xy, int max_pts, int* out_len, int max_paths)
{
  ClipperLib::Path A, B;
  for (int i = 0; i < na; i++) A << ClipperLib::IntPoint(xa[i], ya[i]);
  for (int i = 0; i < nb; i++) B << ClipperLib::IntPoint(xb[i], yb[i]);
  ClipperLib::Clipper c;
  ClipperLib::Paths res;
  c.AddPath(A, ClipperLib::ptClip, true);
  c.AddPath(B, ClipperLib::ptSubject, true);
  c.Execute(ClipperLib::ctIntersection, res, ClipperLib::pftNonZero, ClipperLib::pftNonZero);
  int np = 0, k = 0;
  for (size_t r = 0; r < res.size(); r++) {
    if (np >= max_paths) return -1;
    out_len[np++] = (int)res[r].size();
    for (size_t i = 0; i < res[r].size(); i++) {
      if (k >= max_pts) return -1;
      out_xy[2*k] = res[r][i].X; out_xy[2*k+1] = res[r][i].Y; k++;
    }
  }
  return np;
}

// area exactly as stardist2d.cpp:128-138 + :161-164 (float accumulation in path order,
// abs per path, float sum over paths)
float clipper_ref_area(const int64_t* xa, const int64_t* ya, int na,
                       const int64_t* xb, const int64_t* yb, int nb)
{
  ClipperLib::Path A, B;
  for (int i = 0; i < na; i++) A << ClipperLib::IntPoint(xa[i], ya[i]);
  for (int i = 0; i < nb; i++) B << ClipperLib::IntPoint(xb[i], yb[i]);
  ClipperLib::Clipper c;
  ClipperLib::Paths res;
  c.AddPath(A, ClipperLib::ptClip, true);
  c.AddPath(B, ClipperLib::ptSubject, true);
  c.Execute(ClipperLib::ctIntersection, res, ClipperLib::pftNonZero, ClipperLib::pftNonZero);
  float area_inter = 0;
  for (size_t r = 0; r < res.size(); r++) {
    const ClipperLib::Path& p = res[r];
    float area = 0; const int n = (int)p.size();
    for (int i = 0; i < n; i++)
      area += p[i].X * p[(i+1)%n].Y - p[i].Y * p[(i+1)%n].X;
    area = 0.5 * std::abs(area);
    area_inter += area;
  }
  return area_inter;
}
}
