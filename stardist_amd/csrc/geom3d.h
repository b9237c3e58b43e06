// geom3d.h -- device-side restatement of the reference's 3D geometry predicates
// (stardist/lib/stardist3d_impl.cpp:76-636).  fp32 throughout, operation order as in the
// reference; the library is compiled with -ffp-contract=off because the reference is built
// for baseline x86-64 (no FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace sd3 {

__device__ __forceinline__ int round_to_int(float r) { return __float2int_rn(r); }   // lrint :76-78

// inside_halfspace :89-106  (det >= 0)
__device__ __forceinline__ bool inside_halfspace(float z, float y, float x, float Az, float Ay, float Ax, float Bz, float By,
                                                 float Bx, float Cz, float Cy, float Cx) {
  const float M00 = Bz - Az, M01 = By - Ay, M02 = Bx - Ax;
  const float M10 = Cz - Az, M11 = Cy - Ay, M12 = Cx - Ax;
  const float M20 = z - Az, M21 = y - Ay, M22 = x - Ax;
  const float det = M00 * (M11 * M22 - M21 * M12) - M01 * (M10 * M22 - M12 * M20) + M02 * (M10 * M21 - M11 * M20);
  return det >= 0;
}

// inside_tetrahedron :109-148
__device__ __forceinline__ bool inside_tetrahedron(float z, float y, float x, float Rz, float Ry, float Rx, float Az, float Ay,
                                                   float Ax, float Bz, float By, float Bx, float Cz, float Cy, float Cx) {
  return inside_halfspace(z, y, x, Az, Ay, Ax, Bz, By, Bx, Cz, Cy, Cx) &&
         inside_halfspace(z, y, x, Rz, Ry, Rx, Bz, By, Bx, Az, Ay, Ax) &&
         inside_halfspace(z, y, x, Rz, Ry, Rx, Cz, Cy, Cx, Bz, By, Bx) &&
         inside_halfspace(z, y, x, Rz, Ry, Rx, Az, Ay, Ax, Cz, Cy, Cx);
}

// inside_polyhedron :153-191 : union of the tetrahedra (centre, face)
__device__ __forceinline__ bool inside_polyhedron(float z, float y, float x, float Rz, float Ry, float Rx,
                                                  const float* __restrict__ pv, const int* __restrict__ faces, int n_faces) {
  for (int i = 0; i < n_faces; ++i) {
    const int iA = faces[3 * i], iB = faces[3 * i + 1], iC = faces[3 * i + 2];
    if (inside_tetrahedron(z, y, x, Rz, Ry, Rx, pv[3 * iA], pv[3 * iA + 1], pv[3 * iA + 2], pv[3 * iB], pv[3 * iB + 1],
                           pv[3 * iB + 2], pv[3 * iC], pv[3 * iC + 1], pv[3 * iC + 2]))
      return true;
  }
  return false;
}

// ---- cone map: which faces can contain a given direction ---------------------------------------------------------------------
// inside_polyhedron tests the point against the tetrahedron (centre, face) of EVERY face until one contains it.  The three side
// planes of that tetrahedron pass through the centre and two ray directions: the set of directions they admit (the face's cone)
// is the same for every polyhedron of a model, because vertex k lies at centre + dist_k * ray_k.  The cone map stores, for each
// cell of a cube map over directions, the faces whose cone meets the cell widened by a safety margin; a point is then tested --
// with the reference's own fp32 predicate -- against those faces only.  Faces not listed fail one side-plane test by at least the
// margin (angle >= ~0.0115 rad), far beyond what fp32 rounding of the vertices and determinants can turn around while
// dist_k >= 1 and coordinates stay below 8192 (the callers fall back to the full loop otherwise, and next to the centre).
#define SD_CM_G 16          // cells per cube-face axis
#define SD_CM_CAP 16        // faces listed per cell; a fuller cell is marked -1 (full loop)
#define SD_CM_CELLS (6 * SD_CM_G * SD_CM_G)
__device__ __forceinline__ int cone_map_cell(float dz, float dy, float dx) {
  const float az = fabsf(dz), ay = fabsf(dy), ax = fabsf(dx);
  int m; float maj, a, b;
  if (az >= ay && az >= ax) { m = 0; maj = dz; a = dy; b = dx; }
  else if (ay >= ax) { m = 1; maj = dy; a = dz; b = dx; }
  else { m = 2; maj = dx; a = dz; b = dy; }
  const float inv = 1.f / fabsf(maj);
  int ia = (int)((a * inv + 1.f) * (0.5f * SD_CM_G)), ib = (int)((b * inv + 1.f) * (0.5f * SD_CM_G));
  ia = ia < 0 ? 0 : (ia > SD_CM_G - 1 ? SD_CM_G - 1 : ia);
  ib = ib < 0 ? 0 : (ib > SD_CM_G - 1 ? SD_CM_G - 1 : ib);
  return ((2 * m + (maj < 0.f ? 1 : 0)) * SD_CM_G + ia) * SD_CM_G + ib;
}
struct ConeMap { const unsigned short* list; const signed char* count; };
// inside_polyhedron restricted to the faces the cone map lists for the point's direction; identical result (see above)
__device__ __forceinline__ bool inside_polyhedron_mapped(float z, float y, float x, float Rz, float Ry, float Rx, const float* __restrict__ pv,
                                                         const int* __restrict__ faces, int n_faces, const ConeMap& cm, bool safe) {
  const float dz = z - Rz, dy = y - Ry, dx = x - Rx;
  if (!safe || !(dz * dz + dy * dy + dx * dx >= 0.25f)) return inside_polyhedron(z, y, x, Rz, Ry, Rx, pv, faces, n_faces);
  const int cell = cone_map_cell(dz, dy, dx);
  const int n = cm.count[cell];
  if (n < 0) return inside_polyhedron(z, y, x, Rz, Ry, Rx, pv, faces, n_faces);
  const unsigned short* l = cm.list + (size_t)cell * SD_CM_CAP;
  for (int k = 0; k < n; ++k) {
    const int i = l[k];
    const int iA = faces[3 * i], iB = faces[3 * i + 1], iC = faces[3 * i + 2];
    if (inside_tetrahedron(z, y, x, Rz, Ry, Rx, pv[3 * iA], pv[3 * iA + 1], pv[3 * iA + 2], pv[3 * iB], pv[3 * iB + 1],
                           pv[3 * iB + 2], pv[3 * iC], pv[3 * iC + 1], pv[3 * iC + 2]))
      return true;
  }
  return false;
}
// one thread per cell.  A face is left out only if one of its side determinants (the reference's own orientation:
// inside_halfspace(p; R, B, A), (p; R, C, B), (p; R, A, C), linear in the direction u = p - R) is below -tau * |normal| at all
// four corners of the cell widened by `grow` (cube-map units) -- hence below -tau/sqrt(3) for every unit direction of the cell.
__device__ __forceinline__ void cone_map_build_cell(int cell, const float* __restrict__ verts, const int* __restrict__ faces, int F,
                                                    unsigned short* __restrict__ list, signed char* __restrict__ count) {
  const double tau = 0.02, grow = 1e-3;
  const int ib = cell % SD_CM_G, ia = (cell / SD_CM_G) % SD_CM_G, ms = cell / (SD_CM_G * SD_CM_G);
  const int m = ms >> 1;
  const double sgn = (ms & 1) ? -1.0 : 1.0;
  const double a0 = -1.0 + 2.0 * ia / SD_CM_G - grow, a1 = -1.0 + 2.0 * (ia + 1) / SD_CM_G + grow;
  const double b0 = -1.0 + 2.0 * ib / SD_CM_G - grow, b1 = -1.0 + 2.0 * (ib + 1) / SD_CM_G + grow;
  double U[4][3];
  for (int c = 0; c < 4; ++c) {
    const double a = (c & 1) ? a1 : a0, b = (c & 2) ? b1 : b0;
    if (m == 0) { U[c][0] = sgn; U[c][1] = a; U[c][2] = b; }
    else if (m == 1) { U[c][0] = a; U[c][1] = sgn; U[c][2] = b; }
    else { U[c][0] = a; U[c][1] = b; U[c][2] = sgn; }
  }
  int n = 0;
  for (int f = 0; f < F && n >= 0; ++f) {
    const int id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    double v[3][3];
    for (int t = 0; t < 3; ++t) for (int d = 0; d < 3; ++d) v[t][d] = (double)verts[3 * id[t] + d];
    // side k: rows (r0, r1, u) with (r0, r1) = (B, A), (C, B), (A, C)
    const int r0i[3] = {1, 2, 0}, r1i[3] = {0, 1, 2};
    bool out = false;
    for (int k = 0; k < 3 && !out; ++k) {
      const double* r0 = v[r0i[k]]; const double* r1 = v[r1i[k]];
      const double nz = r0[1] * r1[2] - r0[2] * r1[1], ny = r0[2] * r1[0] - r0[0] * r1[2], nx = r0[0] * r1[1] - r0[1] * r1[0];   // r0 x r1
      const double nn = sqrt(nz * nz + ny * ny + nx * nx);
      if (!(nn > 1e-12)) continue;                           // degenerate side: never excludes
      double mx = -1e300;
      for (int c = 0; c < 4; ++c) { const double sv = (nz * U[c][0] + ny * U[c][1] + nx * U[c][2]) / nn; mx = sv > mx ? sv : mx; }
      if (mx < -tau) out = true;
    }
    if (!out) { if (n < SD_CM_CAP) list[(size_t)cell * SD_CM_CAP + n++] = (unsigned short)f; else n = -1; }
  }
  count[cell] = (signed char)n;
}

// inside_polyhedron_kernel :195-231
__device__ __forceinline__ bool inside_polyhedron_kernel(float z, float y, float x, const float* __restrict__ pv,
                                                         const int* __restrict__ faces, int n_faces) {
  for (int i = 0; i < n_faces; ++i) {
    const int iA = faces[3 * i], iB = faces[3 * i + 1], iC = faces[3 * i + 2];
    if (!inside_halfspace(z, y, x, pv[3 * iA], pv[3 * iA + 1], pv[3 * iA + 2], pv[3 * iB], pv[3 * iB + 1], pv[3 * iB + 2],
                          pv[3 * iC], pv[3 * iC + 1], pv[3 * iC + 2]))
      return false;
  }
  return true;
}

// tetrahedron_volume :234-253 with R = origin
__device__ __forceinline__ float tetrahedron_volume0(float Az, float Ay, float Ax, float Bz, float By, float Bx, float Cz,
                                                     float Cy, float Cx) {
  const float M00 = Bz - Az, M01 = By - Ay, M02 = Bx - Ax;
  const float M10 = Cz - Az, M11 = Cy - Ay, M12 = Cx - Ax;
  const float M20 = 0.f - Az, M21 = 0.f - Ay, M22 = 0.f - Ax;
  const float det = M00 * (M11 * M22 - M21 * M12) - M01 * (M10 * M22 - M12 * M20) + M02 * (M10 * M21 - M11 * M20);
  return det / 6.f;
}

// build_halfspace :744-764 : fp32 normal, widened to double; inside <=> hs.(z,y,x,1) <= 0
__device__ __forceinline__ void build_halfspace(const float* A, const float* B, const float* C, double* hs) {
  const float Az = A[0], Ay = A[1], Ax = A[2];
  const float Bz = B[0], By = B[1], Bx = B[2];
  const float Cz = C[0], Cy = C[1], Cx = C[2];
  const float Pz = Bz - Az, Py = By - Ay, Px = Bx - Ax;
  const float Qz = Cz - Az, Qy = Cy - Ay, Qx = Cx - Ax;
  const float Nz = -(Py * Qx - Px * Qy);
  const float Ny = -(Px * Qz - Pz * Qx);
  const float Nx = -(Pz * Qy - Py * Qz);
  hs[0] = Nz; hs[1] = Ny; hs[2] = Nx;
  hs[3] = -(Az * Nz + Ay * Ny + Ax * Nx);
}

// intersect_sphere_isotropic :494-520 (mixed float/double exactly as written)
__device__ __forceinline__ float intersect_sphere_isotropic(float r1, const float* p1, float r2, const float* p2, const float* an) {
  const float dz = an[0] * (p1[0] - p2[0]);
  const float dy = an[1] * (p1[1] - p2[1]);
  const float dx = an[2] * (p1[2] - p2[2]);
  const float d = sqrtf(dz * dz + dy * dy + dx * dx);
  const float rmin = fminf(r1, r2), rmax = fmaxf(r1, r2);
  if (d > (r1 + r2)) return 0;
  if ((double)rmax >= (double)(d + rmin) - 1.e-10) return (float)(M_PI * 4.f / 3 * rmin * rmin * rmin);
  const float t = (r1 + r2 - d) / 2 / d;
  const float h1 = (r2 - r1 + d) * t;
  const float h2 = (r1 - r2 + d) * t;
  const float v1 = (float)(M_PI / 3 * h1 * h1 * (3 * r1 - h1));
  const float v2 = (float)(M_PI / 3 * h2 * h2 * (3 * r2 - h2));
  return (v1 + v2) / (an[0] * an[1] * an[2]);
}

// intersect_bbox :523-532 (inclusive int boxes, no +1)
__device__ __forceinline__ float intersect_bbox(const int* b1, const int* b2) {
  const float wz = (float)fmax(0.0, fmin((double)b1[1], (double)b2[1]) - fmax((double)b1[0], (double)b2[0]));
  const float wy = (float)fmax(0.0, fmin((double)b1[3], (double)b2[3]) - fmax((double)b1[2], (double)b2[2]));
  const float wx = (float)fmax(0.0, fmin((double)b1[5], (double)b2[5]) - fmax((double)b1[4], (double)b2[4]));
  return wx * wy * wz;
}

}  // namespace sd3
