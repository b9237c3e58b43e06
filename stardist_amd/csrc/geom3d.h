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
