// oracle/qhull_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Pair-level probe of the reference's two Qhull volume stages.  qhull_overlap_kernel and qhull_overlap_convex_hulls are
// `inline` functions inside stardist/lib/stardist3d_impl.cpp (:830-869, :872-939), so this shim compiles that translation unit
// where it lies (see oracle/Makefile: -I$(REF)) and forwards to them, building the vertices with the reference's own
// polyhedron_polyverts (:570-584).
#include "stardist3d_impl.cpp"

extern "C" void ref_pair_volumes(const float* dist, const float* points, const float* verts, const int* faces, const int* pairs,
                                 int n_pairs, int n_rays, int n_faces, float* vol_kernel, float* vol_hull) {
  float* pv1 = new float[3 * n_rays];
  float* pv2 = new float[3 * n_rays];
  for (int p = 0; p < n_pairs; ++p) {
    const int i = pairs[2 * p], j = pairs[2 * p + 1];
    const float* c1 = &points[3 * i];
    const float* c2 = &points[3 * j];
    polyhedron_polyverts(&dist[(size_t)i * n_rays], c1, verts, n_rays, pv1);
    polyhedron_polyverts(&dist[(size_t)j * n_rays], c2, verts, n_rays, pv2);
    if (vol_kernel) vol_kernel[p] = qhull_overlap_kernel(pv1, c1, pv2, c2, faces, n_rays, n_faces);
    if (vol_hull) vol_hull[p] = qhull_overlap_convex_hulls(pv1, c1, pv2, c2, faces, n_rays, n_faces);
  }
  delete[] pv1;
  delete[] pv2;
}
