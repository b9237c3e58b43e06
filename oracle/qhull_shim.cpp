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

// Every quantity the cascade of _COMMON_non_maximum_suppression_sparse (:1207-1318) looks at for one pair, computed with the reference's own
// functions: out[8 p + ...] = volume i, volume j, upper bound (outer spheres / boxes :1213-1219), lower bound (inner spheres :1232-1237),
// kernel stage, hull stage, rendered overlap (the FULL voxel count: overlap_maximal = infinity), 0.  tools/diag_cartesian.py uses it to
// find the stage at which a keep flag of the device NMS leaves the reference.
extern "C" void ref_pair_cascade(const float* dist, const float* points, const float* verts, const int* faces, const int* pairs,
                                 int n_pairs, int n_rays, int n_faces, const float* anisotropy, float* out) {
  float* pv1 = new float[3 * n_rays];
  float* pv2 = new float[3 * n_rays];
  for (int p = 0; p < n_pairs; ++p) {
    const int i = pairs[2 * p], j = pairs[2 * p + 1];
    const float* c1 = &points[3 * i];
    const float* c2 = &points[3 * j];
    const float* d1 = &dist[(size_t)i * n_rays];
    const float* d2 = &dist[(size_t)j * n_rays];
    float* o = out + 8 * (size_t)p;
    int b1[6], b2[6];
    polyhedron_bbox(d1, c1, verts, n_rays, b1);
    polyhedron_bbox(d2, c2, verts, n_rays, b2);
    polyhedron_polyverts(d1, c1, verts, n_rays, pv1);
    polyhedron_polyverts(d2, c2, verts, n_rays, pv2);
    o[0] = polyhedron_volume(d1, verts, faces, n_rays, n_faces);
    o[1] = polyhedron_volume(d2, verts, faces, n_rays, n_faces);
    o[2] = fmin(intersect_sphere_isotropic(bounding_radius_outer_isotropic(d1, verts, n_rays, anisotropy), c1,
                                           bounding_radius_outer_isotropic(d2, verts, n_rays, anisotropy), c2, anisotropy),
                intersect_bbox(b1, b2));
    o[3] = intersect_sphere_isotropic(bounding_radius_inner_isotropic(d1, verts, faces, n_rays, n_faces, anisotropy), c1,
                                      bounding_radius_inner_isotropic(d2, verts, faces, n_rays, n_faces, anisotropy), c2, anisotropy);
    o[4] = qhull_overlap_kernel(pv1, c1, pv2, c2, faces, n_rays, n_faces);
    o[5] = qhull_overlap_convex_hulls(pv1, c1, pv2, c2, faces, n_rays, n_faces);
    const int Nz = b1[1] - b1[0] + 1, Ny = b1[3] - b1[2] + 1, Nx = b1[5] - b1[4] + 1;
    bool* rendered = new bool[(size_t)Nz * Ny * Nx];
    render_polyhedron(d1, c1, b1, pv1, faces, n_rays, n_faces, rendered, Nz, Ny, Nx);
    o[6] = (float)overlap_render_polyhedron(d2, c2, b1, pv2, faces, n_rays, n_faces, rendered, Nz, Ny, Nx, 1e30f);
    o[7] = 0.f;
    delete[] rendered;
  }
  delete[] pv1;
  delete[] pv2;
}
