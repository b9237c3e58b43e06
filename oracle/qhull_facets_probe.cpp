// oracle/qhull_facets_probe.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
// Prints the facets of the convex hull the reference's halfspaces_convex (stardist/lib/stardist3d_impl.cpp:767-795: Qhull "convex hull", no
// options) builds for one polyhedron: simplicial flag, orientation, hyperplane (normal, offset) in full precision and the facet's vertices IN THE
// ORDER OF ITS VERTEX SET -- the order qh_setfacetplane hands to qh_sethyperplane_det.  stdin: n, then n x 3 float coordinates.
// tools/qhull_facet_plane_order.py uses it to show what the hull-facet voxels of render mode "full" depend on.
#include "libqhullcpp/Qhull.h"
#include "libqhullcpp/QhullFacetList.h"
#include "libqhullcpp/QhullVertexSet.h"
#include "libqhullcpp/QhullHyperplane.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace orgQhull;
int main(int argc, char** argv) {
  int n; if (scanf("%d", &n) != 1) return 1;
  std::vector<double> pts(3 * n);
  for (int i = 0; i < 3 * n; ++i) { float f; if (scanf("%f", &f) != 1) return 1; pts[i] = f; }
  Qhull q("convex hull", 3, n, pts.data(), "");
  int nf = 0, nsimp = 0;
  for (auto f : q.facetList()) {
    ++nf;
    facetT* ft = f.getFacetT();
    QhullHyperplane h = f.hyperplane();
    QhullVertexSet vs = f.vertices();
    printf("facet %d simplicial %d toporient %d nverts %d plane %.17g %.17g %.17g %.17g verts", f.id(), (int)ft->simplicial, (int)ft->toporient, (int)vs.size(), h[0], h[1], h[2], h.offset());
    for (auto v : vs) printf(" %d", v.point().id());
    printf("\n");
    nsimp += ft->simplicial;
  }
  printf("facets %d simplicial %d\n", nf, nsimp);
  return 0;
}
