// nms3d.hip -- PLACEHOLDER until the 3D cascade lands (entry points fail loudly).
#include "common.h"
#include "../../include/stardist_hip.h"
extern "C" int sd_nms3d_device(const float*, const float*, const float*, int, int, int, const float*, const int*, float, int,
                               int, int, uint8_t*, int64_t*, void*) {
  sd::set_error("sd_nms3d_device: not implemented yet");
  return -1;
}
extern "C" void _LIB_non_maximum_suppression_sparse(const float*, const float*, const float*, const int, const int, const int,
                                                    const float*, const int*, const float, const int, const int, const int, bool*) {
  fprintf(stderr, "_LIB_non_maximum_suppression_sparse: not implemented yet\n");
  abort();
}
extern "C" int sd_polyhedron_to_label_device(const float*, const float*, const float*, const int*, int, int, int, const int*, int,
                                             int, int, int, int, int, int, int*, void*) {
  sd::set_error("sd_polyhedron_to_label_device: not implemented yet");
  return -1;
}
extern "C" void _LIB_polyhedron_to_label(const float*, const float*, const float*, const int*, const int, const int, const int,
                                         const int*, const int, const int, const int, const int, const int, const int, const int, int*) {
  fprintf(stderr, "_LIB_polyhedron_to_label: not implemented yet\n");
  abort();
}
