// tests/host/beam_prep_lib.cpp -- the host build of prepare_polygon (stardist_amd/csrc/clip_beam.h) as a tiny shared
// library, so the GPU test can compare the records the device kernel writes with the host's byte for byte.
#include "../../stardist_amd/csrc/clip_sweep.h"
#include "../../stardist_amd/csrc/clip_beam.h"
#include <cstring>

template <int MAXV>
static void run(const int* x, const int* y, int n, int R, void* out) {
  sdclip::PolyPrep<MAXV>* o = (sdclip::PolyPrep<MAXV>*)out;
  for (int i = 0; i < n; ++i) {
    sdclip::PrepWork<sdclip::PlainStorage, MAXV> w;
    w.prepare(x + (size_t)i * R, y + (size_t)i * R, R, o + i);
  }
}
extern "C" long beam_prep_record_bytes(int R) {
  return R <= 32 ? sizeof(sdclip::PolyPrep<32>) : R <= 64 ? sizeof(sdclip::PolyPrep<64>) : R <= 128 ? sizeof(sdclip::PolyPrep<128>) : sizeof(sdclip::PolyPrep<256>);
}
// fields of a record that are defined: header, v[0..n], ecode/mpair/hlast[0..n), lm[0..n_lm); everything else is left as the caller initialised it
extern "C" void beam_prepare_host(const int* x, const int* y, int n, int R, void* out) {
  if (R <= 32) run<32>(x, y, n, R, out); else if (R <= 64) run<64>(x, y, n, R, out);
  else if (R <= 128) run<128>(x, y, n, R, out); else run<256>(x, y, n, R, out);
}
