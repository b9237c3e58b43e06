// tests/host/clip_check.cpp -- CPU harness: the product's scan-beam sweep
// (stardist_amd/csrc/clip_sweep.h, compiled for the host) against the reference's vendored
// Clipper (oracle/_ref/libclipper_ref.so) on seeded random star-polygon pairs.
// Build/run: see tests/test_clip_host.py.
#include "../../stardist_amd/csrc/clip_sweep.h"
#include "../../stardist_amd/csrc/clip_sweep_full.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

extern "C" float clipper_ref_area(const int64_t*, const int64_t*, int, const int64_t*, const int64_t*, int);
extern "C" int clipper_ref_intersect(const int64_t*, const int64_t*, int, const int64_t*, const int64_t*, int,
                                     int64_t*, int, int*, int);

typedef sdclip::Sweep<128, 512, 128> SweepT;
typedef sdclip::SweepFull<128, 512, 128, 1024, 256> SweepF;
// LDS-style storage policy exercised on the host (interleaved arrays, recomputed slopes, int8 indices)
typedef sdclip::LdsStorage<4> LdsP;
typedef sdclip::Sweep<32, 64, 32, LdsP> SweepL;
typedef sdclip::SweepFull<32, 64, 32, 192, 64, LdsP> SweepFL;

static void make_poly(std::mt19937& rng, int n_rays, float radius, float noise, float cy, float cx,
                      std::vector<int64_t>& xs, std::vector<int64_t>& ys) {
  // vertices exactly as stardist2d.cpp:419,447-471 (float math, truncation to cInt)
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  const float ANGLE_PI = 2 * M_PI / n_rays;
  xs.resize(n_rays); ys.resize(n_rays);
  for (int k = 0; k < n_rays; k++) {
    float d = radius * (1.f + noise * U(rng));
    if (d < 1e-3f) d = 1e-3f;
    const float y = (float)(cy + d * sinf(ANGLE_PI * k));
    const float x = (float)(cx + d * cosf(ANGLE_PI * k));
    xs[k] = (int64_t)x; ys[k] = (int64_t)y;
  }
}

int main(int argc, char** argv) {
  long n_pairs = argc > 1 ? atol(argv[1]) : 100000;
  int n_rays = argc > 2 ? atoi(argv[2]) : 32;
  float radius = argc > 3 ? atof(argv[3]) : 10.f;
  float noise = argc > 4 ? atof(argv[4]) : 0.1f;
  unsigned seed = argc > 5 ? atoi(argv[5]) : 1;
  float offset = argc > 6 ? atof(argv[6]) : 0.f;      // coordinate offset (large-coordinate regime)
  int verbose = argc > 7 ? atoi(argv[7]) : 0;
  int all_full = argc > 8 ? atoi(argv[8]) : 0;
  int lds_mode = argc > 9 ? atoi(argv[9]) : 0;   // 1: run the LdsStorage policy variants (n_rays <= 32)
  static char lds_buf[1 << 16];
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U01(0.f, 1.f);
  std::vector<int64_t> xa, ya, xb, yb;
  static SweepT sw;
  static SweepF sf;
  long full_runs = 0, mism_fast_nojoin = 0;
  long mism = 0, with_joins = 0, mism_joins = 0, flagged = 0, nonzero = 0, inexact = 0;
  double max_rel = 0;
  for (long p = 0; p < n_pairs; p++) {
    float cy = offset + 50 + (int)(U01(rng) * 20), cx = offset + 50 + (int)(U01(rng) * 20);
    float sep = U01(rng) * 2.2f * radius;
    float ang = U01(rng) * 6.2831853f;
    float cy2 = cy + (float)(int)(sep * sinf(ang)), cx2 = cx + (float)(int)(sep * cosf(ang));
    float r2 = radius * (0.5f + U01(rng));
    make_poly(rng, n_rays, radius, noise, cy, cx, xa, ya);
    make_poly(rng, n_rays, r2, noise, cy2, cx2, xb, yb);
    float ref = clipper_ref_area(xa.data(), ya.data(), n_rays, xb.data(), yb.data(), n_rays);
    long long twice; int st; int nj;
    if (!lds_mode) {
      sw.reset_state();
      sw.add_path(xa.data(), ya.data(), n_rays, sdclip::kClip, 0);
      sw.add_path(xb.data(), yb.data(), n_rays, sdclip::kSubject, 128);
      twice = sw.execute(); st = sw.status; nj = sw.n_joins;
      sw.n_joins = nj;
    } else {
      SweepL sl; sdclip::HostLds::base() = lds_buf; sdclip::HostLds::tid() = (int)(p & 3);
      if (SweepL::lds_bytes() > sizeof(lds_buf)) { printf("lds_buf too small\n"); return 2; }
      sl.reset_state();
      sl.add_path(xa.data(), ya.data(), n_rays, sdclip::kClip, 0);
      sl.add_path(xb.data(), yb.data(), n_rays, sdclip::kSubject, 32);
      twice = sl.execute(); st = sl.status; nj = sl.n_joins;
      if (p == 0) printf("LdsStorage<4> bytes per 4 threads: %u\n", SweepL::lds_bytes());
    }
    if (nj == 0 && 0.5f * (float)twice != ref) mism_fast_nojoin++;
    if (nj > 0 || all_full) {
      if (!lds_mode) {
        sf.reset_state();
        sf.add_path(xa.data(), ya.data(), n_rays, sdclip::kClip, 0);
        sf.add_path(xb.data(), yb.data(), n_rays, sdclip::kSubject, 128);
        twice = sf.execute(); st = sf.status;
      } else {
        static SweepFL sfl; sdclip::HostLds::base() = lds_buf; sdclip::HostLds::tid() = (int)(p & 3);
        sfl.reset_state();
        sfl.add_path(xa.data(), ya.data(), n_rays, sdclip::kClip, 0);
        sfl.add_path(xb.data(), yb.data(), n_rays, sdclip::kSubject, 32);
        twice = sfl.execute(); st = sfl.status;
      }
      full_runs++;
    }
    float mine = 0.5f * (float)twice;
    if (st) flagged++;
    if (nj) with_joins++;
    if (ref != 0) nonzero++;
    if (sw.sum_abs_terms >= (1ll << 24)) inexact++;
    if (mine != ref) {
      mism++;
      if (nj) mism_joins++;
      double rel = fabs(mine - ref) / (fabs(ref) + 1e-9);
      if (rel > max_rel) max_rel = rel;
      if (verbose && mism <= verbose) {
        printf("MISMATCH pair %ld: ref=%.1f mine=%.1f status=%d joins=%d\nA:", p, ref, mine, st, nj);
        for (int k = 0; k < n_rays; k++) printf(" (%ld,%ld)", (long)xa[k], (long)ya[k]);
        printf("\nB:");
        for (int k = 0; k < n_rays; k++) printf(" (%ld,%ld)", (long)xb[k], (long)yb[k]);
        printf("\n");
        int64_t out[4096]; int lens[64];
        int np = clipper_ref_intersect(xa.data(), ya.data(), n_rays, xb.data(), yb.data(), n_rays, out, 2048, lens, 64);
        int k = 0;
        for (int r = 0; r < np; r++) { printf("ref path %d:", r); for (int i = 0; i < lens[r]; i++, k++) printf(" (%ld,%ld)", (long)out[2*k], (long)out[2*k+1]); printf("\n"); }
      }
    }
  }
  printf("pairs=%ld nonzero=%ld mismatches=%ld (with_joins=%ld, mism_with_joins=%ld, fast_nojoin_mism=%ld, full_runs=%ld) flagged=%ld inexact_risk=%ld max_rel=%.3g\n",
         n_pairs, nonzero, mism, with_joins, mism_joins, mism_fast_nojoin, full_runs, flagged, inexact, max_rel);
  return mism ? 1 : 0;
}
