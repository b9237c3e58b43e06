// tests/host/beam_check.cpp -- CPU harness for the bound-slot sweep (stardist_amd/csrc/clip_beam.h,
// compiled for the host): every pair is evaluated by
//   (1) the reference's vendored Clipper (oracle/_ref/libclipper_ref.so),
//   (2) the per-pair sweep of clip_sweep.h (already pinned to (1)),
//   (3) prepare_polygon + Beam (the GPU layout, PlainStorage and the LDS-interleaved policy),
// and (3) must reproduce (2) exactly -- area, number of join records, failure status -- and (1) whenever no join is
// recorded.  Capacity flags (ST_OVERFLOW_*) are counted separately: such pairs go to the general path on the GPU.
// Build/run: see tests/test_cpu_clip_host.py.
#include "../../stardist_amd/csrc/clip_sweep.h"
#include "../../stardist_amd/csrc/clip_beam.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

extern "C" float clipper_ref_area(const int64_t*, const int64_t*, int, const int64_t*, const int64_t*, int);

#ifndef BEAM_MAXV
#define BEAM_MAXV 32
#endif
#ifndef BEAM_K
#define BEAM_K 8
#endif
#ifndef BEAM_MAXIL
#define BEAM_MAXIL 16
#endif
#ifndef BEAM_MAXREC
#define BEAM_MAXREC 8
#endif

typedef sdclip::Sweep<128, 512, 128> SweepT;
typedef sdclip::PolyPrep<BEAM_MAXV> Prep;
typedef sdclip::PrepWork<sdclip::PlainStorage, BEAM_MAXV> PrepW;
typedef sdclip::Beam<BEAM_MAXV, BEAM_K, BEAM_MAXIL, BEAM_MAXREC> BeamT;
typedef sdclip::LdsStorage<4> LdsP;
typedef sdclip::PrepWork<LdsP, BEAM_MAXV> PrepWL;
typedef sdclip::Beam<BEAM_MAXV, BEAM_K, BEAM_MAXIL, BEAM_MAXREC, LdsP> BeamL;
typedef sdclip::LdsStorage16<4> Lds16P;                 // 16-bit coordinates relative to the pair's origin, slopes recomputed (tier 1 on the GPU)
typedef sdclip::Beam<BEAM_MAXV, BEAM_K, BEAM_MAXIL, BEAM_MAXREC, Lds16P> BeamL16;

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
  float offset = argc > 6 ? atof(argv[6]) : 0.f;
  int verbose = argc > 7 ? atoi(argv[7]) : 0;
  int lds_mode = argc > 8 ? atoi(argv[8]) : 0;
  if (n_rays > BEAM_MAXV) { printf("n_rays > BEAM_MAXV\n"); return 2; }
  alignas(64) static char lds_buf[1 << 18];
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U01(0.f, 1.f);
  std::vector<int64_t> xa, ya, xb, yb;
  static SweepT sw;
  static Prep pa, pb;
  long mism_old = 0, mism_ref = 0, flagged_cap = 0, flagged_other = 0, with_joins = 0, nonzero = 0, join_mism = 0, fail_mism = 0;
  long max_ael = 0; long capbits[8] = {0};
  FILE* fin = getenv("BEAM_PAIRS_FILE") ? fopen(getenv("BEAM_PAIRS_FILE"), "r") : nullptr;   // rows: xa[R] ya[R] xb[R] yb[R]
  for (long p = 0; p < n_pairs; p++) {
    if (fin) {
      xa.resize(n_rays); ya.resize(n_rays); xb.resize(n_rays); yb.resize(n_rays);
      bool ok = true;
      for (auto* v : {&xa, &ya, &xb, &yb}) for (int k = 0; k < n_rays; k++) { long t; if (fscanf(fin, "%ld", &t) != 1) ok = false; (*v)[k] = t; }
      if (!ok) { n_pairs = p; break; }
    } else {
    float cy = offset + 50 + (int)(U01(rng) * 20), cx = offset + 50 + (int)(U01(rng) * 20);
    float sep = U01(rng) * 2.2f * radius;
    float ang = U01(rng) * 6.2831853f;
    float cy2 = cy + (float)(int)(sep * sinf(ang)), cx2 = cx + (float)(int)(sep * cosf(ang));
    float r2 = radius * (0.5f + U01(rng));
    make_poly(rng, n_rays, radius, noise, cy, cx, xa, ya);
    make_poly(rng, n_rays, r2, noise, cy2, cx2, xb, yb);
    }
    const float ref = clipper_ref_area(xa.data(), ya.data(), n_rays, xb.data(), yb.data(), n_rays);
    sw.reset_state();
    sw.add_path(xa.data(), ya.data(), n_rays, sdclip::kClip, 0);
    sw.add_path(xb.data(), yb.data(), n_rays, sdclip::kSubject, 128);
    const long long t_old = sw.execute();
    const int st_old = sw.status, nj_old = sw.n_joins;

    long long t_new; int st_new, nj_new;
    if (!lds_mode) {
      PrepW w;
      w.prepare(xa.data(), ya.data(), n_rays, &pa);
      w.prepare(xb.data(), yb.data(), n_rays, &pb);
      BeamT bm;
      bm.reset_state(&pa, &pb);
      t_new = bm.execute(); st_new = bm.status; nj_new = bm.n_joins;
    } else if (lds_mode == 2) {
      sdclip::HostLds::base() = lds_buf; sdclip::HostLds::tid() = (int)(p & 3);
      PrepWL w;
      w.prepare(xa.data(), ya.data(), n_rays, &pa);
      w.prepare(xb.data(), yb.data(), n_rays, &pb);
      BeamL16 bm;
      bm.reset_state(&pa, &pb);
      t_new = bm.execute(); st_new = bm.status; nj_new = bm.n_joins;
      if (p == 0) printf("LdsStorage16<4>: beam %u bytes per 4 threads (32-bit form: %u)\n", BeamL16::lds_bytes(), BeamL::lds_bytes());
    } else {
      sdclip::HostLds::base() = lds_buf; sdclip::HostLds::tid() = (int)(p & 3);
      if (BeamL::lds_bytes() > sizeof(lds_buf) || PrepWL::lds_bytes() > sizeof(lds_buf)) { printf("lds_buf too small\n"); return 2; }
      PrepWL w;
      w.prepare(xa.data(), ya.data(), n_rays, &pa);
      w.prepare(xb.data(), yb.data(), n_rays, &pb);
      BeamL bm;
      bm.reset_state(&pa, &pb);
      t_new = bm.execute(); st_new = bm.status; nj_new = bm.n_joins;
      if (p == 0) printf("LdsStorage<4>: beam %u bytes, prep %u bytes per 4 threads\n", BeamL::lds_bytes(), PrepWL::lds_bytes());
    }
    if (ref != 0) nonzero++;
    if (nj_old) with_joins++;
    const int cap = sdclip::ST_OVERFLOW_IL | sdclip::ST_OVERFLOW_REC | sdclip::ST_OVERFLOW_AEL | sdclip::ST_OVERFLOW_LM | sdclip::ST_OVERFLOW_GJ;
    if (st_new & cap) { flagged_cap++; for (int b = 0; b < 8; ++b) if (st_new & (1 << b)) capbits[b]++; continue; }          // goes to the general path
    if (st_new & ~(cap | sdclip::ST_FAIL)) flagged_other++;
    bool bad = false;
    if (t_new != t_old) { mism_old++; bad = true; }
    if ((nj_new > 0) != (nj_old > 0)) { join_mism++; bad = true; }
    if ((st_new & sdclip::ST_FAIL) != (st_old & sdclip::ST_FAIL)) { fail_mism++; bad = true; }
    if (nj_new == 0 && 0.5f * (float)t_new != ref) { mism_ref++; bad = true; }
    if (bad && verbose > 0) {
      --verbose;
      printf("MISMATCH pair %ld: ref=%.1f old=%lld (st %d, joins %d) new=%lld (st %d, joins %d)\nA:", p, ref, t_old, st_old, nj_old, t_new, st_new, nj_new);
      for (int k = 0; k < n_rays; k++) printf(" (%ld,%ld)", (long)xa[k], (long)ya[k]);
      printf("\nB:");
      for (int k = 0; k < n_rays; k++) printf(" (%ld,%ld)", (long)xb[k], (long)yb[k]);
      printf("\n");
    }
  }
  printf("pairs=%ld nonzero=%ld with_joins=%ld mism_vs_sweep=%ld join_flag_mism=%ld fail_mism=%ld mism_vs_clipper=%ld capacity_flagged=%ld other_flagged=%ld\n",
         n_pairs, nonzero, with_joins, mism_old, join_mism, fail_mism, mism_ref, flagged_cap, flagged_other);
  printf("  capacity flags: IL=%ld REC=%ld AEL=%ld LM=%ld GJ=%ld\n", capbits[0], capbits[1], capbits[5], capbits[6], capbits[7]);
#ifdef BEAM_COUNT
  { const BeamCounters& c = beam_counters(); const double n = (double)n_pairs;
    printf("  per pair: scanbeams %.1f  mean AEL %.2f  top_x %.1f  beams-with-intersections %.2f  intersections %.2f  intersect_edges %.2f  out pts %.1f  edge updates %.1f  maxima %.2f  lm %.2f  horizontals %.2f  static edge loads %.1f\n",
           c.beams / n, (double)c.ael_sum / c.beams, c.topx / n, c.il_beams / n, c.isect_pt / n, c.isect_edges / n, c.outpt / n, c.update / n, c.maxima / n, c.lm / n, c.horz / n, c.es_loads / n); }
#endif
  (void)max_ael;
  return (mism_old || join_mism || fail_mism || mism_ref || flagged_other) ? 1 : 0;
}
