// conv_check.cpp -- host emulation of conv3x3.hip's data flow (unit staging -> LDS tile, packed weight block -> LDS, per-lane
// operand fetch, v_mfma_f32_32x32x2_f32 semantics, accumulator -> pixel map), built from the SAME index functions the device code
// uses (stardist_amd/csrc/conv3x3_layout.h), checked against a direct 2D / 3D convolution in double precision.
// usage: conv_check   (exit code 0 = all configurations agree)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../stardist_amd/csrc/conv3x3_layout.h"

using namespace sdconv;

struct SrcH { const float* p; int stride, shz, shy, shx; };

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

// one output tile of one output-channel group, emulating a workgroup
static void emulate_tile(const SrcH* src, int n_chunks, int kz, int g, int D, int H, int W, const float* wp, const float* bias, int c_out, int act,
                         int t, int tiles_x, int tiles_plane, float* out) {
  const int nt = nt_for(c_out), n_units = n_chunks * kz;
  std::vector<float> tileL(TILE_FLOATS), Wl(wunit_floats(nt));
  std::vector<float> acc((size_t)4 * 2 * nt * 64 * 16);
  auto A = [&](int wave, int p, int ct, int lane, int r) -> float& { return acc[((((size_t)wave * 2 + p) * nt + ct) * 64 + lane) * 16 + r]; };
  for (int wave = 0; wave < 4; ++wave) for (int p = 0; p < 2; ++p) for (int ct = 0; ct < nt; ++ct) for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 16; ++r) A(wave, p, ct, lane, r) = bias ? bias[g * 32 * nt + ct * 32 + (lane & 31)] : 0.f;
  const int tz = t / tiles_plane, tr = t - tz * tiles_plane;
  const int ty0 = (tr / tiles_x) * TH - 1, tx0 = (tr % tiles_x) * TW - 1;
  for (int u = 0; u < n_units; ++u) {
    const int c = u / kz, dz = kz == 3 ? u - c * 3 - 1 : 0;
    const SrcH& S = src[c];
    const int z = tz + dz;
    const int ws = W >> S.shx, hs = H >> S.shy;
    for (int e = 0; e < TILE_F4; ++e) {
      int ty, tx, q4; stage_elem(e, ty, tx, q4);
      const int gy = ty0 + ty, gx = tx0 + tx;
      for (int k = 0; k < 4; ++k) {
        float v = 0.f;
        const bool fast = z >= 0 && z < D && ty0 >= 0 && ty0 + HALO_H <= H && tx0 >= 0 && tx0 + HALO_W <= W;      // the kernel's interior-tile path
        if (fast)
          v = S.p[(((size_t)(z >> S.shz) * hs + src_base(ty0, S.shy) + src_rel(ty, S.shy)) * ws + src_base(tx0, S.shx) + src_rel(tx, S.shx)) * S.stride + q4 * 4 + k];
        else if (z >= 0 && z < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = S.p[(((size_t)(z >> S.shz) * hs + (gy >> S.shy)) * ws + (gx >> S.shx)) * S.stride + q4 * 4 + k];
        tileL[tile_off(ty, tx, q4 * 4) + k] = v;
      }
    }
    for (int e = 0; e < wunit_floats(nt); ++e) Wl[e] = wp[((size_t)g * n_units + u) * wunit_floats(nt) + e];     // linear copy (LDS-direct)
    const float* wl = Wl.data();
    for (int wave = 0; wave < 4; ++wave)
      for (int gi = 0; gi < 12; ++gi) {
        const int dx = gi >> 2, j = gi & 3;
        for (int dy = 0; dy < 3; ++dy)
          for (int e = 0; e < 4; ++e)
            for (int ct = 0; ct < nt; ++ct)
              for (int p = 0; p < 2; ++p) {
                // D[m][n] = fma(A[m][1], B[1][n], fma(A[m][0], B[0][n], C[m][n])); lane (i, h): a = A[i][h], b = B[h][i]
                float a[64], b[64];
                for (int lane = 0; lane < 64; ++lane) {
                  const int i = lane & 31, h = lane >> 5;
                  a[lane] = tileL[a_off(wave * 2, dy + p, dx, j, i, h) + e];
                  b[lane] = wl[wl_off(dy * 3 + dx, j, h, ct, i, nt) + e];
                }
                for (int lane = 0; lane < 64; ++lane) {
                  const int n = lane & 31, h = lane >> 5;
                  for (int r = 0; r < 16; ++r) {
                    const int m = acc_col(r, h);
                    float v = A(wave, p, ct, lane, r);
                    v = fmaf(a[m], b[n], v);
                    v = fmaf(a[32 + m], b[32 + n], v);
                    A(wave, p, ct, lane, r) = v;
                  }
                }
              }
      }
  }
  const int x0 = (tr % tiles_x) * TW;
  for (int wave = 0; wave < 4; ++wave) for (int p = 0; p < 2; ++p) {
    const int y = (tr / tiles_x) * TH + wave * 2 + p;
    if (y >= H) continue;
    for (int ct = 0; ct < nt; ++ct) for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
      const int x = x0 + acc_col(r, lane >> 5);
      float v = A(wave, p, ct, lane, r);
      if (act == 1) v = fmaxf(v, 0.f);
      if (x < W) out[(((size_t)tz * H + y) * W + x) * c_out + g * 32 * nt + ct * 32 + (lane & 31)] = v;
    }
  }
}


// ---- split-bf16 variant (conv3x3_bf16.hip): halo staging with the three-way split, packed sub-unit weight blocks, per-lane 16-byte
// operand fetch, v_mfma_f32_32x32x16_bf16 semantics (8 values per lane and operand; slot (h, j) of A meets slot (h, j) of B), the six
// plane pairs, same accumulator map
static float bf16_to_f(unsigned short v) { return u2f((unsigned)v << 16); }
static void emulate_tile_bf16(const SrcH* src, int n_chunks, int kz, int g, int D, int H, int W, const unsigned short* wp, const float* bias, int c_out,
                              int act, int t, int tiles_x, int tiles_plane, float* out) {
  const int n_units = n_chunks * kz;
  std::vector<unsigned char> tileB(BTILE_BYTES);
  std::vector<float> acc((size_t)4 * 2 * 64 * 16);
  auto A = [&](int wave, int p, int lane, int r) -> float& { return acc[(((size_t)wave * 2 + p) * 64 + lane) * 16 + r]; };
  for (int wave = 0; wave < 4; ++wave) for (int p = 0; p < 2; ++p) for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 16; ++r) A(wave, p, lane, r) = bias ? bias[g * 32 + (lane & 31)] : 0.f;
  const int tz = t / tiles_plane, tr = t - tz * tiles_plane;
  const int ty0 = (tr / tiles_x) * TH - 1, tx0 = (tr % tiles_x) * TW - 1;
  for (int u = 0; u < n_units; ++u) {
    const int c = u / kz, dz = kz == 3 ? u - c * 3 - 1 : 0;
    const SrcH& S = src[c];
    const int z = tz + dz;
    const int ws = W >> S.shx, hs = H >> S.shy;
    for (int e = 0; e < TILE_F4; ++e) {
      int ty, tx, q4; stage_elem(e, ty, tx, q4);
      const int gy = ty0 + ty, gx = tx0 + tx;
      for (int k = 0; k < 4; ++k) {
        float v = 0.f;
        if (z >= 0 && z < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
          v = S.p[(((size_t)(z >> S.shz) * hs + (gy >> S.shy)) * ws + (gx >> S.shx)) * S.stride + q4 * 4 + k];
        unsigned pl[3]; split3(v, pl[0], pl[1], pl[2]);
        for (int p = 0; p < 3; ++p) {
          const unsigned short b = (unsigned short)(pl[p] >> 16);
          memcpy(&tileB[btile_store_off(ty, tx, p, q4) + k * 2], &b, 2);
        }
      }
    }
    for (int dy = 0; dy < 3; ++dy) {
      const unsigned char* w = (const unsigned char*)wp + (((size_t)g * n_units + u) * 3 + dy) * BWSUB_BYTES;
      static const int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
      for (int wave = 0; wave < 4; ++wave)
        for (int gi = 0; gi < 6; ++gi) {
          const int dx = gi >> 1, b = gi & 1;
          for (int k = 0; k < 6; ++k)
            for (int p = 0; p < 2; ++p) {
              float a[64][8], bb[64][8];
              for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                unsigned short av[8], bv[8];
                memcpy(av, &tileB[btile_off(wave * 2 + dy + p, i + dx, PA[k], b, h)], 16);
                memcpy(bv, w + bw_off(dx, b, PB[k], h, i), 16);
                for (int j = 0; j < 8; ++j) { a[lane][j] = bf16_to_f(av[j]); bb[lane][j] = bf16_to_f(bv[j]); }
              }
              for (int lane = 0; lane < 64; ++lane) {
                const int n = lane & 31, h = lane >> 5;
                for (int r = 0; r < 16; ++r) {
                  const int m = acc_col(r, h);
                  float v = A(wave, p, lane, r);
                  for (int hh = 0; hh < 2; ++hh) for (int j = 0; j < 8; ++j) v += a[hh * 32 + m][j] * bb[hh * 32 + n][j];   // exact products, f32 sums
                  A(wave, p, lane, r) = v;
                }
              }
            }
        }
    }
  }
  const int x0 = (tr % tiles_x) * TW;
  for (int wave = 0; wave < 4; ++wave) for (int p = 0; p < 2; ++p) {
    const int y = (tr / tiles_x) * TH + wave * 2 + p;
    if (y >= H) continue;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 16; ++r) {
      const int x = x0 + acc_col(r, lane >> 5);
      float v = A(wave, p, lane, r);
      if (act == 1) v = fmaxf(v, 0.f);
      if (x < W) out[(((size_t)tz * H + y) * W + x) * c_out + g * 32 + (lane & 31)] = v;
    }
  }
}

// up masks: bit 0 x, bit 1 y, bit 2 z
static int run_case(int D, int H, int W, int kz, int c0, int up0, int c1, int up1, int c_out, int act) {
  const int c_in = c0 + c1;
  auto sh = [](int up, int b) { return (up >> b) & 1; };
  auto vol = [&](int up) { return (size_t)(D >> sh(up, 2)) * (H >> sh(up, 1)) * (W >> sh(up, 0)); };
  std::vector<float> s0(vol(up0) * c0), s1(c1 ? vol(up1) * c1 : 0);
  for (auto& v : s0) v = frand();
  for (auto& v : s1) v = frand();
  std::vector<float> w((size_t)c_out * c_in * 9 * kz), bias(c_out);
  for (auto& v : w) v = frand() * 0.1f;
  for (auto& v : bias) v = frand();
  std::vector<float> wp(packed_floats(c_in, c_out, kz));
  pack_weights(w.data(), c_in, c_out, kz, wp.data());
  SrcH src[MAX_CHUNKS]; int nc = 0;
  for (int k = 0; k < c0 / 32; ++k) src[nc++] = SrcH{s0.data() + k * 32, c0, sh(up0, 2), sh(up0, 1), sh(up0, 0)};
  for (int k = 0; k < c1 / 32; ++k) src[nc++] = SrcH{s1.data() + k * 32, c1, sh(up1, 2), sh(up1, 1), sh(up1, 0)};
  const int nt = nt_for(c_out), groups = c_out / (32 * nt);
  const int tiles_x = (W + TW - 1) / TW, tiles_plane = tiles_x * ((H + TH - 1) / TH), n_tiles = tiles_plane * D;
  std::vector<float> out((size_t)D * H * W * c_out, NAN), outb((size_t)D * H * W * c_out, NAN);
  std::vector<unsigned short> wpb(bpacked_bytes(c_in, c_out, kz) / 2);
  pack_weights_bf16(w.data(), c_in, c_out, kz, wpb.data());
  for (int g = 0; g < groups; ++g) for (int t = 0; t < n_tiles; ++t) {
    emulate_tile(src, nc, kz, g, D, H, W, wp.data(), bias.data(), c_out, act, t, tiles_x, tiles_plane, out.data());
    emulate_tile_bf16(src, nc, kz, g, D, H, W, wpb.data(), bias.data(), c_out, act, t, tiles_x, tiles_plane, outb.data());
  }
  double worst = 0, worstb = 0;
  auto in = [&](int z, int y, int x, int ci) -> double {
    if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return 0.0;
    const bool first = ci < c0;
    const int up = first ? up0 : up1, cc = first ? c0 : c1, cj = first ? ci : ci - c0;
    const float* p = first ? s0.data() : s1.data();
    return p[(((size_t)(z >> sh(up, 2)) * (H >> sh(up, 1)) + (y >> sh(up, 1))) * (W >> sh(up, 0)) + (x >> sh(up, 0))) * cc + cj];
  };
  for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) for (int co = 0; co < c_out; ++co) {
    double s = bias[co];
    for (int ci = 0; ci < c_in; ++ci) for (int a = 0; a < kz; ++a) for (int tap = 0; tap < 9; ++tap)
      s += in(z + (kz == 3 ? a - 1 : 0), y + tap / 3 - 1, x + tap % 3 - 1, ci) * w[(((size_t)co * c_in + ci) * kz + a) * 9 + tap];
    if (act == 1 && s < 0) s = 0;
    const double d = fabs(s - out[(((size_t)z * H + y) * W + x) * c_out + co]);
    if (!(d <= worst)) worst = d;        // catches NaN (unwritten outputs)
    const double db = fabs(s - outb[(((size_t)z * H + y) * W + x) * c_out + co]);
    if (!(db <= worstb)) worstb = db;
  }
  printf("D=%d H=%d W=%d kz=%d  %d(up%d)+%d(up%d) -> %d act=%d  groups=%d units=%d  max|err| f32 %.3g  split-bf16 %.3g\n", D, H, W, kz, c0, up0, c1, up1, c_out, act, groups,
         nc * kz, worst, worstb);
  return worst < 3e-5 && worstb < 3e-5 ? 0 : 1;
}

int main() {
  srand(1);
  int bad = 0;
  bad += run_case(1, 16, 64, 1, 32, 0, 0, 0, 32, 1);
  bad += run_case(1, 32, 128, 1, 32, 3, 32, 0, 32, 1);        // interior tiles with an up-sampled source
  bad += run_case(4, 24, 96, 3, 32, 5, 32, 0, 32, 0);         // 3D, z and x up-sampled only
  bad += run_case(1, 13, 45, 1, 32, 0, 0, 0, 64, 0);
  bad += run_case(1, 10, 34, 1, 64, 0, 0, 0, 32, 1);
  bad += run_case(1, 12, 36, 1, 32, 3, 32, 0, 32, 1);
  bad += run_case(1, 8, 32, 1, 128, 3, 128, 0, 64, 1);
  bad += run_case(4, 9, 33, 3, 32, 0, 0, 0, 32, 1);
  bad += run_case(4, 8, 34, 3, 32, 7, 32, 0, 32, 1);
  bad += run_case(2, 6, 20, 3, 64, 6, 64, 0, 64, 0);
  printf(bad ? "FAILED\n" : "OK\n");
  return bad;
}
