// conv_f16_check.cpp -- host emulation of conv3x3_f16.hip's data flow, built from the index functions the device code uses
// (stardist_amd/csrc/conv3x3_layout.h):
//   1. with the two-plane tile (144 bytes per halo pixel) the 8-byte LDS stores of any 16 or 32 consecutive lanes of a full block hit
//      distinct banks, and the 16 lanes of every ds_read_b128 group of an A-operand read hit 16 distinct bank quadruples;
//   2. the kernel's offset rule -- tile base (src_base) + per-element offset derived from the halo coordinates (src_rel) -- addresses
//      the pixel a direct index computes, on interior and border tiles, full- and half-resolution sources;
//   3. split2_f16 reproduces a float to 2^-21 (2^-35 absolute below the normal fp16 range) and the packer puts every
//      (output channel, input channel, tap, plane) where hw_off reads it;
//   4. a whole layer through the LDS tile (htile_store_off / htile_off), the packed weights (hw_off) and the three fp16 products with
//      two accumulators agrees with a float64 convolution to 1e-5 of the output scale.
// usage: conv_f16_check   (exit code 0 = all checks pass)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../stardist_amd/csrc/conv3x3_layout.h"

using namespace sdconv;

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

static int check_banks() {
  // stores: ds_write_b64, 2 dwords per lane
  for (int n = 0; n < (TILE_F4 >> 8); ++n)
    for (int wave = 0; wave < 4; ++wave)
      for (int grp = 32; grp >= 16; grp >>= 1)
        for (int l0 = 0; l0 < 64; l0 += grp) {
          int bank[64] = {0};
          for (int l = l0; l < l0 + grp; ++l) {
            int ty, tx, q4;
            stage_elem_b(n * THREADS + wave * 64 + l, ty, tx, q4);
            if (q4 != ((wave * 64 + l) & 7)) { printf("q4 is not tid & 7\n"); return 1; }
            const int b = (htile_store_off(ty, tx, 0, q4) / 4) & 63;
            if (bank[b]++ || bank[(b + 1) & 63]++) { printf("store bank conflict: block %d wave %d lanes %d..%d\n", n, wave, l0, l0 + grp - 1); return 1; }
          }
        }
  // A-operand reads: ds_read_b128, lane groups of MI355X_MICROARCH.md (16 lanes each), lane = h * 32 + i reads pixel i + dx
  static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  for (int dx = 0; dx < 3; ++dx)
    for (int p = 0; p < 2; ++p)
      for (int b = 0; b < 2; ++b)
        for (int gi = 0; gi < 4; ++gi) {
          int quad[16] = {0};
          for (int k = 0; k < 16; ++k) {
            const int lane = groups[gi][k], i = lane & 31, h = lane >> 5;
            const int qd = (htile_off(3, i + dx, p, b, h) / 16) & 15;
            if (quad[qd]++) { printf("read bank conflict: dx %d plane %d block %d group %d\n", dx, p, b, gi); return 1; }
          }
        }
  return 0;
}

static int check_offsets() {
  const int H = 27, W = 70;
  for (int sh = 0; sh < 2; ++sh) {
    const int He = sh ? 28 : H, We = sh ? 70 : W;
    const int hs = He >> sh, ws = We >> sh, stride = 40;
    const int tiles_x = (We + TW - 1) / TW, tiles_y = (He + TH - 1) / TH;
    for (int t = 0; t < tiles_x * tiles_y; ++t) {
      const int row = t / tiles_x, ty0 = row * TH - 1, tx0 = (t - row * tiles_x) * TW - 1;
      const long long base = ((long long)src_base(ty0, sh) * ws + src_base(tx0, sh)) * stride * 4;      // bytes (tile_addr)
      for (int e = 0; e < TILE_F4; ++e) {
        int ty, tx, q4;
        stage_elem_b(e, ty, tx, q4);
        const unsigned row_bytes = (unsigned)ws * stride * 4, pix_bytes = stride * 4;
        const unsigned ry = (unsigned)(((ty - sh) >> sh) + sh), rx = (unsigned)(((tx - sh) >> sh) + sh);   // halo_off_one
        const long long off = (long long)(ry * row_bytes + (rx * pix_bytes + (unsigned)q4 * 16u));
        if ((int)ry != src_rel(ty, sh) || (int)rx != src_rel(tx, sh)) { printf("src_rel formula\n"); return 1; }
        const int gy = ty0 + ty, gx = tx0 + tx;
        if (!((unsigned)gy < (unsigned)He && (unsigned)gx < (unsigned)We)) continue;                     // offset beyond the range: zeros
        const long long want = (((long long)(gy >> sh) * ws + (gx >> sh)) * stride + q4 * 4) * 4;
        if (base + off != want || want < 0 || want >= (long long)hs * ws * stride * 4) {
          printf("offset rule: sh %d tile %d element %d: %lld + %lld != %lld\n", sh, t, e, base, off, want);
          return 1;
        }
      }
    }
  }
  return 0;
}

static int check_split_and_pack() {
  for (int k = 0; k < 200000; ++k) {
    const float x = frand() * expf(frand() * 12.f);
    if (fabsf(x) > 65504.f) continue;
    unsigned short hi, lo;
    split2_f16(x, hi, lo);
    const double r = (double)x - ((double)f16_value(hi) + (double)f16_value(lo) / 2048.0);
    const double bound = fmax(ldexp(fabs((double)x), -21), ldexp(1.0, -35));
    if (fabs(r) > bound) { printf("split2_f16(%g): remainder %g\n", x, r); return 1; }
  }
  const int c_in = 64, c_out = 64, kz = 3;
  std::vector<float> w((size_t)c_out * c_in * kz * 9);
  for (auto& v : w) v = frand();
  std::vector<unsigned short> packed(hpacked_bytes(c_in, c_out, kz) / 2);
  pack_weights_f16(w.data(), c_in, c_out, kz, packed.data());
  for (int g = 0; g < c_out / 32; ++g)
    for (int u = 0; u < (c_in / 32) * kz; ++u)
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          for (int b = 0; b < 2; ++b)
            for (int h = 0; h < 2; ++h)
              for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 8; ++j) {
                  const int c = u / kz, z = u - c * kz, co = g * 32 + i, ci = c * 32 + b * 16 + h * 8 + j;
                  const float x = w[(((size_t)co * c_in + ci) * kz + z) * 9 + dy * 3 + dx];
                  const size_t sub = (((size_t)g * (c_in / 32) * kz + u) * 3 + dy) * HWSUB_BYTES;
                  const float sum = f16_value(packed[(sub + hw_off(dx, b, 0, h, i)) / 2 + j]) + f16_value(packed[(sub + hw_off(dx, b, 1, h, i)) / 2 + j]) / 2048.f;
                  if (fabsf(sum - x) > ldexpf(fabsf(x), -20)) { printf("packed weight (%d,%d,%d,%d,%d): %g != %g\n", co, ci, z, dy, dx, sum, x); return 1; }
                }
  return 0;
}

static int check_layer() {
  const int H = 16, W = 64, c_in = 64, c_out = 32;
  std::vector<float> x((size_t)H * W * c_in), w((size_t)c_out * c_in * 9), bias(c_out);
  for (auto& v : x) v = frand() * 3.f;
  for (auto& v : w) v = frand() * 0.1f;
  for (auto& v : bias) v = frand();
  std::vector<unsigned short> packed(hpacked_bytes(c_in, c_out, 1) / 2);
  pack_weights_f16(w.data(), c_in, c_out, 1, packed.data());
  double worst = 0, scale = 0;
  for (int t = 0; t < 4; ++t) {
    const int ty0 = (t / 2) * TH - 1, tx0 = (t % 2) * TW - 1;
    std::vector<float> acc0((size_t)TH * TW * 32), acc1((size_t)TH * TW * 32, 0.f);
    for (size_t k = 0; k < acc0.size(); ++k) acc0[k] = bias[k & 31];
    for (int u = 0; u < c_in / 32; ++u) {
      std::vector<unsigned short> tile(HTILE_BYTES / 2, 0);
      for (int e = 0; e < TILE_F4; ++e) {
        int ty, tx, q4;
        stage_elem_b(e, ty, tx, q4);
        const int gy = ty0 + ty, gx = tx0 + tx;
        for (int k = 0; k < 4; ++k) {
          const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)gy * W + gx) * c_in + u * 32 + q4 * 4 + k] : 0.f;
          unsigned short pl[2];
          split2_f16(v, pl[0], pl[1]);
          for (int p = 0; p < 2; ++p) tile[htile_store_off(ty, tx, p, q4) / 2 + k] = pl[p];
        }
      }
      for (int dy = 0; dy < 3; ++dy) {
        const unsigned short* wsub = packed.data() + ((size_t)u * 3 + dy) * HWSUB_BYTES / 2;
        for (int row = 0; row < TH; ++row)
          for (int dx = 0; dx < 3; ++dx)
            for (int b = 0; b < 2; ++b)
              for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                  float s_hl = 0, s_lh = 0, s_hh = 0;
                  for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 8; ++j) {
                      const float ah = f16_value(tile[htile_off(row + dy, m + dx, 0, b, h) / 2 + j]), al = f16_value(tile[htile_off(row + dy, m + dx, 1, b, h) / 2 + j]);
                      const float bh = f16_value(wsub[hw_off(dx, b, 0, h, n) / 2 + j]), bl = f16_value(wsub[hw_off(dx, b, 1, h, n) / 2 + j]);
                      s_hl += ah * bl; s_lh += al * bh; s_hh += ah * bh;
                    }
                  acc1[((size_t)row * TW + m) * 32 + n] += s_hl;
                  acc1[((size_t)row * TW + m) * 32 + n] += s_lh;
                  acc0[((size_t)row * TW + m) * 32 + n] += s_hh;
                }
      }
    }
    for (int row = 0; row < TH; ++row)
      for (int m = 0; m < TW; ++m)
        for (int n = 0; n < 32; ++n) {
          const int y = ty0 + 1 + row, xx = tx0 + 1 + m;
          double ref = bias[n];
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int gy = y + dy, gx = xx + dx;
              if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
              for (int ci = 0; ci < c_in; ++ci) ref += (double)x[((size_t)gy * W + gx) * c_in + ci] * (double)w[((size_t)n * c_in + ci) * 9 + (dy + 1) * 3 + dx + 1];
            }
          const size_t k = ((size_t)row * TW + m) * 32 + n;
          const float got = acc0[k] + acc1[k] * 4.8828125e-4f;
          worst = fmax(worst, fabs(ref - (double)got));
          scale = fmax(scale, fabs(ref));
        }
  }
  if (!(worst <= 3e-6 * scale)) { printf("layer: max |error| %g at output scale %g\n", worst, scale); return 1; }
  printf("layer: max |error| / scale = %.3g\n", worst / scale);
  return 0;
}

int main() {
  srand(1);
  if (check_banks() || check_offsets() || check_split_and_pack() || check_layer()) return 1;
  printf("OK\n");
  return 0;
}
