// conv_bf16_check.cpp -- host emulation of conv3x3_bf16.hip's data flow, built from the index functions the device code uses
// (stardist_amd/csrc/conv3x3_layout.h):
//   1. stage_elem_b deals every float4 element of the halo tile to exactly one (thread, slot), and the 8-byte LDS stores of any 16 or
//      32 consecutive lanes of a full block hit distinct banks;
//   2. the kernel's address rule -- per-tile offset (src_base) + per-thread offset (src_rel) for every element that lies inside the
//      image, on interior AND border tiles, full- and half-resolution sources -- addresses the pixel a direct index computes;
//   3. split3 reproduces a float to 2^-24 and the weight packer puts every (output channel, input channel, tap, plane) where bw_off reads it;
//   4. a whole layer through the LDS tile (btile_store_off / btile_off), the packed weights (bw_off) and the six bf16 products with f32
//      accumulation agrees with a float64 convolution to 1e-5 of the output scale.
// usage: conv_bf16_check   (exit code 0 = all checks pass)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../stardist_amd/csrc/conv3x3_layout.h"

using namespace sdconv;

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
static float bf(unsigned short v) { return u2f((unsigned)v << 16); }

static int check_assignment() {
  std::vector<int> seen(TILE_F4, 0);
  for (int e = 0; e < TILE_F4; ++e) {
    int ty, tx, q4;
    stage_elem_b(e, ty, tx, q4);
    if (ty < 0 || ty >= HALO_H || tx < 0 || tx >= HALO_W || q4 < 0 || q4 > 7) { printf("stage_elem_b(%d) out of range\n", e); return 1; }
    seen[(ty * HALO_W + tx) * 8 + q4]++;
  }
  for (int k = 0; k < TILE_F4; ++k) if (seen[k] != 1) { printf("halo element %d assigned %d times\n", k, seen[k]); return 1; }
  // bank conflicts of the plane stores (8 bytes = 2 banks of 4 bytes, 64 banks), full blocks only
  for (int n = 0; n < (TILE_F4 >> 8); ++n)
    for (int wave = 0; wave < 4; ++wave)
      for (int grp = 32; grp >= 16; grp >>= 1)
        for (int l0 = 0; l0 < 64; l0 += grp) {
          int bank[64] = {0};
          for (int l = l0; l < l0 + grp; ++l) {
            int ty, tx, q4;
            stage_elem_b(n * THREADS + wave * 64 + l, ty, tx, q4);
            const int b = (btile_store_off(ty, tx, 0, q4) / 4) & 63;
            if (bank[b]++ || bank[(b + 1) & 63]++) { printf("bank conflict: block %d wave %d lanes %d..%d\n", n, wave, l0, l0 + grp - 1); return 1; }
          }
        }
  return 0;
}

static int check_addresses() {
  const int H = 27, W = 70;                              // ragged: 4 x 3 tiles, the last row / column partial; even sizes not needed for sh = 0
  for (int sh = 0; sh < 2; ++sh) {
    const int He = sh ? 28 : H, We = sh ? 70 : W;         // an up-sampled axis needs an even output size
    const int hs = He >> sh, ws = We >> sh, stride = 40;
    const int tiles_x = (We + TW - 1) / TW, tiles_y = (He + TH - 1) / TH;
    for (int t = 0; t < tiles_x * tiles_y; ++t) {
      const int row = t / tiles_x, ty0 = row * TH - 1, tx0 = (t - row * tiles_x) * TW - 1;
      const long long off = ((long long)src_base(ty0, sh) * ws + src_base(tx0, sh)) * stride;      // tile_addr
      for (int e = 0; e < TILE_F4; ++e) {
        int ty, tx, q4;
        stage_elem_b(e, ty, tx, q4);
        const long long goff = ((long long)src_rel(ty, sh) * ws + src_rel(tx, sh)) * stride + q4 * 4;   // goff_init (floats)
        const int gy = ty0 + ty, gx = tx0 + tx;
        const bool inside = (unsigned)gy < (unsigned)He && (unsigned)gx < (unsigned)We;
        if (!inside) continue;                                                                          // reads the zero block
        const long long want = ((long long)(gy >> sh) * ws + (gx >> sh)) * stride + q4 * 4;
        if (off + goff != want || want < 0 || want >= (long long)hs * ws * stride) {
          printf("address rule: sh %d tile %d element %d: %lld + %lld != %lld\n", sh, t, e, off, goff, want);
          return 1;
        }
      }
    }
  }
  return 0;
}

static int check_split_and_pack() {
  for (int k = 0; k < 100000; ++k) {
    const float x = frand() * expf(frand() * 20.f);
    unsigned hi, mid, lo;
    split3(x, hi, mid, lo);
    const double r = (double)x - ((double)u2f(hi) + (double)u2f(mid) + (double)u2f(lo));
    if (fabs(r) > ldexp(fabs((double)x), -23)) { printf("split3(%g): remainder %g\n", x, r); return 1; }
    if ((hi | mid | lo) & 0xFFFFu) { printf("split3(%g): a term is not a bf16 value\n", x); return 1; }
  }
  const int c_in = 64, c_out = 64, kz = 3;
  std::vector<float> w((size_t)c_out * c_in * kz * 9);
  for (auto& v : w) v = frand();
  std::vector<unsigned short> packed(bpacked_bytes(c_in, c_out, kz) / 2);
  pack_weights_bf16(w.data(), c_in, c_out, kz, packed.data());
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
                  const size_t sub = (((size_t)g * (c_in / 32) * kz + u) * 3 + dy) * BWSUB_BYTES;
                  float sum = 0;
                  for (int p = 0; p < 3; ++p) sum += bf(packed[(sub + bw_off(dx, b, p, h, i)) / 2 + j]);
                  if (fabsf(sum - x) > ldexpf(fabsf(x), -22)) { printf("packed weight (%d,%d,%d,%d,%d): %g != %g\n", co, ci, z, dy, dx, sum, x); return 1; }
                }
  return 0;
}

// one 8 x 32 output tile of one 32-channel group through the kernel's LDS layouts: 2D layer, c_in = 64 (two units)
static int check_layer() {
  const int H = 16, W = 64, c_in = 64, c_out = 32;
  std::vector<float> x((size_t)H * W * c_in), w((size_t)c_out * c_in * 9);
  for (auto& v : x) v = frand();
  for (auto& v : w) v = frand() * 0.1f;
  std::vector<unsigned short> packed(bpacked_bytes(c_in, c_out, 1) / 2);
  pack_weights_bf16(w.data(), c_in, c_out, 1, packed.data());
  double worst = 0, scale = 0;
  for (int t = 0; t < 4; ++t) {
    const int ty0 = (t / 2) * TH - 1, tx0 = (t % 2) * TW - 1;
    std::vector<float> acc((size_t)TH * TW * 32, 0.f);
    for (int u = 0; u < c_in / 32; ++u) {
      std::vector<unsigned short> tile(BTILE_BYTES / 2, 0);
      for (int e = 0; e < TILE_F4; ++e) {
        int ty, tx, q4;
        stage_elem_b(e, ty, tx, q4);
        const int gy = ty0 + ty, gx = tx0 + tx;
        for (int k = 0; k < 4; ++k) {
          const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)gy * W + gx) * c_in + u * 32 + q4 * 4 + k] : 0.f;
          unsigned pl[3];
          split3(v, pl[0], pl[1], pl[2]);
          for (int p = 0; p < 3; ++p) tile[btile_store_off(ty, tx, p, q4) / 2 + k] = (unsigned short)(pl[p] >> 16);
        }
      }
      for (int dy = 0; dy < 3; ++dy) {
        const unsigned short* wsub = packed.data() + ((size_t)u * 3 + dy) * BWSUB_BYTES / 2;
        for (int row = 0; row < TH; ++row)
          for (int dx = 0; dx < 3; ++dx)
            for (int b = 0; b < 2; ++b) {
              constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // the kernel's six plane pairs
              for (int k = 0; k < 6; ++k)
                for (int m = 0; m < 32; ++m)            // pixel column (MFMA row)
                  for (int n = 0; n < 32; ++n) {        // output channel (MFMA column)
                    float s = 0;
                    for (int h = 0; h < 2; ++h)
                      for (int j = 0; j < 8; ++j)
                        s += bf(tile[btile_off(row + dy, m + dx, PA[k], b, h) / 2 + j]) * bf(wsub[bw_off(dx, b, PB[k], h, n) / 2 + j]);
                    acc[((size_t)row * TW + m) * 32 + n] += s;
                  }
            }
      }
    }
    for (int row = 0; row < TH; ++row)
      for (int m = 0; m < TW; ++m)
        for (int n = 0; n < 32; ++n) {
          const int y = ty0 + 1 + row, xx = tx0 + 1 + m;
          double ref = 0;
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int gy = y + dy, gx = xx + dx;
              if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
              for (int ci = 0; ci < c_in; ++ci) ref += (double)x[((size_t)gy * W + gx) * c_in + ci] * (double)w[((size_t)n * c_in + ci) * 9 + (dy + 1) * 3 + dx + 1];
            }
          worst = fmax(worst, fabs(ref - (double)acc[((size_t)row * TW + m) * 32 + n]));
          scale = fmax(scale, fabs(ref));
        }
  }
  if (!(worst <= 1e-5 * scale)) { printf("layer: max |error| %g at output scale %g\n", worst, scale); return 1; }
  return 0;
}

int main() {
  srand(1);
  if (check_assignment() || check_addresses() || check_split_and_pack() || check_layer()) return 1;
  printf("OK\n");
  return 0;
}
