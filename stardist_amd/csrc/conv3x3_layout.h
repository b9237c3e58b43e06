// conv3x3_layout.h -- index algebra of the hand-written 3x3 / 3x3x3 convolution (conv3x3.hip), shared between the device code,
// the host-side weight packer of the C ABI and the host emulation harness (tests/host/conv_check.cpp), so that the layout is
// written down exactly once.
//
// The reference's network (csbdeep unet_block as called from stardist/models/model2d.py:310-349, model3d.py:360-399) is a chain
// of 3x3(x3) 'same' convolutions, each followed by bias + ReLU, with nearest-neighbour up-sampling and channel concatenation on the
// way up.  Here one such layer is an implicit GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32, exact float32 = one fma
// chain per output):
//
//   output tile  = TH x TW = 8 x 32 pixels of one z plane per workgroup (4 waves x 2 rows of 32 pixels), 32*NT output channels
//   M (A rows)   = the 32 pixels of one tile row,  N (B columns) = 32 output channels,  K = taps x input channels
//   unit         = (32-channel input chunk c, z tap kz): the (TH+2) x (TW+2) halo tile of that chunk in plane z + kz - 1 and the
//                  matching 9-tap weight block, both staged in LDS per unit (a 3x3x3 convolution = 3 units per chunk, a 3x3 one 1).
//                  Tile pixel stride 36 floats: the 4 floats of padding make the 16 lanes of a ds_read_b128 group hit 16
//                  different bank quadruples.
//   weights      = packed so that lane (i, h) reads the four k-steps of one (tap, j) group as one float4
//   k order      = unit-major (chunk, then kz); inside a unit: column tap dx, channel group j, row tap dy, step e; lane half
//                  h = 0 / 1 feeds channel h*16 + j*4 + e (the MFMA's two k rows)
#pragma once
#include <stddef.h>

#if defined(__HIPCC__)
#define SDC_HD __host__ __device__ inline
#else
#define SDC_HD inline
#endif

namespace sdconv {

constexpr int TH = 8, TW = 32;                    // output tile (rows, columns)
constexpr int HALO_H = TH + 2, HALO_W = TW + 2;
constexpr int CHUNK = 32;                         // input channels per unit
constexpr int PIX_STRIDE = 36;                    // floats per pixel of the LDS tile
constexpr int TILE_FLOATS = HALO_H * HALO_W * PIX_STRIDE;
constexpr int TILE_F4 = HALO_H * HALO_W * (CHUNK / 4);       // float4 elements of one staged halo tile
constexpr int THREADS = 256;
constexpr int PRE_F4 = (TILE_F4 + THREADS - 1) / THREADS;    // float4 registers per thread for the prefetch of the next tile
constexpr int MAX_CHUNKS = 16;                    // up to 512 input channels (depth-4 U-Net: 256 up-sampled + 256 skip)

// 32-wide output-channel tiles a workgroup computes at once (registers: 2*NT accumulator tiles per wave, 9*NT float4 of weight
// prefetch per thread; LDS: NT*36 KiB of weights).  NT = 2 halves the input staging per output channel but its prefetch registers
// spill (256 arch VGPRs), so every layer runs as c_out/32 groups of 32 output channels; the groups of one tile are co-scheduled
// on one XCD (conv3x3.hip) so that the re-read of the input tile is an L2 hit.
SDC_HD int nt_for(int c_out) { (void)c_out; return 1; }
// packed weight floats of one (group, unit): 9 taps x 4 j x 2 h x (32 NT) output channels x 4 e
SDC_HD int wunit_floats(int nt) { return 9 * 4 * 2 * 32 * nt * 4; }
SDC_HD size_t packed_floats(int c_in, int c_out, int kz) { return (size_t)c_out * c_in * 9 * kz; }
// channel (within the chunk) that k-step (j, e) of lane half h multiplies
SDC_HD int chan_of(int j, int h, int e) { return h * 16 + j * 4 + e; }
// position of W[group g, unit u, tap, j, h, n (output channel within the group), e] in the packed array
SDC_HD size_t wpack_index(int g, int u, int tap, int j, int h, int n, int e, int n_units, int nt) {
  return ((((((size_t)g * n_units + u) * 9 + tap) * 4 + j) * 2 + h) * (32 * nt) + n) * 4 + e;
}
// float offset, inside one (group, unit) block in LDS, of the float4 lane (i, h) reads as B operand for (tap, j, output tile ct)
SDC_HD int wl_off(int tap, int j, int h, int ct, int i, int nt) { return (((tap * 4 + j) * 2 + h) * (32 * nt) + ct * 32 + i) * 4; }
// float offset in the LDS tile of (halo row ty, halo column tx, channel ch)
SDC_HD int tile_off(int ty, int tx, int ch) { return (ty * HALO_W + tx) * PIX_STRIDE + ch; }
// float offset of the float4 lane (i, h) reads as A operand: output row `row` of the tile (0..TH-1), tap (dy, dx), group j
SDC_HD int a_off(int row, int dy, int dx, int j, int i, int h) { return tile_off(row + dy, i + dx, h * 16 + j * 4); }
// accumulator register r of lane half h holds tile column ...
SDC_HD int acc_col(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// staging: float4 element e (0..TILE_F4-1) of a halo tile -> halo pixel and channel quad
SDC_HD void stage_elem(int e, int& ty, int& tx, int& q4) {
  const int pix = e >> 3;
  q4 = e & 7;
  ty = pix / HALO_W;
  tx = pix - ty * HALO_W;
}

// The split-bf16 kernel's assignment (conv3x3_bf16.hip): its LDS tile has a pixel stride of 52 dwords and a thread writes 8 bytes per
// plane, so 8 lanes cover 16 banks of one pixel; pixels 4 apart start 16 banks apart (4 * 52 mod 64), adjacent ones overlap in 4 banks.
// Inside each full block of 256 elements (32 pixels) the 8-lane groups of a wave therefore take pixels 4 apart (wave w: pixels
// w, w + 4, ..., w + 28 of the block): every 16- or 32-lane group of a ds_write_b64 hits distinct banks.  The global loads are
// unaffected (8 lanes still read one pixel's 128 contiguous bytes).  The last, partial block keeps the plain order.
SDC_HD void stage_elem_b(int e, int& ty, int& tx, int& q4) {
  q4 = e & 7;
  const int n = e >> 8;
  const int pix = n < (TILE_F4 >> 8) ? 32 * n + ((e >> 3) & 7) * 4 + ((e >> 6) & 3) : e >> 3;
  ty = pix / HALO_W;
  tx = pix - ty * HALO_W;
}

// Halo coordinates of a tile start at t0 = 8k - 1 (rows) / 32m - 1 (columns): always odd.  For a half-resolution source (sh = 1)
// halo index t maps to source index (t0 + t) >> 1 = src_base(t0, 1) + src_rel(t, 1), both parts non-negative inside the image;
// for a full-resolution source (sh = 0) to t0 + t.  The kernel's fast path adds a wave-uniform base built from src_base to
// per-thread constants built from src_rel.
SDC_HD int src_rel(int t, int sh) { return sh ? ((t - 1) >> 1) + 1 : t; }
SDC_HD int src_base(int t0, int sh) { return sh ? ((t0 + 1) >> 1) - 1 : t0; }

// Pack a PyTorch / Keras-converted kernel w[c_out][c_in][kz][3][3] (float32; kz = 1: Conv2D, 3: Conv3D) for the device kernel.
// c_in, c_out multiples of 32.  `out` holds packed_floats(c_in, c_out, kz) floats.
inline void pack_weights(const float* w, int c_in, int c_out, int kz, float* out) {
  const int nt = nt_for(c_out), n_chunks = c_in / CHUNK, groups = c_out / (32 * nt), n_units = n_chunks * kz;
  for (int g = 0; g < groups; ++g)
    for (int c = 0; c < n_chunks; ++c)
      for (int z = 0; z < kz; ++z)
        for (int tap = 0; tap < 9; ++tap)
          for (int j = 0; j < 4; ++j)
            for (int h = 0; h < 2; ++h)
              for (int n = 0; n < 32 * nt; ++n)
                for (int e = 0; e < 4; ++e) {
                  const int co = g * 32 * nt + n, ci = c * CHUNK + chan_of(j, h, e);
                  out[wpack_index(g, c * kz + z, tap, j, h, n, e, n_units, nt)] = w[(((size_t)co * c_in + ci) * kz + z) * 9 + tap];
                }
}

// ---- split-bf16 variant (conv3x3_bf16.hip) -------------------------------------------------------------------------------------
// Every f32 operand x is written as hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest
// even; the two remainders are exact in f32), and a product a*b is evaluated as the six bf16 x bf16 products hi*hi + hi*mid + mid*hi +
// hi*lo + lo*hi + mid*mid (each exact in f32, accumulated in f32 by v_mfma_f32_32x32x16_bf16); the three dropped terms are below
// 2^-24 of the product.  bf16 MFMA runs at 16x the f32-MFMA rate, so six of them are 2.7x faster than one f32 MFMA.
//   LDS tile   : 208 bytes per halo pixel = 3 planes (hi, mid, lo) x 32 channels x 2 bytes + 16 bytes of padding (52 dwords:
//                the 16 lanes of a ds_read_b128 group hit 16 distinct bank quadruples)
//   k order    : a 32x32x16 MFMA takes 8 values per lane; lane half h of 16-channel block b holds channels b*16 + h*8 + 0..7
//                for BOTH operands (which k index the hardware gives a slot is irrelevant as long as A and B agree)
//   sub-unit   : the weights of one (unit, row tap dy) = 3 dx x 2 blocks x 3 planes x 2 h x 32 output channels x 16 bytes = 18 KiB
constexpr int BPIX = 208;
constexpr int BTILE_BYTES = HALO_H * HALO_W * BPIX;
constexpr int BWSUB_BYTES = 3 * 2 * 3 * 2 * 32 * 16;
// byte offset of the 16 bytes (8 channels) lane half h reads as A operand: halo pixel (ty, tx), plane p, 16-channel block b
SDC_HD int btile_off(int ty, int tx, int p, int b, int h) { return (ty * HALO_W + tx) * BPIX + p * 64 + b * 32 + h * 16; }
// byte offset where the 4 channels q4*4 .. q4*4+3 of plane p of halo pixel (ty, tx) are stored (8 bytes)
SDC_HD int btile_store_off(int ty, int tx, int p, int q4) { return (ty * HALO_W + tx) * BPIX + p * 64 + q4 * 8; }
// byte offset, inside a sub-unit block, of the 16 bytes lane (i = output channel, h) reads as B operand
SDC_HD int bw_off(int dx, int b, int p, int h, int i) { return ((((dx * 2 + b) * 3 + p) * 2 + h) * 32 + i) * 16; }
SDC_HD size_t bpacked_bytes(int c_in, int c_out, int kz) { return (size_t)(c_out / 32) * (c_in / CHUNK) * kz * 3 * BWSUB_BYTES; }

SDC_HD unsigned f2u(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
SDC_HD float u2f(unsigned u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
// bf16(f), round to nearest even, as f32 bits (low 16 bits zero)
SDC_HD unsigned bf16_bits(float f) { unsigned u = f2u(f); u += 0x7FFFu + ((u >> 16) & 1u); return u & 0xFFFF0000u; }
SDC_HD void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = bf16_bits(x);
  const float r = x - u2f(hi);
  mid = bf16_bits(r);
  lo = bf16_bits(r - u2f(mid));
}

// Pack w[c_out][c_in][kz][3][3] (float32) into the three bf16 planes of the device layout: [group][unit][dy][sub-unit block].
inline void pack_weights_bf16(const float* w, int c_in, int c_out, int kz, unsigned short* out) {
  const int n_chunks = c_in / CHUNK, groups = c_out / 32, n_units = n_chunks * kz;
  for (int g = 0; g < groups; ++g)
    for (int c = 0; c < n_chunks; ++c)
      for (int z = 0; z < kz; ++z)
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx)
            for (int b = 0; b < 2; ++b)
              for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 32; ++i)
                  for (int j = 0; j < 8; ++j) {
                    const int co = g * 32 + i, ci = c * CHUNK + b * 16 + h * 8 + j;
                    unsigned pl[3];
                    split3(w[(((size_t)co * c_in + ci) * kz + z) * 9 + dy * 3 + dx], pl[0], pl[1], pl[2]);
                    const size_t sub = (((size_t)g * n_units + c * kz + z) * 3 + dy) * BWSUB_BYTES;
                    for (int p = 0; p < 3; ++p) out[(sub + bw_off(dx, b, p, h, i)) / 2 + j] = (unsigned short)(pl[p] >> 16);
                  }
}

// ---- split-fp16 variant (conv3x3_f16.hip) --------------------------------------------------------------------------------------
// Every f32 operand x is written as hi + lo' * 2^-11 with hi = fp16(x) and lo' = fp16((x - hi) * 2^11) (round to nearest even; the
// remainder x - hi is exact in f32, and scaling it by 2^11 keeps lo' in the normal fp16 range whenever hi is).  fp16 carries 11
// significant bits, so hi + lo' * 2^-11 reproduces x to 2^-22, and a product a*b is evaluated as
//     hi_a * hi_b  +  2^-11 * (hi_a * lo'_b + lo'_a * hi_b)
// -- three fp16 x fp16 products (each exact in f32) on v_mfma_f32_32x32x16_f16 with two f32 accumulators (the cross terms are summed
// in their own accumulator and scaled once, in the epilogue); the dropped lo * lo term is below 2^-22 of the product.  Half the
// matrix-core work of the six-product bf16 form at ~2.5x its split error, still 5x below the f32 accumulation error of any f32 kernel
// (tools/split_study.py: 2.3e-7 vs 9.5e-8 on the 2D network, where f32 accumulation itself is at 1.2e-6).
// Range: fp16 overflows above 65504.  Weights are checked when they are packed; the kernel raises a device flag when an activation
// exceeds the range (models/unet.py then re-evaluates the network with the bf16 form, loudly).  Values below 2^-14 have a subnormal
// hi, i.e. an ABSOLUTE error floor of 2^-36 after the lo' term -- harmless.
//   LDS tile   : 144 bytes per halo pixel = 2 planes (hi, lo') x 32 channels x 2 bytes + 16 bytes of padding (36 dwords: the 16 lanes
//                of a ds_read_b128 group hit 16 distinct bank quadruples)
//   sub-unit   : the weights of one (unit, row tap dy) = 3 dx x 2 blocks x 2 planes x 2 h x 32 output channels x 16 bytes = 12 KiB
constexpr int HPIX = 144;
constexpr int HTILE_BYTES = HALO_H * HALO_W * HPIX;
constexpr int HWSUB_BYTES = 3 * 2 * 2 * 2 * 32 * 16;
SDC_HD int htile_off(int ty, int tx, int p, int b, int h) { return (ty * HALO_W + tx) * HPIX + p * 64 + b * 32 + h * 16; }
SDC_HD int htile_store_off(int ty, int tx, int p, int q4) { return (ty * HALO_W + tx) * HPIX + p * 64 + q4 * 8; }
SDC_HD int hw_off(int dx, int b, int p, int h, int i) { return ((((dx * 2 + b) * 2 + p) * 2 + h) * 32 + i) * 16; }
SDC_HD size_t hpacked_bytes(int c_in, int c_out, int kz) { return (size_t)(c_out / 32) * (c_in / CHUNK) * kz * 3 * HWSUB_BYTES; }

// fp16(f), round to nearest even, subnormals kept, overflow -> infinity: the bits (what v_cvt_pk_f16_f32 / v_cvt_f16_f32 produce)
SDC_HD unsigned short f16_bits(float f) {
  const unsigned u = f2u(f), sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) return (unsigned short)(sign | 0x7C00u | ((a > 0x7F800000u) ? 0x200u : 0u));   // inf / nan
  if (a >= 0x477FF000u) return (unsigned short)(sign | 0x7C00u);                                       // rounds to >= 2^16: infinity
  if (a < 0x33000001u) return (unsigned short)sign;                                                     // <= 2^-25: rounds to zero
  int e = (int)(a >> 23) - 127;
  unsigned m = (a & 0x7FFFFFu) | 0x800000u;                    // 24-bit significand
  int shift = e >= -14 ? 13 : 13 + (-14 - e);                   // bits dropped (subnormal results drop more)
  const unsigned keep = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  unsigned r = keep + ((rem > half || (rem == half && (keep & 1u))) ? 1u : 0u);
  // normal: exponent field e + 15 with the implicit bit inside r (r in [2^10, 2^11]); subnormal: exponent field 0
  const unsigned bits = e >= -14 ? ((unsigned)(e + 14) << 10) + r : r;      // (a carry out of r bumps the exponent by itself)
  return (unsigned short)(sign | bits);
}
SDC_HD float f16_value(unsigned short h) {
  const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
  if (e == 31u) return u2f(sign | 0x7F800000u | (m << 13));
  if (e == 0u) {                                                // zero / subnormal: m * 2^-24
    float v = (float)m * 5.9604644775390625e-08f;
    return (sign ? -v : v);
  }
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
SDC_HD void split2_f16(float x, unsigned short& hi, unsigned short& lo) {
  hi = f16_bits(x);
  lo = f16_bits((x - f16_value(hi)) * 2048.f);
}

// Pack w[c_out][c_in][kz][3][3] (float32) into the two fp16 planes of the device layout: [group][unit][dy][sub-unit block].
// Returns the largest |w| (the caller refuses weights beyond the fp16 range).
inline float pack_weights_f16(const float* w, int c_in, int c_out, int kz, unsigned short* out) {
  const int n_chunks = c_in / CHUNK, groups = c_out / 32, n_units = n_chunks * kz;
  float wmax = 0.f;
  for (int g = 0; g < groups; ++g)
    for (int c = 0; c < n_chunks; ++c)
      for (int z = 0; z < kz; ++z)
        for (int dy = 0; dy < 3; ++dy)
          for (int dx = 0; dx < 3; ++dx)
            for (int b = 0; b < 2; ++b)
              for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 32; ++i)
                  for (int j = 0; j < 8; ++j) {
                    const int co = g * 32 + i, ci = c * CHUNK + b * 16 + h * 8 + j;
                    const float x = w[(((size_t)co * c_in + ci) * kz + z) * 9 + dy * 3 + dx];
                    const float ax = x < 0 ? -x : x;
                    if (!(ax <= wmax)) wmax = ax;               // (a NaN ends up in wmax)
                    unsigned short pl[2];
                    split2_f16(x, pl[0], pl[1]);
                    const size_t sub = (((size_t)g * n_units + c * kz + z) * 3 + dy) * HWSUB_BYTES;
                    for (int p = 0; p < 2; ++p) out[(sub + hw_off(dx, b, p, h, i)) / 2 + j] = pl[p];
                  }
  return wmax;
}

}  // namespace sdconv
