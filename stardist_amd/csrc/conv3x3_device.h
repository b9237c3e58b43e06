// conv3x3_device.h -- device-side pieces shared by the two convolution kernels (conv3x3.hip: exact f32 MFMA; conv3x3_bf16.hip:
// split-bf16 MFMA): the launch descriptor, the workgroup -> (output-channel group, tile slot) map, and the halo fetch.
#pragma once
#include <hip/hip_runtime.h>

#include "conv3x3_layout.h"

namespace sdconvdev {

using namespace sdconv;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));      // native 16-byte vector: plain loads/stores in any address space

struct Src {
  const float* p;      // channels-last [D >> shz][H >> shy][W >> shx][stride]
  int stride;          // floats per pixel
  int shz, shy, shx;   // 1: the source is half resolution along that axis (nearest-neighbour up-sampling by 2)
};

struct Params {
  Src kind[2];                         // the (at most two) source tensors
  int chunk_kind[MAX_CHUNKS];          // 32-channel chunk c comes from kind[chunk_kind[c]] ...
  int chunk_choff[MAX_CHUNKS];         // ... starting at this channel
  const float* zero;                   // 16 bytes of zeros (tail of the packed weights): where out-of-image halo elements are read from
  int D, H, W;
  int kz;              // z taps: 1 (2D) or 3
  int n_units;         // chunks * kz
  const float* wp;     // packed weights [groups][n_units][WUNIT]
  const float* bias;
  float* out;          // [D][H][W][c_out]
  int c_out, act;
  const float* res;    // optional residual [D][H][W][res_stride]: out = act(conv + bias + res) -- the Add + Activation that closes a
  int res_stride;      // csbdeep resnet_block, folded into the epilogue (nullptr: none)
  int tiles_x, tiles_plane, n_tiles, groups;
};

// workgroup -> (output-channel group g, tile slot q): consecutive workgroups go round-robin over the 8 XCDs, so the `groups`
// workgroups b, b+8, b+16, ... (same XCD, same L2) take the same tile sequence and differ in g
__device__ __forceinline__ void wg_slot(const Params& P, int& g, int& q, int& Q) {
  const int b = blockIdx.x, span = 8 * P.groups, blk = b / span, rem = b - blk * span;
  if ((blk + 1) * span <= (int)gridDim.x) { g = rem >> 3; q = blk * 8 + (rem & 7); }
  else { const int tail = gridDim.x - blk * span, per = tail / P.groups; g = rem / per; q = blk * 8 + rem % per; }   // last partial span
  Q = gridDim.x / P.groups;
}

// Per-thread fetch constants, computed once per kernel: for tiles whose halo lies inside the image, the byte offsets of this thread's
// PRE_F4 float4 elements from the halo's first source pixel, per source tensor (a half-resolution source maps halo row ty to source
// row ((ty - 1) >> 1) + 1 relative to the row of halo row 0, because a tile's first halo row/column is odd: 8k - 1 / 32m - 1).
__device__ __forceinline__ void goff_init(const Params& P, unsigned (&goff)[2][PRE_F4], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    int e = tid + n * THREADS;
    e = e < TILE_F4 ? e : TILE_F4 - 1;
    int ty, tx, q4;
    stage_elem(e, ty, tx, q4);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const Src S = P.kind[k];
      const int ry = src_rel(ty, S.shy), rx = src_rel(tx, S.shx);
      goff[k][n] = (unsigned)(((ry * (P.W >> S.shx) + rx) * S.stride + q4 * 4) * 4);
    }
  }
}

// Halo tile of unit u of output tile t -> registers (16 bytes per (thread, n)).
// Only ADDRESSES differ between tiles: inside the image (the rule) an element's address is a wave-uniform base + the precomputed
// per-thread offset; on border tiles it is computed per element, and elements outside the volume (the zero padding of 'same')
// point at a 16-byte block of zeros.  The loads themselves are unconditional and issue back to back after the addresses are
// known, so nothing has to be selected, merged or waited for before the matrix cores start on the current unit.
__device__ __forceinline__ void halo_fetch(const Params& P, const unsigned (&goff)[2][PRE_F4], int t, int u, v4f (&pre)[PRE_F4], int tid) {
  const int c = u / P.kz, dz = P.kz == 3 ? u - c * 3 - 1 : 0;
  const int k = P.chunk_kind[c];
  const Src S = P.kind[k];
  const float* sp = S.p + P.chunk_choff[c];
  const int tz = t / P.tiles_plane, tr = t - tz * P.tiles_plane;
  const int ty0 = (tr / P.tiles_x) * TH - 1, tx0 = (tr % P.tiles_x) * TW - 1;
  const int z = tz + dz;
  const bool zin = z >= 0 && z < P.D;
  const int ws = P.W >> S.shx, hs = P.H >> S.shy;
  const float* plane = sp + (size_t)(min(max(z, 0), P.D - 1) >> S.shz) * hs * ws * S.stride;
  typedef const __attribute__((address_space(1))) char* gptr;      // explicitly global: the asm fence below hides the provenance
  gptr addr[PRE_F4];
  if (zin && ty0 >= 0 && ty0 + HALO_H <= P.H && tx0 >= 0 && tx0 + HALO_W <= P.W) {
    const int by = src_base(ty0, S.shy), bx = src_base(tx0, S.shx);
    gptr base = (gptr)(plane + ((size_t)by * ws + bx) * S.stride);
#pragma unroll
    for (int n = 0; n < PRE_F4; ++n) addr[n] = base + (k ? goff[1][n] : goff[0][n]);
  } else {
#pragma unroll
    for (int n = 0; n < PRE_F4; ++n) {
      int e = tid + n * THREADS;
      e = e < TILE_F4 ? e : TILE_F4 - 1;
      int ty, tx, q4;
      stage_elem(e, ty, tx, q4);
      const int gy = ty0 + ty, gx = tx0 + tx;
      const bool inside = zin && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
      const int cy = min(max(gy, 0), P.H - 1) >> S.shy, cx = min(max(gx, 0), P.W - 1) >> S.shx;
      addr[n] = inside ? (gptr)(plane + ((size_t)cy * ws + cx) * S.stride + q4 * 4) : (gptr)P.zero;
    }
  }
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) asm volatile("" : "+v"(addr[n]));        // addresses are final here: the loads below stay below
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) pre[n] = *(const __attribute__((address_space(1))) v4f*)addr[n];
}

// Epilogue shared by both kernels: activation, then a transpose through LDS so that a lane stores 16 bytes (4 channels of one
// pixel) instead of 4: 8 global_store_dwordx4 per wave instead of 32 global_store_dword.  An accumulator register holds ONE
// channel (i) of 16 pixels (register r: tile column (r & 3) + 8 (r >> 2) + 4 h), channels-last memory wants 32 channels of one
// pixel together.  `scr` = this wave's eight 1-KiB scratch chunks, chunk n at scr + n * chunk_stride floats; the caller guarantees
// that no other wave touches them (see the kernels).
template <int chunk_stride>
__device__ __forceinline__ void tile_to_scratch(const Params& P, const f32x16 (&acc)[2], float* scr, int lane) {
  const int i = lane & 31, h = lane >> 5;
  const bool relu = P.act == 1 && !P.res;           // with a residual the activation follows the addition (scratch_to_global)
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pp = p * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // pixel of the wave's 64 (two rows of 32)
      const float v = relu ? fmaxf(acc[p][r], 0.f) : acc[p][r];
      scr[(pp >> 3) * chunk_stride + (pp & 7) * 32 + i] = v;
    }
}
template <int chunk_stride>
__device__ __forceinline__ void scratch_to_global(const Params& P, const float* scr, int g, int t, int wave, int lane) {
  const int tz = t / P.tiles_plane, tr = t - tz * P.tiles_plane;
  const int y0 = (tr / P.tiles_x) * TH + wave * 2, x0 = (tr % P.tiles_x) * TW;
  const bool xfull = x0 + TW <= P.W;
  const int px = lane >> 3, c4 = lane & 7;                      // this lane's pixel within a chunk's 8, its channel quad
  v4f vv[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) vv[n] = *(const v4f*)(scr + n * chunk_stride + lane * 4);
#pragma unroll
  for (int n = 0; n < 8; ++n) asm volatile("" : "+v"(vv[n]));      // all eight reads in flight before the first store
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int y = y0 + (n >> 2), x = x0 + (n & 3) * 8 + px;
    if (y < P.H && (xfull || x < P.W)) {
      const size_t pix = ((size_t)tz * P.H + y) * P.W + x;
      v4f o = vv[n];
      if (P.res) {
        o += *(const v4f*)(P.res + pix * P.res_stride + g * 32 + c4 * 4);
        if (P.act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      }
      *(v4f*)(P.out + pix * P.c_out + g * 32 + c4 * 4) = o;
    }
  }
}
template <int chunk_stride>
__device__ __forceinline__ void store_tile(const Params& P, const f32x16 (&acc)[2], float* scr, int g, int t, int wave, int lane) {
  tile_to_scratch<chunk_stride>(P, acc, scr, lane);
  scratch_to_global<chunk_stride>(P, scr, g, t, wave, lane);
}

}  // namespace sdconvdev
