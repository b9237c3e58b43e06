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
  long long plane;     // floats per z plane of the source: (H >> shy) * (W >> shx) * stride
  int stride;          // floats per pixel
  int shz, shy, shx;   // 1: the source is half resolution along that axis (nearest-neighbour up-sampling by 2)
};
// up: bit 0 x, bit 1 y, bit 2 z; H, W: the OUTPUT's plane size
inline Src make_src(const float* p, int stride, int up, int H, int W) {
  const int shz = (up >> 2) & 1, shy = (up >> 1) & 1, shx = up & 1;
  return Src{p, (long long)(H >> shy) * (W >> shx) * stride, stride, shz, shy, shx};
}

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
  int n_chunks0;       // chunks of kind[0] (the first c0 / 32 chunks; the others come from kind[1]): chunk_kind / chunk_choff in closed form
  int* flag;           // conv3x3_f16.hip: OR-ed with 1 when an activation is outside the fp16 range (nullptr: not reported)
  const float* dotw;   // conv3x3_f16.hip, fused one-channel head (the probability head behind the features layer): weights [c_out] ...
  float* dotp;         // ... and the partial dot products [groups][D * H * W]: dotp[g][pixel] = sum over the 32 channels of group g of
                       // out[pixel][c] * dotw[c], taken while the tile is in registers (nullptr: none)
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
template <bool BF = false>      // BF: the split-bf16 kernel's element order (stage_elem_b)
__device__ __forceinline__ void goff_init(const Params& P, unsigned (&goff)[2][PRE_F4], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    int e = tid + n * THREADS;
    e = e < TILE_F4 ? e : TILE_F4 - 1;
    int ty, tx, q4;
    if (BF) stage_elem_b(e, ty, tx, q4); else stage_elem(e, ty, tx, q4);
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

// ---- the same fetch with the per-tile part of the address arithmetic hoisted out of the unit loop (conv3x3_bf16.hip) ----------------
// A unit's halo differs from the previous unit's only in the source tensor / channel offset / z plane: wave-uniform terms.  The
// tile's own terms (the two integer divisions that turn a tile number into coordinates, the source offsets of its first halo pixel
// per source tensor, whether the halo lies inside the image) are computed once per tile.  On border tiles the same per-thread
// offsets apply to every element that lies inside the image (the base may then point in front of the image: it is never dereferenced
// for elements outside, which read the 16 zero bytes instead), so a border element costs two compares and a select.
struct TileAddr {
  int tz, ty0, tx0;
  bool interior;
  long long off[2];      // floats from a source plane's first pixel to the source pixel of halo pixel (0, 0), per source tensor
};
__device__ __forceinline__ void tile_addr(const Params& P, int t, TileAddr& T) {
  T.tz = t / P.tiles_plane;
  const int tr = t - T.tz * P.tiles_plane, row = tr / P.tiles_x;
  T.ty0 = row * TH - 1;
  T.tx0 = (tr - row * P.tiles_x) * TW - 1;
  T.interior = T.ty0 >= 0 && T.ty0 + HALO_H <= P.H && T.tx0 >= 0 && T.tx0 + HALO_W <= P.W;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const Src S = P.kind[k];
    T.off[k] = ((long long)src_base(T.ty0, S.shy) * (P.W >> S.shx) + src_base(T.tx0, S.shx)) * S.stride;
  }
}
// A persistent workgroup walks the tiles q, q + Q, q + 2Q, ...: the two integer divisions of tile_addr (and their ~150 vector
// instructions: 7-12 % of a unit in the round-5 phase profile) are paid once per kernel; afterwards the position advances by the
// step's digits (plane, tile row, tile column) with carries -- a handful of scalar instructions.
struct TileWalk {
  int tz, row, col;          // position of the tile whose address record is current
  int sz, srow, scol;        // digits of the step Q
  int tiles_y;
};
__device__ __forceinline__ void tile_at(const Params& P, const TileWalk& w, TileAddr& T) {
  T.tz = w.tz;
  T.ty0 = w.row * TH - 1;
  T.tx0 = w.col * TW - 1;
  T.interior = T.ty0 >= 0 && T.ty0 + HALO_H <= P.H && T.tx0 >= 0 && T.tx0 + HALO_W <= P.W;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const Src S = P.kind[k];
    T.off[k] = ((long long)src_base(T.ty0, S.shy) * (P.W >> S.shx) + src_base(T.tx0, S.shx)) * S.stride;
  }
}
__device__ __forceinline__ void walk_init(const Params& P, int q, int Q, TileWalk& w) {
  w.tiles_y = P.tiles_plane / P.tiles_x;
  w.tz = q / P.tiles_plane;
  int tr = q - w.tz * P.tiles_plane;
  w.row = tr / P.tiles_x;
  w.col = tr - w.row * P.tiles_x;
  w.sz = Q / P.tiles_plane;
  tr = Q - w.sz * P.tiles_plane;
  w.srow = tr / P.tiles_x;
  w.scol = tr - w.srow * P.tiles_x;
  w.tz = __builtin_amdgcn_readfirstlane(w.tz); w.row = __builtin_amdgcn_readfirstlane(w.row); w.col = __builtin_amdgcn_readfirstlane(w.col);
  w.sz = __builtin_amdgcn_readfirstlane(w.sz); w.srow = __builtin_amdgcn_readfirstlane(w.srow); w.scol = __builtin_amdgcn_readfirstlane(w.scol);
  w.tiles_y = __builtin_amdgcn_readfirstlane(w.tiles_y);
}
__device__ __forceinline__ void walk_step(const Params& P, TileWalk& w) {
  w.col += w.scol;
  int c = w.col >= P.tiles_x;
  w.col -= c ? P.tiles_x : 0;
  w.row += w.srow + c;
  c = w.row >= w.tiles_y;
  w.row -= c ? w.tiles_y : 0;
  w.tz += w.sz + c;
}
// field-wise choice (a reference picked at run time would put both records into scratch memory)
__device__ __forceinline__ TileAddr tile_select(bool second, const TileAddr& a, const TileAddr& b) {
  TileAddr T;
  T.tz = second ? b.tz : a.tz; T.ty0 = second ? b.ty0 : a.ty0; T.tx0 = second ? b.tx0 : a.tx0; T.interior = second ? b.interior : a.interior;
  T.off[0] = second ? b.off[0] : a.off[0]; T.off[1] = second ? b.off[1] : a.off[1];
  return T;
}
// halo coordinates of this thread's elements: ty | tx << 8
template <bool BF = false>
__device__ __forceinline__ void tyx_init(unsigned (&tyx)[PRE_F4], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    int e = tid + n * THREADS;
    e = e < TILE_F4 ? e : TILE_F4 - 1;
    int ty, tx, q4;
    if (BF) stage_elem_b(e, ty, tx, q4); else stage_elem(e, ty, tx, q4);
    tyx[n] = (unsigned)ty | ((unsigned)tx << 8);
  }
}
// The same in three steps, so that the arithmetic can be issued in the shadow of the matrix instructions of the previous sub-unit
// (conv3x3_bf16.hip): the wave-uniform part, one element's address (branch-free: inside the image or the zero block), the loads.
typedef const __attribute__((address_space(1))) char* halo_gptr;
struct HaloBase {
  halo_gptr base;
  int k;                 // source tensor
  bool zin;
};
__device__ __forceinline__ HaloBase halo_base(const Params& P, const TileAddr& T, int u) {
  const int c = P.kz == 3 ? (u * 21846) >> 16 : u;           // u / 3 (exact for u < 4096)
  const int dz = P.kz == 3 ? u - c * 3 - 1 : 0;
  HaloBase B;
  B.k = P.chunk_kind[c];
  const Src S = P.kind[B.k];
  const int z = T.tz + dz;
  B.zin = z >= 0 && z < P.D;
  B.base = (halo_gptr)(S.p + P.chunk_choff[c] + (long long)(min(max(z, 0), P.D - 1) >> S.shz) * S.plane + (B.k ? T.off[1] : T.off[0]));
  return B;
}
// everything halo_addr_one needs, in scalar registers BEFORE the matrix instructions start: a scalar load from the kernel arguments
// inside that sequence would wait on lgkmcnt(0), i.e. for every LDS operand read in flight
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {      // a wave-uniform value the compiler may hold in VGPRs
  return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
         ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
}
struct HaloScalars {
  halo_gptr base, zero;
  unsigned kmask;        // all ones: second source tensor
  int ty0, tx0, H, W;    // zin folded in: H = 0 for a z plane outside the volume (no element is inside)
};
__device__ __forceinline__ HaloScalars halo_scalars(const Params& P, const HaloBase& B, const TileAddr& T) {
  HaloScalars s;
  s.base = (halo_gptr)uniform64((unsigned long long)B.base); s.zero = (halo_gptr)uniform64((unsigned long long)P.zero);
  s.kmask = (unsigned)__builtin_amdgcn_readfirstlane((int)(0u - (unsigned)B.k));
  s.ty0 = __builtin_amdgcn_readfirstlane(T.ty0); s.tx0 = __builtin_amdgcn_readfirstlane(T.tx0);
  s.H = __builtin_amdgcn_readfirstlane(B.zin ? P.H : 0); s.W = __builtin_amdgcn_readfirstlane(P.W);
  asm volatile("" : "+s"(s.base), "+s"(s.zero), "+s"(s.kmask), "+s"(s.ty0), "+s"(s.tx0), "+s"(s.H), "+s"(s.W));
  return s;
}
__device__ __forceinline__ halo_gptr halo_addr_one(const HaloScalars& s, unsigned goff0, unsigned goff1, unsigned tyx) {
  const bool inside = (int)((unsigned)(s.ty0 + (int)(tyx & 255u)) < (unsigned)s.H) & (int)((unsigned)(s.tx0 + (int)(tyx >> 8)) < (unsigned)s.W);
  halo_gptr a = inside ? s.base + ((goff1 & s.kmask) | (goff0 & ~s.kmask)) : s.zero;   // (bit select, not a branch on the wave-uniform k)
  asm volatile("" : "+v"(a));                                 // computed HERE, not where it is used
  return a;
}
__device__ __forceinline__ void halo_load(halo_gptr (&addr)[PRE_F4], v4f (&pre)[PRE_F4]) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) asm volatile("" : "+v"(addr[n]));        // addresses are final here: the loads below stay below
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) pre[n] = *(const __attribute__((address_space(1))) v4f*)addr[n];
}

__device__ __forceinline__ void halo_fetch_at(const Params& P, const unsigned (&goff)[2][PRE_F4], const unsigned (&tyx)[PRE_F4], const TileAddr& T, int u,
                                              v4f (&pre)[PRE_F4]) {
  const int c = P.kz == 3 ? (u * 21846) >> 16 : u;           // u / 3 (exact for u < 4096)
  const int dz = P.kz == 3 ? u - c * 3 - 1 : 0;
  const int k = P.chunk_kind[c];
  const Src S = P.kind[k];
  const int z = T.tz + dz;
  const bool zin = z >= 0 && z < P.D;
  typedef const __attribute__((address_space(1))) char* gptr;
  gptr base = (gptr)(S.p + P.chunk_choff[c] + (long long)(min(max(z, 0), P.D - 1) >> S.shz) * S.plane + (k ? T.off[1] : T.off[0]));
  gptr addr[PRE_F4];
  if (zin && T.interior) {
#pragma unroll
    for (int n = 0; n < PRE_F4; ++n) addr[n] = base + (k ? goff[1][n] : goff[0][n]);
  } else {
#pragma unroll
    for (int n = 0; n < PRE_F4; ++n) {
      const bool inside = zin && (unsigned)(T.ty0 + (int)(tyx[n] & 255u)) < (unsigned)P.H && (unsigned)(T.tx0 + (int)(tyx[n] >> 8)) < (unsigned)P.W;
      addr[n] = inside ? base + (k ? goff[1][n] : goff[0][n]) : (gptr)P.zero;
    }
  }
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) asm volatile("" : "+v"(addr[n]));
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
// `vv` = the eight 16-byte registers the stores read: handed out so that a caller can keep them alive (hold_stores) until it has
// work behind it that may wait for the stores -- a register that an outstanding global store still reads cannot be overwritten
// before the store has completed (vmcnt), and the register allocator would otherwise reuse them at once
// RES = false: the kernel instance for layers without a residual contains no load here at all (a load under `if (P.res)` makes the
// compiler wait, on every path, for the outstanding memory operations before the registers it names are written again)
// (tz, ty, tx) = the tile's z plane and its first output pixel
template <int chunk_stride, bool RES = true>
__device__ __forceinline__ void scratch_to_global(const Params& P, const float* scr, int g, int tz, int ty, int tx, int wave, int lane, v4f (&vv)[8]) {
  const int y0 = ty + wave * 2, x0 = tx;
  const bool xfull = x0 + TW <= P.W;
  const int px = lane >> 3, c4 = lane & 7;                      // this lane's pixel within a chunk's 8, its channel quad
#pragma unroll
  for (int n = 0; n < 8; ++n) vv[n] = *(const v4f*)(scr + n * chunk_stride + lane * 4);
#pragma unroll
  for (int n = 0; n < 8; ++n) asm volatile("" : "+v"(vv[n]));      // all eight reads in flight before the first store
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int y = y0 + (n >> 2), x = x0 + (n & 3) * 8 + px;
    if (y < P.H && (xfull || x < P.W)) {
      const size_t pix = ((size_t)tz * P.H + y) * P.W + x;
      if (RES && P.res) {
        vv[n] += *(const v4f*)(P.res + pix * P.res_stride + g * 32 + c4 * 4);
        if (P.act == 1) { vv[n].x = fmaxf(vv[n].x, 0.f); vv[n].y = fmaxf(vv[n].y, 0.f); vv[n].z = fmaxf(vv[n].z, 0.f); vv[n].w = fmaxf(vv[n].w, 0.f); }
      }
      *(v4f*)(P.out + pix * P.c_out + g * 32 + c4 * 4) = vv[n];
    }
  }
}
__device__ __forceinline__ void hold_stores(const v4f (&vv)[8]) {
#pragma unroll
  for (int n = 0; n < 8; ++n) asm volatile("" ::"v"(vv[n]));
}
template <int chunk_stride>
__device__ __forceinline__ void scratch_to_global(const Params& P, const float* scr, int g, int t, int wave, int lane) {
  v4f vv[8];
  const int tz = t / P.tiles_plane, tr = t - tz * P.tiles_plane;
  scratch_to_global<chunk_stride>(P, scr, g, tz, (tr / P.tiles_x) * TH, (tr % P.tiles_x) * TW, wave, lane, vv);
}
template <int chunk_stride>
__device__ __forceinline__ void store_tile(const Params& P, const f32x16 (&acc)[2], float* scr, int g, int t, int wave, int lane) {
  tile_to_scratch<chunk_stride>(P, acc, scr, lane);
  scratch_to_global<chunk_stride>(P, scr, g, t, wave, lane);
}

}  // namespace sdconvdev
