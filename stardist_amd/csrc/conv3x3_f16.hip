// conv3x3_f16.hip -- the 3x3 / 3x3x3 convolution of conv3x3.hip with every f32 product evaluated as THREE fp16 x fp16 products on
// v_mfma_f32_32x32x16_f16 (f32 accumulation): x = hi + lo' * 2^-11 with hi = fp16(x), lo' = fp16((x - hi) * 2^11), and
//     a * b  ~  hi_a * hi_b + 2^-11 * (hi_a * lo'_b + lo'_a * hi_b)              (conv3x3_layout.h: error 2^-22 of the product)
// The network's default kernel since round 4 (models/unet.py conv_mode()).
//
// Why: the six-product bf16 form (conv3x3_bf16.hip) spends 5/6 of its matrix-core time on precision emulation; fp16 carries 11
// significant bits instead of 8, so two terms per operand and three products reach the same "f32-accurate" bar (networks within 3e-6 of
// a float64 evaluation; tools/split_study.py) with half the matrix instructions, two LDS planes instead of three and 12-KiB instead of
// 18-KiB weight blocks.  The cross terms are accumulated in their own tile (acc1) and scaled by 2^-11 once, in the epilogue; the bias is
// the initial value of acc0.
//
// Same decomposition as the other two kernels (8 x 32 output tile, 32 output channels per workgroup, (chunk, kz) units walked as three
// sub-units, persistent workgroups, halo tile prefetched through registers: conv3x3_device.h).  What the smaller footprint buys:
//   * 79.8 KiB of LDS per workgroup (two 12-KiB weight buffers + the 47.8-KiB two-plane tile + 8 KiB) and <= 256 registers per lane, so TWO
//     workgroups share a CU (two waves per SIMD): while one wave waits for a barrier, for its halo or for the LDS, the other one
//     feeds the matrix pipe.  The bf16 kernel (137 KiB, one wave per SIMD) had the matrix pipe idle 46 % of the time.
//   * the epilogue transposes a tile through a wave-private 2-KiB scratch in four rounds (the other kernels' 32-KiB scratch would not
//     fit twice), so that a lane stores 16 bytes of one pixel's channels.
// Range: an activation beyond the fp16 range (|x| > 65504) cannot be split; the kernel ORs a flag into *P.flag when it sees one and
// the caller re-evaluates with the bf16 kernel (models/unet.py).  Weights are checked by the packer.
#include <stdlib.h>

#include "common.h"
#include "conv3x3_device.h"
#include "stardist_hip.h"

#ifdef SD_CONV_PROFILE
__device__ unsigned long long g_conv_prof[20];   // [0] total, [1..16] phases, [19] units
#define PROF_DECL unsigned long long pf_t = __builtin_amdgcn_s_memtime(), pf_acc[16] = {}; const unsigned long long pf_t0 = pf_t; unsigned long long pf_units = 0
#define PROF(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_acc[k] += n_ - pf_t; pf_t = n_; } while (0)
#define PROF_UNIT() (++pf_units)
#define PROF_END() do { if (threadIdx.x == 0) { atomicAdd(&g_conv_prof[0], __builtin_amdgcn_s_memtime() - pf_t0); \
  for (int k_ = 0; k_ < 16; ++k_) atomicAdd(&g_conv_prof[1 + k_], pf_acc[k_]); atomicAdd(&g_conv_prof[19], pf_units); } } while (0)
#else
#define PROF_DECL
#define PROF(k)
#define PROF_UNIT()
#define PROF_END()
#endif

// Build-time experiment switches (tools/build_variant.sh): bit 0 = per-thread halo offsets precomputed per source tensor and an
// interior-tile fast path (split16 sources only: their staging keeps no split registers, so the 22 offsets fit)
// bit 1 = matrix instructions issued so that consecutive ones never share an accumulator
// bits 2, 3 = ENERGY PROBES, wrong results: only the first two operand groups of a sub-unit read their A (bit 2) / B (bit 3) operands from LDS
#ifndef SD_CONV_EXP
#define SD_CONV_EXP 3
#endif

namespace {

using namespace sdconvdev;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Per-thread constants of the halo staging (registers are the scarce resource at two waves per SIMD): only the halo coordinates
// ty | tx << 8 of this thread's PRE_F4 elements.  The source offset of an element and its LDS address are derived from them where they
// are needed (in the shadow of the matrix instructions); the channel quad of every element of a thread is q4 = tid & 7 (THREADS is a
// multiple of 8).
struct StageH {
  unsigned tyx[PRE_F4];
};

__device__ __forceinline__ void stage_init_h(StageH& st, int tid) { tyx_init<true>(st.tyx, tid); }
__device__ __forceinline__ int lds_off_of(unsigned tyx, int tid) {
  asm volatile("" : "+v"(tyx));         // derived where it is used: hoisted out of the tile loop the eleven offsets would be spilled
  return htile_store_off((int)(tyx & 255u), (int)(tyx >> 8), 0, tid & 7);
}

// The halo is read with BUFFER loads: a unit's source plane is described by one wave-uniform resource (base = the source address of the
// halo's first pixel, which may lie in front of the tensor on border tiles: nothing is read there), an element is a 32-bit offset, and an
// element outside the image gets an offset beyond the resource's range, for which the hardware returns zeros -- the zero padding of
// 'same' costs one select, no 64-bit address arithmetic and half the address registers of the pointer form (conv3x3_device.h).
// Offset of halo pixel (ty, tx), channel quad q4, in a source with half-resolution flags (shy, shx): conv3x3_layout.h src_rel,
//   (src_rel(ty, shy) * row_bytes + src_rel(tx, shx) * pix_bytes) + q4 * 16,   src_rel(t, sh) = ((t - sh) >> sh) + sh.
constexpr unsigned HALO_RANGE = 0x80000000u, HALO_OUTSIDE = 0xFFFFFFF0u;
struct HaloRsrc {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned row_bytes, pix_bytes;
  int shy, shx;
  int ty0, tx0, H, W;    // zin folded in: H = 0 for a z plane outside the volume (no element is inside)
};
__device__ __forceinline__ HaloRsrc halo_rsrc(const Params& P, const HaloBase& B, const TileAddr& T) {
  HaloRsrc s;
  const Src S = P.kind[B.k];
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)uniform64((unsigned long long)B.base), 0, (int)HALO_RANGE, 0x00020000);
  s.pix_bytes = (unsigned)__builtin_amdgcn_readfirstlane(S.stride * 4);
  s.row_bytes = (unsigned)__builtin_amdgcn_readfirstlane((P.W >> S.shx) * S.stride * 4);
  s.shy = __builtin_amdgcn_readfirstlane(S.shy); s.shx = __builtin_amdgcn_readfirstlane(S.shx);
  s.ty0 = __builtin_amdgcn_readfirstlane(T.ty0); s.tx0 = __builtin_amdgcn_readfirstlane(T.tx0);
  s.H = __builtin_amdgcn_readfirstlane(B.zin ? P.H : 0); s.W = __builtin_amdgcn_readfirstlane(P.W);
  asm volatile("" : "+s"(s.pix_bytes), "+s"(s.row_bytes), "+s"(s.shy), "+s"(s.shx), "+s"(s.ty0), "+s"(s.tx0), "+s"(s.H), "+s"(s.W));
  return s;
}
__device__ __forceinline__ unsigned halo_off_one(const HaloRsrc& s, unsigned tyx, unsigned q4off) {
  const int ty = (int)(tyx & 255u), tx = (int)(tyx >> 8);
  const bool inside = (int)((unsigned)(s.ty0 + ty) < (unsigned)s.H) & (int)((unsigned)(s.tx0 + tx) < (unsigned)s.W);
  const unsigned ry = (unsigned)(((ty - s.shy) >> s.shy) + s.shy), rx = (unsigned)(((tx - s.shx) >> s.shx) + s.shx);
  unsigned a = inside ? ry * s.row_bytes + (rx * s.pix_bytes + q4off) : HALO_OUTSIDE;
  asm volatile("" : "+v"(a));                                 // computed HERE, not where it is used
  return a;
}
__device__ __forceinline__ v4f halo_load_one(const HaloRsrc& s, unsigned off) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, (int)off, 0, 0));
}

// The same descriptor from values held in scalar registers (the two source records are read from the kernel arguments ONCE; the chunk ->
// source map in closed form): no dependent scalar loads per unit, so the whole computation can sit between matrix instructions.
struct SrcS {
  unsigned long long p;        // address of the tensor
  long long plane;             // floats per z plane
  int stride, shz, shy, shx, ws;   // ws = W >> shx
};
__device__ __forceinline__ SrcS src_scalars(const Params& P, int k) {
  const Src S = P.kind[k];
  SrcS r;
  r.p = uniform64((unsigned long long)S.p);
  r.plane = (long long)uniform64((unsigned long long)S.plane);
  r.stride = __builtin_amdgcn_readfirstlane(S.stride); r.shz = __builtin_amdgcn_readfirstlane(S.shz);
  r.shy = __builtin_amdgcn_readfirstlane(S.shy); r.shx = __builtin_amdgcn_readfirstlane(S.shx);
  r.ws = __builtin_amdgcn_readfirstlane(P.W >> S.shx);
  asm volatile("" : "+s"(r.p), "+s"(r.plane), "+s"(r.stride), "+s"(r.shz), "+s"(r.shy), "+s"(r.shx), "+s"(r.ws));
  return r;
}
struct DimS { int D, H, W, kz, n0; };
__device__ __forceinline__ void tile_at_s(const SrcS& S0, const SrcS& S1, const DimS& d, const TileWalk& w, TileAddr& T) {
  T.tz = w.tz;
  T.ty0 = w.row * TH - 1;
  T.tx0 = w.col * TW - 1;
  T.interior = T.ty0 >= 0 && T.ty0 + HALO_H <= d.H && T.tx0 >= 0 && T.tx0 + HALO_W <= d.W;
  // pixels from a plane's first pixel to the source pixel of halo pixel (0, 0): 32-bit (a plane has fewer than 2^31 pixels), times the stride
  T.off[0] = (long long)(src_base(T.ty0, S0.shy) * S0.ws + src_base(T.tx0, S0.shx)) * S0.stride;
  T.off[1] = (long long)(src_base(T.ty0, S1.shy) * S1.ws + src_base(T.tx0, S1.shx)) * S1.stride;
}
struct UnitS {
  __amdgpu_buffer_rsrc_t rsrc;
  int ty0, tx0, H, W;          // H = 0 for a z plane outside the volume
  int k, in;                   // source tensor; every element of the halo lies inside the image
};
__device__ __forceinline__ UnitS unit_scalars(const SrcS& S0, const SrcS& S1, const DimS& d, const TileAddr& T, int u) {
  const int c = d.kz == 3 ? (u * 21846) >> 16 : u;           // u / 3 (exact for u < 4096)
  const int dz = d.kz == 3 ? u - c * 3 - 1 : 0;
  const int k = c >= d.n0;
  const int choff = (c - (k ? d.n0 : 0)) * 32;
  const int z = T.tz + dz;
  const bool zin = z >= 0 && z < d.D;
  const int zc = min(max(z, 0), d.D - 1) >> (k ? S1.shz : S0.shz);
  const long long fl = (long long)choff + (long long)zc * (k ? S1.plane : S0.plane) + (k ? T.off[1] : T.off[0]);
  UnitS r;
  r.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((k ? S1.p : S0.p) + (unsigned long long)(fl * 4)), 0, (int)HALO_RANGE, 0x00020000);
  r.ty0 = T.ty0; r.tx0 = T.tx0; r.H = zin ? d.H : 0; r.W = d.W;
  r.k = k; r.in = (int)(T.interior && zin);
  return r;
}

// weights of sub-unit (u, dy) of group g: 12 KiB = 768 x 16 bytes through registers (three per thread), as in conv3x3_bf16.hip
constexpr int WREG = (HWSUB_BYTES / 16 + THREADS - 1) / THREADS;
static_assert(WREG * THREADS == HWSUB_BYTES / 16, "a sub-unit's weights are a whole number of 16-byte elements per thread");
__device__ __forceinline__ void weights_fetch(const Params& P, int g, int u, int dy, v4f (&wreg)[WREG], int tid) {
  // buffer loads: wave-uniform block address in the resource, one 32-bit offset register per thread (no 64-bit address per element)
  const char* blk = (const char*)P.wp + (((size_t)g * P.n_units + u) * 3 + dy) * HWSUB_BYTES;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)uniform64((unsigned long long)blk), 0, HWSUB_BYTES, 0x00020000);
#pragma unroll
  for (int n = 0; n < WREG; ++n) wreg[n] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, n * THREADS * 16, 0));
}
__device__ __forceinline__ void weights_store(char* __restrict__ wnext, const v4f (&wreg)[WREG], int tid) {
#pragma unroll
  for (int n = 0; n < WREG; ++n) ((v4f*)wnext)[tid + n * THREADS] = wreg[n];
}

// two f32 values -> their two fp16 terms, packed (a in the low half); round to nearest even like split2_f16 of conv3x3_layout.h;
// `amax` collects max |x| for the range flag
__device__ __forceinline__ void split2_pair(float a, float b, unsigned& hi, unsigned& lo, float& amax) {
  const f32x2 x = {a, b};
  const f16x2 h = __builtin_convertvector(x, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * 2048.f;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
}
__device__ __forceinline__ void split_elem(const v4f x, u32x2 (&pl)[2], float& amax) {
  unsigned a[2], b[2];
  split2_pair(x.x, x.y, a[0], a[1], amax);
  split2_pair(x.z, x.w, b[0], b[1], amax);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    asm volatile("" : "+v"(a[p]), "+v"(b[p]));                // computed HERE (the compiler would sink the arithmetic to the stores)
    pl[p].x = a[p]; pl[p].y = b[p];
  }
}
__device__ __forceinline__ void store_planes(const StageH& st, char* __restrict__ tileH, const u32x2 (&pk)[PRE_F4][2], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n)
    if (n < PRE_F4 - 1 || tid < TILE_F4 - (PRE_F4 - 1) * THREADS) {
#pragma unroll
      for (int p = 0; p < 2; ++p) *(u32x2*)(tileH + lds_off_of(st.tyx[n], tid) + p * 64) = pk[n][p];
    }
}

// split16 sources (INP): an element is already 8 fp16 values of one plane (conv3x3_layout.h "split16"): the pixel's 128 bytes are copied as they are
__device__ __forceinline__ void store_copy(const StageH& st, char* __restrict__ tileH, const v4f (&pre)[PRE_F4], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n)
    if (n < PRE_F4 - 1 || tid < TILE_F4 - (PRE_F4 - 1) * THREADS) {
      // element tid + n * THREADS lies in halo pixel 32 n + p(tid) of the staging order (conv3x3_layout.h stage_elem_b; the last, partial
      // block keeps the plain order), quad tid & 7: one per-thread base, the rest is an immediate offset
      const int pb = n < (TILE_F4 >> 8) ? 32 * n + ((tid >> 3) & 7) * 4 + ((tid >> 6) & 3) : (tid >> 3) + n * (THREADS / 8);
      *(v4f*)(tileH + pb * HPIX + (tid & 7) * 16) = pre[n];
    }
}

// One sub-unit (row tap dy): 6 operand groups (dx, block); a group = 2 halo rows x 2 planes (A) + 2 planes (B) = 6 ds_read_b128 feeding
// 6 MFMAs (three plane pairs x two output rows).  The operands of group g+1 are read while the matrix cores work on group g.
// `extra(gi)`: vector-ALU work / global loads that do not depend on the matrix instructions, interleaved with them.
template <int NV, int NLD, class Extra>
__device__ __forceinline__ void compute_sub(const char* __restrict__ tileH, const char* __restrict__ w, int dy, f32x16 (&acc0)[2], f32x16 (&acc1)[2],
                                            int wave, int i, int h, Extra extra) {
  u32x4 A[2][2][2], B[2][2];
  const char* arow = tileH + (wave * 2 + dy) * HALO_W * HPIX;
#define SD_LOAD_GROUP_H(gi, buf)                                                                                        \
  do {                                                                                                                   \
    const int dx_ = (gi) >> 1, b_ = (gi) & 1;                                                                            \
    if (!(SD_CONV_EXP & 4) || (gi) < 2) {                                                                               \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                        \
      _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) A[buf][p][pl] = *(const u32x4*)(arow + htile_off(p, i + dx_, pl, b_, h)); \
    }                                                                                                                    \
    if (!(SD_CONV_EXP & 8) || (gi) < 2) {                                                                               \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) B[buf][pl] = *(const u32x4*)(w + hw_off(dx_, b_, pl, h, i));        \
    }                                                                                                                    \
  } while (0)
  SD_LOAD_GROUP_H(0, 0);
#pragma unroll
  for (int gi = 0; gi < 6; ++gi) {
    const int buf = gi & 1;
    if (gi + 1 < 6) SD_LOAD_GROUP_H(gi + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    extra(gi);
    const f16x8 bh = __builtin_bit_cast(f16x8, B[buf][0]), bl = __builtin_bit_cast(f16x8, B[buf][1]);
    // Issue order: no two consecutive matrix instructions share an accumulator (every tile's own sequence of additions is unchanged, so
    // the results are the same bits): a dependent pair issued back to back stalls for the first one's passes whenever anything else
    // -- a vector instruction of the shadow work, an operand read -- lands between them (MI355X_MICROARCH: +43 cycles per such slot).
    {
      const f16x8 a0h = __builtin_bit_cast(f16x8, A[buf][0][0]), a0l = __builtin_bit_cast(f16x8, A[buf][0][1]);
      const f16x8 a1h = __builtin_bit_cast(f16x8, A[buf][1][0]), a1l = __builtin_bit_cast(f16x8, A[buf][1][1]);
#if SD_CONV_EXP & 2
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, acc1[1], 0, 0, 0);
      acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, acc0[0], 0, 0, 0);
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, acc1[1], 0, 0, 0);
      acc0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, acc0[1], 0, 0, 0);
#else
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, acc1[0], 0, 0, 0);
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, acc1[0], 0, 0, 0);
      acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, acc0[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, acc1[1], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, acc1[1], 0, 0, 0);
      acc0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, acc0[1], 0, 0, 0);
#endif
    }
    if (NV > 0 || NLD > 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA ...
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                      // ... then NV vector-ALU instructions
        if (NLD > 0 && k % (6 / NLD) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ... or a global load
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SD_LOAD_GROUP_H
}

// Results of one tile -> HBM.  An accumulator register holds ONE channel (i) of 16 pixels (register r of lane (i, h): tile column
// (r & 3) + 8 (r >> 2) + 4 h, conv3x3_layout.h acc_col); channels-last memory wants the 32 channels of a pixel together, and a store
// instruction costs the memory pipeline per LANE ADDRESS, not per byte: 32 global_store_dword per wave (two pixels' 128 bytes each) took
// 19 % of the 2D 32 -> 32 layer.  So the tile is transposed through a wave-private 2-KiB LDS scratch in four rounds of 16 pixels
// (8 ds_write_b32 + 2 ds_read_b128 + 2 buffer_store_dwordx4 each: a lane stores 4 channels of one pixel); LDS operations of one wave
// execute in order, so the rounds need no barrier and no wait between a round's reads and the next round's writes.
// DOT: the one-channel head behind this layer (the object probability, model2d.py:338-341 / model3d.py:436-439), first stage: from the four
// values the lane is about to store, their products with the head's weights summed in channel order -- exactly the per-lane term of
// sd_bias_act_dot_device (unet_ops.hip) -- written to dotp[pixel][c_out / 4] (4 bytes per lane, 1/4 of the features' bytes).
// sd_dot_combine_device then adds a pixel's c_out / 4 terms in the order of that kernel's xor butterfly, the bias and the logistic function:
// the probabilities are BIT-IDENTICAL to the two-pass form, and the 128-channel features are not read a second time for them.
// The head's 32 weights of this workgroup's channels are staged in LDS (behind the epilogue scratch) at kernel start.
// NOST (with DOT): the tile itself is NOT stored -- the sparse prediction path needs the features of the candidate pixels only and
// recomputes exactly those afterwards (k_conv3_f16_rows below, bit-identical), so the dense 128-channel tensor (2 GiB at 2048^2, 8.6 GB at
// 256^3) is never written.
template <bool RES, bool DOT, bool NOST = false>
__device__ __forceinline__ void store_tile(const Params& P, const f32x16 (&acc0)[2], const f32x16 (&acc1)[2], float* __restrict__ scr, int g, int tz,
                                           int ty, int tx, int wave, int lane) {
  const int i = lane & 31, h = lane >> 5;
  const int px = lane >> 3, c4 = lane & 7;                      // as a reader: this lane's pixel within a chunk's 8, its channel quad
  const unsigned pix_bytes = (unsigned)P.c_out * 4u, res_bytes = (unsigned)P.res_stride * 4u;
  const unsigned chan_off = (unsigned)(g * 32 + c4 * 4) * 4u;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int y = ty + wave * 2 + p;
    if (y >= P.H) continue;                                            // (wave-uniform)
    const size_t row = ((size_t)tz * P.H + y) * P.W;
    // buffer stores: one resource per output row (wave-uniform), 32-bit offsets inside the row; a pixel beyond the end of the row lies
    // outside the resource and the hardware drops the store (partial last tile column)
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform64((unsigned long long)(P.out + row * P.c_out)), 0, (int)((unsigned)P.W * pix_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rr = ro;
    if (RES && P.res)
      rr = __builtin_amdgcn_make_buffer_rsrc((void*)uniform64((unsigned long long)(P.res + row * P.res_stride)), 0, (int)((unsigned)P.W * res_bytes), 0x00020000);
    __amdgpu_buffer_rsrc_t rd = ro;
    if (DOT)
      rd = __builtin_amdgcn_make_buffer_rsrc((void*)uniform64((unsigned long long)(P.dotp + row * (size_t)(P.c_out >> 2))), 0, (int)((unsigned)P.W * (unsigned)P.c_out), 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                      // columns 16 q .. 16 q + 15
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int r = q * 8 + r8;
        const int pl = (r8 & 3) + 8 * (r8 >> 2) + 4 * h;               // pixel within the round's 16
        scr[pl * 32 + i] = acc0[p][r] + acc1[p][r] * 4.8828125e-4f;    // 2^-11
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        v4f v = *(const v4f*)(scr + n * 256 + lane * 4);
        const unsigned x = (unsigned)(tx + q * 16 + n * 8 + px);
        if (RES && P.res) v += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(x * res_bytes + chan_off), 0, 0));
        if (P.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        const int so = (int)(x * pix_bytes + chan_off);
        if (!NOST) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, so, 0, 0);
        if (DOT) {
          const v4f hw = *(const v4f*)(scr + (4 - wave) * 512 + c4 * 4);    // (LDS, behind the four waves' scratch; held in registers across the unit loop it would be spilled)
          float d = v.x * hw.x;
          d += v.y * hw.y; d += v.z * hw.z; d += v.w * hw.w;
          // dotp[pixel][c_out / 4]: this lane's slot is a quarter of the byte offset of its 16-byte feature store (x >= W: dropped)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d), rd, so >> 2, 0, 0);
        }
      }
    }
  }
}

// The same tile as a split16 tensor (OUTP): the consumer's two fp16 terms of every value, made ONCE here instead of once per consumer
// workgroup and unit.  The transposition hands a lane EIGHT consecutive channels of one pixel (two conflict-free ds_read_b128: the
// scratch holds channels o*8 .. o*8+3 of the round's 16 pixels in its first KiB, o*8+4 .. o*8+7 in the second), which it splits
// (round to nearest even, exactly split2_pair of the consumer) and stores as the pixel's hi element and lo' element (16 bytes each,
// 64 bytes apart).  `amax` collects max |x| of what is stored: a value beyond the fp16 range cannot be represented (range flag bit 1).
__device__ __forceinline__ void store_tile_split(const Params& P, const f32x16 (&acc0)[2], const f32x16 (&acc1)[2], float* __restrict__ scr, int g, int tz,
                                                 int ty, int tx, int wave, int lane, float& amax) {
  const int i = lane & 31, h = lane >> 5;
  const int pr = lane >> 2, o = lane & 3;                       // as a reader: this lane's pixel within the round's 16, its channel octet
  const unsigned pix_bytes = (unsigned)P.c_out * 4u;
  const unsigned chan_off = (unsigned)(g * 128 + o * 16);
  const int wofs = ((i >> 2) & 1) * 256 + (i >> 3) * 4 + (i & 3);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int y = ty + wave * 2 + p;
    if (y >= P.H) continue;                                            // (wave-uniform)
    const size_t row = ((size_t)tz * P.H + y) * P.W;
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc((void*)uniform64((unsigned long long)(P.out + row * P.c_out)), 0, (int)((unsigned)P.W * pix_bytes), 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                      // columns 16 q .. 16 q + 15
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int r = q * 8 + r8;
        const int pl = (r8 & 3) + 8 * (r8 >> 2) + 4 * h;               // pixel within the round's 16
        scr[pl * 16 + wofs] = acc0[p][r] + acc1[p][r] * 4.8828125e-4f;    // 2^-11
      }
      v4f va = *(const v4f*)(scr + lane * 4), vb = *(const v4f*)(scr + 256 + lane * 4);
      if (P.act == 1) {
        va.x = fmaxf(va.x, 0.f); va.y = fmaxf(va.y, 0.f); va.z = fmaxf(va.z, 0.f); va.w = fmaxf(va.w, 0.f);
        vb.x = fmaxf(vb.x, 0.f); vb.y = fmaxf(vb.y, 0.f); vb.z = fmaxf(vb.z, 0.f); vb.w = fmaxf(vb.w, 0.f);
      }
      unsigned hw[4], lw[4];
      split2_pair(va.x, va.y, hw[0], lw[0], amax);
      split2_pair(va.z, va.w, hw[1], lw[1], amax);
      split2_pair(vb.x, vb.y, hw[2], lw[2], amax);
      split2_pair(vb.z, vb.w, hw[3], lw[3], amax);
      const u32x4 hi = {hw[0], hw[1], hw[2], hw[3]}, lo = {lw[0], lw[1], lw[2], lw[3]};
      const int so = (int)((unsigned)(tx + q * 16 + pr) * pix_bytes + chan_off);     // (x >= W: outside the row's resource, dropped)
      __builtin_amdgcn_raw_buffer_store_b128(hi, ro, so, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(lo, ro, so + 64, 0, 0);
    }
  }
}

// TWO workgroups per CU (79.8 KiB of LDS each, <= 256 registers per lane): two waves per SIMD
// (WPE = 1: the same code compiled for one wave per SIMD -- 512 registers -- as the A/B partner of option conv_f16_workgroups_per_cu = 1)
// INP: the sources are split16 tensors (no split here, the halo is copied); OUTP: the output is written as a split16 tensor
template <bool RES, int WPE, bool DOT = false, bool INP = false, bool OUTP = false, bool NOST = false>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_conv3_f16(const Params P) {
  static_assert(!(OUTP && (RES || DOT)), "a split16 output has neither residual nor fused head");
  static_assert(!NOST || DOT, "only the fused-head form can do without its tile store");
  extern __shared__ float4 smem4h[];
  // LDS map (bytes): two weight buffers of one sub-unit each | halo tile, 2 fp16 planes | 4 x 2 KiB wave-private epilogue scratch
  char* W = (char*)smem4h;
  char* tileH = W + 2 * HWSUB_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* scr = (float*)(tileH + HTILE_BYTES) + wave * 512;
  int g, q, Q;
  wg_slot(P, g, q, Q);
  if (q >= P.n_tiles) return;
  if (DOT && tid < 32) ((float*)(tileH + HTILE_BYTES) + 4 * 512)[tid] = P.dotw[g * 32 + tid];    // the head's 32 weights of this group, behind the epilogue scratch (visible behind the prologue's barrier)
  const float bias_r = P.bias ? P.bias[g * 32 + (lane & 31)] : 0.f;
  StageH st;
  stage_init_h(st, tid);
  const unsigned q4off = (unsigned)(tid & 7) * 16u;
  // split16 sources: the element offsets relative to the halo's first source pixel depend on the source tensor only (resolution flags,
  // strides), not on the tile -- kept per thread for both tensors; a tile whose halo lies inside the image then needs one select per element
  constexpr bool PREOFF = INP && (SD_CONV_EXP & 1);
  unsigned offk[PREOFF ? 2 : 1][PREOFF ? PRE_F4 : 1];
  if constexpr (PREOFF) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const Src S = P.kind[k];
      const unsigned rowb = (unsigned)((P.W >> S.shx) * S.stride * 4), pixb = (unsigned)(S.stride * 4);
#pragma unroll
      for (int n = 0; n < PRE_F4; ++n) {
        const int ty = (int)(st.tyx[n] & 255u), tx = (int)(st.tyx[n] >> 8);
        offk[k][n] = (unsigned)(((ty - S.shy) >> S.shy) + S.shy) * rowb + ((unsigned)(((tx - S.shx) >> S.shx) + S.shx) * pixb + q4off);
      }
    }
  }
  const SrcS S0 = src_scalars(P, 0), S1 = src_scalars(P, 1);
  DimS dim;
  dim.D = __builtin_amdgcn_readfirstlane(P.D); dim.H = __builtin_amdgcn_readfirstlane(P.H); dim.W = __builtin_amdgcn_readfirstlane(P.W);
  dim.kz = __builtin_amdgcn_readfirstlane(P.kz); dim.n0 = __builtin_amdgcn_readfirstlane(P.n_chunks0);
  TileAddr Tc, Tn;                                      // the tile being computed, the tile whose first unit is fetched next
  TileWalk walk;
  walk_init(P, q, Q, walk);
  tile_at(P, walk, Tc);
  Tn = Tc;
  float amax = 0.f, amax_out = 0.f;
  {
    v4f pre[PRE_F4], wreg[WREG];
    weights_fetch(P, g, 0, 0, wreg, tid);
    {
      const HaloRsrc hs = halo_rsrc(P, halo_base(P, Tc, 0), Tc);
#pragma unroll
      for (int n = 0; n < PRE_F4; ++n) pre[n] = halo_load_one(hs, halo_off_one(hs, st.tyx[n], q4off));
    }
    weights_store(W, wreg, tid);
    if (INP) store_copy(st, tileH, pre, tid);
    else {
      u32x2 pk[PRE_F4][2];
#pragma unroll
      for (int n = 0; n < PRE_F4; ++n) split_elem(pre[n], pk[n], amax);
      store_planes(st, tileH, pk, tid);
    }
  }
  __syncthreads();
  PROF_DECL;
  int wb = 0;
  for (int t = q; t < P.n_tiles; t += Q) {
    f32x16 acc0[2], acc1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[p][r] = bias_r; acc1[p][r] = 0.f; }
    for (int u = 0; u < P.n_units; ++u) {
      const bool last = u == P.n_units - 1;
      const int tn = last ? t + Q : t, un = last ? 0 : u + 1;
      const bool have = tn < P.n_tiles;
      v4f pre[PRE_F4];
      unsigned addr[PRE_F4];
      u32x2 pk[INP ? 1 : PRE_F4][2];
      HaloRsrc hs;
      UnitS us;                         // (PREOFF) the next unit's source: resource, tile position, source tensor, "halo inside the image"
      // dy 0: its matrix instructions hide the address arithmetic of the next unit's halo; dy 1 issues its weight loads FIRST and the
      // halo loads after them (the wait for the weights leaves the halo in flight); dy 2 splits the halo elements into their fp16
      // planes (registers) as they arrive; after its barrier only the LDS stores are left.
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const bool lastdy = dy == 2, have_w = lastdy ? have : true;
        v4f wreg[WREG];
        // (always issued, from a valid address even when there is nothing left to fetch)
        weights_fetch(P, g, lastdy ? (have ? un : u) : u, lastdy ? (have ? 0 : dy) : dy + 1, wreg, tid);      // next sub-unit's weights
        __builtin_amdgcn_sched_barrier(0);
        PROF(dy);
        if (!PREOFF && dy == 0 && last && have) { walk_step(P, walk); tile_at(P, walk, Tn); }   // (once per tile; no division)
        __builtin_amdgcn_sched_barrier(0);
        if (dy == 0) PROF(14);
        const char* wcur = W + wb * HWSUB_BYTES;
        if (PREOFF && dy == 1 && !us.in) {
          // a tile at the image border (wave-uniform): elements outside the image get the offset the hardware answers with zeros
#pragma unroll
          for (int n = 0; n < PRE_F4; ++n) {
            const int ty = (int)(st.tyx[n] & 255u), tx = (int)(st.tyx[n] >> 8);
            const bool inside = (int)((unsigned)(us.ty0 + ty) < (unsigned)us.H) & (int)((unsigned)(us.tx0 + tx) < (unsigned)us.W);
            addr[n] = inside ? addr[n] : HALO_OUTSIDE;
          }
        }
        if (dy == 0 && PREOFF) {
          // the next unit's source (scalar registers only: the next tile's position once per tile, the unit's plane and tensor) and its
          // element offsets (one select each) -- all of it between this sub-unit's matrix instructions
          compute_sub<1, 0>(tileH, wcur, dy, acc0, acc1, wave, lane & 31, lane >> 5, [&](int gi) {
            if (gi == 0) {
              if (last && have) { walk_step(P, walk); tile_at_s(S0, S1, dim, walk, Tn); }
              us = unit_scalars(S0, S1, dim, tile_select(last && have, Tc, Tn), have ? un : u);
            } else {
              const bool k1 = us.k != 0;
#pragma unroll
              for (int n = (gi - 1) * 3; n < (gi - 1) * 3 + 3; ++n)
                if (n < PRE_F4) { unsigned a = k1 ? offk[1][PREOFF ? n : 0] : offk[0][PREOFF ? n : 0]; asm volatile("" : "+v"(a)); addr[n] = a; }
            }
          });
        } else if (dy == 0) {
          const TileAddr T = tile_select(last && have, Tc, Tn);
          const HaloBase hb = halo_base(P, T, have ? un : u);
          hs = halo_rsrc(P, hb, T);
          PROF(8);
          compute_sub<5, 0>(tileH, wcur, dy, acc0, acc1, wave, lane & 31, lane >> 5, [&](int gi) {
#pragma unroll
            for (int n = gi * 2; n < gi * 2 + 2; ++n)
              if (n < PRE_F4) addr[n] = halo_off_one(hs, st.tyx[n], q4off);
          });
        } else if (dy == 1) {
          compute_sub<0, 2>(tileH, wcur, dy, acc0, acc1, wave, lane & 31, lane >> 5, [&](int gi) {      // the next unit's halo loads
#pragma unroll
            for (int n = gi * 2; n < gi * 2 + 2; ++n)
              if (n < PRE_F4) pre[n] = PREOFF ? __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(us.rsrc, (int)addr[n], 0, 0)) : halo_load_one(hs, addr[n]);
          });
        } else if (INP) {
          compute_sub<0, 0>(tileH, wcur, dy, acc0, acc1, wave, lane & 31, lane >> 5, [&](int) {});
        } else {
          compute_sub<6, 0>(tileH, wcur, dy, acc0, acc1, wave, lane & 31, lane >> 5, [&](int gi) {
#pragma unroll
            for (int n = (gi - 2) * 3; n < (gi - 2) * 3 + 3; ++n)
              if (gi >= 2 && n < PRE_F4) split_elem(pre[n], pk[INP ? 0 : n], amax);
          });
        }
        PROF(dy == 0 ? 3 : 11 + dy);
        if (have_w) weights_store(W + (wb ^ 1) * HWSUB_BYTES, wreg, tid);                    // the other buffer: nobody reads it now
        if (lastdy) {                                                                        // (a use on every path, see conv3x3_bf16.hip)
#pragma unroll
          for (int n = 0; n < WREG; ++n) asm volatile("" ::"v"(wreg[n]));
        }
        PROF(4 + dy);
        __syncthreads();
        PROF(7);
        wb ^= 1;
      }
      PROF(8);
      if (have) {                                                  // (every wave is past the barrier behind the last sub-unit)
        if constexpr (INP) store_copy(st, tileH, pre, tid);
        else store_planes(st, tileH, pk, tid);
      }
      PROF(9);
      __syncthreads();
      PROF(10);
      PROF_UNIT();
    }
    if constexpr (OUTP) store_tile_split(P, acc0, acc1, scr, g, Tc.tz, Tc.ty0 + 1, Tc.tx0 + 1, wave, lane, amax_out);
    else store_tile<RES, DOT, NOST>(P, acc0, acc1, scr, g, Tc.tz, Tc.ty0 + 1, Tc.tx0 + 1, wave, lane);
    PROF(11);
    Tc = Tn;
  }
  // an activation outside the fp16 range (or not finite): tell the host (one atomic per offending wave, normally none)
  if (!INP && P.flag && !(amax <= 65504.f)) atomicOr(P.flag, 1);
  // ... or a value this layer was to store as a split16 element (bit 1: the OUTPUT is not valid, whatever reads it)
  if (OUTP && P.flag && !(amax_out <= 65504.f)) atomicOr(P.flag, 2);
  PROF_END();
}

// ---- the same layer on SELECTED pixels (the sparse prediction path's features) --------------------------------------------------------
// out[r][:] = act(bias + conv(x))[rows[r]] for n_rows pixels given by their linear index in the [D][H][W] grid -- what the dense kernel
// would have stored there, BIT FOR BIT: the same matrix instruction with the same operand slots in the same order per accumulator
// (units (chunk, z tap), row tap dy, (dx, 16-channel block); cross terms ah*bl, al*bh into acc1, ah*bh into acc0; acc0 starts from the
// bias; acc0 + acc1 2^-11; activation).  A row of the MFMA's A operand is a pixel, rows are independent, so gathering 32 arbitrary
// candidates into one operand changes nothing for any of them.  One wave = 32 candidates x NG groups of 32 output channels (the gathered
// A operand is used for all groups); operands straight from global memory (the candidates of one object share cache lines, the packed
// weights of a sub-unit are 12 KiB that every wave reads).  One full-resolution source; f32 or split16.
struct RowsParams {
  const float* src; int c_in;            // [D][H][W][c_in]
  int D, H, W, kz, n_units;
  const float* wp; const float* bias;
  int c_out, act, g0;                    // first output-channel group of this launch
  const long long* rows; long long n_rows;
  float* out;                            // [n_rows][c_out]
};
template <int NG, bool INP>
__global__ void __launch_bounds__(256) k_conv3_f16_rows(const RowsParams P) {
  // Per sub-unit (unit, row tap dy): the workgroup stages the NG weight blocks (12 KiB each) in LDS once for its four waves, every lane
  // has its 6 x 2 operand elements of the gathered pixels in flight meanwhile (unconditional loads from a clamped address, zeroed
  // afterwards outside the image: nothing waits on a branch), then 6 x 3 x NG matrix instructions read their B operands from LDS.
  extern __shared__ float4 rows_lds4[];
  char* wl = (char*)rows_lds4;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const long long wbase = ((long long)blockIdx.x * 4 + (tid >> 6)) * 32;
  const long long ri = wbase + i;
  const bool live = ri < P.n_rows;
  long long pix = live ? P.rows[ri] : 0;
  const int x = (int)(pix % P.W); pix /= P.W;
  const int y = (int)(pix % P.H);
  const int z = (int)(pix / P.H);
  f32x16 acc0[NG], acc1[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float b = P.bias ? P.bias[(P.g0 + g) * 32 + i] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[g][r] = b; acc1[g][r] = 0.f; }
  }
  const size_t pix_floats = (size_t)P.c_in;
  float amax = 0.f;
  for (int u = 0; u < P.n_units; ++u) {
    const int c = P.kz == 3 ? u / 3 : u, dz = P.kz == 3 ? u - 3 * c - 1 : 0;
    const int zz = z + dz;
    const bool zin = live && zz >= 0 && zz < P.D;
    const int zc = min(max(zz, 0), P.D - 1);
#pragma unroll 1
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      const bool yin = zin && yy >= 0 && yy < P.H;
      const int yc = min(max(yy, 0), P.H - 1);
      const float* prow = P.src + (((size_t)zc * P.H + yc) * P.W) * pix_floats + (size_t)c * 32;
      // the gathered A operands of the six (dx, block) groups
      u32x4 ah[6], al[6];
      v4f f0[INP ? 1 : 6], f1[INP ? 1 : 6];
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        const int dx = gi >> 1, b = gi & 1;
        const int xc = min(max(x + dx - 1, 0), P.W - 1);
        const float* px = prow + (size_t)xc * pix_floats;
        if (INP) {                                                    // split16: element p * 4 + (b * 2 + h) of the pixel's chunk
          ah[gi] = *(const u32x4*)(px + (b * 2 + h) * 4);
          al[gi] = *(const u32x4*)(px + 16 + (b * 2 + h) * 4);
        } else {
          f0[INP ? 0 : gi] = *(const v4f*)(px + b * 16 + h * 8); f1[INP ? 0 : gi] = *(const v4f*)(px + b * 16 + h * 8 + 4);
        }
      }
      // the weights of this sub-unit for the NG groups -> LDS (the previous sub-unit's reads are done behind the first barrier)
      __syncthreads();
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const v4f* wsrc = (const v4f*)((const char*)P.wp + ((((size_t)(P.g0 + g) * P.n_units + u) * 3 + dy) * HWSUB_BYTES));
#pragma unroll
        for (int n = 0; n < HWSUB_BYTES / 16 / 256; ++n) ((v4f*)(wl + g * HWSUB_BYTES))[tid + n * 256] = wsrc[tid + n * 256];
      }
      __syncthreads();
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        const int dx = gi >> 1, b = gi & 1;
        const int xx = x + dx - 1;
        const bool inside = yin && xx >= 0 && xx < P.W;
        u32x4 a_h, a_l;
        if (INP) { a_h = ah[gi]; a_l = al[gi]; }
        else {
          const v4f v0 = f0[INP ? 0 : gi], v1 = f1[INP ? 0 : gi];
          unsigned hw[4], lw[4];
          split2_pair(v0.x, v0.y, hw[0], lw[0], amax); split2_pair(v0.z, v0.w, hw[1], lw[1], amax);
          split2_pair(v1.x, v1.y, hw[2], lw[2], amax); split2_pair(v1.z, v1.w, hw[3], lw[3], amax);
          a_h = u32x4{hw[0], hw[1], hw[2], hw[3]}; a_l = u32x4{lw[0], lw[1], lw[2], lw[3]};
        }
        if (!inside) { a_h = u32x4{0u, 0u, 0u, 0u}; a_l = u32x4{0u, 0u, 0u, 0u}; }
        const f16x8 fah = __builtin_bit_cast(f16x8, a_h), fal = __builtin_bit_cast(f16x8, a_l);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const f16x8 bh = __builtin_bit_cast(f16x8, *(const u32x4*)(wl + g * HWSUB_BYTES + hw_off(dx, b, 0, h, i)));
          const f16x8 bl = __builtin_bit_cast(f16x8, *(const u32x4*)(wl + g * HWSUB_BYTES + hw_off(dx, b, 1, h, i)));
          acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bl, acc1[g], 0, 0, 0);
          acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, bh, acc1[g], 0, 0, 0);
          acc0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bh, acc0[g], 0, 0, 0);
        }
      }
    }
  }
  // register r of lane (i, h): candidate (r & 3) + 8 (r >> 2) + 4 h of the wave, output channel i of the group
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long ro = wbase + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc0[g][r] + acc1[g][r] * 4.8828125e-4f;
      if (P.act == 1) v = fmaxf(v, 0.f);
      if (ro < P.n_rows) P.out[(size_t)ro * P.c_out + (P.g0 + g) * 32 + i] = v;
    }
}

}  // namespace

extern "C" long long sd_conv3_f16x3_packed_floats(int c_in, int c_out, int kz) {
  if ((kz != 1 && kz != 3) || c_in <= 0 || c_in % 32 || c_in > 32 * sdconv::MAX_CHUNKS || c_out <= 0 || c_out % 32) return -1;
  return (long long)(sdconv::hpacked_bytes(c_in, c_out, kz) / 4) + 4;      // + 16 bytes of zeros (the zero-padding source)
}

extern "C" int sd_conv3_f16x3_pack_weights_host(const float* w, int c_in, int c_out, int kz, float* packed) {
  const long long n = sd_conv3_f16x3_packed_floats(c_in, c_out, kz);
  if (!w || !packed || n < 0) {
    sd::set_error("sd_conv3_f16x3_pack_weights: kz 1|3, c_in a multiple of 32 up to 512, c_out a multiple of 32");
    return -1;
  }
  const float wmax = sdconv::pack_weights_f16(w, c_in, c_out, kz, (unsigned short*)packed);
  for (int k = 0; k < 4; ++k) packed[n - 4 + k] = 0.f;
  if (!(wmax <= 65504.f)) {
    sd::set_error("sd_conv3_f16x3_pack_weights: a weight of magnitude %g is outside the fp16 range (use the bf16x6 form for this layer)", (double)wmax);
    return -2;
  }
  return 0;
}

static int conv3_f16x3_launch(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                              int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias,
                              const float* d_res, int res_stride, int c_out, int act, float* d_out, int* d_range_flag,
                              const float* d_dot_w, float* d_dot_partial, void* stream_, int in_split = 0, int out_split = 0) {
  hipStream_t s = (hipStream_t)stream_;
  const bool no_store = d_out == nullptr && d_dot_w != nullptr && d_dot_partial != nullptr && !d_res && !out_split;   // fused head without the feature store
  if ((in_split & ~1) || (out_split & ~1) || (out_split && (d_res || d_dot_w)) || (in_split && d_res) ||
      (in_split && (stride0 != c0 || (d_src1 && stride1 != c1)))) {
    sd::set_error("sd_conv3_f16x3: split16 tensors are dense (stride == channels); a split16 output takes neither residual nor fused head, a residual layer no split16 input");
    return -1;
  }
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  const int c_in = c0 + (d_src1 ? c1 : 0);
  const long long n_packed = sd_conv3_f16x3_packed_floats(c_in, c_out, kz);
  if (!d_src0 || !d_wpacked || (!d_out && !no_store) || (act != 0 && act != 1) || n_packed < 0 || (kz == 1 && D != 1) ||
      (((uintptr_t)d_src0 | (uintptr_t)d_src1 | (uintptr_t)d_wpacked | (uintptr_t)d_out | (uintptr_t)d_bias) & 15) || ((uintptr_t)d_range_flag & 3)) {
    sd::set_error("sd_conv3_f16x3: unsupported channel counts (%d + %d -> %d), kz, act or misaligned pointers", c0, d_src1 ? c1 : 0, c_out);
    return -1;
  }
  const int ups[2] = {up0, d_src1 ? up1 : 0};
  for (int k = 0; k < 2; ++k)
    if (ups[k] < 0 || ups[k] > 7 || ((ups[k] & 1) && (W & 1)) || ((ups[k] & 2) && (H & 1)) || ((ups[k] & 4) && (D & 1))) {
      sd::set_error("sd_conv3_f16x3: up is a bit mask (1: x, 2: y, 4: z); an up-sampled axis needs an even output size");
      return -1;
    }
  if ((c0 % 32) || (d_src1 && (c1 % 32)) || stride0 < c0 || (stride0 & 3) || (d_src1 && (stride1 < c1 || (stride1 & 3)))) {
    sd::set_error("sd_conv3_f16x3: sources must hold multiples of 32 channels, strides multiples of 4 floats");
    return -1;
  }
  Params P;
  int nc = 0;
  P.kind[0] = make_src(d_src0, stride0, up0, H, W);
  P.kind[1] = d_src1 ? make_src(d_src1, stride1, up1, H, W) : P.kind[0];
  for (int k = 0; k < MAX_CHUNKS; ++k) { P.chunk_kind[k] = 0; P.chunk_choff[k] = 0; }
  for (int k = 0; k < c0 / 32; ++k) { P.chunk_kind[nc] = 0; P.chunk_choff[nc++] = k * 32; }
  if (d_src1) for (int k = 0; k < c1 / 32; ++k) { P.chunk_kind[nc] = 1; P.chunk_choff[nc++] = k * 32; }
  P.D = D; P.H = H; P.W = W; P.kz = kz; P.n_units = nc * kz; P.n_chunks0 = c0 / 32;
  P.zero = d_wpacked + (n_packed - 4);
  if (d_res && (res_stride < c_out || (res_stride & 3) || ((uintptr_t)d_res & 15))) {
    sd::set_error("sd_conv3_f16x3: the residual needs 16-byte alignment and a stride >= c_out");
    return -1;
  }
  P.res = d_res; P.res_stride = res_stride;
  P.wp = d_wpacked; P.bias = d_bias; P.out = d_out; P.c_out = c_out; P.act = act;
  P.flag = d_range_flag;
  if ((d_dot_w != nullptr) != (d_dot_partial != nullptr) || (d_dot_w && d_res) || ((uintptr_t)d_dot_w & 15) || ((uintptr_t)d_dot_partial & 3) ||
      (d_dot_w && (long long)W * c_out >= 0x7fffffffLL)) {
    sd::set_error("sd_conv3_f16x3: the fused head needs both its weights (16-byte aligned) and its partial-sum buffer, and no residual");
    return -1;
  }
  P.dotw = d_dot_w; P.dotp = d_dot_partial;
  P.tiles_x = (W + TW - 1) / TW;
  P.tiles_plane = P.tiles_x * ((H + TH - 1) / TH);
  const long long nt_ll = (long long)P.tiles_plane * D;
  if (nt_ll > 0x7fffffffLL) { sd::set_error("sd_conv3_f16x3: too many tiles"); return -1; }
  // 32-bit offsets inside one halo tile's rows (buffer loads) and inside one output row (buffer stores)
  const long long row0 = (long long)(W >> (up0 & 1)) * stride0 * 4, row1 = d_src1 ? (long long)(W >> (up1 & 1)) * stride1 * 4 : 0;
  if ((HALO_H + 1) * (row0 > row1 ? row0 : row1) >= 0x7fffffffLL || (long long)W * c_out * 4 >= 0x7fffffffLL ||
      (d_res && (long long)W * res_stride * 4 >= 0x7fffffffLL)) {
    sd::set_error("sd_conv3_f16x3: an image row of %d pixels is too long for 32-bit offsets", W);
    return -1;
  }
  P.n_tiles = (int)nt_ll;
  P.groups = c_out / 32;
  static bool attr_set[16] = {};
  static int n_cu[16] = {};
  int dev = 0;
  SD_CHECK(hipGetDevice(&dev));
  const size_t lds = (size_t)2 * HWSUB_BYTES + HTILE_BYTES + 4 * 2048 + 128;    // 79.9 KiB: two workgroups per CU (the last 128 bytes: the fused head's weights)
  typedef void (*kern_t)(const Params);
  // [variant][one workgroup per CU]: plain, residual, fused head; then the split16 forms (in, out, in + out, in + fused head)
  static const kern_t kern[9][2] = {
      {k_conv3_f16<false, 2>, k_conv3_f16<false, 1>},
      {k_conv3_f16<true, 2>, k_conv3_f16<true, 1>},
      {k_conv3_f16<false, 2, true>, k_conv3_f16<false, 1, true>},
      {k_conv3_f16<false, 2, false, true, false>, k_conv3_f16<false, 1, false, true, false>},
      {k_conv3_f16<false, 2, false, false, true>, k_conv3_f16<false, 1, false, false, true>},
      {k_conv3_f16<false, 2, false, true, true>, k_conv3_f16<false, 1, false, true, true>},
      {k_conv3_f16<false, 2, true, true, false>, k_conv3_f16<false, 1, true, true, false>},
      {k_conv3_f16<false, 2, true, false, false, true>, k_conv3_f16<false, 1, true, false, false, true>},
      {k_conv3_f16<false, 2, true, true, false, true>, k_conv3_f16<false, 1, true, true, false, true>}};
  if (dev >= 16 || !attr_set[dev]) {
    for (int v = 0; v < 9; ++v)
      for (int w = 0; w < 2; ++w) SD_CHECK(hipFuncSetAttribute((const void*)kern[v][w], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev < 16) attr_set[dev] = true;
  }
  int cus = dev < 16 ? n_cu[dev] : 0;
  if (cus <= 0) {
    SD_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    if (dev < 16) n_cu[dev] = cus;
  }
  const int per_cu = sd::option(sd::OPT_CONV_F16_WGS) == 1 ? 1 : 2;          // (1: A/B probe of the one-workgroup-per-CU launch)
  long long blocks = (long long)(per_cu * cus / P.groups) * P.groups;
  if (blocks < P.groups) blocks = P.groups;
  const long long want = (long long)P.n_tiles * P.groups;
  if (blocks > want) blocks = want;
  const int variant = d_dot_w ? ((in_split ? 6 : 2) + (no_store ? (in_split ? 2 : 5) : 0)) : d_res ? 1 : (in_split ? (out_split ? 5 : 3) : (out_split ? 4 : 0));
  hipLaunchKernelGGL(kern[variant][per_cu == 1 ? 1 : 0], dim3((unsigned)blocks), dim3(THREADS), lds, s, P);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_conv3_f16x3_res_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                               int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias,
                                               const float* d_res, int res_stride, int c_out, int act, float* d_out, int* d_range_flag,
                                               void* stream_) {
  return conv3_f16x3_launch(d_src0, c0, stride0, up0, d_src1, c1, stride1, up1, D, H, W, kz, d_wpacked, d_bias, d_res, res_stride, c_out, act, d_out,
                            d_range_flag, nullptr, nullptr, stream_);
}

extern "C" int sd_conv3_f16x3_dot_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                               int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out, int act,
                                               float* d_out, int* d_range_flag, const float* d_dot_w, float* d_dot_partial, void* stream_) {
  if (!d_dot_w || !d_dot_partial) { sd::set_error("sd_conv3_f16x3_dot: head weights and partial-sum buffer required"); return -1; }
  return conv3_f16x3_launch(d_src0, c0, stride0, up0, d_src1, c1, stride1, up1, D, H, W, kz, d_wpacked, d_bias, nullptr, 0, c_out, act, d_out,
                            d_range_flag, d_dot_w, d_dot_partial, stream_);
}

extern "C" int sd_conv3_f16x3_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                                           int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out, int act,
                                           float* d_out, int* d_range_flag, void* stream_) {
  return sd_conv3_f16x3_res_ndhwc_device(d_src0, c0, stride0, up0, d_src1, c1, stride1, up1, D, H, W, kz, d_wpacked, d_bias, nullptr, 0,
                                         c_out, act, d_out, d_range_flag, stream_);
}

// The same layers over split16 tensors (conv3x3_layout.h): in_split16 != 0: every source holds, per pixel and 32-channel chunk, the
// 32 fp16 hi terms followed by the 32 fp16 lo' terms of its values (the 128 bytes the chunk's 32 floats would take: same strides, same
// addresses) -- what this kernel's consumer side derives from an f32 tensor anyway, made once by the producer; out_split16 != 0: the
// output is written in that form (after bias + activation; d_range_flag |= 2 when a value cannot be represented: |x| > 65504).
// Results are bit-identical to the f32-tensor entry points (the consumer's operands are the same 22 bits either way).
// d_dot_w / d_dot_partial (both or neither): the fused one-channel head of sd_conv3_f16x3_dot_ndhwc_device (f32 output only).
extern "C" int sd_conv3_f16x3_fmt_ndhwc_device(const float* d_src0, int c0, int up0, const float* d_src1, int c1, int up1, int D, int H, int W, int kz,
                                               const float* d_wpacked, const float* d_bias, int c_out, int act, float* d_out, int in_split16,
                                               int out_split16, int* d_range_flag, const float* d_dot_w, float* d_dot_partial, void* stream_) {
  return conv3_f16x3_launch(d_src0, c0, c0, up0, d_src1, c1, c1, up1, D, H, W, kz, d_wpacked, d_bias, nullptr, 0, c_out, act, d_out, d_range_flag,
                            d_dot_w, d_dot_partial, stream_, in_split16 ? 1 : 0, out_split16 ? 1 : 0);
}

// The layer's output on selected pixels only: d_out[r][0 .. c_out) = act(bias + conv(src))[d_rows[r]], d_rows = linear pixel indices into the
// [D][H][W] grid (int64), bit-identical to what sd_conv3_f16x3_*_ndhwc_device stores at those pixels.  One full-resolution source of c_in
// channels (f32, or split16 with in_split16 != 0); weights as for the dense entry points.  The sparse prediction path evaluates the
// features layer this way for the candidate pixels (10 % of a 2D tile, 1 % of a volume) after a dense pass that keeps only the probability
// head's partial sums (sd_conv3_f16x3_fmt_ndhwc_device with d_out == NULL), cf. stardist/models/base.py:553-610 (predict_sparse).
extern "C" int sd_conv3_f16x3_rows_device(const float* d_src, int c_in, int in_split16, int D, int H, int W, int kz, const float* d_wpacked,
                                          const float* d_bias, int c_out, int act, const long long* d_rows, long long n_rows, float* d_out,
                                          void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_rows <= 0) return 0;
  if (!d_src || !d_wpacked || !d_rows || !d_out || (act != 0 && act != 1) || sd_conv3_f16x3_packed_floats(c_in, c_out, kz) < 0 || (kz == 1 && D != 1) ||
      D <= 0 || H <= 0 || W <= 0 || (((uintptr_t)d_src | (uintptr_t)d_wpacked) & 15) || ((uintptr_t)d_rows & 7) || ((uintptr_t)d_out & 3)) {
    sd::set_error("sd_conv3_f16x3_rows: channel counts multiples of 32 (c_in <= 512), kz 1|3, act 0|1, aligned pointers");
    return -1;
  }
  RowsParams P;
  P.src = d_src; P.c_in = c_in; P.D = D; P.H = H; P.W = W; P.kz = kz; P.n_units = (c_in / 32) * kz;
  P.wp = d_wpacked; P.bias = d_bias; P.c_out = c_out; P.act = act; P.rows = d_rows; P.n_rows = n_rows; P.out = d_out;
  const long long waves = (n_rows + 31) / 32;
  const dim3 grid((unsigned)((waves + 3) / 4));
  const int groups = c_out / 32;
  for (int g0 = 0; g0 < groups;) {
    const int ng = groups - g0 >= 4 ? 4 : (groups - g0 >= 2 ? 2 : 1);
    P.g0 = g0;
    const size_t lds = (size_t)ng * HWSUB_BYTES;
    if (in_split16) {
      if (ng == 4) hipLaunchKernelGGL((k_conv3_f16_rows<4, true>), grid, dim3(256), lds, s, P);
      else if (ng == 2) hipLaunchKernelGGL((k_conv3_f16_rows<2, true>), grid, dim3(256), lds, s, P);
      else hipLaunchKernelGGL((k_conv3_f16_rows<1, true>), grid, dim3(256), lds, s, P);
    } else {
      if (ng == 4) hipLaunchKernelGGL((k_conv3_f16_rows<4, false>), grid, dim3(256), lds, s, P);
      else if (ng == 2) hipLaunchKernelGGL((k_conv3_f16_rows<2, false>), grid, dim3(256), lds, s, P);
      else hipLaunchKernelGGL((k_conv3_f16_rows<1, false>), grid, dim3(256), lds, s, P);
    }
    g0 += ng;
  }
  SD_LAUNCH_CHECK();
  return 0;
}
