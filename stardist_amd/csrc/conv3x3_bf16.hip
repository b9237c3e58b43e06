// conv3x3_bf16.hip -- the 3x3 / 3x3x3 convolution of conv3x3.hip with every f32 product evaluated as six bf16 x bf16 products on
// v_mfma_f32_32x32x16_bf16 (f32 accumulation).  The network's default kernel (models/unet.py conv_mode(); STARDIST_AMD_CONV=hand
// selects the exact-f32 kernel of conv3x3.hip).
//
// Why: the exact kernel runs at the f32-MFMA roof (conv3x3.hip: 116-122 TFLOP/s of a 157 spec / ~126 sustained), and bf16 MFMA is
// 16x that rate.  With x = hi + mid + lo (three bf16 terms, the remainders exact in f32) the six leading cross products reproduce the
// f32 product to ~2^-24 (conv3x3_layout.h) -- on the reference's 2D network the outputs are as close to a float64 evaluation as the
// plain f32 ones (max |dprob| 3.3e-7 vs 4.9e-7, max rel |ddist| 1.2e-6 vs 1.1e-6) -- at 6/16 of the matrix-core time.
//
// Same decomposition as the exact kernel (8 x 32 output tile, 32 output channels per workgroup, (chunk, kz) units, persistent
// workgroups, halo tile prefetched through registers, transposed epilogue: conv3x3_device.h); differences:
//   * the f32 activations are split into the three bf16 planes while they are written to LDS (208 bytes per halo pixel);
//   * a unit's weights (3 planes x 18 KiB) do not fit next to that tile twice, so a unit is walked as three SUB-UNITS (row tap dy),
//     each with its own 18-KiB weight block, double buffered, staged through registers;
//   * per (column tap, 16-channel block): 6 + 3 ds_read_b128 feed 12 MFMAs.
//
// One workgroup per CU (137 KiB of LDS) means one wave per SIMD: nothing hides a wave's own latencies, and an MFMA occupies the matrix
// pipe for 32 cycles while an ALU instruction issues in 4.  Everything that is not a matrix instruction is therefore placed BEHIND one
// (sched_group_barrier patterns in compute_sub): the address arithmetic of the next unit's halo in the first sub-unit, the halo loads
// themselves in the second, the f32 -> 3 x bf16 split of the arrived halo (v_cvt_pk_bf16_f32) in the third; after the last barrier of a
// unit only the LDS stores of the split planes are left.  Per-tile quantities (tile coordinates: two integer divisions; source
// offsets) are computed once per tile, and every scalar the in-shadow code needs sits in an SGPR before the first MFMA (a scalar load
// from the kernel arguments inside the sequence waits on lgkmcnt(0), i.e. for all LDS operand reads in flight).
// tools/conv_phase_profile.hip stamps the phases with s_memtime; profiles/r03_conv_bf16_phases_*.txt hold the before / after tables
// (16 600 -> 13 300 cycles per unit of 6 912 MFMA cycles on the 256^3 32 -> 32 layer).
#include <stdlib.h>

#include "common.h"
#include "conv3x3_device.h"
#include "stardist_hip.h"

// phase timing for tools/conv_phase_profile.hip (never defined in the library build)
#ifdef SD_CONV_PROFILE
__device__ unsigned long long g_conv_prof[16];   // [0] total, [1..14] phases, [15] units
#define PROF_DECL unsigned long long pf_t = __builtin_amdgcn_s_memtime(), pf_acc[14] = {}; const unsigned long long pf_t0 = pf_t; unsigned long long pf_units = 0
#define PROF(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_acc[k] += n_ - pf_t; pf_t = n_; } while (0)
#define PROF_UNIT() (++pf_units)
#define PROF_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PROF_END() do { if (threadIdx.x == 0) { atomicAdd(&g_conv_prof[0], __builtin_amdgcn_s_memtime() - pf_t0); \
  for (int k_ = 0; k_ < 14; ++k_) atomicAdd(&g_conv_prof[1 + k_], pf_acc[k_]); atomicAdd(&g_conv_prof[15], pf_units); } } while (0)
#else
#define PROF_DECL
#define PROF(k)
#define PROF_UNIT()
#define PROF_DRAIN()
#define PROF_END()
#endif

namespace {

using namespace sdconvdev;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct StageB {
  int lds[PRE_F4];               // byte offset of this thread's element n in plane 0 of the LDS tile
  unsigned goff[2][PRE_F4];
  unsigned tyx[PRE_F4];
};

__device__ __forceinline__ void stage_init_b(const Params& P, StageB& st, int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    int e = tid + n * THREADS;
    e = e < TILE_F4 ? e : TILE_F4 - 1;
    int ty, tx, q4;
    stage_elem_b(e, ty, tx, q4);
    st.lds[n] = btile_store_off(ty, tx, 0, q4);
  }
  goff_init<true>(P, st.goff, tid);
  tyx_init<true>(st.tyx, tid);
}

// weights of sub-unit (u, dy) of group g: 18 KiB = 1152 x 16 bytes through registers (five per thread).  Not LDS-direct here: with
// LDS-direct loads in flight the compiler drains EVERY outstanding load at each barrier, and a sub-unit is too short (~1 us) to cover
// the halo prefetch that is in flight next to it; register loads are waited for individually, oldest first -- the weight loads of
// a unit's first sub-unit are issued BEFORE the halo loads, so waiting for them leaves the halo in flight for two sub-units.
constexpr int WREG = (BWSUB_BYTES / 16 + THREADS - 1) / THREADS;
__device__ __forceinline__ void weights_fetch(const Params& P, int g, int u, int dy, v4f (&wreg)[WREG], int tid) {
  const v4f* wsrc = (const v4f*)((const char*)P.wp + (((size_t)g * P.n_units + u) * 3 + dy) * BWSUB_BYTES);
#pragma unroll
  for (int n = 0; n < WREG; ++n) {
    const int e = tid + n * THREADS;
    wreg[n] = wsrc[e < BWSUB_BYTES / 16 ? e : BWSUB_BYTES / 16 - 1];
  }
}
__device__ __forceinline__ void weights_store(char* __restrict__ wnext, const v4f (&wreg)[WREG], int tid) {
#pragma unroll
  for (int n = 0; n < WREG; ++n) {
    const int e = tid + n * THREADS;
    if (e < BWSUB_BYTES / 16) ((v4f*)wnext)[e] = wreg[n];
  }
}

// two f32 values -> their three bf16 terms, packed (a in the low half): v_cvt_pk_bf16_f32 rounds to nearest even like split3 of
// conv3x3_layout.h (which packs the weights on the host); the remainders are exact in f32
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 x = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 r = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xffff0000u)};
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  const f32x2 r2 = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}
// f32 halo elements -> three bf16 planes (8 bytes per plane and element) ...
__device__ __forceinline__ void split_elem(const v4f x, u32x2 (&pl)[3]) {
  unsigned a[3], b[3];
  split3_pair(x.x, x.y, a[0], a[1], a[2]);
  split3_pair(x.z, x.w, b[0], b[1], b[2]);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    asm volatile("" : "+v"(a[p]), "+v"(b[p]));                // computed HERE (the compiler would sink the arithmetic to the stores)
    pl[p].x = a[p]; pl[p].y = b[p];
  }
}
// ... and into the LDS tile
__device__ __forceinline__ void store_planes(const StageB& st, char* __restrict__ tileB, const u32x2 (&pk)[PRE_F4][3], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n)
    if (n < PRE_F4 - 1 || tid < TILE_F4 - (PRE_F4 - 1) * THREADS) {
#pragma unroll
      for (int p = 0; p < 3; ++p) *(u32x2*)(tileB + st.lds[n] + p * 64) = pk[n][p];
    }
}
__device__ __forceinline__ void store_split(const StageB& st, char* __restrict__ tileB, const v4f (&pre)[PRE_F4], int tid) {
  u32x2 pk[PRE_F4][3];
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) split_elem(pre[n], pk[n]);
  store_planes(st, tileB, pk, tid);
}

// One sub-unit (row tap dy): 6 operand groups (dx, block); a group = 2 halo rows x 3 planes (A) + 3 planes (B) = 9 ds_read_b128 feeding
// 12 MFMAs (six plane pairs x two output rows).  The operands of group g+1 are read while the matrix cores work on group g.
// `extra(gi)`: vector-ALU work that does not depend on the matrix instructions (the next halo's addresses, the split of the halo that
// has arrived); NV of its instructions are placed behind each MFMA, where the wave -- alone on its SIMD -- would otherwise only wait
// for the matrix pipe (8 passes = 32 cycles per MFMA, an ALU instruction issues in 4).
struct NoExtra { __device__ __forceinline__ void operator()(int) const {} };
template <int NV, int NLD, class Extra>      // per MFMA: NV vector-ALU instructions; per operand group: NLD global loads
__device__ __forceinline__ void compute_sub(const char* __restrict__ tileB, const char* __restrict__ w, int dy, f32x16 (&acc)[2], int wave, int i,
                                            int h, Extra extra) {
  u32x4 A[2][2][3], B[2][3];
  const char* arow = tileB + (wave * 2 + dy) * HALO_W * BPIX;
#define SD_LOAD_GROUP_B(gi, buf)                                                                                        \
  do {                                                                                                                   \
    const int dx_ = (gi) >> 1, b_ = (gi) & 1;                                                                            \
    _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                        \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) A[buf][p][pl] = *(const u32x4*)(arow + btile_off(p, i + dx_, pl, b_, h)); \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) B[buf][pl] = *(const u32x4*)(w + bw_off(dx_, b_, pl, h, i));        \
  } while (0)
  SD_LOAD_GROUP_B(0, 0);
#pragma unroll
  for (int gi = 0; gi < 6; ++gi) {
    const int buf = gi & 1;
    if (gi + 1 < 6) SD_LOAD_GROUP_B(gi + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    extra(gi);
    // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bf16x8 bv = __builtin_bit_cast(bf16x8, B[buf][PB[k]]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[buf][0][PA[k]]), bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[buf][1][PA[k]]), bv, acc[1], 0, 0, 0);
    }
    if (NV > 0 || NLD > 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA ...
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                      // ... then NV vector-ALU instructions
        if (NLD > 0 && k % (12 / NLD) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // ... or a global load
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SD_LOAD_GROUP_B
}

// One workgroup per CU by LDS footprint (140 KiB): one wave per SIMD, so the whole register file is this wave's
template <bool RES>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv3_bf16(const Params P) {
  extern __shared__ float4 smem4b[];
  // LDS map (bytes): two weight buffers of one sub-unit each | halo tile, 3 bf16 planes | 4 x 8 KiB epilogue scratch
  char* W = (char*)smem4b;
  char* tileB = W + 2 * BWSUB_BYTES;
  float* scratch = (float*)(tileB + BTILE_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int g, q, Q;
  wg_slot(P, g, q, Q);
  if (q >= P.n_tiles) return;
  const float bias_r = P.bias ? P.bias[g * 32 + (lane & 31)] : 0.f;
  StageB st;
  stage_init_b(P, st, tid);
  TileAddr Tc, Tn;                                      // the tile being computed, the tile whose first unit is fetched next
  tile_addr(P, q, Tc);
  Tn = Tc;
  {
    v4f pre[PRE_F4], wreg[WREG];
    weights_fetch(P, g, 0, 0, wreg, tid);
    halo_fetch_at(P, st.goff, st.tyx, Tc, 0, pre);
    weights_store(W, wreg, tid);
    store_split(st, tileB, pre, tid);
  }
  __syncthreads();
  PROF_DECL;
  int wb = 0, pt = -1, ptz = 0, pty = 0, ptx = 0;       // previous tile (its results are still in the scratch): number, z plane, first pixel
  float* scr = scratch + wave * 2048;                  // this wave's private transpose scratch (8 chunks of 1 KiB)
  for (int t = q; t < P.n_tiles; t += Q) {
    f32x16 acc[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = bias_r;
    for (int u = 0; u < P.n_units; ++u) {
      const bool last = u == P.n_units - 1;
      const int tn = last ? t + Q : t, un = last ? 0 : u + 1;
      const bool have = tn < P.n_tiles;
      v4f pre[PRE_F4];
      halo_gptr addr[PRE_F4];
      u32x2 pk[PRE_F4][3];
      v4f vv[8];                                                    // data of the previous tile's output stores (u == 0)
      // Order of the memory operations inside a unit (the compiler's wait counting is exact only while loads are the only pending
      // kind): dy 0 carries the previous tile's output stores and ends with a full drain, its matrix instructions hide the address
      // arithmetic of the next unit's halo; dy 1 issues its weight loads FIRST and the halo loads after them, so the wait for the
      // weights leaves the halo in flight; dy 2 splits the halo elements into their bf16 planes (registers) in the shadow of its last
      // four operand groups, as they arrive; after its barrier only the LDS stores are left.
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const bool lastdy = dy == 2, have_w = lastdy ? have : true;
        v4f wreg[WREG];
        // (always issued, from a valid address even when there is nothing left to fetch: a load under a condition would make the
        // compiler's merged wait counts for the loads around it collapse to "drain everything")
        weights_fetch(P, g, lastdy ? (have ? un : u) : u, lastdy ? (have ? 0 : dy) : dy + 1, wreg, tid);      // next sub-unit's weights
        __builtin_amdgcn_sched_barrier(0);                                                    // ... issued before anything below
        if (dy == 0 && u == 0 && pt >= 0) scratch_to_global<256, RES>(P, scr, g, ptz, pty, ptx, wave, lane, vv);   // previous tile's results -> HBM
        if (dy == 0 && last && have) tile_addr(P, tn, Tn);                                     // (once per tile)
        __builtin_amdgcn_sched_barrier(0);
        PROF(dy);
        const char* wcur = W + wb * BWSUB_BYTES;
        if (dy == 0) {
          const TileAddr T = tile_select(last && have, Tc, Tn);
          const HaloScalars hs = halo_scalars(P, halo_base(P, T, have ? un : u), T);
          compute_sub<2, 0>(tileB, wcur, dy, acc, wave, lane & 31, lane >> 5, [&](int gi) {
#pragma unroll
            for (int n = gi * 2; n < gi * 2 + 2; ++n)
              if (n < PRE_F4) addr[n] = halo_addr_one(hs, st.goff[0][n], st.goff[1][n], st.tyx[n]);
          });
        } else if (dy == 1) {
          compute_sub<0, 2>(tileB, wcur, dy, acc, wave, lane & 31, lane >> 5, [&](int gi) {      // the next unit's halo loads
#pragma unroll
            for (int n = gi * 2; n < gi * 2 + 2; ++n)
              if (n < PRE_F4) pre[n] = *(const __attribute__((address_space(1))) v4f*)addr[n];
          });
        } else {
          compute_sub<5, 0>(tileB, wcur, dy, acc, wave, lane & 31, lane >> 5, [&](int gi) {
#pragma unroll
            for (int n = (gi - 2) * 3; n < (gi - 2) * 3 + 3; ++n)
              if (gi >= 2 && n < PRE_F4) split_elem(pre[n], pk[n]);
          });
        }
        if (dy == 0 && u == 0 && pt >= 0) hold_stores(vv);         // (their registers stay untouched while the matrix cores run)
        PROF(dy == 0 ? 3 : 11 + dy);
        if (have_w) weights_store(W + (wb ^ 1) * BWSUB_BYTES, wreg, tid);                    // the other buffer: nobody reads it now
        // (a use on every path: without it the loads look pending around the loop's back edge on the `!have` path, and the compiler
        // guards every later write to these registers with a wait that also holds back the output stores' registers)
        if (lastdy) {
#pragma unroll
          for (int n = 0; n < WREG; ++n) asm volatile("" ::"v"(wreg[n]));
        }
        PROF(4 + dy);
        __syncthreads();
        PROF(7);
        wb ^= 1;
      }
      PROF(8);
      if (have) store_planes(st, tileB, pk, tid);                  // (every wave is past the barrier behind the last sub-unit)
      PROF(9);
      __syncthreads();
      PROF(10);
      PROF_UNIT();
    }
    // epilogue, first half: accumulators -> this wave's scratch in channels-last order (LDS only); the stores to HBM are issued
    // inside the next tile's first sub-unit (or after the loop)
    tile_to_scratch<256>(P, acc, scr, lane);
    PROF(11);
    pt = t; ptz = Tc.tz; pty = Tc.ty0 + 1; ptx = Tc.tx0 + 1;
    Tc = Tn;
  }
  if (pt >= 0) {
    v4f vv[8];
    scratch_to_global<256, RES>(P, scr, g, ptz, pty, ptx, wave, lane, vv);
  }
  PROF_END();
}

}  // namespace

extern "C" long long sd_conv3_bf16x6_packed_floats(int c_in, int c_out, int kz) {
  if ((kz != 1 && kz != 3) || c_in <= 0 || c_in % 32 || c_in > 32 * sdconv::MAX_CHUNKS || c_out <= 0 || c_out % 32) return -1;
  return (long long)(sdconv::bpacked_bytes(c_in, c_out, kz) / 4) + 4;      // + 16 bytes of zeros (the zero-padding source)
}

extern "C" int sd_conv3_bf16x6_pack_weights_host(const float* w, int c_in, int c_out, int kz, float* packed) {
  const long long n = sd_conv3_bf16x6_packed_floats(c_in, c_out, kz);
  if (!w || !packed || n < 0) {
    sd::set_error("sd_conv3_bf16x6_pack_weights: kz 1|3, c_in a multiple of 32 up to 512, c_out a multiple of 32");
    return -1;
  }
  sdconv::pack_weights_bf16(w, c_in, c_out, kz, (unsigned short*)packed);
  for (int k = 0; k < 4; ++k) packed[n - 4 + k] = 0.f;
  return 0;
}

extern "C" int sd_conv3_bf16x6_res_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1,
                                                int up1, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias,
                                                const float* d_res, int res_stride, int c_out, int act, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  const int c_in = c0 + (d_src1 ? c1 : 0);
  const long long n_packed = sd_conv3_bf16x6_packed_floats(c_in, c_out, kz);
  if (!d_src0 || !d_wpacked || !d_out || (act != 0 && act != 1) || n_packed < 0 || (kz == 1 && D != 1) ||
      (((uintptr_t)d_src0 | (uintptr_t)d_src1 | (uintptr_t)d_wpacked | (uintptr_t)d_out | (uintptr_t)d_bias) & 15)) {
    sd::set_error("sd_conv3_bf16x6: unsupported channel counts (%d + %d -> %d), kz, act or misaligned pointers", c0, d_src1 ? c1 : 0, c_out);
    return -1;
  }
  const int ups[2] = {up0, d_src1 ? up1 : 0};
  for (int k = 0; k < 2; ++k)
    if (ups[k] < 0 || ups[k] > 7 || ((ups[k] & 1) && (W & 1)) || ((ups[k] & 2) && (H & 1)) || ((ups[k] & 4) && (D & 1))) {
      sd::set_error("sd_conv3_bf16x6: up is a bit mask (1: x, 2: y, 4: z); an up-sampled axis needs an even output size");
      return -1;
    }
  if ((c0 % 32) || (d_src1 && (c1 % 32)) || stride0 < c0 || (stride0 & 3) || (d_src1 && (stride1 < c1 || (stride1 & 3)))) {
    sd::set_error("sd_conv3_bf16x6: sources must hold multiples of 32 channels, strides multiples of 4 floats");
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
    sd::set_error("sd_conv3_bf16x6: the residual needs 16-byte alignment and a stride >= c_out");
    return -1;
  }
  P.res = d_res; P.res_stride = res_stride;
  P.dotw = nullptr; P.dotp = nullptr;
  P.wp = d_wpacked; P.bias = d_bias; P.out = d_out; P.c_out = c_out; P.act = act;
  P.tiles_x = (W + TW - 1) / TW;
  P.tiles_plane = P.tiles_x * ((H + TH - 1) / TH);
  const long long nt_ll = (long long)P.tiles_plane * D;
  if (nt_ll > 0x7fffffffLL) { sd::set_error("sd_conv3_bf16x6: too many tiles"); return -1; }
  P.n_tiles = (int)nt_ll;
  P.groups = c_out / 32;
  static bool attr_set[16] = {};
  static int n_cu[16] = {};
  int dev = 0;
  SD_CHECK(hipGetDevice(&dev));
  const size_t lds = (size_t)2 * BWSUB_BYTES + BTILE_BYTES + 4 * 8192;     // 137 KiB
  if (dev >= 16 || !attr_set[dev]) {
    SD_CHECK(hipFuncSetAttribute((const void*)k_conv3_bf16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SD_CHECK(hipFuncSetAttribute((const void*)k_conv3_bf16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev < 16) attr_set[dev] = true;
  }
  int cus = dev < 16 ? n_cu[dev] : 0;
  if (cus <= 0) {
    SD_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    if (dev < 16) n_cu[dev] = cus;
  }
  long long blocks = (long long)(cus / P.groups) * P.groups;
  if (blocks < P.groups) blocks = P.groups;
  const long long want = (long long)P.n_tiles * P.groups;
  if (blocks > want) blocks = want;
  if (d_res) hipLaunchKernelGGL(k_conv3_bf16<true>, dim3((unsigned)blocks), dim3(THREADS), lds, s, P);
  else hipLaunchKernelGGL(k_conv3_bf16<false>, dim3((unsigned)blocks), dim3(THREADS), lds, s, P);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_conv3_bf16x6_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                                            int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out, int act,
                                            float* d_out, void* stream_) {
  return sd_conv3_bf16x6_res_ndhwc_device(d_src0, c0, stride0, up0, d_src1, c1, stride1, up1, D, H, W, kz, d_wpacked, d_bias, nullptr, 0,
                                          c_out, act, d_out, stream_);
}
