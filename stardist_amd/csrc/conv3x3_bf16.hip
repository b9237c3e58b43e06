// conv3x3_bf16.hip -- the 3x3 / 3x3x3 convolution of conv3x3.hip with every f32 product evaluated as six bf16 x bf16 products on
// v_mfma_f32_32x32x16_bf16 (f32 accumulation).  OPT-IN (STARDIST_AMD_CONV=bf16x6 on the Python side); the exact-f32 kernel stays
// the default network path.
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
#include <stdlib.h>

#include "common.h"
#include "conv3x3_device.h"
#include "stardist_hip.h"

namespace {

using namespace sdconvdev;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct StageB {
  int lds[PRE_F4];               // byte offset of this thread's element n in plane 0 of the LDS tile
  unsigned goff[2][PRE_F4];
};

__device__ __forceinline__ void stage_init_b(const Params& P, StageB& st, int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    int e = tid + n * THREADS;
    e = e < TILE_F4 ? e : TILE_F4 - 1;
    int ty, tx, q4;
    stage_elem(e, ty, tx, q4);
    st.lds[n] = btile_store_off(ty, tx, 0, q4);
  }
  goff_init(P, st.goff, tid);
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

// f32 halo elements -> three bf16 planes in LDS (8 bytes per plane and element)
__device__ __forceinline__ void store_split(const StageB& st, char* __restrict__ tileB, const v4f (&pre)[PRE_F4], int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    if (n < PRE_F4 - 1 || tid < TILE_F4 - (PRE_F4 - 1) * THREADS) {
      unsigned pl[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) split3(pre[n][k], pl[k][0], pl[k][1], pl[k][2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x2 v;
        v.x = (pl[0][p] >> 16) | pl[1][p];
        v.y = (pl[2][p] >> 16) | pl[3][p];
        *(u32x2*)(tileB + st.lds[n] + p * 64) = v;
      }
    }
  }
}

// One sub-unit (row tap dy): 6 operand groups (dx, block); a group = 2 halo rows x 3 planes (A) + 3 planes (B) = 9 ds_read_b128 feeding
// 12 MFMAs (six plane pairs x two output rows).  The operands of group g+1 are read while the matrix cores work on group g.
__device__ __forceinline__ void compute_sub(const char* __restrict__ tileB, const char* __restrict__ w, int dy, f32x16 (&acc)[2], int wave, int i,
                                            int h) {
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
    // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bf16x8 bv = __builtin_bit_cast(bf16x8, B[buf][PB[k]]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[buf][0][PA[k]]), bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[buf][1][PA[k]]), bv, acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SD_LOAD_GROUP_B
}

// One workgroup per CU by LDS footprint (140 KiB): one wave per SIMD, so the whole register file is this wave's
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv3_bf16(const Params P) {
  extern __shared__ float4 smem4b[];
  // LDS map (bytes): two weight buffers (LDS-direct destinations, below 64 KiB) | halo tile, 3 bf16 planes | 4 x 8 KiB epilogue scratch
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
  {
    v4f pre[PRE_F4], wreg[WREG];
    weights_fetch(P, g, 0, 0, wreg, tid);
    halo_fetch(P, st.goff, q, 0, pre, tid);
    weights_store(W, wreg, tid);
    store_split(st, tileB, pre, tid);
  }
  __syncthreads();
  int wb = 0, pt = -1;
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
      // Order of the memory operations inside a unit (the compiler's wait counting is exact only while loads are the only pending
      // kind): dy 0 carries the previous tile's output stores and ends with a full drain; dy 1 issues its weight loads FIRST and the
      // next unit's halo loads after them, so the wait for the weights leaves the halo in flight; dy 2's wait for its weights
      // (younger than the halo loads) is where the halo has to have arrived -- two sub-units after it was requested.
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const bool lastdy = dy == 2, have_w = lastdy ? have : true;
        v4f wreg[WREG];
        // (always issued, from a valid address even when there is nothing left to fetch: a load under a condition would make the
        // compiler's merged wait counts for the loads around it collapse to "drain everything")
        weights_fetch(P, g, lastdy ? (have ? un : u) : u, lastdy ? (have ? 0 : dy) : dy + 1, wreg, tid);      // next sub-unit's weights
        __builtin_amdgcn_sched_barrier(0);                                                    // ... issued before anything below
        if (dy == 0 && u == 0 && pt >= 0) scratch_to_global<256>(P, scr, g, pt, wave, lane);   // previous tile's results -> HBM
        if (dy == 1) halo_fetch(P, st.goff, have ? tn : t, have ? un : u, pre, tid);          // next unit's halo
        __builtin_amdgcn_sched_barrier(0);
        compute_sub(tileB, W + wb * BWSUB_BYTES, dy, acc, wave, lane & 31, lane >> 5);
        if (have_w) weights_store(W + (wb ^ 1) * BWSUB_BYTES, wreg, tid);                    // the other buffer: nobody reads it now
        __syncthreads();
        wb ^= 1;
      }
      if (have) store_split(st, tileB, pre, tid);                  // (every wave is past the barrier behind the last sub-unit)
      __syncthreads();
    }
    // epilogue, first half: accumulators -> this wave's scratch in channels-last order (LDS only); the stores to HBM are issued
    // inside the next tile's first sub-unit (or after the loop)
    tile_to_scratch<256>(P, acc, scr, lane);
    pt = t;
  }
  if (pt >= 0) scratch_to_global<256>(P, scr, g, pt, wave, lane);
}

}  // namespace

extern "C" long long sd_conv3_bf16x6_packed_floats(int c_in, int c_out, int kz) {
  if ((kz != 1 && kz != 3) || c_in <= 0 || c_in % 32 || c_in > 32 * sdconv::MAX_CHUNKS || c_out <= 0 || c_out % 32) return -1;
  return (long long)(sdconv::bpacked_bytes(c_in, c_out, kz) / 4) + 4;      // + 16 bytes of zeros (the zero-padding source)
}

extern "C" int sd_conv3_bf16x6_pack_weights_host(const float* w, int c_in, int c_out, int kz, float* packed) {
  const long long n = sd_conv3_bf16x6_packed_floats(c_in, c_out, kz);
  if (!w || !packed || n < 0) {
    sd::set_error("sd_conv3_bf16x6_pack_weights: kz 1|3, c_in a multiple of 32 up to 256, c_out a multiple of 32");
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
  P.kind[0] = Src{d_src0, stride0, (up0 >> 2) & 1, (up0 >> 1) & 1, up0 & 1};
  P.kind[1] = d_src1 ? Src{d_src1, stride1, (up1 >> 2) & 1, (up1 >> 1) & 1, up1 & 1} : P.kind[0];
  for (int k = 0; k < MAX_CHUNKS; ++k) { P.chunk_kind[k] = 0; P.chunk_choff[k] = 0; }
  for (int k = 0; k < c0 / 32; ++k) { P.chunk_kind[nc] = 0; P.chunk_choff[nc++] = k * 32; }
  if (d_src1) for (int k = 0; k < c1 / 32; ++k) { P.chunk_kind[nc] = 1; P.chunk_choff[nc++] = k * 32; }
  P.D = D; P.H = H; P.W = W; P.kz = kz; P.n_units = nc * kz;
  P.zero = d_wpacked + (n_packed - 4);
  if (d_res && (res_stride < c_out || (res_stride & 3) || ((uintptr_t)d_res & 15))) {
    sd::set_error("sd_conv3_bf16x6: the residual needs 16-byte alignment and a stride >= c_out");
    return -1;
  }
  P.res = d_res; P.res_stride = res_stride;
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
    SD_CHECK(hipFuncSetAttribute((const void*)k_conv3_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
  hipLaunchKernelGGL(k_conv3_bf16, dim3((unsigned)blocks), dim3(THREADS), lds, s, P);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_conv3_bf16x6_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                                            int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out, int act,
                                            float* d_out, void* stream_) {
  return sd_conv3_bf16x6_res_ndhwc_device(d_src0, c0, stride0, up0, d_src1, c1, stride1, up1, D, H, W, kz, d_wpacked, d_bias, nullptr, 0,
                                          c_out, act, d_out, stream_);
}
