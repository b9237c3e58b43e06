// conv3x3.hip -- the network's 3x3 (2D) and 3x3x3 (3D) convolutions as hand-written implicit GEMMs on the f32 matrix cores.
//
// Replaces Keras Conv2D / Conv3D(kernel 3, 'same') + bias + ReLU of the csbdeep unet_block the reference builds in
// stardist/models/model2d.py:310-349 and model3d.py:360-399 (and the UpSampling + Concatenate in front of the first convolution
// of an up level):
//   * k_conv3<NT>: input channels in chunks of 32, each chunk from its own source tensor so that Concatenate([up, skip]) never
//     exists in memory; a source may be half resolution along any axis = nearest 2x up-sampling folded into the operand fetch;
//     32*NT output channels per workgroup, bias + activation in the epilogue.  A 3x3x3 convolution is the sum of three z planes
//     of 3x3 taps: the work list of an output tile is (chunk, kz) "units", each = one halo tile + one 9-tap weight block in LDS.
//     Layout and k order: conv3x3_layout.h.  Persistent workgroups (one per CU); the next unit's halo tile and weight block are
//     fetched (tile: into registers, weights: LDS-direct into the second weight buffer) while the matrix cores work on the current one.  Exact float32: every output is ONE fma chain in a fixed order (bias first), so results do not depend on
//     tiling, launch geometry or run.
//   * k_conv3_c1: the first layer (1 input channel, K = 9 or 27): HBM-write bound, plain FMAs.
// Bound: MFMA (f32: 64 FLOP/clk/SIMD); algorithmic bytes 4*(C_in + C_out) per pixel.
#include "common.h"
#include "conv3x3_device.h"
#include "stardist_hip.h"

namespace {

using namespace sdconvdev;

constexpr int WUNIT = 9 * 4 * 2 * 32 * 4;     // floats of one weight block (NT = 1)
constexpr int WMAIN = 8 * 4 * 2 * 32 * 4;     // ... of its taps 0..7

// Per-thread staging constants, computed once per kernel: where this thread's PRE_F4 float4 elements of a halo tile live in LDS,
// and their source offsets on interior tiles (goff_init, conv3x3_device.h)
struct Stage {
  int lds[PRE_F4];
  unsigned goff[2][PRE_F4];
};

__device__ __forceinline__ void stage_init(const Params& P, Stage& st, int tid) {
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n) {
    int e = tid + n * THREADS;
    e = e < TILE_F4 ? e : TILE_F4 - 1;
    int ty, tx, q4;
    stage_elem(e, ty, tx, q4);
    st.lds[n] = tile_off(ty, tx, q4 * 4);
  }
  goff_init(P, st.goff, tid);
}

// Halo tile of unit u of output tile t -> registers (halo_fetch), the unit's weight block -> LDS.
template <int NT>
__device__ __forceinline__ void load_unit(const Params& P, const Stage& st, int g, int t, int u, v4f (&pre)[PRE_F4], v4f& w8,
                                          float* __restrict__ Wnext, int tid, int wave) {
  halo_fetch(P, st.goff, t, u, pre, tid);
  // the unit's weight block (36 KiB, tap-major): taps 0..7 = 32 KiB global -> LDS without passing through registers
  // (global_load_lds_dwordx4: LDS destination = wave-uniform base + lane * 16; wave w, instruction n moves the n*4+w-th KiB); both
  // 32 KiB buffers lie in the first 64 KiB of LDS, whatever width of M0 the DMA honours.  Tap 8 (4 KiB) rides along in a register.
  const float* wsrc = P.wp + ((size_t)g * P.n_units + u) * WUNIT;
  const int lane = tid & 63;
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int kib = n * 4 + wave;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + kib * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(Wnext + kib * 256), 16, 0, 0);
  }
  w8 = ((const v4f*)(wsrc + WMAIN))[tid];
}

__device__ __forceinline__ void store_unit(const Stage& st, float* __restrict__ tileL, float* __restrict__ W8next, const v4f (&pre)[PRE_F4],
                                           const v4f& w8, int tid) {
  ((v4f*)W8next)[tid] = w8;
#pragma unroll
  for (int n = 0; n < PRE_F4; ++n)
    if (n < PRE_F4 - 1 || tid < TILE_F4 - (PRE_F4 - 1) * THREADS) *(v4f*)(tileL + st.lds[n]) = pre[n];
}

// the two accumulator tiles of a wave as the plain array store_tile takes (NT == 1)
template <int NT>
__device__ __forceinline__ const f32x16 (&acc_rows(const f32x16 (&acc)[2][NT]))[2] {
  static_assert(NT == 1, "");
  return reinterpret_cast<const f32x16(&)[2]>(acc);
}

__device__ __forceinline__ float comp(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// One unit: 12 operand groups (dx, j); a group = the four halo rows the wave's two output rows touch (A) and the three taps dy of
// that column (B) = 7 ds_read_b128 feeding 24 MFMAs.  The operands of group g+1 are read while the matrix cores work on group g
// (two register sets).
template <int NT>
__device__ __forceinline__ void compute_unit(const float* __restrict__ tileL, const float* __restrict__ wl, const float* __restrict__ w8, f32x16 (&acc)[2][NT],
                                             int wave, int i, int h) {
  float4 A[2][4], B[2][3][NT];
#define SD_LOAD_GROUP(gi, buf)                                                                          \
  do {                                                                                                   \
    const int dx_ = (gi) >> 2, j_ = (gi) & 3;                                                            \
    _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) A[buf][rr] = *(const float4*)(tileL + a_off(wave * 2, rr, dx_, j_, i, h)); \
    _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                     \
      _Pragma("unroll") for (int ct = 0; ct < NT; ++ct) B[buf][dy][ct] = *(const float4*)((dy * 3 + dx_ == 8 ? w8 - WMAIN : wl) + wl_off(dy * 3 + dx_, j_, h, ct, i, NT)); \
  } while (0)
  SD_LOAD_GROUP(0, 0);
#pragma unroll
  for (int gi = 0; gi < 12; ++gi) {
    const int buf = gi & 1;
    if (gi + 1 < 12) SD_LOAD_GROUP(gi + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);        // keep the reads of the next group ahead of this group's MFMAs
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
          const float bv = comp(B[buf][dy][ct], e);
          acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A[buf][dy], e), bv, acc[0][ct], 0, 0, 0);
          acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(A[buf][dy + 1], e), bv, acc[1][ct], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef SD_LOAD_GROUP
}

// One step of the pipeline: start fetching the next unit (halo tile -> registers, weight block -> LDS-direct into `wnext`), run the
// matrix cores on the current unit (tile + `wcur`), then move the fetched tile into LDS.  The LDS regions are disjoint; the
// __restrict__ qualifiers carry that to the waitcnt insertion, which otherwise orders every ds_read behind the in-flight
// LDS-direct loads (vmcnt(0) in front of the first MFMA group = no overlap).
template <int NT>
__device__ __forceinline__ void unit_step(const Params& P, const Stage& st, int g, bool have, int tn, int un, float* __restrict__ tileL,
                                          const float* __restrict__ wcur, const float* __restrict__ w8cur, float* __restrict__ wnext,
                                          float* __restrict__ w8next, f32x16 (&acc)[2][NT], int tid, int wave, int i, int h) {
  v4f pre[PRE_F4], w8 = {0.f, 0.f, 0.f, 0.f};
  if (have) load_unit<NT>(P, st, g, tn, un, pre, w8, wnext, tid, wave);
  __builtin_amdgcn_sched_barrier(0);
  compute_unit<NT>(tileL, wcur, w8cur, acc, wave, i, h);
  __syncthreads();
  if (have) store_unit(st, tileL, w8next, pre, w8, tid);
  __syncthreads();
}

// One workgroup per CU by LDS footprint (120 KiB): one wave per SIMD, so the whole register file is this wave's
template <int NT>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) k_conv3(const Params P) {
  extern __shared__ float4 smem4[];
  // LDS map (floats): [0, 2 WMAIN) the two weight buffers' taps 0..7 (DMA destinations, below 64 KiB), then their tap-8 blocks, then the halo tile
  float* Wl = (float*)smem4;
  float* W8 = Wl + 2 * WMAIN;
  float* tileL = W8 + 2 * (WUNIT - WMAIN);
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // wave-uniform: row bases and DMA destinations stay scalar
  int g, q, Q;
  wg_slot(P, g, q, Q);
  if (q >= P.n_tiles) return;
  float bias_r[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) bias_r[ct] = P.bias ? P.bias[g * 32 * NT + ct * 32 + i] : 0.f;
  static_assert(NT == 1, "LDS map and weight staging are written for 32 output channels per workgroup");
  Stage st;
  stage_init(P, st, tid);
  {
    v4f pre[PRE_F4], w8;
    load_unit<NT>(P, st, g, q, 0, pre, w8, Wl, tid, wave);
    store_unit(st, tileL, W8, pre, w8, tid);
  }
  __syncthreads();                                     // (the compiler drains the LDS-direct loads before the barrier)
  int wb = 0;
  for (int t = q; t < P.n_tiles; t += Q) {
    f32x16 acc[2][NT];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][ct][r] = bias_r[ct];
    for (int u = 0; u < P.n_units; ++u) {
      const bool last = u == P.n_units - 1;
      const int tn = last ? t + Q : t, un = last ? 0 : u + 1;
      unit_step<NT>(P, st, g, tn < P.n_tiles, tn, un, tileL, Wl + wb * WMAIN, W8 + wb * (WUNIT - WMAIN), Wl + (wb ^ 1) * WMAIN,
                    W8 + (wb ^ 1) * (WUNIT - WMAIN), acc, tid, wave, i, h);
      wb ^= 1;
    }
    // epilogue (store_tile): scratch = the weight buffer the matrix cores have just finished with (every wave is past the barrier
    // behind that compute); a wave uses exactly the eight 1-KiB chunks n*4 + wave that its own LDS-direct loads refill in the next
    // step, so no barrier is needed -- program order within the wave is enough.  The stores drain while the next tile is computed.
    store_tile<1024>(P, acc_rows(acc), Wl + (wb ^ 1) * WMAIN + wave * 256, g, t, wave, lane);
  }
}

// first layer: one input channel.  thread = (pixel, 4 output channels); w4 is [taps][c_out/4] float4 (tap-major, taps = 9 kz)
template <int KZ>
__global__ void __launch_bounds__(256) k_conv3_c1(const float* __restrict__ x, int D, int H, int W, const float4* __restrict__ w4,
                                                  const float4* __restrict__ bias4, int c4, int act, float4* __restrict__ out) {
  const long long n = (long long)D * H * W * c4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const int q = (int)(idx % c4);
    const long long pix = idx / c4;
    const int xx = (int)(pix % W);
    const long long rest = pix / W;
    const int y = (int)(rest % H), z = (int)(rest / H);
    float4 s = bias4 ? bias4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9 * KZ; ++tap) {
      const int gz = z + (KZ == 3 ? tap / 9 - 1 : 0), gy = y + (tap % 9) / 3 - 1, gx = xx + tap % 3 - 1;
      const float v = (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)gz * H + gy) * W + gx] : 0.f;
      const float4 wv = w4[tap * c4 + q];
      s.x = __builtin_fmaf(v, wv.x, s.x); s.y = __builtin_fmaf(v, wv.y, s.y);
      s.z = __builtin_fmaf(v, wv.z, s.z); s.w = __builtin_fmaf(v, wv.w, s.w);
    }
    if (act == 1) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    out[idx] = s;
  }
}

// first layer, 32 output channels (the networks' case): thread = one pixel, all 32 channels.  The weights are wave-uniform (scalar
// loads, SGPR operands of the FMAs), the 9 / 27 input taps are read once per pixel instead of once per channel quad, and the
// 128 bytes a thread produces are transposed through LDS (row pitch 36 floats) so that a wave writes its 64 pixels as eight
// fully coalesced 1-KiB stores.  HBM-write bound (128 B/pixel).
// SPLIT: the output is written as a split16 tensor (conv3x3_layout.h; the form the split-fp16 kernel's consumer side reads without
// splitting): a lane then takes 8 consecutive channels of one pixel from the transposition and stores their 8 hi terms and 8 lo' terms
// (16 bytes each, 64 bytes apart); *flag |= 2 when a value is beyond the fp16 range.
typedef _Float16 c1_f16x2 __attribute__((ext_vector_type(2)));
typedef float c1_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int c1_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void c1_split2(float a, float b, unsigned& hi, unsigned& lo, float& amax) {      // split2_pair of conv3x3_f16.hip
  const c1_f32x2 x = {a, b};
  const c1_f16x2 h = __builtin_convertvector(x, c1_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const c1_f32x2 r = (x - __builtin_convertvector(h, c1_f32x2)) * 2048.f;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, c1_f16x2));
  amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)));
}
template <int KZ, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_conv3_c1x32(const float* __restrict__ x, int D, int H, int W, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int act, float* __restrict__ out, int* __restrict__ flag = nullptr) {
  __shared__ float rows[4][64 * 36];
  const long long n_pix = (long long)D * H * W;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long wbase = ((long long)blockIdx.x * 4 + wave) * 64;       // first pixel of this wave
  if (wbase >= n_pix) return;
  const long long pix = wbase + lane;
  const bool live = pix < n_pix;
  const long long pc = live ? pix : n_pix - 1;
  const int xx = (int)(pc % W);
  const long long rest = pc / W;
  const int y = (int)(rest % H), z = (int)(rest / H);
  float acc[32];
#pragma unroll
  for (int co = 0; co < 32; ++co) acc[co] = bias ? bias[co] : 0.f;
  // rows of three taps: the loop over (dz, dy) stays rolled so that only one row's 96 weights occupy scalar registers at a time
#pragma unroll 1
  for (int r3 = 0; r3 < 3 * KZ; ++r3) {
    const int gz = z + (KZ == 3 ? r3 / 3 - 1 : 0), gy = y + r3 % 3 - 1;
    const bool rin = gz >= 0 && gz < D && gy >= 0 && gy < H;
    const float* xr = x + ((size_t)min(max(gz, 0), D - 1) * H + min(max(gy, 0), H - 1)) * W;
    const float* wr = w + r3 * 96;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int gx = xx + dx - 1;
      const float v = (rin && gx >= 0 && gx < W) ? xr[min(max(gx, 0), W - 1)] : 0.f;
#pragma unroll
      for (int co = 0; co < 32; ++co) acc[co] = __builtin_fmaf(v, wr[dx * 32 + co], acc[co]);
    }
  }
  float* row = rows[wave] + lane * 36;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v4f o = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
    if (act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *(v4f*)(row + q * 4) = o;
  }
  // (only this wave touches rows[wave]: LDS operations of one wave complete in order)
  if (SPLIT) {
    const int pr = lane >> 2, oc = lane & 3;
    float amax = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* src = rows[wave] + (k * 16 + pr) * 36 + oc * 8;
      const v4f va = *(const v4f*)src, vb = *(const v4f*)(src + 4);
      unsigned hw[4], lw[4];
      c1_split2(va.x, va.y, hw[0], lw[0], amax); c1_split2(va.z, va.w, hw[1], lw[1], amax);
      c1_split2(vb.x, vb.y, hw[2], lw[2], amax); c1_split2(vb.z, vb.w, hw[3], lw[3], amax);
      const long long p2 = wbase + k * 16 + pr;
      if (p2 < n_pix) {
        *(c1_u32x4*)(out + p2 * 32 + oc * 4) = c1_u32x4{hw[0], hw[1], hw[2], hw[3]};
        *(c1_u32x4*)(out + p2 * 32 + 16 + oc * 4) = c1_u32x4{lw[0], lw[1], lw[2], lw[3]};
      }
    }
    if (flag && !(amax <= 65504.f)) atomicOr(flag, 2);
    return;
  }
  const int px = lane >> 3, c4 = lane & 7;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const v4f o = *(const v4f*)(rows[wave] + (k * 8 + px) * 36 + c4 * 4);
    const long long p2 = wbase + k * 8 + px;
    if (p2 < n_pix) *(v4f*)(out + p2 * 32 + c4 * 4) = o;
  }
}

template <int NT>
int launch_conv(const Params& P, hipStream_t s) {
  static bool attr_set[16] = {};
  static int n_cu[16] = {};
  int dev = 0;
  SD_CHECK(hipGetDevice(&dev));
  const size_t lds = (size_t)(TILE_FLOATS + 2 * wunit_floats(NT)) * sizeof(float);
  if (dev >= 16 || !attr_set[dev]) {
    SD_CHECK(hipFuncSetAttribute((const void*)k_conv3<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev < 16) attr_set[dev] = true;
  }
  int cus = dev < 16 ? n_cu[dev] : 0;
  if (cus <= 0) {
    SD_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    if (dev < 16) n_cu[dev] = cus;
  }
  long long blocks = (long long)(cus / P.groups) * P.groups;      // one persistent workgroup per CU, a whole number per group
  if (blocks < P.groups) blocks = P.groups;
  const long long want = (long long)P.n_tiles * P.groups;
  if (blocks > want) blocks = want;
  hipLaunchKernelGGL((k_conv3<NT>), dim3((unsigned)blocks), dim3(THREADS), lds, s, P);
  SD_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" long long sd_conv3_packed_floats(int c_in, int c_out, int kz) {
  if (kz != 1 && kz != 3) return -1;
  if (c_in == 1) return c_out % 4 == 0 && c_out > 0 ? 9LL * kz * c_out : -1;
  if (c_in <= 0 || c_in % 32 || c_in > 32 * sdconv::MAX_CHUNKS || c_out <= 0 || c_out % 32) return -1;
  return (long long)sdconv::packed_floats(c_in, c_out, kz) + 4;      // + 16 bytes of zeros (the kernel's zero padding source)
}

extern "C" int sd_conv3_pack_weights_host(const float* w, int c_in, int c_out, int kz, float* packed) {
  if (!w || !packed || sd_conv3_packed_floats(c_in, c_out, kz) < 0) {
    sd::set_error("sd_conv3_pack_weights: kz 1|3, c_in 1 (c_out %% 4 == 0) or a multiple of 32 up to 512 (c_out %% 32 == 0)");
    return -1;
  }
  if (c_in == 1) {
    const int taps = 9 * kz;
    for (int tap = 0; tap < taps; ++tap)
      for (int co = 0; co < c_out; ++co) packed[tap * c_out + co] = w[(size_t)co * taps + tap];
    return 0;
  }
  sdconv::pack_weights(w, c_in, c_out, kz, packed);
  for (int k = 0; k < 4; ++k) packed[sdconv::packed_floats(c_in, c_out, kz) + k] = 0.f;
  return 0;
}

extern "C" int sd_conv3_res_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                                         int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, const float* d_res,
                                         int res_stride, int c_out, int act, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  const int c_in = c0 + (d_src1 ? c1 : 0);
  if (!d_src0 || !d_wpacked || !d_out || (act != 0 && act != 1) || sd_conv3_packed_floats(c_in, c_out, kz) < 0 || (kz == 1 && D != 1) ||
      (((uintptr_t)d_src0 | (uintptr_t)d_src1 | (uintptr_t)d_wpacked | (uintptr_t)d_out | (uintptr_t)d_bias) & 15)) {
    sd::set_error("sd_conv3_ndhwc: unsupported channel counts (%d + %d -> %d), kz, act or misaligned pointers", c0, d_src1 ? c1 : 0, c_out);
    return -1;
  }
  if (d_res && (c_in == 1 || res_stride < c_out || (res_stride & 3) || ((uintptr_t)d_res & 15))) {
    sd::set_error("sd_conv3_ndhwc: the residual needs a 32-channel-chunk layer, 16-byte alignment and a stride >= c_out");
    return -1;
  }
  if (c_in == 1) {
    if (d_src1 || up0 || stride0 != 1) { sd::set_error("sd_conv3_ndhwc: the one-channel layer takes one full-resolution source"); return -1; }
    if (c_out == 32) {
      const long long waves = ((long long)D * H * W + 63) / 64;
      const dim3 g1((unsigned)((waves + 3) / 4));
      if (kz == 1) hipLaunchKernelGGL((k_conv3_c1x32<1, false>), g1, dim3(256), 0, s, d_src0, D, H, W, d_wpacked, d_bias, act, d_out, (int*)nullptr);
      else hipLaunchKernelGGL((k_conv3_c1x32<3, false>), g1, dim3(256), 0, s, d_src0, D, H, W, d_wpacked, d_bias, act, d_out, (int*)nullptr);
      SD_LAUNCH_CHECK();
      return 0;
    }
    const int c4 = c_out / 4;
    const long long n = (long long)D * H * W * c4;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (kz == 1)
      hipLaunchKernelGGL(k_conv3_c1<1>, dim3((unsigned)blocks), dim3(256), 0, s, d_src0, D, H, W, (const float4*)d_wpacked, (const float4*)d_bias, c4,
                         act, (float4*)d_out);
    else
      hipLaunchKernelGGL(k_conv3_c1<3>, dim3((unsigned)blocks), dim3(256), 0, s, d_src0, D, H, W, (const float4*)d_wpacked, (const float4*)d_bias, c4,
                         act, (float4*)d_out);
    SD_LAUNCH_CHECK();
    return 0;
  }
  const int ups[2] = {up0, d_src1 ? up1 : 0};
  for (int k = 0; k < 2; ++k) {
    const int up = ups[k];
    if (up < 0 || up > 7 || ((up & 1) && (W & 1)) || ((up & 2) && (H & 1)) || ((up & 4) && (D & 1))) {
      sd::set_error("sd_conv3_ndhwc: up is a bit mask (1: x, 2: y, 4: z); an up-sampled axis needs an even output size");
      return -1;
    }
  }
  if ((c0 % 32) || (d_src1 && (c1 % 32)) || stride0 < c0 || (stride0 & 3) || (d_src1 && (stride1 < c1 || (stride1 & 3)))) {
    sd::set_error("sd_conv3_ndhwc: sources must hold multiples of 32 channels, strides multiples of 4 floats");
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
  P.zero = d_wpacked + sdconv::packed_floats(c_in, c_out, kz);
  P.res = d_res; P.res_stride = res_stride;
  P.dotw = nullptr; P.dotp = nullptr;
  P.wp = d_wpacked; P.bias = d_bias; P.out = d_out; P.c_out = c_out; P.act = act;
  P.tiles_x = (W + TW - 1) / TW;
  P.tiles_plane = P.tiles_x * ((H + TH - 1) / TH);
  const long long nt_ll = (long long)P.tiles_plane * D;
  if (nt_ll > 0x7fffffffLL) { sd::set_error("sd_conv3_ndhwc: too many tiles"); return -1; }
  P.n_tiles = (int)nt_ll;
  const int nt = sdconv::nt_for(c_out);
  P.groups = c_out / (32 * nt);
  (void)nt;
  return launch_conv<1>(P, s);
}

// first layer (one input channel -> 32) with the output written as a split16 tensor: what sd_conv3_f16x3_fmt_ndhwc_device reads with in_split16
extern "C" int sd_conv3_c1x32_split16_device(const float* d_src, int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int act,
                                             float* d_out, int* d_range_flag, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  if (!d_src || !d_wpacked || !d_out || (act != 0 && act != 1) || (kz != 1 && kz != 3) || (kz == 1 && D != 1) ||
      (((uintptr_t)d_wpacked | (uintptr_t)d_out | (uintptr_t)d_bias) & 15) || ((uintptr_t)d_range_flag & 3)) {
    sd::set_error("sd_conv3_c1x32_split16: kz 1|3, act 0|1, 16-byte aligned pointers");
    return -1;
  }
  const long long waves = ((long long)D * H * W + 63) / 64;
  const dim3 g1((unsigned)((waves + 3) / 4));
  if (kz == 1) hipLaunchKernelGGL((k_conv3_c1x32<1, true>), g1, dim3(256), 0, s, d_src, D, H, W, d_wpacked, d_bias, act, d_out, d_range_flag);
  else hipLaunchKernelGGL((k_conv3_c1x32<3, true>), g1, dim3(256), 0, s, d_src, D, H, W, d_wpacked, d_bias, act, d_out, d_range_flag);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_conv3_ndhwc_device(const float* d_src0, int c0, int stride0, int up0, const float* d_src1, int c1, int stride1, int up1,
                                     int D, int H, int W, int kz, const float* d_wpacked, const float* d_bias, int c_out, int act,
                                     float* d_out, void* stream_) {
  return sd_conv3_res_ndhwc_device(d_src0, c0, stride0, up0, d_src1, c1, stride1, up1, D, H, W, kz, d_wpacked, d_bias, nullptr, 0, c_out, act,
                                   d_out, stream_);
}
