// unet_ops.hip -- element-wise epilogue of the network convolutions.
// The reference's Keras Conv layers (csbdeep unet_block / resnet_block, called from stardist/models/model2d.py:310-349 and
// model3d.py:360-447) apply bias and activation inside the layer; MIOpen's convolution leaves both to the framework, which
// costs two extra passes over the activation tensor.  This kernel does both in one in-place pass (HBM-bound: 8 B/element).
#include "common.h"

#include "stardist_hip.h"
#include <math.h>

namespace {

// channels-last: x is [n_pix][C], C % 4 == 0
template <bool ADD>
__global__ void __launch_bounds__(256) k_bias_act_cl4(float4* __restrict__ x, const float4* __restrict__ addend, const float4* __restrict__ bias,
                                                      long long n4, int C4, int act) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = x[i];
    if (ADD) { const float4 a = addend[i]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
    const float4 b = bias[(int)(i % C4)];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    x[i] = v;
  }
}

// generic: x is [n_outer][C][inner]
template <bool ADD>
__global__ void __launch_bounds__(256) k_bias_act(float* __restrict__ x, const float* __restrict__ addend, const float* __restrict__ bias, long long n,
                                                  int C, long long inner, int act) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = x[i];
    if (ADD) v += addend[i];
    v += bias[(int)((i / inner) % C)];
    if (act == 1) v = fmaxf(v, 0.f);
    x[i] = v;
  }
}

// ---- features epilogue + probability head ----------------------------------------------------------------------------------
// channels-last [n_pix][C], C = 4*LPP: LPP lanes hold one pixel.  out = act(in + bias) (in place when out == in) and, when w is
// given, dot[p] = (sigmoid)(sum_c out[p][c] * w[c] + wb): the 1x1 convolution to ONE channel (the object-probability head,
// model2d.py:338-341 / model3d.py:436-439) done while the features are in registers instead of a second pass over them.
// Fixed summation order (4 products per lane in channel order, then an xor butterfly), the same for every caller.
// bias == nullptr: nothing is added; out == nullptr: the features are only read (they already hold bias + activation, written by
// the hand-written convolution's epilogue) and just the dot product is produced.
template <int LPP>
__global__ void __launch_bounds__(256) k_bias_act_dot(const float4* __restrict__ in, float4* __restrict__ out, const float4* __restrict__ bias,
                                                      long long n_pix, int act, const float4* __restrict__ w, const float* __restrict__ wb,
                                                      int sigm, float* __restrict__ dot) {
  const int s = threadIdx.x % LPP;
  const long long groups = (long long)gridDim.x * (256 / LPP);
  const float4 b = bias ? bias[s] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
  float w0 = 0.f;
  if (w) { wv = w[s]; w0 = wb ? wb[0] : 0.f; }
  for (long long p = (long long)blockIdx.x * (256 / LPP) + threadIdx.x / LPP; p < n_pix; p += groups) {
    float4 v = in[p * LPP + s];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (out) out[p * LPP + s] = v;
    if (w) {
      float d = v.x * wv.x;
      d += v.y * wv.y; d += v.z * wv.z; d += v.w * wv.w;
#pragma unroll
      for (int o = LPP / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      if (s == 0) {
        d += w0;
        dot[p] = sigm ? 1.f / (1.f + expf(-d)) : d;
      }
    }
  }
}

// ---- distance head on selected rows -----------------------------------------------------------------------------------------
// out[i][r] = max(clamp, bias[r] + sum_k feat[rows[i]][k] * W[r][k]): the 1x1 convolution to n_rays channels (model2d.py:342-343 /
// model3d.py:440-441) as an fp32-MFMA GEMM over the rows that are asked for -- every pixel (rows == nullptr: the dense head) or the
// candidate pixels only (the sparse prediction path, base.py:553-610: no dense distance tensor is ever written).
// One wave = 32 rows x all columns: v_mfma_f32_32x32x2_f32, lane (i = lane&31, h = lane>>5) feeds row i with the channels of its
// half h*C/2.. in order, so a row's result is ONE fixed fma chain (bias first) whatever other rows share the tile: the dense and the
// sparse path agree bit for bit.  W^T is staged once per workgroup in LDS as [k][32*CT (+1 pad)].
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CT>
__global__ void __launch_bounds__(256) k_head_rows(const float* __restrict__ feat, int C, const long long* __restrict__ rows, long long n_rows,
                                                   const float* __restrict__ W, const float* __restrict__ bias, int R, float clampv,
                                                   float* __restrict__ out) {
  extern __shared__ float Wl[];
  constexpr int RP = CT * 32 + 1;
  for (int e = threadIdx.x; e < C * CT * 32; e += 256) {
    const int c = e / C, k = e - c * C;                       // coalesced read of W[c][k], transposed store
    Wl[k * RP + c] = c < R ? W[e] : 0.f;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int KH = C >> 1;
  float bcol[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) bcol[ct] = (bias && ct * 32 + i < R) ? bias[ct * 32 + i] : 0.f;
  const long long n_tiles = (n_rows + 31) >> 5;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
    const long long ri = tile * 32 + i;
    const long long src = ri < n_rows ? (rows ? rows[ri] : ri) : (rows ? rows[0] : 0);
    const float4* ap = (const float4*)(feat + src * C + h * KH);
    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = bcol[ct];
    const float* wl = Wl + (size_t)h * KH * RP + i;
    for (int j4 = 0; j4 < KH / 4; ++j4) {
      const float4 a = ap[j4];
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wk = wl + (size_t)(j4 * 4 + e) * RP;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wk[ct * 32], acc[ct], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = ct * 32 + i;
      if (col < R) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (orow < n_rows) out[orow * R + col] = fmaxf(acc[ct][r], clampv);
        }
      }
    }
  }
}

// ---- max pooling, channels-last ----------------------------------------------------------------------------------------------
// Keras MaxPooling2D / 3D(pool), padding 'valid', stride = pool (csbdeep unet_block between the levels; the grid > 1 stages in front
// of the U-Net, stardist/models/model2d.py:317-325): out[zo][yo][xo][c] = max over the pz x py x px window.  One thread per (output
// pixel, channel quad); 64-bit indexing (a 32-channel 416^3 level has more than 2^31 elements).  HBM-bound: 4 B read per input element.
__global__ void __launch_bounds__(256) k_maxpool_cl4(const float4* __restrict__ in, float4* __restrict__ out, long long n_out4, int C4, int Ho, int Wo,
                                                     int H, int W, int pz, int py, int px) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_out4; idx += stride) {
    const int q = (int)(idx % C4);
    long long pix = idx / C4;
    const int xo = (int)(pix % Wo); pix /= Wo;
    const int yo = (int)(pix % Ho);
    const long long zo = pix / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dz = 0; dz < pz; ++dz)
      for (int dy = 0; dy < py; ++dy) {
        const float4* row = in + (((zo * pz + dz) * H + ((long long)yo * py + dy)) * W + (long long)xo * px) * C4 + q;
        for (int dx = 0; dx < px; ++dx) {
          const float4 v = row[(long long)dx * C4];
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    out[idx] = m;
  }
}

// ---- split16 tensors (conv3x3_layout.h) -----------------------------------------------------------------------------------------
// Per pixel and 32-channel chunk: 8 elements of 16 bytes, element p * 4 + o = the 8 fp16 terms of plane p (0: hi, 1: lo') of channels
// o * 8 .. o * 8 + 7; value = hi + lo' * 2^-11 (exact in f32).
typedef _Float16 sp_h8 __attribute__((ext_vector_type(8)));
typedef float sp_f8 __attribute__((ext_vector_type(8)));

// Max pooling of a split16 tensor: the element of the window whose VALUE is largest is copied (both of its terms).  x -> (hi, lo') is
// monotone, so this is the pair the consumer would derive from max(x): the pooled split16 tensor equals split(maxpool(f32 tensor))
// bit for bit.  Two pairs with the same value (hi + half a step, hi' - half a step; or -0 and +0) can only come from x < x': the larger
// hi wins, in the total order that puts -0 below +0 (what v_max_f32 does with the f32 values).  Thread per (output pixel, chunk, octet).
__device__ __forceinline__ unsigned sp_order_key(_Float16 h) {
  const unsigned b = __builtin_bit_cast(unsigned short, h);
  return (b & 0x8000u) ? (~b & 0xFFFFu) : (b | 0x8000u);
}
__global__ void __launch_bounds__(256) k_maxpool_split16(const sp_h8* __restrict__ in, sp_h8* __restrict__ out, long long n_out, int C32, int Ho, int Wo,
                                                         int H, int W, int pz, int py, int px) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int per_pix = C32 * 4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_out; idx += stride) {
    const int q = (int)(idx % per_pix), c = q >> 2, o = q & 3;
    long long pix = idx / per_pix;
    const int xo = (int)(pix % Wo); pix /= Wo;
    const int yo = (int)(pix % Ho);
    const long long zo = pix / Ho;
    // per channel: the pair with the largest (value, hi) in the total order of their bit patterns, as ONE 48-bit integer key
    sp_h8 bh = {}, bl = {};
    unsigned long long bk[8] = {};
    for (int dz = 0; dz < pz; ++dz)
      for (int dy = 0; dy < py; ++dy) {
        const sp_h8* row = in + ((((zo * pz + dz) * H + ((long long)yo * py + dy)) * W + (long long)xo * px) * C32 + c) * 8 + o;
        for (int dx = 0; dx < px; ++dx) {
          const sp_h8 h = row[(long long)dx * C32 * 8], l = row[(long long)dx * C32 * 8 + 4];
          const sp_f8 v = __builtin_convertvector(h, sp_f8) + __builtin_convertvector(l, sp_f8) * 4.8828125e-4f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float vk = v[k];                          // (a scalar copy: __builtin_bit_cast on the vector ELEMENT compiled to element 0's bits for every k)
            const unsigned fb = __float_as_uint(vk);
            const unsigned fk = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
            const unsigned long long key = (((unsigned long long)fk << 16) | (unsigned long long)sp_order_key(h[k])) + 1ull;      // (> 0: beats the empty slot)
            const bool take = key > bk[k];
            bk[k] = take ? key : bk[k];
            bh[k] = take ? h[k] : bh[k];
            bl[k] = take ? l[k] : bl[k];
          }
        }
      }
    sp_h8* dst = out + (idx / per_pix * C32 + c) * 8 + o;
    dst[0] = bh; dst[4] = bl;
  }
}

// f32 -> split16 (the producer-side split as its own pass: tests, and a tensor that reaches a split-fp16 layer from an f32 producer);
// thread per (pixel, chunk, octet)
__global__ void __launch_bounds__(256) k_split16_pack(const float4* __restrict__ in, sp_h8* __restrict__ out, long long n, int* __restrict__ flag) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  float amax = 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const long long chunk = idx >> 2;
    const int o = (int)(idx & 3);
    const float4 a = in[chunk * 8 + o * 2], b = in[chunk * 8 + o * 2 + 1];
    const sp_f8 x = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const sp_h8 h = __builtin_convertvector(x, sp_h8);
    const sp_f8 r = (x - __builtin_convertvector(h, sp_f8)) * 2048.f;
    out[chunk * 8 + o] = h;
    out[chunk * 8 + 4 + o] = __builtin_convertvector(r, sp_h8);
#pragma unroll
    for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(x[k]));
  }
  if (flag && !(amax <= 65504.f)) atomicOr(flag, 2);
}
// split16 -> f32: hi + lo' * 2^-11 (the 22 bits a split-fp16 layer reads; for a consumer that takes f32 tensors only)
__global__ void __launch_bounds__(256) k_split16_unpack(const sp_h8* __restrict__ in, float4* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
    const long long chunk = idx >> 2;
    const int o = (int)(idx & 3);
    const sp_f8 v = __builtin_convertvector(in[chunk * 8 + o], sp_f8) + __builtin_convertvector(in[chunk * 8 + 4 + o], sp_f8) * 4.8828125e-4f;
    out[chunk * 8 + o * 2] = make_float4(v[0], v[1], v[2], v[3]);
    out[chunk * 8 + o * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// UpSampling (nearest, x2 along the axes of `up`) + Concatenate([up-sampled a, b]) as ONE channels-last tensor: the coverage path of an
// up level whose channel counts the fused 3x3 kernels do not take (csbdeep unet_block, e.g. n_filter_base = 48); thread per
// (output pixel, channel quad)
__global__ void __launch_bounds__(256) k_upcat_cl4(const float4* __restrict__ a, int ca4, int shz, int shy, int shx, const float4* __restrict__ b, int cb4,
                                                   int H, int W, long long n_out4, float4* __restrict__ out) {
  const int C4 = ca4 + cb4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int Ha = H >> shy, Wa = W >> shx;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n_out4; idx += stride) {
    const int q = (int)(idx % C4);
    long long pix = idx / C4;
    const int x = (int)(pix % W); pix /= W;
    const int y = (int)(pix % H);
    const long long z = pix / H;
    if (q < ca4) out[idx] = a[(((z >> shz) * Ha + (y >> shy)) * (long long)Wa + (x >> shx)) * ca4 + q];
    else out[idx] = b[((z * H + y) * (long long)W + x) * cb4 + (q - ca4)];
  }
}

}  // namespace

extern "C" int sd_upcat_ndhwc_device(const float* d_a, int ca, int up, const float* d_b, int cb, int D, int H, int W, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  if (!d_a || !d_b || !d_out || ca <= 0 || cb <= 0 || (ca % 4) || (cb % 4) || up < 0 || up > 7 || ((up & 1) && (W & 1)) || ((up & 2) && (H & 1)) || ((up & 4) && (D & 1)) ||
      (((uintptr_t)d_a | (uintptr_t)d_b | (uintptr_t)d_out) & 15)) {
    sd::set_error("sd_upcat_ndhwc: channel counts must be multiples of 4, pointers 16-byte aligned, up a bit mask (1: x, 2: y, 4: z) over even output sizes");
    return -1;
  }
  const long long n4 = (long long)D * H * W * ((ca + cb) / 4);
  long long blocks = (n4 + 255) / 256;
  if (blocks > 262144) blocks = 262144;
  hipLaunchKernelGGL(k_upcat_cl4, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)d_a, ca / 4, (up >> 2) & 1, (up >> 1) & 1, up & 1, (const float4*)d_b, cb / 4, H, W, n4,
                     (float4*)d_out);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_maxpool_ndhwc_device(const float* d_in, int n_channels, int D, int H, int W, int pz, int py, int px, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (pz <= 0 || py <= 0 || px <= 0 || n_channels <= 0 || n_channels % 4 || !d_in || !d_out || (((uintptr_t)d_in | (uintptr_t)d_out) & 15)) {
    sd::set_error("sd_maxpool_ndhwc: channels must be a multiple of 4, pointers 16-byte aligned, pool sizes positive");
    return -1;
  }
  const int Do = D / pz, Ho = H / py, Wo = W / px;
  if (Do <= 0 || Ho <= 0 || Wo <= 0) return 0;
  const long long n4 = (long long)Do * Ho * Wo * (n_channels / 4);
  long long blocks = (n4 + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_maxpool_cl4, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)d_in, (float4*)d_out, n4, n_channels / 4, Ho, Wo, H, W, pz, py, px);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_maxpool_split16_ndhwc_device(const float* d_in, int n_channels, int D, int H, int W, int pz, int py, int px, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (pz <= 0 || py <= 0 || px <= 0 || n_channels <= 0 || n_channels % 32 || !d_in || !d_out || (((uintptr_t)d_in | (uintptr_t)d_out) & 15)) {
    sd::set_error("sd_maxpool_split16_ndhwc: channels must be a multiple of 32, pointers 16-byte aligned, pool sizes positive");
    return -1;
  }
  const int Do = D / pz, Ho = H / py, Wo = W / px;
  if (Do <= 0 || Ho <= 0 || Wo <= 0) return 0;
  const long long n = (long long)Do * Ho * Wo * (n_channels / 8);
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_maxpool_split16, dim3((unsigned)blocks), dim3(256), 0, s, (const sp_h8*)d_in, (sp_h8*)d_out, n, n_channels / 32, Ho, Wo, H, W, pz, py, px);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_split16_pack_device(const float* d_in, long long n_pix, int n_channels, float* d_out, int* d_range_flag, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_pix <= 0) return 0;
  if (n_channels <= 0 || n_channels % 32 || !d_in || !d_out || d_in == d_out || (((uintptr_t)d_in | (uintptr_t)d_out) & 15) || ((uintptr_t)d_range_flag & 3)) {
    sd::set_error("sd_split16_pack: channels must be a multiple of 32, pointers 16-byte aligned and distinct");
    return -1;
  }
  const long long n = n_pix * (n_channels / 8);
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_split16_pack, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)d_in, (sp_h8*)d_out, n, d_range_flag);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_split16_unpack_device(const float* d_in, long long n_pix, int n_channels, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_pix <= 0) return 0;
  if (n_channels <= 0 || n_channels % 32 || !d_in || !d_out || d_in == d_out || (((uintptr_t)d_in | (uintptr_t)d_out) & 15)) {
    sd::set_error("sd_split16_unpack: channels must be a multiple of 32, pointers 16-byte aligned and distinct");
    return -1;
  }
  const long long n = n_pix * (n_channels / 8);
  long long blocks = (n + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_split16_unpack, dim3((unsigned)blocks), dim3(256), 0, s, (const sp_h8*)d_in, (float4*)d_out, n);
  SD_LAUNCH_CHECK();
  return 0;
}

static int bias_act_impl(float* d_x, const float* d_add, const float* d_bias, long long n_outer, int n_channels, long long inner, int act, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_outer <= 0 || inner <= 0) return 0;
  if (n_channels <= 0 || (act != 0 && act != 1)) { sd::set_error("sd_bias_act: bad arguments"); return -1; }
  const long long n = n_outer * (long long)n_channels * inner;
  if (inner == 1 && n_channels % 4 == 0 && ((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_bias & 15) == 0 && ((uintptr_t)d_add & 15) == 0) {
    const long long n4 = n / 4;
    const long long blocks = (n4 + 255) / 256;
    const dim3 g((unsigned int)(blocks < 65536 ? blocks : 65536));
    if (d_add) hipLaunchKernelGGL(k_bias_act_cl4<true>, g, dim3(256), 0, s, (float4*)d_x, (const float4*)d_add, (const float4*)d_bias, n4, n_channels / 4, act);
    else hipLaunchKernelGGL(k_bias_act_cl4<false>, g, dim3(256), 0, s, (float4*)d_x, (const float4*)nullptr, (const float4*)d_bias, n4, n_channels / 4, act);
  } else {
    const long long blocks = (n + 255) / 256;
    const dim3 g((unsigned int)(blocks < 65536 ? blocks : 65536));
    if (d_add) hipLaunchKernelGGL(k_bias_act<true>, g, dim3(256), 0, s, d_x, d_add, d_bias, n, n_channels, inner, act);
    else hipLaunchKernelGGL(k_bias_act<false>, g, dim3(256), 0, s, d_x, (const float*)nullptr, d_bias, n, n_channels, inner, act);
  }
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_bias_act_device(float* d_x, const float* d_bias, long long n_outer, int n_channels, long long inner, int act, void* stream_) {
  return bias_act_impl(d_x, nullptr, d_bias, n_outer, n_channels, inner, act, stream_);
}

extern "C" int sd_add_bias_act_device(float* d_x, const float* d_addend, const float* d_bias, long long n_outer, int n_channels, long long inner, int act,
                                      void* stream_) {
  if (!d_addend) { sd::set_error("sd_add_bias_act: null addend"); return -1; }
  return bias_act_impl(d_x, d_addend, d_bias, n_outer, n_channels, inner, act, stream_);
}

extern "C" int sd_bias_act_dot_device(const float* d_in, float* d_out, const float* d_bias, long long n_pix, int n_channels, int act,
                                      const float* d_w, const float* d_wbias, int sigmoid, float* d_dot, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_pix <= 0) return 0;
  const int lpp = n_channels / 4;
  if (n_channels % 4 || (lpp != 8 && lpp != 16 && lpp != 32 && lpp != 64) || (act != 0 && act != 1) || !d_in || (!d_out && !d_w) || (d_w && !d_dot) ||
      (((uintptr_t)d_in | (uintptr_t)d_out | (uintptr_t)d_bias | (uintptr_t)d_w) & 15)) {
    sd::set_error("sd_bias_act_dot: n_channels must be 32, 64, 128 or 256, pointers 16-byte aligned, act 0|1");
    return -1;
  }
  const long long per_block = 256 / lpp;
  long long blocks = (n_pix + per_block - 1) / per_block;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 g((unsigned int)blocks), b(256);
#define SD_BAD(L) hipLaunchKernelGGL(k_bias_act_dot<L>, g, b, 0, s, (const float4*)d_in, (float4*)d_out, (const float4*)d_bias, n_pix, act, (const float4*)d_w, d_wbias, sigmoid, d_dot)
  if (lpp == 8) SD_BAD(8); else if (lpp == 16) SD_BAD(16); else if (lpp == 32) SD_BAD(32); else SD_BAD(64);
#undef SD_BAD
  SD_LAUNCH_CHECK();
  return 0;
}

// second stage of the fused one-channel head (conv3x3_f16.hip store_tile<.., DOT> wrote, per pixel, the LPP = n_channels / 4 per-lane terms
// of k_bias_act_dot): the terms of a pixel are added in the order of that kernel's xor butterfly -- d[s] += d[s ^ o] for o = LPP / 2 .. 1,
// i.e. first across the 32-channel groups (o >= 8, in registers: term s = g * 8 + c of lane c), then across the 8 lanes of a group -- then
// the head's bias and (sigm) the logistic function: bit-identical to k_bias_act_dot on the same features.
template <int G>
__global__ void __launch_bounds__(256) k_dot_combine(const float* __restrict__ part, long long n_pix, const float* __restrict__ wb, int sigm,
                                                     float* __restrict__ out) {
  const int c = threadIdx.x & 7;
  const float w0 = wb ? wb[0] : 0.f;
  const long long per = (long long)gridDim.x * 32;
  for (long long p0 = (long long)blockIdx.x * 32; p0 < n_pix; p0 += per) {
    const long long p = p0 + (threadIdx.x >> 3);
    float v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = p < n_pix ? part[(size_t)p * (G * 8) + g * 8 + c] : 0.f;
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
      float t[G];
#pragma unroll
      for (int g = 0; g < G; ++g) t[g] = v[g] + v[g ^ o];
#pragma unroll
      for (int g = 0; g < G; ++g) v[g] = t[g];
    }
    float d = v[0];
    d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
    if (c == 0 && p < n_pix) {
      d += w0;
      out[p] = sigm ? 1.f / (1.f + expf(-d)) : d;
    }
  }
}

extern "C" int sd_dot_combine_device(const float* d_partial, int groups, long long n_pix, const float* d_wbias, int sigmoid, float* d_out, void* stream_) {
  if (n_pix <= 0) return 0;
  if (!d_partial || !d_out || (groups != 1 && groups != 2 && groups != 4 && groups != 8)) { sd::set_error("sd_dot_combine: groups must be 1, 2, 4 or 8"); return -1; }
  long long blocks = (n_pix + 31) / 32;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const dim3 gd((unsigned int)blocks), bd(256);
  hipStream_t s = (hipStream_t)stream_;
  if (groups == 1) hipLaunchKernelGGL(k_dot_combine<1>, gd, bd, 0, s, d_partial, n_pix, d_wbias, sigmoid, d_out);
  else if (groups == 2) hipLaunchKernelGGL(k_dot_combine<2>, gd, bd, 0, s, d_partial, n_pix, d_wbias, sigmoid, d_out);
  else if (groups == 4) hipLaunchKernelGGL(k_dot_combine<4>, gd, bd, 0, s, d_partial, n_pix, d_wbias, sigmoid, d_out);
  else hipLaunchKernelGGL(k_dot_combine<8>, gd, bd, 0, s, d_partial, n_pix, d_wbias, sigmoid, d_out);
  SD_LAUNCH_CHECK();
  return 0;
}

extern "C" int sd_head_rows_device(const float* d_feat, int n_channels, const long long* d_rows, long long n_rows, const float* d_w, const float* d_bias,
                                   int n_out, float clamp_min, float* d_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (n_rows <= 0 || n_out <= 0) return 0;
  const int ct = (n_out + 31) / 32;
  const size_t lds = (size_t)n_channels * (ct * 32 + 1) * sizeof(float);
  if (n_channels < 8 || n_channels % 8 || ct > 4 || lds > 64 * 1024 || !d_feat || !d_w || !d_out || ((uintptr_t)d_feat & 15)) {
    sd::set_error("sd_head_rows: n_channels must be a multiple of 8, n_out <= 128, n_channels * n_out <= 16k, feat 16-byte aligned");
    return -1;
  }
  const long long tiles = (n_rows + 31) / 32;
  long long blocks = (tiles + 3) / 4;
  if (blocks > 256 * 3) blocks = 256 * 3;
  const dim3 g((unsigned int)blocks), b(256);
#define SD_HR(T) hipLaunchKernelGGL(k_head_rows<T>, g, b, lds, s, d_feat, n_channels, d_rows, n_rows, d_w, d_bias, n_out, clamp_min, d_out)
  if (ct == 1) SD_HR(1); else if (ct == 2) SD_HR(2); else if (ct == 3) SD_HR(3); else SD_HR(4);
#undef SD_HR
  SD_LAUNCH_CHECK();
  return 0;
}
