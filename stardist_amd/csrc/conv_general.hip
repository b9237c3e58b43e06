// conv_general.hip -- every convolution of the reference's networks that is NOT a stride-1 3x3(x3) layer over 32-channel chunks
// (those are conv3x3.hip), as deterministic implicit GEMMs on the f32 matrix cores:
//   * the ResNet backbone of StarDist3D (stardist/models/model3d.py:400-447 + csbdeep resnet_block): the 7x7x7 stem on the raw
//     image (model3d.py:414), the first convolution of a block strided by `pool` with TensorFlow's asymmetric 'same' padding
//     (resnet_block: conv_layer(..., strides=pool)), the strided 1x1x1 projection of the shortcut, and the Add + Activation that
//     closes a block (folded into the epilogue of whichever kernel produces the last convolution);
//   * the first layer of a model with n_channel_in not in {1, 32k} (the 3-channel H&E models, model2d.py:310-316);
//   * 1x1 heads with few output channels (prob_class, model2d.py:345-347) and any other kernel size a config may ask for.
// Arbitrary odd or even kernel (kz, ky, kx), stride, padding-before, input / output channel counts; channels-last float32.
//     out[zo][yo][xo][co] = act(bias[co] + sum_{dz,dy,dx,ci} in[zo*sz-pz+dz][yo*sy-py+dy][xo*sx-px+dx][ci] * w[co][ci][dz][dy][dx]
//                               (+ res[zo][yo][xo][co]))                                  (zero outside the input)
// GEMM view: M = 32 consecutive output pixels of a row, N = 32 output channels, K = taps x input channels in (tap, channel) order, on
// v_mfma_f32_32x32x2_f32.  Each output is ONE fma chain in a fixed order (bias first, the residual added last), independent of the
// launch geometry: repeatable bit for bit on every box -- no library solver is involved anywhere.
// No LDS staging of operands: a wave reads its A operands (16 B per lane per four k-steps) and its packed B operands (contiguous
// 1 KiB per wave) straight from L1/L2; two output rows per wave share the B operands.  These layers are a few percent of a
// network's FLOPs; the kernel is bound by the matrix pipe once C_in >= 32.
#include "common.h"
#include "stardist_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int ROWS = 2;            // output rows per wave
constexpr int WAVES = 4;

struct GParams {
  const float* x; int xs, c_in, D, H, W;
  int kz, ky, kx, sz, sy, sx, pz, py, px;
  int Do, Ho, Wo;
  const float* w; const int* tab; const float* bias; const float* res; int rs; float* out; int os; int c_out, act;
  int tiles_x; long long rows;
  int n_k4, kp;
};

__device__ __forceinline__ int acc_col(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct RowCtx {
  long long ro[ROWS];
  bool rv[ROWS];
  int iz0[ROWS], iy0[ROWS];
};

__device__ __forceinline__ void rows_init(const GParams& P, long long tr, int wave, RowCtx& R) {
#pragma unroll
  for (int p = 0; p < ROWS; ++p) {
    long long ro = tr * (WAVES * ROWS) + wave * ROWS + p;
    R.rv[p] = ro < P.rows;
    ro = R.rv[p] ? ro : P.rows - 1;
    R.ro[p] = ro;
    const int zo = (int)(ro / P.Ho), yo = (int)(ro - (long long)zo * P.Ho);
    R.iz0[p] = zo * P.sz - P.pz;
    R.iy0[p] = yo * P.sy - P.py;
  }
}

__device__ __forceinline__ void epilogue(const GParams& P, const RowCtx& R, const f32x16 (&acc)[ROWS], int g, int xo0, int i, int h) {
  const int co = g * 32 + i;
  if (co >= P.c_out) return;
#pragma unroll
  for (int p = 0; p < ROWS; ++p) {
    if (!R.rv[p]) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int xo = xo0 + acc_col(r, h);
      if (xo < P.Wo) {
        const size_t pix = (size_t)R.ro[p] * P.Wo + xo;
        float v = acc[p][r];
        if (P.res) v += P.res[pix * P.rs + co];
        if (P.act == 1) v = fmaxf(v, 0.f);
        P.out[pix * P.os + co] = v;
      }
    }
  }
}

// C_in a multiple of 32.  packed weights (v4f index): ((((g*T + tap)*nch + c)*4 + j)*2 + h)*32 + i  = w[g*32+i][c*32 + h*16 + j*4 + e][tap], e = 0..3
__global__ void __launch_bounds__(256) k_convg_vec(const GParams P) {
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y;
  const long long tile = blockIdx.x;
  const int tx = (int)(tile % P.tiles_x);
  const long long tr = tile / P.tiles_x;
  RowCtx R;
  rows_init(P, tr, wave, R);
  const int xo0 = tx * 32;
  const int ix0 = (xo0 + i) * P.sx - P.px;
  f32x16 acc[ROWS];
  {
    const int co = g * 32 + i;
    const float b = (P.bias && co < P.c_out) ? P.bias[co] : 0.f;
#pragma unroll
    for (int p = 0; p < ROWS; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = b;
  }
  const int nch = P.c_in >> 5;
  const int T = P.kz * P.ky * P.kx;
  const v4f* wg = (const v4f*)P.w + (size_t)g * T * nch * 256 + h * 32 + i;
  int tap = 0;
  for (int dz = 0; dz < P.kz; ++dz)
    for (int dy = 0; dy < P.ky; ++dy) {
      bool rowv[ROWS];
      size_t rowb[ROWS];
#pragma unroll
      for (int p = 0; p < ROWS; ++p) {
        const int iz = R.iz0[p] + dz, iy = R.iy0[p] + dy;
        rowv[p] = iz >= 0 && iz < P.D && iy >= 0 && iy < P.H;
        rowb[p] = ((size_t)min(max(iz, 0), P.D - 1) * P.H + min(max(iy, 0), P.H - 1)) * P.W;
      }
      for (int dx = 0; dx < P.kx; ++dx, ++tap) {
        const int ix = ix0 + dx;
        const bool xv = ix >= 0 && ix < P.W;
        const int cx = min(max(ix, 0), P.W - 1);
        const float* ap[ROWS];
        bool av[ROWS];
#pragma unroll
        for (int p = 0; p < ROWS; ++p) {
          ap[p] = P.x + (rowb[p] + cx) * P.xs + h * 16;
          av[p] = rowv[p] && xv;
        }
        const v4f* wt = wg + (size_t)tap * nch * 256;
        for (int c = 0; c < nch; ++c) {
          v4f A[ROWS][4], B[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            B[j] = wt[(c * 4 + j) * 64];
#pragma unroll
            for (int p = 0; p < ROWS; ++p) {
              v4f a = *(const v4f*)(ap[p] + c * 32 + j * 4);
              const v4f z = {0.f, 0.f, 0.f, 0.f};
              A[p][j] = av[p] ? a : z;
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int p = 0; p < ROWS; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[p][j][e], B[j][e], acc[p], 0, 0, 0);
        }
      }
    }
  epilogue(P, R, acc, g, xo0, i, h);
}

// any C_in (meant for 1..8): k = tap * C_in + ci, padded to a multiple of 8 with zero weights.  tab[k] = dz | dy << 6 | dx << 12 | ci << 18,
// bit 31 = padding.  packed weights (v4f index): ((g*n_k4 + m)*2 + h)*32 + i = w[g*32+i][k = 8m + 2e + h], e = 0..3
__global__ void __launch_bounds__(256) k_convg_small(const GParams P) {
  extern __shared__ int lds[];
  int* offs = lds;
  int* pack = lds + P.kp;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int k = tid; k < P.kp; k += 256) {
    const int t = P.tab[k];
    const int dz = t & 63, dy = (t >> 6) & 63, dx = (t >> 12) & 63, ci = (t >> 18) & 0x1fff;
    pack[k] = t;
    offs[k] = t < 0 ? 0 : ((dz * P.H + dy) * P.W + dx) * P.xs + ci;
  }
  __syncthreads();
  const int g = blockIdx.y;
  const long long tile = blockIdx.x;
  const int tx = (int)(tile % P.tiles_x);
  const long long tr = tile / P.tiles_x;
  RowCtx R;
  rows_init(P, tr, wave, R);
  const int xo0 = tx * 32;
  const int ix0 = (xo0 + i) * P.sx - P.px;
  f32x16 acc[ROWS];
  {
    const int co = g * 32 + i;
    const float b = (P.bias && co < P.c_out) ? P.bias[co] : 0.f;
#pragma unroll
    for (int p = 0; p < ROWS; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = b;
  }
  long long base[ROWS];
#pragma unroll
  for (int p = 0; p < ROWS; ++p) base[p] = (((long long)R.iz0[p] * P.H + R.iy0[p]) * P.W + ix0) * P.xs;
  const v4f* wg = (const v4f*)P.w + ((size_t)g * P.n_k4 * 2 + h) * 32 + i;
  for (int m = 0; m < P.n_k4; ++m) {
    const v4f B = wg[(size_t)m * 64];
    float a[ROWS][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = m * 8 + e * 2 + h;
      const int t = pack[k], o = offs[k];
      const int dz = t & 63, dy = (t >> 6) & 63, dx = (t >> 12) & 63;
      const int ix = ix0 + dx;
      const bool kv = t >= 0 && ix >= 0 && ix < P.W;
#pragma unroll
      for (int p = 0; p < ROWS; ++p) {
        const int iz = R.iz0[p] + dz, iy = R.iy0[p] + dy;
        const bool v = kv && iz >= 0 && iz < P.D && iy >= 0 && iy < P.H;
        a[p][e] = v ? P.x[base[p] + o] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int p = 0; p < ROWS; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][e], B[e], acc[p], 0, 0, 0);
  }
  epilogue(P, R, acc, g, xo0, i, h);
}

bool vec_form(int c_in) { return c_in > 0 && c_in % 32 == 0; }
long long taps_of(int kz, int ky, int kx) { return (long long)kz * ky * kx; }
bool shape_ok(int c_in, int c_out, int kz, int ky, int kx) {
  if (c_in <= 0 || c_out <= 0 || kz <= 0 || ky <= 0 || kx <= 0 || kz > 63 || ky > 63 || kx > 63) return false;
  if (vec_form(c_in)) return true;
  return c_in < 8192 && taps_of(kz, ky, kx) * c_in <= 6144;        // offsets + tap table of the small form: <= 48 KiB of LDS
}

}  // namespace

extern "C" long long sd_convg_packed_floats(int c_in, int c_out, int kz, int ky, int kx) {
  if (!shape_ok(c_in, c_out, kz, ky, kx)) return -1;
  const long long groups = (c_out + 31) / 32, T = taps_of(kz, ky, kx);
  if (vec_form(c_in)) return groups * T * (c_in / 32) * 1024;
  const long long kp = (T * c_in + 7) / 8 * 8;
  return groups * (kp / 8) * 256 + kp;
}

extern "C" int sd_convg_pack_weights_host(const float* w, int c_in, int c_out, int kz, int ky, int kx, float* packed) {
  if (!w || !packed || !shape_ok(c_in, c_out, kz, ky, kx)) {
    sd::set_error("sd_convg_pack_weights: kernel sizes 1..63, c_in a multiple of 32, or taps * c_in <= 6144");
    return -1;
  }
  const int groups = (c_out + 31) / 32, T = kz * ky * kx;
  if (vec_form(c_in)) {
    const int nch = c_in / 32;
    for (int g = 0; g < groups; ++g)
      for (int tap = 0; tap < T; ++tap)
        for (int c = 0; c < nch; ++c)
          for (int j = 0; j < 4; ++j)
            for (int h = 0; h < 2; ++h)
              for (int i = 0; i < 32; ++i)
                for (int e = 0; e < 4; ++e) {
                  const int co = g * 32 + i, ci = c * 32 + h * 16 + j * 4 + e;
                  const size_t idx = (((((size_t)g * T + tap) * nch + c) * 4 + j) * 2 + h) * 32 + i;
                  packed[idx * 4 + e] = co < c_out ? w[((size_t)co * c_in + ci) * T + tap] : 0.f;
                }
    return 0;
  }
  const int K = T * c_in, kp = (K + 7) / 8 * 8, n_k4 = kp / 8;
  for (int g = 0; g < groups; ++g)
    for (int m = 0; m < n_k4; ++m)
      for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 32; ++i)
          for (int e = 0; e < 4; ++e) {
            const int co = g * 32 + i, k = m * 8 + e * 2 + h;
            const size_t idx = (((size_t)g * n_k4 + m) * 2 + h) * 32 + i;
            float v = 0.f;
            if (co < c_out && k < K) { const int tap = k / c_in, ci = k % c_in; v = w[((size_t)co * c_in + ci) * T + tap]; }
            packed[idx * 4 + e] = v;
          }
  int* tab = (int*)(packed + (size_t)groups * n_k4 * 256);
  for (int k = 0; k < kp; ++k) {
    if (k >= K) { tab[k] = (int)0x80000000u; continue; }
    const int tap = k / c_in, ci = k % c_in;
    const int dz = tap / (ky * kx), dy = (tap / kx) % ky, dx = tap % kx;
    tab[k] = dz | (dy << 6) | (dx << 12) | (ci << 18);
  }
  return 0;
}

extern "C" int sd_convg_ndhwc_device(const float* d_src, int c_in, int src_stride, int D, int H, int W, int kz, int ky, int kx, int sz, int sy,
                                     int sx, int pz, int py, int px, int Do, int Ho, int Wo, const float* d_wpacked, const float* d_bias,
                                     const float* d_res, int res_stride, int c_out, int act, float* d_out, int out_stride, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (Do <= 0 || Ho <= 0 || Wo <= 0) return 0;
  if (!d_src || !d_wpacked || !d_out || !shape_ok(c_in, c_out, kz, ky, kx) || (act != 0 && act != 1) || D <= 0 || H <= 0 || W <= 0 ||
      sz <= 0 || sy <= 0 || sx <= 0 || pz < 0 || py < 0 || px < 0 || src_stride < c_in || out_stride < c_out || (d_res && res_stride < c_out)) {
    sd::set_error("sd_convg_ndhwc: bad arguments (c_in %d, c_out %d, kernel %dx%dx%d, stride %dx%dx%d)", c_in, c_out, kz, ky, kx, sz, sy, sx);
    return -1;
  }
  // every input element an output touches must be addressable with the kernel's index arithmetic
  if ((long long)(Do - 1) * sz - pz + kz - 1 > (long long)D + 64 || (long long)(Ho - 1) * sy - py + ky - 1 > (long long)H + 64 ||
      (long long)(Wo - 1) * sx - px + kx - 1 > (long long)W + 64) {
    sd::set_error("sd_convg_ndhwc: output extent does not match input extent, stride and padding");
    return -1;
  }
  GParams P;
  P.x = d_src; P.xs = src_stride; P.c_in = c_in; P.D = D; P.H = H; P.W = W;
  P.kz = kz; P.ky = ky; P.kx = kx; P.sz = sz; P.sy = sy; P.sx = sx; P.pz = pz; P.py = py; P.px = px;
  P.Do = Do; P.Ho = Ho; P.Wo = Wo;
  P.w = d_wpacked; P.bias = d_bias; P.res = d_res; P.rs = res_stride; P.out = d_out; P.os = out_stride; P.c_out = c_out; P.act = act;
  P.tiles_x = (Wo + 31) / 32;
  P.rows = (long long)Do * Ho;
  const long long tiles = (long long)P.tiles_x * ((P.rows + WAVES * ROWS - 1) / (WAVES * ROWS));
  if (tiles > 0x7fffffffLL) { sd::set_error("sd_convg_ndhwc: too many tiles"); return -1; }
  const int groups = (c_out + 31) / 32;
  if (vec_form(c_in)) {
    if ((((uintptr_t)d_src | (uintptr_t)d_wpacked) & 15) || (src_stride & 3)) {
      sd::set_error("sd_convg_ndhwc: the 32-channel form needs 16-byte aligned sources (stride a multiple of 4 floats)");
      return -1;
    }
    P.tab = nullptr; P.n_k4 = 0; P.kp = 0;
    hipLaunchKernelGGL(k_convg_vec, dim3((unsigned)tiles, groups), dim3(256), 0, s, P);
  } else {
    const long long K = taps_of(kz, ky, kx) * c_in;
    P.kp = (int)((K + 7) / 8 * 8);
    P.n_k4 = P.kp / 8;
    if (((uintptr_t)d_wpacked & 15) || (long long)(kz - 1) * H * W * src_stride + (long long)(ky - 1) * W * src_stride > 0x3fffffffLL) {
      sd::set_error("sd_convg_ndhwc: misaligned weights or tap offsets beyond 2^30 elements");
      return -1;
    }
    P.tab = (const int*)(d_wpacked + (size_t)groups * P.n_k4 * 256);
    hipLaunchKernelGGL(k_convg_small, dim3((unsigned)tiles, groups), dim3(256), (size_t)P.kp * 8, s, P);
  }
  SD_LAUNCH_CHECK();
  return 0;
}
