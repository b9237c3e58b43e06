// edt.hip -- per-object normalised Euclidean distance transform of a label image: the training target `edt_prob` of the reference
// (stardist/utils.py:71-125; called by the data generators, stardist/models/model2d.py / model3d.py StarDistData*.__getitem__).
//
// The reference computes, object by object, scipy's exact EDT of the object's mask on its bounding box grown by one pixel
// (where the box does not touch the image border) and divides by the object's maximum.  The grown box always contains the
// nearest non-object pixel of every object pixel (clamping a farther pixel onto the box's frame gives a closer non-object pixel),
// so that is the distance to the nearest pixel INSIDE THE IMAGE whose label differs -- the image border itself is not background.
//
// GPU formulation (exact, separable, label-aware; float64 like scipy):
//   pass x : s1(p) = (sx * distance along the row to the nearest pixel with another label)^2   (infinite if the run reaches both borders)
//   pass y : s2(p) = min over rows y' of (sy (y - y'))^2 + (label(y', x) == label(p) ? s1(y', x) : 0)
//   pass z : s3(p) = min over planes z' of (sz (z - z'))^2 + (label(z', y, x) == label(p) ? s2(z', y, x) : 0)
//   d = sqrt(s), per-label maximum (atomicMax on the bit pattern of a non-negative double), prob = float(d / (max + 1e-10)).
// The outward scans stop as soon as the axis term alone exceeds the best value, i.e. after about one object radius.
// Bound: HBM (2 B read + 4 B written per pixel and pass) for the small objects of nuclei images.
#include "common.h"
#include "stardist_hip.h"

namespace {

__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

// along x (innermost axis)
__global__ void __launch_bounds__(256) k_edt_x(const int* __restrict__ lbl, long long n, int X, double sx, double* __restrict__ out) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int l = lbl[p];
  if (l == 0) { out[p] = 0.0; return; }
  const int x = (int)(p % X);
  const int* row = lbl + (p - x);
  int dl = -1, dr = -1;
  for (int k = 1; x - k >= 0; ++k) if (row[x - k] != l) { dl = k; break; }
  for (int k = 1; x + k < X; ++k) { if (dl > 0 && k >= dl) break; if (row[x + k] != l) { dr = k; break; } }
  const int d = dl < 0 ? dr : (dr < 0 ? dl : (dl < dr ? dl : dr));
  if (d < 0) { out[p] = dinf(); return; }
  const double t = sx * (double)d;
  out[p] = t * t;
}

// along an outer axis with element stride `stride` and extent `len`; idx = position of p along that axis
__global__ void __launch_bounds__(256) k_edt_axis(const int* __restrict__ lbl, const double* __restrict__ in, long long n, long long stride, int len,
                                                  double s, double* __restrict__ out) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int l = lbl[p];
  if (l == 0) { out[p] = 0.0; return; }
  const int idx = (int)((p / stride) % len);
  double best = in[p];
  bool up = true, down = true;
  for (int k = 1; up || down; ++k) {
    const double t = s * (double)k, c = t * t;
    if (c >= best) break;
    if (down) {
      if (idx - k < 0) down = false;
      else {
        const long long q = p - (long long)k * stride;
        const double v = c + (lbl[q] == l ? in[q] : 0.0);
        if (v < best) best = v;
      }
    }
    if (up) {
      if (idx + k >= len) up = false;
      else {
        const long long q = p + (long long)k * stride;
        const double v = c + (lbl[q] == l ? in[q] : 0.0);
        if (v < best) best = v;
      }
    }
  }
  out[p] = best;
}

__global__ void __launch_bounds__(256) k_edt_max(const int* __restrict__ lbl, double* __restrict__ sq, long long n, int max_label,
                                                 unsigned long long* __restrict__ mx) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int l = lbl[p];
  if (l <= 0 || l > max_label) return;
  const double d = sqrt(sq[p]);
  sq[p] = d;
  atomicMax(&mx[l], (unsigned long long)__double_as_longlong(d));      // non-negative doubles order like their bit patterns
}

__global__ void __launch_bounds__(256) k_edt_norm(const int* __restrict__ lbl, const double* __restrict__ d, long long n, int max_label,
                                                  const unsigned long long* __restrict__ mx, float* __restrict__ prob) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int l = lbl[p];
  if (l <= 0 || l > max_label) { prob[p] = 0.f; return; }
  const double m = __longlong_as_double((long long)mx[l]);
  prob[p] = (float)(d[p] / (m + 1e-10));
}

}  // namespace

extern "C" int sd_edt_prob_device(const int32_t* d_lbl, int Z, int Y, int X, double sz, double sy, double sx, int max_label, float* d_prob,
                                  void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (Z <= 0 || Y <= 0 || X <= 0) return 0;
  if (!d_lbl || !d_prob || max_label < 0 || !(sz > 0) || !(sy > 0) || !(sx > 0)) {
    sd::set_error("sd_edt_prob: null pointer, negative max_label or non-positive sampling");
    return -1;
  }
  const long long n = (long long)Z * Y * X;
  sd::Arena& A = sd::arena();
  if (A.begin(s)) return -1;
  double* a = A.take_n<double>(n);
  double* b = A.take_n<double>(n);
  unsigned long long* mx = A.take_n<unsigned long long>((size_t)max_label + 1);
  if (!a || !b || !mx) return -1;
  SD_CHECK(hipMemsetAsync(mx, 0, ((size_t)max_label + 1) * sizeof(unsigned long long), s));
  const dim3 g((unsigned)((n + 255) / 256)), t(256);
  hipLaunchKernelGGL(k_edt_x, g, t, 0, s, d_lbl, n, X, sx, a);
  double* cur = a; double* nxt = b;
  if (Y > 1) { hipLaunchKernelGGL(k_edt_axis, g, t, 0, s, d_lbl, cur, n, (long long)X, Y, sy, nxt); double* tmp = cur; cur = nxt; nxt = tmp; }
  if (Z > 1) { hipLaunchKernelGGL(k_edt_axis, g, t, 0, s, d_lbl, cur, n, (long long)X * Y, Z, sz, nxt); double* tmp = cur; cur = nxt; nxt = tmp; }
  hipLaunchKernelGGL(k_edt_max, g, t, 0, s, d_lbl, cur, n, max_label, mx);
  hipLaunchKernelGGL(k_edt_norm, g, t, 0, s, d_lbl, cur, n, max_label, mx, d_prob);
  SD_LAUNCH_CHECK();
  return 0;
}
