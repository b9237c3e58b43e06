// Does a kernel without any communication between its waves return the same bytes for the same input when SEVERAL PROCESSES time-slice the
// device?  A persistent grid (2048 workgroups x 256 threads) of long-lived waves: every thread runs a chain of float operations (fma, rcp,
// compare / select, a ballot and a cross-lane shuffle per step) for about a millisecond and writes one float.  The kernel is launched twice
// on the same input and the outputs are compared on the host; REPS such pairs.  Start P copies of this program at once on one device
// (tools/gpu_r06_q15.sh) -- DESIGN.md 4 item 18.    build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/probe_timeslice tools/probe_timeslice.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// mode bit 0: a gather from global memory in every step; bit 1: the neighbour's value through LDS (write own, read neighbour's) in every step
__global__ void __launch_bounds__(256) k_chain(const float* __restrict__ in, float* __restrict__ out, int n, int steps, int mode) {
  __shared__ float sh[256];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float a = in[i], b = in[(i * 7 + 3) % n], acc = 0.f;
    int par = 0;
    for (int k = 0; k < steps; ++k) {
      const float c = a * b - b * 0.37f, d = __builtin_amdgcn_rcpf(1.5f + c * c);
      const bool pos = c > 0.f;
      const unsigned long long m = __ballot(pos);
      par ^= (int)(__popcll(m) & 1);
      acc += pos ? d : -d;
      float nb = __shfl(b, (threadIdx.x + 1) & 63);
      if (mode & 2) { sh[threadIdx.x] = b; __builtin_amdgcn_wave_barrier(); nb = sh[(threadIdx.x & ~63) | ((threadIdx.x + 1) & 63)]; __builtin_amdgcn_wave_barrier(); }
      if (mode & 1) { const unsigned int g = ((unsigned int)i * 2654435761u + (unsigned int)k * 40503u) % (unsigned int)n; nb += 0.25f * in[g]; }
      a = a * 0.999f + 0.001f * nb;
      b = b * 0.998f + 0.002f * d;
    }
    out[i] = acc + (float)par;
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 100, steps = argc > 2 ? atoi(argv[2]) : 4000, mode = argc > 3 ? atoi(argv[3]) : 0;
  const int n = 2048 * 256 * 2;
  std::vector<float> h(n);
  unsigned int s = 12345u;
  for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.f * 2.f - 1.f; }
  float *d_in, *d_a, *d_b;
  CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_a, n * 4)); CK(hipMalloc(&d_b, n * 4));
  CK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<float> ha(n), hb(n), ref;
  long long bad_pairs = 0, bad_vs_first = 0, bad_elems = 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_total = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_chain, dim3(2048), dim3(256), 0, 0, d_in, d_a, n, steps, mode);
    hipLaunchKernelGGL(k_chain, dim3(2048), dim3(256), 0, 0, d_in, d_b, n, steps, mode);
    CK(hipEventRecord(e1, 0));
    CK(hipMemcpy(ha.data(), d_a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms;
    if (r == 0) ref = ha;
    if (memcmp(ha.data(), hb.data(), n * 4)) { ++bad_pairs; for (int i = 0; i < n; ++i) bad_elems += memcmp(&ha[i], &hb[i], 4) != 0; }
    if (memcmp(ha.data(), ref.data(), n * 4) || memcmp(hb.data(), ref.data(), n * 4)) ++bad_vs_first;
  }
  printf("mode %d, pid %d: %d launch pairs of %.2f ms each (%d steps): %lld pairs whose two launches differ (%lld elements in all), %lld pairs that differ from the first result\n",
         mode, (int)getpid(), reps, ms_total / reps, steps, bad_pairs, bad_elems, bad_vs_first);
  return 0;
}
