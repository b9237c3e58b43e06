// probe_memtime.hip -- what does s_memtime count?  A wave spins until s_memtime has advanced by N ticks; wall time (HIP events) gives ticks/s --
// once on an otherwise idle chip (one wave), once with every SIMD busy with dependent FMAs.  hipcc --offload-arch=gfx950 -O3 tools/probe_memtime.hip -o tools/bin/probe_memtime
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long n, unsigned long long* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t = t0;
  float x = threadIdx.x;
  while (t - t0 < n) { for (int k = 0; k < 64; ++k) x = __builtin_fmaf(x, 1.0001f, 0.5f); t = __builtin_amdgcn_s_memtime(); }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t - t0; out[1] = (unsigned long long)x; }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int busy = 0; busy < 2; ++busy)
    for (int rep = 0; rep < 2; ++rep) {
      const unsigned long long n = 20000000ull;
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(spin, dim3(busy ? 256 * 8 : 1), dim3(busy ? 256 : 64), 0, 0, n, d);
      hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%s: %llu ticks in %.3f ms = %.1f MHz\n", busy ? "all SIMDs busy" : "one wave", h[0], ms, h[0] / ms * 1e-3);
    }
  return 0;
}
