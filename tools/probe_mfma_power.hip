// probe_mfma_power.hip -- the matrix pipe's PRACTICAL roof on real data.  k_conv3_f16 is power-limited (tools/conv_f16_phase_profile.hip: the same
// instruction stream on all-zero data runs 31-37 % faster at a 45 % higher shader clock), so its roof is not 2.5 PFLOP/s x busy fraction but what the
// chip sustains when v_mfma_f32_32x32x16_f16 is fed operands that toggle like the kernel's (fp16 hi terms and their 2^11-scaled remainders).
// This loop has nothing but the matrix instructions: operands in registers, two waves per SIMD (512 workgroups of 256 threads), the kernel's
// accumulator pattern (acc1 += ah*bl; acc1 += al*bh; acc0 += ah*bh for two tiles).  It prints TFLOP/s executed and the shader clock
// (s_memtime ticks / wall time) for zero operands, for random operands held constant, and for random operands that change every iteration
// (rotated between registers, as fresh LDS reads would).
// hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_power.hip -o tools/bin/probe_mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k(const f16x8* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* ticks) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f16x8 a[8], b[4];
#pragma unroll
  for (int n = 0; n < 8; ++n) a[n] = src[(size_t)tid * 12 + n];
#pragma unroll
  for (int n = 0; n < 4; ++n) b[n] = src[(size_t)tid * 12 + 8 + n];
  f32x16 acc0[2] = {}, acc1[2] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {                       // two operand groups per iteration: 12 matrix instructions
      const f16x8 a0h = a[g * 4], a0l = a[g * 4 + 1], a1h = a[g * 4 + 2], a1l = a[g * 4 + 3], bh = b[g * 2], bl = b[g * 2 + 1];
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, acc1[1], 0, 0, 0);
      acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, acc0[0], 0, 0, 0);
      acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, acc1[0], 0, 0, 0);
      acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, acc1[1], 0, 0, 0);
      acc0[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, acc0[1], 0, 0, 0);
    }
    if (ROT) {                                          // new operand values next time (a permutation of the registers: no arithmetic)
      const f16x8 t = a[0];
#pragma unroll
      for (int n = 0; n < 7; ++n) a[n] = a[n + 1];
      a[7] = t;
      const f16x8 u = b[0];
#pragma unroll
      for (int n = 0; n < 3; ++n) b[n] = b[n + 1];
      b[3] = u;
#pragma unroll
      for (int n = 0; n < 8; ++n) asm volatile("" : "+v"(a[n]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[0][r] + acc0[1][r] + acc1[0][r] + acc1[1][r];
  out[tid] = s;
  if (tid == 0) ticks[0] = t1 - t0;
}

int main() {
  const int blocks = 512, n = blocks * 256, iters = 20000;
  std::vector<_Float16> h((size_t)n * 12 * 8);
  f16x8* d; float* o; unsigned long long* t;
  (void)hipMalloc(&d, h.size() * 2); (void)hipMalloc(&o, n * 4); (void)hipMalloc(&t, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    unsigned s = 99u;
    for (size_t i = 0; i < h.size(); ++i) {
      s = s * 1664525u + 1013904223u;
      const float v = ((float)(s >> 8) / 16777216.f - 0.5f) * ((i / 8) % 2 ? 0.002f : 2.f);       // hi-like and remainder-like magnitudes
      h[i] = mode == 0 ? (_Float16)0.f : (_Float16)v;
    }
    (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, 0);
      if (mode == 2) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, o, iters, t);
      else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, o, iters, t);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long ticks; (void)hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    const double flop = (double)n / 64 * iters * 12 * 32768.0;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s executed (%.1f %% of 2500)  shader clock %.2f GHz  matrix pipe busy %.0f %% of the cycles\n",
           mode == 0 ? "zero operands" : mode == 1 ? "random operands, the same every iteration" : "random operands, rotated every iteration", ms, flop / ms * 1e-9,
           flop / ms * 1e-9 / 25.0, ticks / ms * 1e-6, 100.0 * (double)iters * 12 * 32 * 2 / (double)ticks);
  }
  return 0;
}
