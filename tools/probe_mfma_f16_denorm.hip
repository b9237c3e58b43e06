// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal operands on gfx950?  (the split-fp16 convolution's small-value behaviour)
// hipcc --offload-arch=gfx950 tools/probe_mfma_f16_denorm.hip -o /tmp/probe_denorm && /tmp/probe_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short abits, unsigned short bbits) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = __builtin_bit_cast(_Float16, abits); b[j] = __builtin_bit_cast(_Float16, bbits); }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  struct { unsigned short a, b; const char* what; double want; } cases[] = {
      {0x0001, 0x7800, "a = 2^-24 (smallest subnormal), b = 32768", 16 * 5.9604644775390625e-08 * 32768.0},
      {0x03FF, 0x3C00, "a = largest subnormal, b = 1", 16 * 1023 * 5.9604644775390625e-08},
      {0x0400, 0x3C00, "a = smallest normal, b = 1", 16 * 6.103515625e-05},
      {0x0200, 0x0200, "both subnormal 2^-15", 16 * 9.313225746154785e-10}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
    float h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%-45s mfma sum over k=16: %.9g   exact: %.9g   %s\n", c.what, h, c.want, h == (float)c.want ? "kept" : (h == 0 ? "FLUSHED" : "differs"));
  }
  return 0;
}
