// conv_f16_phase_profile.hip -- where do the cycles of k_conv3_f16 go?  Compiles the library's kernel source with SD_CONV_PROFILE
// (s_memtime stamps at the phase boundaries, summed over the workgroups' first lanes) and runs single layers of the bench networks.
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Istardist_amd/csrc tools/conv_f16_phase_profile.hip -o /tmp/cpp16 && /tmp/cpp16 [workgroups per CU: 2 | 1] [split16 tensors: 0 | 1]
#define SD_CONV_PROFILE 1
#include "../stardist_amd/csrc/conv3x3_f16.hip"
#include "../stardist_amd/csrc/lib.hip"
#include "../stardist_amd/csrc/unet_ops.hip"
#include <vector>

static const char* kPhase[16] = {"dy 0 issue: loop top + next weights' loads", "dy 1 issue: weight loads + the next unit's halo loads",
                                 "dy 2 issue: weight loads", "dy 0 compute_sub (LDS operand reads + MFMA; + the next halo's address arithmetic)", "dy 0 weights -> LDS (waits for them)",
                                 "dy 1 weights -> LDS", "dy 2 weights -> LDS (the halo has to be there as well)", "barriers after the sub-units",
                                 "dy 0: the next unit's halo descriptor (scalar)", "LDS stores of the next halo tile's planes", "barrier after those stores",
                                 "epilogue: accumulators -> HBM (per tile)", "dy 1 compute_sub", "dy 2 compute_sub (+ split of the arrived halo)", "dy 0: next tile's coordinates (once per tile)", "(unused)"};

static int g_wgs = 2, g_split = 0, g_zero = 0;      // g_zero: all-zero activations and weights (the power floor: same instructions, no toggling)
static int run(int D, int H, int W, int c_in, int c_out, int kz) {
  const size_t n_in = (size_t)D * H * W * c_in, n_out = (size_t)D * H * W * c_out;
  std::vector<float> hx(n_in), hw((size_t)c_out * c_in * 9 * kz);
  unsigned s = 12345u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.f; }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((float)(s >> 8) / 16777216.f - 0.5f) * 0.1f; }
  if (g_zero) { for (auto& v : hx) v = 0.f; for (auto& v : hw) v = 0.f; }
  const long long np = sd_conv3_f16x3_packed_floats(c_in, c_out, kz);
  std::vector<float> hp((size_t)np);
  if (sd_conv3_f16x3_pack_weights_host(hw.data(), c_in, c_out, kz, hp.data())) { printf("pack: %s\n", sd_last_error()); return 1; }
  float *dx, *dw, *dout, *dxs = nullptr;
  if (hipMalloc(&dx, n_in * 4) || hipMalloc(&dw, (size_t)np * 4) || hipMalloc(&dout, n_out * 4)) return 1;
  hipMemcpy(dx, hx.data(), n_in * 4, hipMemcpyHostToDevice);
  const int out_split = g_split && c_out != 128;          // (the features layer writes f32 for the heads)
  if (g_split) {
    if (hipMalloc(&dxs, n_in * 4)) return 1;
    if (sd_split16_pack_device(dx, (long long)D * H * W, c_in, dxs, nullptr, nullptr)) { printf("pack: %s\n", sd_last_error()); return 1; }
    hipDeviceSynchronize();
  }
  hipMemcpy(dw, hp.data(), (size_t)np * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long zero[20] = {}, prof[20];
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), zero, sizeof(zero));
    hipEventRecord(e0, nullptr);
    if (sd_conv3_f16x3_fmt_ndhwc_device(g_split ? dxs : dx, c_in, 0, nullptr, 0, 0, D, H, W, kz, dw, nullptr, c_out, 1, dout, g_split, out_split, nullptr, nullptr, nullptr, nullptr)) { printf("conv: %s\n", sd_last_error()); return 1; }
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_conv_prof), sizeof(prof));
  const double flops = 2.0 * D * H * W * (double)c_in * c_out * 9 * kz;
  printf("\n%dx%dx%d  %d -> %d  kz %d: %.3f ms  %.1f TFLOP/s f32-equivalent  (%llu workgroups, %llu units, %.0f ticks per unit; shader clock %.2f GHz = ticks of a workgroup / kernel time)\n", D, H, W, c_in, c_out, kz, ms,
         flops / ms * 1e-9, (unsigned long long)g_wgs * 256ull, prof[19], (double)prof[0] / (double)(prof[19] ? prof[19] : 1), (double)prof[0] / ((double)g_wgs * 256.0) / ms * 1e-6);
  double acc = 0;
  for (int k = 0; k < 16; ++k) { printf("   %5.1f %%  %s\n", 100.0 * (double)prof[1 + k] / (double)prof[0], kPhase[k]); acc += (double)prof[1 + k]; }
  printf("   %5.1f %%  outside the unit loop (first tile's staging, last stores)\n", 100.0 * (1.0 - acc / (double)prof[0]));
  hipFree(dx); hipFree(dw); hipFree(dout); if (dxs) hipFree(dxs);
  return 0;
}

int main(int argc, char** argv) {
  int rc = 0;
  if (argc > 1) g_wgs = atoi(argv[1]) == 1 ? 1 : 2;
  if (argc > 2) g_split = atoi(argv[2]) != 0;
  if (argc > 3) g_zero = atoi(argv[3]) != 0;
  sd_set_option("conv_f16_workgroups_per_cu", g_wgs);
  printf("k_conv3_f16, %s tensors, %d workgroup(s) per CU; ticks = s_memtime (shader clock), summed over the workgroups\n", g_split ? "split16" : "f32", g_wgs);
  rc |= run(1, 2048, 2048, 32, 32, 1);
  rc |= run(1, 1024, 1024, 64, 64, 1);
  rc |= run(1, 512, 512, 128, 128, 1);
  rc |= run(256, 256, 256, 32, 32, 3);
  rc |= run(128, 128, 128, 64, 64, 3);
  rc |= run(64, 64, 64, 128, 128, 3);
  rc |= run(256, 256, 256, 32, 128, 3);
  return rc;
}
