// lib.hip -- library-level entry points of libstardist_hip (error state, workspace arena).
#include "common.h"
#include "../../include/stardist_hip.h"
#include <stdarg.h>
#include <stdlib.h>

namespace sd {

// per thread: two threads driving two devices do not overwrite each other's message
static thread_local char g_err[1024] = "";
char* err_buf() { return g_err; }
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int g_opt[OPT_COUNT] = {1, 1, 2, 1, 0, 0, 1, 3, 2, 64, 1, 2, 0, 1, 1, 16384, 1, 3, 1};
static const char* const g_opt_name[OPT_COUNT] = {"nms3d_volume_bounds", "nms3d_cone_map", "nms3d_refine_mesh", "probe_tier", "probe_no_general", "trace", "nms3d_tail_batch", "nms3d_split_exact", "conv_f16_workgroups_per_cu", "nms2d_pair_lanes", "nms2d_area_bounds", "nms2d_defer_undecided", "nms2d_strict", "nms2d_neighbours_single_pass", "nms3d_neighbours_single_pass", "nms2d_defer_max", "nms3d_bounds_reuse", "nms3d_defer_exact", "nms3d_bounds_lean"};
int option(Option o) { return g_opt[o]; }
#ifdef SD_DEBUG_SWITCHES
int tuning_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#endif

// one workspace arena per HIP device: a call carves its slices from the arena of the device that is current when it runs
// (the Python wrappers make the tensors' device current, stardist_amd/lib/_native.py dcall)
enum { kMaxDevices = 64 };
static Arena g_arena[kMaxDevices];
Arena& arena() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) d = 0;
  return g_arena[d];
}

size_t Arena::capacity() const { size_t t = 0; for (int i = 0; i < n_; ++i) t += cap_[i]; return t; }
int Arena::begin(hipStream_t stream) {
  if (n_ > 1) {
    size_t total = capacity();
    SD_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < n_; ++i) { (void)hipFree(base_[i]); base_[i] = nullptr; cap_[i] = 0; }
    n_ = 0;
    size_t want = total + (total >> 3);
    SD_CHECK(hipMalloc(&base_[0], want));
    cap_[0] = want; n_ = 1;
  }
  cur_ = 0; off_ = 0;
  return 0;
}
void* Arena::take(size_t bytes) {
  for (;;) {
    if (cur_ < n_) {
      size_t a = (off_ + 255) & ~size_t(255);
      if (a + bytes <= cap_[cur_]) { off_ = a + bytes; return (char*)base_[cur_] + a; }
      ++cur_; off_ = 0;
      continue;
    }
    if (n_ >= kMaxChunks) { set_error("workspace arena: too many chunks"); return nullptr; }
    size_t want = bytes + (bytes >> 2) + (size_t(4) << 20);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { set_error("workspace arena: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return nullptr; }
    base_[n_] = p; cap_[n_] = want; cur_ = n_; ++n_; off_ = 0;
  }
}
void Arena::release() {
  for (int i = 0; i < n_; ++i) { (void)hipFree(base_[i]); base_[i] = nullptr; cap_[i] = 0; }
  n_ = 0; cur_ = 0; off_ = 0;
}

}  // namespace sd

extern "C" {
int sd_set_option(const char* name, int value) {
  for (int k = 0; name && k < sd::OPT_COUNT; ++k)
    if (!strcmp(name, sd::g_opt_name[k])) { sd::g_opt[k] = value; return 0; }
  sd::set_error("sd_set_option: unknown option '%s'", name ? name : "(null)");
  return -1;
}
int sd_get_option(const char* name) {
  for (int k = 0; name && k < sd::OPT_COUNT; ++k)
    if (!strcmp(name, sd::g_opt_name[k])) return sd::g_opt[k];
  return -1;
}
const char* sd_last_error(void) { return sd::err_buf(); }
int sd_version(void) { return 1; }
int sd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int sd_release_workspace(void) { sd::arena().release(); return 0; }   /* the current device's arena */
}
