// common.h -- shared host-side plumbing for libstardist_hip (error state, workspace arena).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace sd {

// last error message, returned by sd_last_error()
char* err_buf();
void set_error(const char* fmt, ...);

#define SD_CHECK(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      sd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

#define SD_LAUNCH_CHECK() SD_CHECK(hipGetLastError())

// Workspace arena: one device allocation per (process, device) that only grows; calls carve
// 256-byte aligned slices from it. Single-stream, single-caller use (the Python GIL / the
// caller's stream order serialises calls) -- matches the reference natives, which are not
// re-entrant either (SURVEY.md 8b "Threading").
class Arena {
 public:
  // begin a call: rewinds; if the previous call spilled into extra chunks they are merged
  // into one bigger chunk (one-time sync) so steady-state calls never allocate.
  int begin(hipStream_t stream);
  // 256-byte aligned device slice valid until the next begin(); nullptr + error set on OOM
  void* take(size_t bytes);
  template <typename T> T* take_n(size_t n) { return (T*)take((n ? n : 1) * sizeof(T)); }
  void release();
  size_t capacity() const;
 private:
  enum { kMaxChunks = 64 };
  void* base_[kMaxChunks] = {};
  size_t cap_[kMaxChunks] = {};
  int n_ = 0, cur_ = 0;
  size_t off_ = 0;
};
Arena& arena();

static inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

// Result-preserving switches between equivalent formulations (set through sd_set_option of the C ABI, documented there): the parity
// suite runs both settings against the reference.  Nothing in a release build reads the process environment.
enum Option { OPT_NMS3D_VOLUME_BOUNDS, OPT_NMS3D_CONE_MAP, OPT_NMS3D_REFINE_MESH, OPT_PROBE_TIER, OPT_PROBE_NO_GENERAL, OPT_TRACE, OPT_NMS3D_TAIL_BATCH, OPT_NMS3D_SPLIT_EXACT, OPT_CONV_F16_WGS, OPT_NMS2D_PAIR_LANES, OPT_NMS2D_AREA_BOUNDS, OPT_NMS2D_DEFER_UNDECIDED, OPT_NMS2D_STRICT, OPT_NMS2D_NBR_SINGLE, OPT_NMS3D_NBR_SINGLE, OPT_NMS2D_DEFER_MAX, OPT_NMS3D_BOUNDS_REUSE, OPT_NMS3D_DEFER_EXACT, OPT_NMS3D_BOUNDS_LEAN, OPT_COUNT };
int option(Option o);

// A/B tuning knobs of the probe scripts (pair order, tail-batch thresholds, ...): environment variables in builds made with
// -DSD_DEBUG_SWITCHES (tools/build_debug.sh) only; a release build compiles the defaults in.
#ifdef SD_DEBUG_SWITCHES
int tuning_env(const char* name, int dflt);
#else
static inline int tuning_env(const char*, int dflt) { return dflt; }
#endif

}  // namespace sd
