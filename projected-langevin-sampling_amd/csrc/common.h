// Shared by the translation units of libplship.so: error plumbing, the per-launch timeline, small host helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/plship.h"

namespace plship {

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
extern thread_local std::string g_last_error;  // defined in plship.hip

inline int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define PLS_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return fail(PLS_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
  } while (0)

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PLS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return PLS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// per-launch timeline (pls_timeline_begin / _end): events live outside the step path's no-allocation rule because
// they are created in begin(), never inside a launch function
// ---------------------------------------------------------------------------------------------------------------
struct Timeline {
  bool on = false;
  int capacity = 0, count = 0;
  std::vector<hipEvent_t> ev;  // 2 per launch
  std::vector<int> tag;
};
extern thread_local Timeline g_tl;  // defined in plship.hip

struct LaunchScope {  // records the bracketing events of one launch when the timeline is on
  hipStream_t st;
  int slot;
  LaunchScope(int tag, hipStream_t s) : st(s), slot(-1) {
    if (!g_tl.on) return;
    if (g_tl.count < g_tl.capacity) {
      slot = g_tl.count;
      g_tl.tag[slot] = tag;
      (void)hipEventRecord(g_tl.ev[2 * slot], st);
    }
    ++g_tl.count;
  }
  ~LaunchScope() {
    if (slot >= 0) (void)hipEventRecord(g_tl.ev[2 * slot + 1], st);
  }
};

// Kernels that need more than 64 KB of dynamic LDS must be told so once PER DEVICE (the attribute lives in the device's
// copy of the code object); `done` is the per-kernel bit mask of devices already served.  One process per GPU is the
// deployment model, but a process that drives several devices must not find the second one unprepared.
inline int ensure_dynamic_lds(const void *kernel, size_t bytes, std::atomic<uint64_t> &done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return PLS_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(PLS_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  done.fetch_or(bit, std::memory_order_relaxed);
  return PLS_OK;
}

static inline hipStream_t S(void *stream) { return reinterpret_cast<hipStream_t>(stream); }
__host__ __device__ static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace plship
