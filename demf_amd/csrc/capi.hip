// C-ABI plumbing of libdemf_hip.so: version + thread-local error text.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace demf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return DEMF_ELAUNCH;
  }
  return DEMF_OK;
}

}  // namespace demf

extern "C" int demf_version(void) { return DEMF_ABI_VERSION; }
extern "C" const char* demf_last_error(void) { return demf::g_err; }

// A HIP stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask).  Used by the
// training engine to pin the latency-bound FPS pre-pass of the next batch to a few CUs and keep
// the main stream off them: co-resident FPS waves are the oldest on their SIMDs and win issue
// arbitration, which otherwise stretches the tail of every main-stream kernel (measured).
// `cus` lists CU indices; `invert` != 0 selects all CUs EXCEPT those.
extern "C" int demf_stream_create_cu_masked(const int* cus, int n, int invert, void** out) {
  DEMF_REQUIRE(cus && n >= 0 && out, "stream_create_cu_masked: bad arguments");
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    demf::set_error("stream_create_cu_masked: cannot query the device");
    return DEMF_ELAUNCH;
  }
  const int ncu = prop.multiProcessorCount;
  const int words = (ncu + 31) / 32;
  uint32_t mask[16] = {0};
  DEMF_REQUIRE(words <= 16, "stream_create_cu_masked: %d CUs unsupported", ncu);
  for (int i = 0; i < n; ++i)
    if (cus[i] >= 0 && cus[i] < ncu) mask[cus[i] / 32] |= 1u << (cus[i] % 32);
  if (invert)
    for (int i = 0; i < ncu; ++i) mask[i / 32] ^= 1u << (i % 32);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
  if (e != hipSuccess) {
    demf::set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return DEMF_ELAUNCH;
  }
  *out = (void*)s;
  return DEMF_OK;
}

namespace demf {
// wall_clock64 ticks at 100 MHz on gfx9: 100 ticks per microsecond
__global__ __launch_bounds__(64) void spin_us_k(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace demf

extern "C" int demf_spin_us(int microseconds, demf_stream_t stream) {
  DEMF_REQUIRE(microseconds >= 0 && microseconds <= 1000000, "spin_us: %d us out of range", microseconds);
  hipLaunchKernelGGL(demf::spin_us_k, dim3(1), dim3(64), 0, (hipStream_t)stream, 100ll * microseconds);
  return demf::check_launch("spin_us_k");
}
