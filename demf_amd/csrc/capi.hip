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
