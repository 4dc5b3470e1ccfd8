// Error plumbing + version/device probes of the C ABI (include/probpose_hip.h).
#include <stdarg.h>

#include "pp_common.h"

namespace pp {
char *err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return -1;
}
}  // namespace pp

extern "C" int pp_version(void) { return 100; }

extern "C" const char *pp_last_error(void) { return pp::err_buf(); }

extern "C" int pp_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
