// ABI housekeeping entry points of libpita_hip.so.
#include "common.h"

extern "C" int pita_abi_version(void) { return PITA_ABI_VERSION; }

extern "C" int pita_last_error(char* buf, size_t buflen) {
  const char* m = pita::err_buf();
  size_t n = strlen(m);
  if (buf && buflen) {
    size_t k = n < buflen - 1 ? n : buflen - 1;
    memcpy(buf, m, k);
    buf[k] = 0;
  }
  return (int)n;
}

extern "C" int pita_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return pita::fail(PITA_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}
