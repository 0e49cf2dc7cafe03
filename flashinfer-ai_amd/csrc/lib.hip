// Error channel and device queries of libfi_mi355.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace fi {

static thread_local char g_err[1024] = "";

int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}
const char* last_error() { return g_err; }

}  // namespace fi

extern "C" FI_API const char* fi_last_error(void) { return fi::last_error(); }
extern "C" FI_API int fi_abi_version(void) { return FI_ABI_VERSION; }

extern "C" FI_API int fi_num_compute_units(void) {
  if (const char* e = getenv("FI_NUM_CUS")) {
    int v = atoi(e);
    if (v > 0) return v;
  }
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
    return n;
  (void)hipGetLastError();
  return 256;  // MI355X
}
