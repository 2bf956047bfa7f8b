#include "mval_common.h"

static thread_local char g_err[512] = "";

void mval_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mval_last_error(void) { return g_err; }
extern "C" int mval_version(void) { return 100; }

// Compute units of the current device, cached per device (the persistent launchers size their grids with it).
#include <atomic>
int mval_cu_count() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;  // (MI355X: 256)
  cus[dev].store(n, std::memory_order_relaxed);
  return n;
}
