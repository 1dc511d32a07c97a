#include <stdarg.h>

#include "mvsn_common.h"

namespace mvsn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mvsn

extern "C" int mvsn_abi_version(void) { return MVSN_ABI_VERSION; }
extern "C" const char *mvsn_last_error(void) { return mvsn::g_err; }
