// Error reporting and version query of libcvcs_hip.so.
#include <stdarg.h>

#include "common.h"

namespace cvcs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace cvcs

extern "C" const char* cvcs_last_error(void) { return cvcs::g_err; }
extern "C" int cvcs_abi_version(void) { return CVCS_ABI_VERSION; }
extern "C" int cvcs_sizeof_conv_desc(void) { return (int)sizeof(cvcs_conv_desc); }
extern "C" int cvcs_sizeof_wgrad_desc(void) { return (int)sizeof(cvcs_wgrad_desc); }
extern "C" int cvcs_sizeof_conv8_desc(void) { return (int)sizeof(cvcs_conv8_desc); }
