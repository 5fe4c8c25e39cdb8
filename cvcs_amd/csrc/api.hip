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

// ---- cvcs_replay: a recorded launch plan (cvcs_amd/_lib.py Recording) driven from C.  Every launch entry point of this library takes only
// integer-class arguments (pointers, int, int64: general registers, then 8-byte stack slots, in order) and at most 8 floats (xmm0-7, in
// order) and ends with the stream - on the x86-64 System V ABI one prototype with 28 integer slots and 8 float slots therefore calls
// them all: surplus integer arguments sit unread in stack slots the caller cleans up, surplus floats in caller-saved registers.
typedef int (*cvcs_tramp_t)(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                            int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                            float, float, float, float, float, float, float, float);

extern "C" int cvcs_sizeof_call(void) { return (int)sizeof(cvcs_call); }

extern "C" int cvcs_replay(const cvcs_call* calls, int n, void* stream, int* failed_index) {
#if !defined(__x86_64__)
  cvcs::set_error("cvcs_replay: built for the x86-64 System V calling convention only");
  return CVCS_EUNSUPPORTED;
#else
  if (!calls || n < 0) { cvcs::set_error("cvcs_replay: bad argument"); return CVCS_EINVAL; }
  for (int k = 0; k < n; ++k) {
    const cvcs_call& c = calls[k];
    if (!c.fn || c.nint < 0 || c.nint >= CVCS_CALL_MAX_INT || c.nflt < 0 || c.nflt > CVCS_CALL_MAX_FLT) {
      if (failed_index) *failed_index = k;
      cvcs::set_error("cvcs_replay: call %d: bad record (nint=%d, nflt=%d)", k, c.nint, c.nflt);
      return CVCS_EINVAL;
    }
    int64_t a[CVCS_CALL_MAX_INT];
    for (int j = 0; j < CVCS_CALL_MAX_INT; ++j) a[j] = j < c.nint ? c.i[j] : 0;
    a[c.nint] = (int64_t)(intptr_t)stream;                 // the stream is the last argument of every launch entry point
    const float* f = c.f;
    const int rc = ((cvcs_tramp_t)c.fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17],
                                        a[18], a[19], a[20], a[21], a[22], a[23], a[24], a[25], a[26], a[27], f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    if (rc != 0) {
      if (failed_index) *failed_index = k;
      return rc;                                            // (cvcs_last_error holds the callee's message)
    }
  }
  return CVCS_OK;
#endif
}

