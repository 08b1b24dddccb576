// Thread-local error string + ABI version for libfocal_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/focal_hip.h"

static thread_local char g_err[512] = "";

void focal_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// The kernel the GEMM dispatchers launched last on this thread (template name with the parameters that distinguish instantiations in a
// profiler trace): bench.py's in-step tracer labels its launches with it, so that its groups are the rows of `rocprofv3 --stats`.
static thread_local char g_kernel[160] = "";
void focal_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}
extern "C" const char* focal_last_kernel(void) { return g_kernel; }

extern "C" int focal_abi_version(void) { return FOCAL_ABI_VERSION; }
extern "C" const char* focal_last_error(void) { return g_err; }

