// Error text + version of the C ABI (include/ds_kernels.h).  Never throws across the boundary.
#include <stdarg.h>
#include <stdio.h>
#include "ds_kernels.h"

namespace ds {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ds

extern "C" int ds_version(void) { return 1; }
extern "C" const char *ds_last_error(void) { return ds::g_err; }
