// capi.hip -- ABI version and thread-local error message.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace grafp {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace grafp

extern "C" int grafp_abi_version(void) { return GRAFP_ABI_VERSION; }
extern "C" const char *grafp_last_error(void) { return grafp::g_err; }
