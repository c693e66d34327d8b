// runtime.hip -- version / error plumbing of libcloudaae_hip.so.
#include "common.h"
#include "../../include/cloudaae_hip.h"
#include <stdarg.h>
#include <stdio.h>

namespace cloudaae {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace cloudaae

CLOUDAAE_API int cloudaae_version(void) { return 100; }
CLOUDAAE_API const char *cloudaae_last_error(void) { return cloudaae::g_err; }
