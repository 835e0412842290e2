// Error plumbing and version of the C ABI (include/tdship.h).
#include <stdarg.h>
#include <string.h>

#include "tds_common.h"

namespace tds {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace tds

TDS_EXPORT int tds_version(void) { return TDS_ABI_VERSION; }

TDS_EXPORT int tds_last_error(char *buf, size_t n) {
    size_t len = strlen(tds::g_err);
    if (buf && n) {
        size_t k = len < n - 1 ? len : n - 1;
        memcpy(buf, tds::g_err, k);
        buf[k] = 0;
    }
    return (int)len;
}
