// Host-side plumbing shared by every entry point: thread-local error string,
// API version.  (No device code here; compiled by hipcc with the rest.)
#include "wr_common.hpp"

namespace wr {
namespace {
thread_local char g_err[512] = "";
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace wr

extern "C" int wr_api_version(void) { return WR_API_VERSION; }
extern "C" const char *wr_last_error(void) { return wr::g_err; }
