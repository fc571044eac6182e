// Host-side plumbing shared by every entry point: thread-local error string,
// API version.  (No device code here; compiled by hipcc with the rest.)
#include "wr_common.hpp"

#include <atomic>

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

namespace wr {
namespace {
std::atomic<int> g_tune[kTuneCount] = {{0}, {0}, {7}, {16}, {16}, {2}, {0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};
}
int tune_get(int key) { return (key >= 0 && key < kTuneCount) ? g_tune[key].load(std::memory_order_relaxed) : 0; }
}  // namespace wr

extern "C" int wr_tune_set(int key, int value)
{
    if (key < 0 || key >= wr::kTuneCount) {
        wr::set_error("wr_tune_set: unknown key %d", key);
        return WR_EINVAL;
    }
    wr::g_tune[key].store(value, std::memory_order_relaxed);
    return WR_OK;
}

extern "C" int wr_api_version(void) { return WR_API_VERSION; }
extern "C" const char *wr_last_error(void) { return wr::g_err; }
