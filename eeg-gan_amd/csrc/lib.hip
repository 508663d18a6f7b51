// Error plumbing of libeeggan_hip.so.
#include "common.h"

#include <cstring>

namespace eeggan {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

}  // namespace eeggan

extern "C" const char* eeggan_last_error(void) { return eeggan::g_err; }
extern "C" int eeggan_abi_version(void) { return 4; }
