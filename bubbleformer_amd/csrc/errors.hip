// Error reporting for the C ABI: entry points return a negative code, the text is kept per thread.
#include "bf_common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int bf_fail(hipError_t e, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s:%d: HIP error %d (%s)", file, line, (int)e, hipGetErrorString(e));
    return -(int)e - 1000;
}
int bf_fail_msg(const char* msg, const char* file, int line) {
    snprintf(g_err, sizeof(g_err), "%s:%d: %s", file, line, msg);
    return -1;
}
extern "C" const char* bf_last_error(void) { return g_err; }
extern "C" int bf_abi_version(void) { return 1; }
