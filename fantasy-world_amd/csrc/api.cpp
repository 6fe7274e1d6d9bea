// Host-side pieces of the C ABI that are not kernel launchers.
#include "fw_common.h"
#include <string.h>

static thread_local char g_err[256] = "";

void fw_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* fw_last_error(void) { return g_err; }
extern "C" int fw_abi_version(void) { return FW_ABI_VERSION; }
