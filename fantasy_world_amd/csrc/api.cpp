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

// ---- tuning / A-B hooks (never needed for correctness): FW_OPT_* slots, initialised from the environment once.
#include <stdlib.h>
static int g_opt[FW_OPT_COUNT];
static bool g_opt_init = false;
static void opt_init() {
    if (g_opt_init) return;
    g_opt_init = true;
    const char* names[FW_OPT_COUNT] = {"FW_GEMM_TILE", "FW_GEMM_KERNEL", "FW_GEMM_VAR", "FW_ATTN_VAR"};
    const int defaults[FW_OPT_COUNT] = {0, 9, 0, 192};
    for (int i = 0; i < FW_OPT_COUNT; ++i) {
        const char* e = getenv(names[i]);
        g_opt[i] = e ? atoi(e) : defaults[i];
    }
}
int fw_get_option(int opt) {
    opt_init();
    return (opt >= 0 && opt < FW_OPT_COUNT) ? g_opt[opt] : 0;
}
extern "C" int fw_set_option(int opt, int value) {
    opt_init();
    if (opt < 0 || opt >= FW_OPT_COUNT) { fw_set_error("fw_set_option: unknown option"); return FW_E_BADARG; }
    g_opt[opt] = value;
    return 0;
}
