// Error plumbing + ABI version for libfabind_hip.so
#include "common.h"
#include "fabind_hip.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void fabind_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* fabind_last_error(void) { return g_err; }
extern "C" int fabind_abi_version(void) { return FABIND_ABI_VERSION; }
extern "C" int fabind_sizeof_args(int which) {
    switch (which) {
        case 0: return (int)sizeof(FabindGemmArgs);
        case 1: return (int)sizeof(FabindEdgeBwdArgs);
        case 2: return (int)sizeof(FabindPairUpdateArgs);
        case 3: return (int)sizeof(FabindTnJob);
        case 4: return (int)sizeof(FabindAttnFusedBwdArgs);
        default: return -1;
    }
}
