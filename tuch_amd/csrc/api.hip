// Error reporting and version for the C ABI declared in include/tuch_amd.h.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_error[512] = "";

void tuch_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int tuch_check_launch(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        tuch_set_error("%s: %s", what, hipGetErrorString(e));
        return TUCH_ERR_HIP;
    }
    return TUCH_OK;
}

extern "C" const char* tuch_last_error(void) { return g_error; }
extern "C" int tuch_abi_version(void) { return 1; }

#include <stdlib.h>
static int g_deterministic = [] { const char* e = getenv("TUCH_DETERMINISTIC"); return e && atoi(e) != 0 ? 1 : 0; }();
int tuch_deterministic() { return g_deterministic; }
extern "C" void tuch_set_deterministic(int on) { g_deterministic = on ? 1 : 0; }
extern "C" int tuch_get_deterministic(void) { return g_deterministic; }
