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
// on by default since round 6 (SURVEY section 8b asks for a deterministic path; measured cost at batch 64: none --
// 0.4143 against 0.4151 ms per step, BENCH_r05); TUCH_DETERMINISTIC=0 selects the float atomics
static int g_deterministic = [] { const char* e = getenv("TUCH_DETERMINISTIC"); return e && atoi(e) == 0 ? 0 : 1; }();
int tuch_deterministic() { return g_deterministic; }
extern "C" void tuch_set_deterministic(int on) { g_deterministic = on ? 1 : 0; }
extern "C" int tuch_get_deterministic(void) { return g_deterministic; }

// TUCH_HOST_TABLES=1 (read once, when the library is loaded): the model constants stay in HOST memory instead of being
// uploaded -- for running the host-side table builders (tuch_contact_model_create, tuch_hd_model_create,
// tuch_smpl_model_create) under AddressSanitizer / UBSan on a machine without a device (tests/test_sanitized_host.py).
// Handles made this way are good for create / info / export / destroy only; no kernel may be launched on them.
static int g_host_tables = [] { const char* e = getenv("TUCH_HOST_TABLES"); return e && atoi(e) != 0 ? 1 : 0; }();
int tuch_host_tables() { return g_host_tables; }
#include <string.h>
int tuch_table_upload(void** dst, const void* src, size_t bytes)
{
    *dst = nullptr;
    if (bytes == 0) return TUCH_OK;
    if (g_host_tables) {
        *dst = malloc(bytes);
        if (!*dst) { tuch_set_error("model table: malloc(%zu) failed", bytes); return TUCH_ERR_HIP; }
        memcpy(*dst, src, bytes);
        return TUCH_OK;
    }
    if (hipMalloc(dst, bytes) != hipSuccess) {
        *dst = nullptr;
        tuch_set_error("model table: hipMalloc(%zu) failed", bytes);
        return TUCH_ERR_HIP;
    }
    if (hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
        tuch_set_error("model table: hipMemcpy of %zu bytes failed", bytes);
        return TUCH_ERR_HIP;
    }
    return TUCH_OK;
}
int tuch_table_download(void* dst_host, const void* src, size_t bytes)
{
    if (g_host_tables) { memcpy(dst_host, src, bytes); return TUCH_OK; }
    return hipMemcpy(dst_host, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? TUCH_OK : TUCH_ERR_HIP;
}
void tuch_table_free(void* p)
{
    if (!p) return;
    if (g_host_tables) free(p); else (void)hipFree(p);
}
