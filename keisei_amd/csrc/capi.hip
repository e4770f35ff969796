// Error plumbing and library identity for libkeisei_amd.so (C ABI, no C++ exceptions cross it).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void ka_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ka_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ka_set_error("%s: %s", what, hipGetErrorString(e));
        return KA_ERR_HIP;
    }
    return KA_OK;
}

extern "C" const char* ka_last_error(void) { return g_err; }
extern "C" int ka_version(void) { return 1; }
extern "C" const char* ka_target_arch(void) { return "gfx950"; }
