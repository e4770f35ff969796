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
std::atomic<unsigned long long*>& ka_debug_stamps() {
    static std::atomic<unsigned long long*> p{nullptr};
    return p;
}
extern "C" int ka_version(void) { return 1; }
extern "C" const char* ka_target_arch(void) { return "gfx950"; }

// ---- run-time switches: the environment is read once (common.h, KA_OPTIONS) -------------------------------------------------
#include <atomic>
#include <stdlib.h>
static const char* const g_opt_names[KA_OPT_COUNT] = {
#define KA_OPT_NAME(n) "KA_" #n,
    KA_OPTIONS(KA_OPT_NAME)
#undef KA_OPT_NAME
};
static KaOptVal g_opt_tables[2][KA_OPT_COUNT];               // the table in use and the one a reload fills
static std::atomic<const KaOptVal*> g_opts{nullptr};
static std::atomic<int> g_opt_flip{0};
static const KaOptVal* ka_opts_load() {
    KaOptVal* t = g_opt_tables[g_opt_flip.fetch_add(1) & 1];
    for (int i = 0; i < KA_OPT_COUNT; ++i) {
        const char* e = getenv(g_opt_names[i]);
        t[i].set = e != nullptr;
        t[i].val = e ? atoi(e) : 0;
    }
    g_opts.store(t, std::memory_order_release);
    return t;
}
const KaOptVal* ka_opts() {
    const KaOptVal* t = g_opts.load(std::memory_order_acquire);
    return t ? t : ka_opts_load();
}
// re-read every KA_* switch from the environment (a process that changes one after its first launch: tests, A/B tools);
// returns the number of switches set.  Not meant to race with launches of other threads.
extern "C" int ka_options_reload(void) {
    const KaOptVal* t = ka_opts_load();
    int n = 0;
    for (int i = 0; i < KA_OPT_COUNT; ++i) n += t[i].set;
    return n;
}
