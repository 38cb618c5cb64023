#include "common.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

static thread_local char g_err[512] = "ok";

void di2p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* di2p_last_error(void) { return g_err; }
extern "C" int di2p_version(void) { return 6; }

namespace {
struct OptDef { const char* name; const char* env; long long def; };
const OptDef kOpts[DI2P_OPT_COUNT] = {
    {"conv_nosplit", "DI2P_CONV_NOSPLIT", 0},        {"conv_split_blocks", "DI2P_CONV_SPLIT_BLOCKS", 32},
    {"conv_novec", "DI2P_CONV_NOVEC", 0},            {"conv_cfg", "DI2P_CONV_CFG", -1},
    {"conv_depth1", "DI2P_CONV_DEPTH1", 0},          {"index_max_rows", "DI2P_INDEX_MAX_ROWS", 1024},
    {"pw_novec", "DI2P_PW_NOVEC", 0},                {"solver_cfg", "DI2P_SOLVER_CFG", 44},
    {"solver_nocull", "DI2P_SOLVER_NOCULL", 0},      {"solver_noprefilter", "DI2P_SOLVER_NOPREFILTER", 0},
    {"solver_tier_sweeps", "DI2P_SOLVER_TIER_SWEEPS", 0},
    {"wino_cob", "DI2P_WINO_COB", 0},               {"conv_nowinograd", "DI2P_CONV_NOWINOGRAD", 0},
    {"wino_db", "DI2P_WINO_DB", 1},                 {"wino_map", "DI2P_WINO_MAP", 0},
    {"wino_kc", "DI2P_WINO_KC", 0},                 {"conv_nostem", "DI2P_CONV_NOSTEM", 0},
    {"pw_cfg", "DI2P_PW_CFG", 0},                   {"wino_reg", "DI2P_WINO_REG", 0},                 {"wino_reg_min", "DI2P_WINO_REG_MIN", 256},
    {"solver_lds_pad", "DI2P_SOLVER_LDS_PAD", 0},   {"solver_nocache", "DI2P_SOLVER_NOCACHE", 0},
    {"solver_prep_single", "DI2P_SOLVER_PREP_SINGLE", 0}, {"solver_prep_bitonic", "DI2P_SOLVER_PREP_BITONIC", 0},
    {"pw_x3", "DI2P_PW_X3", 1},                     {"pw_nochain", "DI2P_PW_NOCHAIN", 0},
    {"head_reg", "DI2P_HEAD_REG", 0},               {"conv_s2scalar", "DI2P_CONV_S2SCALAR", 0},
    {"conv_x3", "DI2P_CONV_X3", 31},                 {"conv_x3_cfg", "DI2P_CONV_X3_CFG", -1},
    {"head_x3", "DI2P_HEAD_X3", 1},                 {"head_x3_tab", "DI2P_HEAD_X3_TAB", 1},
    {"stem_x3", "DI2P_STEM_X3", 1},                 {"bn_unfused", "DI2P_BN_UNFUSED", 0},
    {"pw_x3_planes", "DI2P_PW_X3_PLANES", 1},        {"conv_dgrad_dense", "DI2P_CONV_DGRAD_DENSE", 0},
    {"rc_tile64", "DI2P_RC_TILE64", 0},
};
long long g_opt[DI2P_OPT_COUNT];
std::once_flag g_opt_once;
void opts_init() {
    for (int i = 0; i < DI2P_OPT_COUNT; ++i) {
        const char* e = getenv(kOpts[i].env);
        g_opt[i] = e ? atoll(e) : kOpts[i].def;
    }
}
}  // namespace

int di2p_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

long long di2p_opt(int id) {
    std::call_once(g_opt_once, opts_init);
    return g_opt[id];
}

// name = lower-case knob name ("conv_nosplit", "solver_nocull", ...).  Returns 0, or -1 for an unknown name.
extern "C" int di2p_set_option(const char* name, long long value) {
    std::call_once(g_opt_once, opts_init);
    for (int i = 0; i < DI2P_OPT_COUNT; ++i)
        if (name && strcmp(name, kOpts[i].name) == 0) { g_opt[i] = value; return 0; }
    di2p_set_error("di2p_set_option: unknown option");
    return -1;
}

extern "C" long long di2p_get_option(const char* name) {
    std::call_once(g_opt_once, opts_init);
    for (int i = 0; i < DI2P_OPT_COUNT; ++i)
        if (name && strcmp(name, kOpts[i].name) == 0) return g_opt[i];
    return -1;
}
