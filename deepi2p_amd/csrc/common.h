// Shared host/device helpers for libdeepi2p_hip.so (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deepi2p_hip.h"

#define DI2P_WAVE 64

void di2p_set_error(const char* fmt, ...);

#define DI2P_CHECK_ARG(cond, msg)                                   \
    do {                                                            \
        if (!(cond)) {                                              \
            di2p_set_error("%s: %s", __func__, msg);                \
            return -1;                                              \
        }                                                           \
    } while (0)

#define DI2P_RETURN_LAUNCH()                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            di2p_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                         \
        }                                                                           \
        return 0;                                                                   \
    } while (0)

static inline int di2p_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
