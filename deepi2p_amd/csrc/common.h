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

// Tuning / test knobs.  Read ONCE from the environment (DI2P_<NAME>) when the library is first used and cached: no
// getenv() on the launch path.  Tests and tools flip them at run time through di2p_set_option().
enum Di2pOption {
    DI2P_OPT_CONV_NOSPLIT = 0,      // 1: never split K in the convolutions
    DI2P_OPT_CONV_SPLIT_BLOCKS,     // per-frame workgroup count below which K is split (default 32)
    DI2P_OPT_CONV_NOVEC,            // 1: scalar stager for the convolutions
    DI2P_OPT_CONV_CFG,              // >= 0: force a convolution tile configuration (experiments)
    DI2P_OPT_CONV_DEPTH1,           // 1: depth-1 register prefetch in the vector convolution engine (default: depth 2; bit-identical)
    DI2P_OPT_INDEX_MAX_ROWS,        // index_max splits rows along N (3 launches) while B*C*S is below this many workgroups
    DI2P_OPT_PW_NOVEC,              // 1: scalar stager for the pointwise GEMMs
    DI2P_OPT_SOLVER_CFG,            // <waves per hypothesis><min waves per SIMD>, default 44
    DI2P_OPT_SOLVER_NOCULL,         // 1: classify every cluster per point (bit-identical by construction)
    DI2P_OPT_SOLVER_NOPREFILTER,    // 1: skip the fp32 pre-filter of the per-point classification (bit-identical by construction)
    DI2P_OPT_SOLVER_TIER_SWEEPS,    // sweeps after which a hypothesis is handed to the wide (16-wave) tail kernel; 0 = never
    DI2P_OPT_WINO_COB,              // 32 / 64: force the output-channel block of the Winograd convolution (0: by grid size)
    DI2P_OPT_CONV_NOWINOGRAD,       // 1: the host layer runs 3x3 stride-1 convolutions on the direct implicit-GEMM kernel (read by networks.py)
    DI2P_OPT_WINO_DB,               // 1 (default): double-buffered operand panels in the Winograd convolution; 0: single (measured slower)
    DI2P_OPT_WINO_MAP,              // workgroup -> XCD mapping of the Winograd convolution: 0 automatic, 1 by tile block, 2 by co-block
    DI2P_OPT_WINO_KC,               // 8 / 4: input channels per K-step of the Winograd convolution (0: 4 up to 256 input channels, else 8)
    DI2P_OPT_CONV_NOSTEM,           // 1: the host layer runs the 7x7 stem on the generic implicit-GEMM kernel (read by networks.py)
    DI2P_OPT_PW_CFG,                // tile of the vector pointwise GEMM: 0/1 = 64x64 (default), 2 = 64x128, 3 = 128x128, >= 16: larger tiles from that many workgroups on
    DI2P_OPT_WINO_REG,              // Winograd kernel: 0 automatic, 1 LDS-panel kernel, 2 register-resident (4 waves), 3 register-resident (2 waves)
    DI2P_OPT_WINO_REG_MIN,          // automatic choice: register-resident Winograd kernel from this many 64-tile workgroups on (default 256: all but the 512-channel stage)
    DI2P_OPT_SOLVER_LDS_PAD,        // bytes of unused dynamic LDS per solver workgroup (caps its workgroups per CU; experiments)
    DI2P_OPT_SOLVER_NOCACHE,        // 1: no classification cache in the cluster walk (bit-identical by construction)
    DI2P_OPT_SOLVER_PREP_SINGLE,    // 1: frame preparation as ONE workgroup per frame (the round-4 kernel; default: five multi-workgroup launches; same results)
    DI2P_OPT_SOLVER_PREP_BITONIC,   // 1: frame preparation always sorts with the bitonic network (default: counting sort + per-bucket ranking; same order)
    DI2P_OPT_PW_X3,                 // 1 (default): the host layer runs the GEMM-shaped pointwise layers (K >= 128, M % 128 == 0) on the bf16x3 kernel (read by ops.py)
    DI2P_OPT_PW_NOCHAIN,            // 1: the host layer runs the narrow PointNet chains as separate launches instead of di2p_point_chain (bit-identical; read by ops.py)
    DI2P_OPT_HEAD_REG,              // 1: di2p_point_head runs the wave-autonomous kernel (one persistent 8-wave workgroup per compute unit: faster alone, slower beside other streams' kernels) instead of the LDS-tile kernel (bit-identical)
    DI2P_OPT_CONV_S2SCALAR,         // 1: stride-2 convolutions stage their operand with four dword loads per row (rounds 1-3) instead of aligned 8-float windows (bit-identical)
    DI2P_OPT_CONV_X3,               // bit mask of the 3x3 layers the host layer runs on di2p_conv3x3_x3 where it supports their shape (bf16 MFMA, exact three-way splits): bit s-1 = stride-1 layers of ResNet stage s, bit 4 = the stride-2 layers with their 1x1 downsample branch; 0: Winograd / direct fp32-MFMA kernels only (read by networks.py)
    DI2P_OPT_CONV_X3_CFG,           // >= 0: force tile configuration 0..3 of di2p_conv3x3_x3 (default -1: cheapest by a cost model)
    DI2P_OPT_HEAD_X3,               // 1 (default): the host layer runs the coarse per-point head on di2p_point_head_x3 (bf16 MFMA, exact three-way splits, wave-autonomous); 0: di2p_point_head (fp32 MFMA, LDS tile; bit-identical to the three separate launches) (read by networks.py)
    DI2P_OPT_HEAD_X3_TAB,           // 1 (default): di2p_point_head_x3 keeps the frame's two node tables in LDS, one workgroup of eight waves (two per SIMD, 256 registers) per compute unit; 2: the same with four waves (one per SIMD, 512 registers); 0: gathers the tables from memory (no LDS: shares its compute units)
    DI2P_OPT_STEM_X3,               // 1 (default): the host layer runs conv1 + bn1 + relu + max-pool of the image branch as ONE launch of di2p_stem_x3 (bf16 MFMA, exact three-way splits) where it supports the image size; 0: di2p_conv7x7s2_stem + di2p_maxpool3x3s2 (fp32 MFMA) (read by networks.py)
    DI2P_OPT_BN_UNFUSED,            // 1: train-mode BatchNorm finalizes its statistics in a launch of its own (rounds 2-5; default: inside the elementwise pass, same results)
    DI2P_OPT_PW_X3_PLANES,          // 1 (default): consecutive bf16x3 pointwise layers hand their activations on as split bf16 planes (di2p_epilogue_t.planes_out -> di2p_pointwise_gemm_x3p) instead of fp32 (bit-identical; read by networks.py); 2: the same, and di2p_pointwise_gemm_x3p always runs its 128-row kernel (default: 256-row tiles with both operands in LDS when M % 256 == 0)
    DI2P_OPT_CONV_DGRAD_DENSE,      // 1: di2p_conv2d_dgrad runs its dense kernel for stride 2 too (default: one parity class of input pixels per workgroup, 1/4 of the matrix work; same results)
    DI2P_OPT_RC_TILE64,             // 1: the reduction GEMMs of the training step (weight gradients) always run 64 x 64 tiles (default: 128 x 128 for strided operand pairs with at least 128 rows and columns)
    DI2P_OPT_COUNT
};
long long di2p_opt(int id);
int di2p_cu_count();      // compute units of the current device (cached per device)
