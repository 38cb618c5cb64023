// The 7x7 / stride 2 / pad 3 stem convolution of ResNet (models/resnet.py:137-139, 197-199: conv1 3 -> 64, BN, ReLU) as a DIRECT
// fp32-MFMA kernel (gfx950).
//
// As an implicit GEMM through the generic stager the stem is the worst layer of the image branch (51 TFLOP/s): K = 147 is three
// input channels deep, so every K-step re-decodes filter taps and re-gathers stride-2 pixels that overlap 12-fold between taps.
// Here a workgroup owns TWO output row segments of 128 pixels x all 64 output channels and stages its whole operand set once:
//   * the packed filter bank W'[k'][co], k' = (ci, ky, kx padded 7 -> 8, zero weights for the pad): 43 KB,
//   * the 3 x 9 input row segments it needs (261 columns), DE-INTERLEAVED by column parity: 28 KB
// then issues 4 x 84 MFMAs per wave without another barrier (each A fragment feeds both rows).  With the parity split, the B operand of v_mfma_f32_32x32x2_f32
// (lane l: k = 2m + (l>>5), pixel = l&31) is patch[ci][ky][parity = l>>5][m + pixel]: one conflict-free ds_read_b32 whose address is
// a per-lane constant plus an IMMEDIATE -- no address arithmetic, no im2col panel, no second pass over the pixels.  The A operand is
// W'[(2m + (l>>5))][co] the same way.  Executed MFMA work is 168/147 of the algorithmic (the zero tap).
// Epilogue: folded BatchNorm scale/shift + ReLU, 128-byte row stores.
#include "common.h"
#include "mfma_tile.h"

namespace {

constexpr int ST_CO = 64, ST_PX = 128, ST_KP = 3 * 7 * 8;            // 168 padded taps
constexpr int ST_PC = 132;                                            // columns per parity plane (131 used)
constexpr int ST_ROWS = 2, ST_LINES = 3 * (2 * ST_ROWS + 5);          // output rows per workgroup; input lines = ci x (2R+5) rows
constexpr int ST_W_FLOATS = ST_KP * ST_CO, ST_P_FLOATS = ST_LINES * 2 * ST_PC;
constexpr int ST_LDS_FLOATS = ST_W_FLOATS + ST_P_FLOATS;          // resident filter bank + one patch: 71.5 KB, two workgroups per CU

// weight[64][3][7][7] -> Wp[(ci*7+ky)*8+kx][co] (kx = 7: zero)
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ Wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ST_KP * ST_CO) return;
    const int co = i % ST_CO, kp = i / ST_CO, kx = kp & 7, cy = kp >> 3;      // cy = ci*7 + ky
    Wp[i] = kx < 7 ? w[(co * 21 + cy) * 7 + kx] : 0.0f;
}

// Persistent: two workgroups per CU walk over (frame, row pair, column tile) work items with the filter bank RESIDENT in LDS (staged
// once per workgroup, not once per 256 pixels); the next item's 28 patch values per thread are fetched into registers before the
// 336 MFMAs of the current one and written to LDS after them; while one workgroup stores its outputs and swaps patches, the other
// one's MFMAs keep the matrix pipe busy.
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ y, int H, int W, int OH, int OW,
                                                        int n_ct, int n_rt, int total, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ws = lds;
    float* Ps = lds + ST_W_FLOATS;                   // [ST_P_FLOATS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    constexpr int LPC = 2 * ST_ROWS + 5, PER = (ST_P_FLOATS + 255) / 256;       // 28 patch values per thread
    for (int f = tid; f < ST_W_FLOATS / 4; f += 256) reinterpret_cast<float4*>(Ws)[f] = reinterpret_cast<const float4*>(Wp)[f];
    float pv[PER];
    auto fetch = [&](int item) __attribute__((always_inline)) {        // flat element e = tid + 256*j -> (line, parity plane, slot)
        const int ct = item % n_ct, rt = (item / n_ct) % n_rt, b = item / (n_ct * n_rt);
        const float* xb = x + (long long)b * 3 * H * W;
        const int ox0 = ct * ST_PX, oy0 = rt * ST_ROWS;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = tid + 256 * j;
            const int line = e / (2 * ST_PC), w = e - line * (2 * ST_PC), par = w / ST_PC, slot = w - par * ST_PC;
            const int ci = line / LPC, r = line - ci * LPC;
            const int iy = 2 * oy0 - 3 + r, c = 2 * ox0 - 3 + 2 * slot + par;
            const bool in = e < ST_P_FLOATS && iy >= 0 && iy < H && c >= 0 && c < W;
            const float v = xb[((long long)min(ci, 2) * H + min(max(iy, 0), H - 1)) * W + min(max(c, 0), W - 1)];
            pv[j] = in ? v : 0.0f;
        }
    };
    auto put = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = tid + 256 * j;
            if (e < ST_P_FLOATS) Ps[e] = pv[j];
        }
    };
    int item = blockIdx.x;
    if (item >= total) return;
    fetch(item);
    put();
    __syncthreads();
    const float* Ab = Ws + half * ST_CO + l31;                     // + k'*64 (+32 for the upper channel half): immediates
    for (; item < total; item += gridDim.x) {
        const int next = item + gridDim.x;
        if (next < total) fetch(next);
        f32x16 acc[ST_ROWS][2];
#pragma unroll
        for (int j = 0; j < ST_ROWS; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][h][r] = 0.0f;
        const float* Bb = Ps + half * ST_PC + wave * 32 + l31;      // + (ci*(2R+5) + 2j + ky)*264 + m: immediates
        DI2P_MFMA_BEGIN();
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int cy = ci * 7 + ky;
                float a0[4], a1[4], bv[ST_ROWS][4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    a0[m] = Ab[(cy * 8 + 2 * m) * ST_CO];
                    a1[m] = Ab[(cy * 8 + 2 * m) * ST_CO + 32];
#pragma unroll
                    for (int j = 0; j < ST_ROWS; ++j) bv[j][m] = Bb[(ci * LPC + 2 * j + ky) * 2 * ST_PC + m];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < ST_ROWS; ++j) {
                        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], bv[j][m], acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], bv[j][m], acc[j][1], 0, 0, 0);
                    }
            }
        DI2P_MFMA_END();
        const int ct = item % n_ct, rt = (item / n_ct) % n_rt, b = item / (n_ct * n_rt);
        const int ox = ct * ST_PX + wave * 32 + l31;
        if (ox < OW) {
#pragma unroll
            for (int j = 0; j < ST_ROWS; ++j) {
                const int oy = rt * ST_ROWS + j;
                if (oy >= OH) break;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = h * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[j][h][r] * scale[co] + shift[co];
                        if (relu) v = fmaxf(v, 0.0f);
                        y[(((long long)b * ST_CO + co) * OH + oy) * OW + ox] = v;
                    }
            }
        }
        __syncthreads();                             // every wave is done with the patch
        if (next < total) { put(); __syncthreads(); }
    }
}

}  // namespace

extern "C" int di2p_stem_pack(const float* weight, float* Wp, void* stream) {
    DI2P_CHECK_ARG(weight && Wp, "null pointer");
    hipLaunchKernelGGL(stem_pack_kernel, dim3(di2p_cdiv(ST_KP * ST_CO, 256)), dim3(256), 0, (hipStream_t)stream, weight, Wp);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_conv7x7s2_stem(const float* x, const float* Wp, const float* scale, const float* shift, float* y, int B, int H, int W, int relu,
                                   void* stream) {
    DI2P_CHECK_ARG(x && Wp && scale && shift && y, "null pointer");
    DI2P_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && ((uintptr_t)Wp & 15) == 0, "bad size / Wp must be 16-byte aligned");
    DI2P_CHECK_ARG((long long)3 * H * W < (1ll << 31), "per-image extent must fit 31 bits");
    if (B == 0) return 0;
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int n_ct = di2p_cdiv(OW, ST_PX), n_rt = di2p_cdiv(OH, ST_ROWS);
    const long long total = (long long)B * n_rt * n_ct;
    DI2P_CHECK_ARG(total < (1ll << 31), "too many work items");
    const long long grid = total < 512 ? total : 512;            // two persistent workgroups per CU
    (void)hipFuncSetAttribute((const void*)stem_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ST_LDS_FLOATS * sizeof(float)));
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)grid), dim3(256), ST_LDS_FLOATS * sizeof(float), (hipStream_t)stream, x, Wp, scale, shift, y, H, W, OH,
                       OW, n_ct, n_rt, (int)total, relu);
    DI2P_RETURN_LAUNCH();
}
