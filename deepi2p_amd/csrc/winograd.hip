// 3x3 / stride 1 / pad 1 convolution as a FUSED Winograd F(2x2, 3x3) kernel on the fp32 MFMA (gfx950).
//
// 26 of the 36 convolutions of the image branch (models/resnet.py:56-72 BasicBlock conv1/conv2, :171-193) have this shape and hold
// 85 % of its multiply-accumulates.  Winograd's minimal filtering computes a 2x2 output tile from a 4x4 input tile with 16
// multiplications per (ci, co) pair instead of 36:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A ,   summed over ci inside the (.) product
// i.e. sixteen independent [Cout x Cin] * [Cin x tiles] contractions, one per position xi of the 4x4 transform domain.  It is exact
// in exact arithmetic and the algorithm vendor libraries pick for fp32 3x3 convolutions; in fp32 its rounding error is a small
// multiple of the direct form's (tests/test_gpu_contractions.py bounds it against an fp64 convolution).
//
// One workgroup (4 waves) = COB output channels x 32 tiles (128 output pixels); wave w owns xi = 4w .. 4w+3.
//   * weights are transformed once at load time (di2p_winograd_weight_transform) to U[xi][ci][co]: the A operand of
//     v_mfma_f32_32x32x2_f32 (lane l: A[co = l&31][k = l>>5]) is a conflict-free ds_read_b32 of a [xi][k][co] panel;
//   * per K-step of 8 input channels every thread loads ONE 4x4 input tile (clamped addresses, zeroed padding), applies B^T d B
//     in registers (32 adds) and writes its 16 values to the [xi][k][tile] panel: the transform-domain input never exists in HBM;
//   * LDS panels double buffered, next K-step's global loads issued before the current step's MFMAs (register prefetch);
//   * epilogue: the 16 xi of a (co, tile) live in four different waves -> they meet in LDS (8 channels x 32 tiles x 16 xi per
//     pass, re-using the operand panels), one thread per (co, tile) applies A^T M A, the folded BatchNorm scale/shift, the residual
//     and the ReLU and stores the 2x2 pixels as two 8-byte stores.
// Workgroups are ordered so that the co-blocks of one tile block run back to back on one XCD (they share the input tiles in L2).
#include "common.h"
#include "mfma_tile.h"

namespace {

constexpr int WG_TILES = 32;
typedef float f32x4 __attribute__((ext_vector_type(4)));      // a plain vector type: its loads / stores stay register values (no memcpy)

template <int COB, bool DB, int KC>
struct WinoLds {
    static constexpr int U_FLOATS = 16 * KC * COB, V_FLOATS = 16 * KC * WG_TILES;
    static constexpr int STAGE = U_FLOATS + V_FLOATS;
    static constexpr int TOTAL = (DB ? 2 : 1) * STAGE;
};

// U[xi][ci][co] = (G g G^T)[xi] of weight[co][ci][3][3];  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
// DGRAD: the filter of the input gradient of a stride-1 "same" convolution -- g'[co' = ci][ci' = co][ky][kx] = g[co][ci][2 - ky][2 - kx] -- read
// straight from the forward filter w[Cin][Cout][3][3] (Cin / Cout are the TRANSFORMED filter's, i.e. the forward layer's Cout / Cin): the
// training step used to flip, transpose and copy the filter with three launches before this one.
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Cin * Cout) return;
    const int co = (int)(i % Cout), ci = (int)(i / Cout);
    const float* g = DGRAD ? w + ((long long)ci * Cout + co) * 9 : w + ((long long)co * Cin + ci) * 9;
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g0 = DGRAD ? g[8 - c] : g[c], g1 = DGRAD ? g[5 - c] : g[3 + c], g2 = DGRAD ? g[2 - c] : g[6 + c];
        t[0][c] = g0;
        t[1][c] = 0.5f * (g0 + g1 + g2);
        t[2][c] = 0.5f * (g0 - g1 + g2);
        t[3][c] = g2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
        const long long base = ((long long)(r * 4) * Cin + ci) * Cout + co;
        U[base] = u0;
        U[base + (long long)Cin * Cout] = u1;
        U[base + 2ll * Cin * Cout] = u2;
        U[base + 3ll * Cin * Cout] = u3;
    }
}

// DB: operand panels double buffered (one barrier per K-step, 64 KB at COB 32: two workgroups per CU) or single buffered (two
// barriers per K-step, 32 KB: four workgroups per CU cover each other's barriers and the grid quantises finer)
// KC: input channels per K-step (8: every thread transforms one tile-channel per step; 4: half the panel bytes, waves 0-1 transform)
template <int COB, bool DB, int KC>
__global__ __launch_bounds__(256) void wino_conv_kernel(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y,
                                                        int Cin, int H, int W, int Cout, int TH, int TW, int total_tiles, int n_tb, int n_cb, int relu, int by_co) {
    using L = WinoLds<COB, DB, KC>;
    constexpr int UP = (16 * KC * COB / 4 + 255) / 256;      // 16-byte U loads per thread and K-step
    constexpr int MT = COB / 32;                 // MFMA row tiles per xi
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD-aware order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs; all co-blocks of a tile block go to one XCD
    const int lin = blockIdx.x, xcd = lin & 7, seq = lin >> 3;
    int cb, tb;
    if (by_co) {        // many co-blocks, large U: every XCD keeps ITS co-blocks' slice of U in its own L2 and streams all the tiles
        const int per = n_cb >> 3;
        cb = xcd * per + seq % per; tb = seq / per;
    } else {            // few co-blocks: all of them for one tile block on one XCD (they share the input tiles)
        cb = seq % n_cb; tb = (seq / n_cb) * 8 + xcd;
    }
    if (tb >= n_tb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int co_blk = cb * COB;

    // ---- loader roles
    const int tl = tid & 31, cl = tid >> 5;      // tile of the block, channel of the K-step
    const int gt = tb * WG_TILES + tl;
    const bool tvalid = gt < total_tiles;
    const int gtc = tvalid ? gt : total_tiles - 1;
    const int b = gtc / (TH * TW), rem = gtc - b * TH * TW, ty = rem / TW, tx = rem - ty * TW;
    // One 16-byte (dword-aligned) load per tile row: image columns cs .. cs+3 with cs = clamp(2tx-1, 0, W-4).  W is even, so the
    // wanted columns 2tx-1 .. 2tx+2 are the loaded ones shifted by -1 (left edge: column -1 is padding), 0, or +1 (right edge:
    // column W is padding); rows outside the image are loaded from a clamped row and zeroed.  (Measured alternatives: 16 scalar
    // loads per tile: 15 % slower; unclamped loads + a zero buffer for the padding rows, fewer selects: 20 % slower -- the
    // compiler then splits the 16-byte loads.)
    const int c0 = 2 * tx - 1, cs = min(max(c0, 0), W - 4);
    const bool left = c0 < cs, right = c0 > cs;
    bool rowok[4];
    const float* rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - 1 + r;
        rowok[r] = tvalid && iy >= 0 && iy < H;
        rp[r] = x + ((long long)b * Cin + cl) * H * W + (long long)min(max(iy, 0), H - 1) * W + cs;
    }
    const long long x_step = (long long)KC * H * W;
    const bool v_on = cl < KC;                   // wave-uniform: with KC = 4 only waves 0-1 stage input tiles
    const float* up[UP];
#pragma unroll
    for (int p = 0; p < UP; ++p) {
        const int f = tid + 256 * p, co4 = f % (COB / 4), row = f / (COB / 4);      // row = xi*KC + k
        up[p] = U + ((long long)(row / KC) * Cin + (row % KC)) * Cout + co_blk + co4 * 4;
    }
    const long long u_step = (long long)KC * Cout;
    const int T = Cin / KC;

    F4u drow[4];
    f32x4 ureg[UP];
    auto gload = [&](int t) __attribute__((always_inline)) {
        if (v_on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) drow[r] = *reinterpret_cast<const F4u*>(rp[r] + t * x_step);
        }
#pragma unroll
        for (int p = 0; p < UP; ++p) ureg[p] = *reinterpret_cast<const f32x4*>(up[p] + t * u_step);
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        float* Us = lds + buf * L::STAGE;
        float* Vs = Us + L::U_FLOATS;
#pragma unroll
        for (int p = 0; p < UP; ++p) {
            const int f = tid + 256 * p, co4 = f % (COB / 4), row = f / (COB / 4);
            *reinterpret_cast<f32x4*>(Us + row * COB + co4 * 4) = ureg[p];
        }
        if (!v_on) return;
        float v[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float l0 = rowok[r] ? drow[r].x : 0.0f, l1 = rowok[r] ? drow[r].y : 0.0f, l2 = rowok[r] ? drow[r].z : 0.0f,
                        l3 = rowok[r] ? drow[r].w : 0.0f;
            v[r * 4 + 0] = left ? 0.0f : (right ? l1 : l0);
            v[r * 4 + 1] = left ? l0 : (right ? l2 : l1);
            v[r * 4 + 2] = left ? l1 : (right ? l3 : l2);
            v[r * 4 + 3] = left ? l2 : (right ? 0.0f : l3);
        }
        float t[16];
#pragma unroll
        for (int s = 0; s < 4; ++s) {            // B^T d : rows
            t[0 * 4 + s] = v[0 * 4 + s] - v[2 * 4 + s];
            t[1 * 4 + s] = v[1 * 4 + s] + v[2 * 4 + s];
            t[2 * 4 + s] = v[2 * 4 + s] - v[1 * 4 + s];
            t[3 * 4 + s] = v[1 * 4 + s] - v[3 * 4 + s];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {            // (.) B : columns
            const float a0 = t[r * 4 + 0], a1 = t[r * 4 + 1], a2 = t[r * 4 + 2], a3 = t[r * 4 + 3];
            Vs[((r * 4 + 0) * KC + cl) * WG_TILES + tl] = a0 - a2;
            Vs[((r * 4 + 1) * KC + cl) * WG_TILES + tl] = a1 + a2;
            Vs[((r * 4 + 2) * KC + cl) * WG_TILES + tl] = a2 - a1;
            Vs[((r * 4 + 3) * KC + cl) * WG_TILES + tl] = a1 - a3;
        }
    };

    f32x16 acc[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < MT; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][h][r] = 0.0f;

    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = DB ? (t & 1) : 0;
        if (t + 1 < T) gload(t + 1);
        const float* Us = lds + buf * L::STAGE;
        const float* Vs = Us + L::U_FLOATS;
        DI2P_MFMA_BEGIN();
#pragma unroll
        for (int j = 0; j < 4; ++j) {            // (hoisting all 32 operand reads of the K-step ahead of the 16 MFMAs measured no gain)
            const int xi = wave * 4 + j;
            float a[KC / 2][MT], bv[KC / 2];
#pragma unroll
            for (int kk = 0; kk < KC; kk += 2) {
#pragma unroll
                for (int h = 0; h < MT; ++h) a[kk / 2][h] = Us[(xi * KC + kk + half) * COB + h * 32 + l31];
                bv[kk / 2] = Vs[(xi * KC + kk + half) * WG_TILES + l31];
            }
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk)
#pragma unroll
                for (int h = 0; h < MT; ++h) acc[j][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][h], bv[kk], acc[j][h], 0, 0, 0);
        }
        DI2P_MFMA_END();
        if (DB) {
            if (t + 1 < T) lstore(buf ^ 1);
            __syncthreads();
        } else {
            __syncthreads();
            if (t + 1 < T) { lstore(0); __syncthreads(); }
        }
    }

    // ---- epilogue: per pass q the 8 output channels {8q .. 8q+7} of every 32-row MFMA tile (accumulator registers 4q .. 4q+3)
    float* Ms = lds;                             // [16 xi][8*MT channels][32 tiles]
    constexpr int CH = 8 * MT;
    const int e_tile = tid & 31, e_ch = tid >> 5;                 // 8 channel slots x 32 tiles per 256 threads; MT passes over the slots
    const int egt = tb * WG_TILES + e_tile;
    const bool e_valid = egt < total_tiles;
    const int egc = e_valid ? egt : total_tiles - 1;
    const int eb = egc / (TH * TW), erem = egc - eb * TH * TW, ety = erem / TW, etx = erem - ety * TW;
    const int oy = 2 * ety, ox = 2 * etx;
    // every load of the epilogue (scale, shift, the residual's two rows for the thread's 4*MT channels) is issued ahead of the exchange
    // passes: inside them they were two serialised memory latencies per pass.  The width is even (entry-point check), so a tile's two
    // output columns always exist and are 8-byte aligned.
    const bool row1 = oy + 1 < H;
    const long long ch_stride = (long long)H * W;
    const long long o_base = (((long long)eb * Cout + co_blk + e_ch) * H + oy) * W + ox;
    float scv[4][MT], shv[4][MT];
    float2 q0[4][MT], q1[4][MT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int pp = 0; pp < MT; ++pp) {
            const int co = co_blk + pp * 32 + 8 * q + e_ch;
            scv[q][pp] = scale[co]; shv[q][pp] = shift[co];
            q0[q][pp] = q1[q][pp] = make_float2(0.0f, 0.0f);
        }
    if (residual && e_valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int pp = 0; pp < MT; ++pp) {
                const long long o = o_base + (pp * 32 + 8 * q) * ch_stride;
                q0[q][pp] = *reinterpret_cast<const float2*>(residual + o);
                q1[q][pp] = *reinterpret_cast<const float2*>(residual + o + (row1 ? W : 0));
            }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xi = wave * 4 + j;
#pragma unroll
            for (int h = 0; h < MT; ++h)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    // accumulator register 4q+rr of MFMA tile h: row (co) = 32h + 8q + rr + 4*half, column (tile) = l31
                    const int chs = h * 8 + rr + 4 * half;                      // channel slot of this pass: 8 per MFMA tile
                    Ms[(xi * CH + chs) * WG_TILES + l31] = acc[j][h][4 * q + rr];
                }
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < MT; ++pp) {
            const int chs = pp * 8 + e_ch;                                      // slot -> MFMA tile pp, row 8q + e_ch within it
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = Ms[(xi * CH + chs) * WG_TILES + e_tile];
            float s0[4], s1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { s0[c] = m[c] + m[4 + c] + m[8 + c]; s1[c] = m[4 + c] - m[8 + c] - m[12 + c]; }
            float o00 = s0[0] + s0[1] + s0[2], o01 = s0[1] - s0[2] - s0[3];
            float o10 = s1[0] + s1[1] + s1[2], o11 = s1[1] - s1[2] - s1[3];
            if (e_valid) {
                const float sc = scv[q][pp], sh = shv[q][pp];
                const long long o = o_base + (pp * 32 + 8 * q) * ch_stride;
                o00 = o00 * sc + sh + q0[q][pp].x; o01 = o01 * sc + sh + q0[q][pp].y;
                o10 = o10 * sc + sh + q1[q][pp].x; o11 = o11 * sc + sh + q1[q][pp].y;
                if (relu) { o00 = fmaxf(o00, 0.0f); o01 = fmaxf(o01, 0.0f); o10 = fmaxf(o10, 0.0f); o11 = fmaxf(o11, 0.0f); }
                *reinterpret_cast<float2*>(y + o) = make_float2(o00, o01);
                if (row1) *reinterpret_cast<float2*>(y + o + W) = make_float2(o10, o11);
            }
        }
        __syncthreads();
    }
}


// ----------------------------------------------------------------------------------------------------------------------------------
// Register-resident variant on v_mfma_f32_16x16x4_f32 (same rate as 32x32x2).  Its B operand -- lane l holds B[k = l>>4][n = l&15] --
// is EXACTLY what a lane that transforms tile (l & 15), input channel k0 + (l >> 4) has in its registers: the 16 values of B^T d B are
// the B fragments of 16 MFMAs (one per xi), with no LDS round trip and no panel.  A wave owns 16 tiles x COB = 32 output channels for
// ALL sixteen xi (2 x 16 accumulators of 4 VGPRs = 128), so the inverse transform A^T M A also happens in the lane's own registers
// (accumulator register r of lane l = channel 4*(l>>4) + r, tile l & 15): no exchange through LDS, no barrier in the epilogue.
// Only the filter panel U[xi][4 channels][32 co] (8 KB per K-step of 4 channels, shared by the NW waves = NW*16 tiles of the
// workgroup) goes through LDS, double buffered, one barrier per K-step; layout [xi][co half][k][16 co]: the 32 lanes of an LDS access
// group hit 32 different banks.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void wino_reg_kernel(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y,
                                                           int Cin, int H, int W, int Cout, int TH, int TW, int total_tiles, int n_tb, int n_cb, int relu) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    constexpr int NT = NW * 64, KC = 4, PANEL = 16 * KC * 32;         // floats per U panel
    constexpr int UP = PANEL / 4 / NT;                                 // 16-byte U loads per thread and K-step (2 at 256 threads, 4 at 128)
    __shared__ __attribute__((aligned(16))) float Us[2][PANEL];
    const int lin = blockIdx.x, xcd = lin & 7, seq = lin >> 3;
    const int cb = seq % n_cb, tb = (seq / n_cb) * 8 + xcd;
    if (tb >= n_tb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int co_blk = cb * 32;
    // this lane's tile and channel-of-the-K-step
    const int gt = tb * (NW * 16) + wave * 16 + l15;
    const bool tvalid = gt < total_tiles;
    const int gtc = tvalid ? gt : total_tiles - 1;
    const int b = gtc / (TH * TW), rem = gtc - b * TH * TW, ty = rem / TW, tx = rem - ty * TW;
    const int c0 = 2 * tx - 1, cs = min(max(c0, 0), W - 4);
    const bool left = c0 < cs, right = c0 > cs;
    bool rowok[4];
    const float* rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - 1 + r;
        rowok[r] = tvalid && iy >= 0 && iy < H;
        rp[r] = x + ((long long)b * Cin + kq) * H * W + (long long)min(max(iy, 0), H - 1) * W + cs;
    }
    const long long x_step = (long long)KC * H * W, u_step = (long long)KC * Cout;
    const float* up[UP];
    int udst[UP];
#pragma unroll
    for (int p = 0; p < UP; ++p) {
        const int f = tid + NT * p, co4 = f & 7, row = f >> 3;           // row = xi*4 + k ; 8 x 16-byte words per 32-co row
        up[p] = U + ((long long)(row >> 2) * Cin + (row & 3)) * Cout + co_blk + co4 * 4;
        udst[p] = (row >> 2) * 128 + (co4 >> 2) * 64 + (row & 3) * 16 + (co4 & 3) * 4;      // [xi][half][k][16]
    }
    F4u drow[4];
    f32x4v ureg[UP];
    float bv[16];
    auto gload = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) drow[r] = *reinterpret_cast<const F4u*>(rp[r] + t * x_step);
#pragma unroll
        for (int p = 0; p < UP; ++p) ureg[p] = *reinterpret_cast<const f32x4v*>(up[p] + t * u_step);
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {           // U panel -> LDS, input tile -> B fragments
#pragma unroll
        for (int p = 0; p < UP; ++p) *reinterpret_cast<f32x4v*>(&Us[buf][udst[p]]) = ureg[p];
        float v[16], t[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float l0 = rowok[r] ? drow[r].x : 0.0f, l1 = rowok[r] ? drow[r].y : 0.0f, l2 = rowok[r] ? drow[r].z : 0.0f,
                        l3 = rowok[r] ? drow[r].w : 0.0f;
            v[r * 4 + 0] = left ? 0.0f : (right ? l1 : l0);
            v[r * 4 + 1] = left ? l0 : (right ? l2 : l1);
            v[r * 4 + 2] = left ? l1 : (right ? l3 : l2);
            v[r * 4 + 3] = left ? l2 : (right ? 0.0f : l3);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0 * 4 + c] = v[0 * 4 + c] - v[2 * 4 + c];
            t[1 * 4 + c] = v[1 * 4 + c] + v[2 * 4 + c];
            t[2 * 4 + c] = v[2 * 4 + c] - v[1 * 4 + c];
            t[3 * 4 + c] = v[1 * 4 + c] - v[3 * 4 + c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bv[r * 4 + 0] = t[r * 4 + 0] - t[r * 4 + 2];
            bv[r * 4 + 1] = t[r * 4 + 1] + t[r * 4 + 2];
            bv[r * 4 + 2] = t[r * 4 + 2] - t[r * 4 + 1];
            bv[r * 4 + 3] = t[r * 4 + 1] - t[r * 4 + 3];
        }
    };
    f32x4v acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[xi][h] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};

    const int T = Cin / KC;
    gload(0);
    stage(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) gload(t + 1);
        const float* Ab = &Us[buf][kq * 16 + l15];
        DI2P_MFMA_BEGIN();
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const float a0 = Ab[xi * 128], a1 = Ab[xi * 128 + 64];
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[xi], acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[xi], acc[xi][1], 0, 0, 0);
        }
        DI2P_MFMA_END();
        if (t + 1 < T) stage(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue, all in registers: lane = tile l15, channels co_blk + 16h + 4*kq + r.  Every load of the epilogue (scale, shift, the
    // residual's two rows for the lane's 8 channels) is issued BEFORE the first inverse transform: as loads inside the per-channel loop
    // they were 16 serialised memory latencies per wave (~1/3 of a wave's lifetime at 64 input channels).
    const int oy = 2 * ty, ox = 2 * tx;
    const bool row1 = oy + 1 < H;
    const long long o_base = (((long long)b * Cout + co_blk + 4 * kq) * H + oy) * W + ox, ch_stride = (long long)H * W;
    float scv[8], shv[8];
    float2 q0[8], q1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int co = co_blk + (i >> 2) * 16 + 4 * kq + (i & 3);
        scv[i] = scale[co]; shv[i] = shift[co];
    }
    if (residual) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long o = o_base + ((i >> 2) * 16 + (i & 3)) * ch_stride;
            q0[i] = *reinterpret_cast<const float2*>(residual + o);
            q1[i] = *reinterpret_cast<const float2*>(residual + o + (row1 ? W : 0));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) q0[i] = q1[i] = make_float2(0.0f, 0.0f);
    }
    if (!tvalid) return;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = h * 4 + r;
            float m[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi] = acc[xi][h][r];
            float s0[4], s1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { s0[c] = m[c] + m[4 + c] + m[8 + c]; s1[c] = m[4 + c] - m[8 + c] - m[12 + c]; }
            float o00 = s0[0] + s0[1] + s0[2], o01 = s0[1] - s0[2] - s0[3];
            float o10 = s1[0] + s1[1] + s1[2], o11 = s1[1] - s1[2] - s1[3];
            const float sc = scv[i], sh = shv[i];
            const long long o = o_base + (h * 16 + r) * ch_stride;
            o00 = o00 * sc + sh + q0[i].x; o01 = o01 * sc + sh + q0[i].y; o10 = o10 * sc + sh + q1[i].x; o11 = o11 * sc + sh + q1[i].y;
            if (relu) { o00 = fmaxf(o00, 0.0f); o01 = fmaxf(o01, 0.0f); o10 = fmaxf(o10, 0.0f); o11 = fmaxf(o11, 0.0f); }
            *reinterpret_cast<float2*>(y + o) = make_float2(o00, o01);
            if (row1) *reinterpret_cast<float2*>(y + o + W) = make_float2(o10, o11);
        }
}

}  // namespace

extern "C" int di2p_winograd_weight_transform(const float* weight, float* U, int Cin, int Cout, void* stream) {
    DI2P_CHECK_ARG(weight && U && Cin >= 1 && Cout >= 1, "bad args");
    hipLaunchKernelGGL(wino_weight_kernel<false>, dim3(di2p_cdiv((long long)Cin * Cout, 256)), dim3(256), 0, (hipStream_t)stream, weight, U, Cin, Cout);
    DI2P_RETURN_LAUNCH();
}

// U f32[16, Cin_g, Cout_g] of the INPUT-GRADIENT filter of a stride-1 3x3 layer, from its forward filter weight f32[Cout_f = Cin_g, Cin_f = Cout_g, 3, 3]
// (flipped taps, swapped channel roles -- one launch instead of flip + transpose + copy + transform).
extern "C" int di2p_winograd_weight_transform_dgrad(const float* weight, float* U, int Cin_g, int Cout_g, void* stream) {
    DI2P_CHECK_ARG(weight && U && Cin_g >= 1 && Cout_g >= 1, "bad args");
    hipLaunchKernelGGL(wino_weight_kernel<true>, dim3(di2p_cdiv((long long)Cin_g * Cout_g, 256)), dim3(256), 0, (hipStream_t)stream, weight, U, Cin_g, Cout_g);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_conv3x3_winograd(const float* x, const float* U, const float* scale, const float* shift, const float* residual, float* y, int B,
                                     int Cin, int H, int W, int Cout, int relu, void* stream) {
    DI2P_CHECK_ARG(x && U && scale && shift && y, "null pointer");
    DI2P_CHECK_ARG(B >= 0 && Cin >= 8 && Cin % 8 == 0 && Cout >= 32 && Cout % 32 == 0 && H >= 1 && W >= 4 && W % 2 == 0,
                   "needs Cin % 8 == 0, Cout % 32 == 0 and an even width >= 4");
    DI2P_CHECK_ARG(((uintptr_t)U & 15) == 0 && ((uintptr_t)y & 7) == 0, "U must be 16-byte and y 8-byte aligned");
    // both kernels read the residual's two pixels of a row as one 8-byte load
    DI2P_CHECK_ARG(((uintptr_t)residual & 7) == 0, "residual must be 8-byte aligned");
    DI2P_CHECK_ARG((long long)Cin * H * W < (1ll << 31), "per-image extent must fit 31 bits");
    if (B == 0) return 0;
    const int TH = (H + 1) / 2, TW = (W + 1) / 2;
    const long long total = (long long)B * TH * TW;
    DI2P_CHECK_ARG(total < (1ll << 31), "too many tiles");
    const int n_tb = di2p_cdiv(total, WG_TILES);
    // co-block 64 when that still leaves >= 3 workgroups per CU, else 32 (more, smaller workgroups for the small late stages)
    const long long opt = di2p_opt(DI2P_OPT_WINO_COB);
    const bool cob64 = opt ? (opt == 64 && Cout % 64 == 0) : (Cout % 64 == 0 && (long long)n_tb * (Cout / 64) >= 768);
    hipStream_t st = (hipStream_t)stream;
    const bool db = di2p_opt(DI2P_OPT_WINO_DB) != 0;
    const int map_opt = (int)di2p_opt(DI2P_OPT_WINO_MAP);      // 0: automatic, 1: tile blocks over the XCDs, 2: co-blocks over the XCDs
#define DI2P_WINO_LAUNCH(COBV, DBV, KCV)                                                                                                     \
    do {                                                                                                                                     \
        const int n_cb = Cout / COBV;                                                                                                        \
        /* the co-block mapping needs n_cb % 8 == 0 (the kernel divides by n_cb / 8): the knob cannot force it elsewhere */                 \
        const int by_co = n_cb % 8 == 0 && (map_opt ? map_opt == 2 : (long long)16 * Cin * Cout * 4 > (2ll << 20));                           \
        const int grid = by_co ? n_cb * n_tb : di2p_cdiv(n_tb, 8) * 8 * n_cb;                                                                \
        const size_t lds = WinoLds<COBV, DBV, KCV>::TOTAL * sizeof(float);                                                                        \
        (void)hipFuncSetAttribute((const void*)wino_conv_kernel<COBV, DBV, KCV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        hipLaunchKernelGGL((wino_conv_kernel<COBV, DBV, KCV>), dim3(grid), dim3(256), lds, st, x, U, scale, shift, residual, y, Cin, H, W, Cout, \
                           TH, TW, (int)total, n_tb, n_cb, relu, by_co);                                                                     \
    } while (0)
    const long long reg_opt = di2p_opt(DI2P_OPT_WINO_REG);      // 0: automatic, 1: LDS-panel kernel, 2: register-resident, 4 waves, 3: 2 waves
    // automatic: the register-resident kernel wherever its 64-tile workgroups number at least 256 (ResNet stages 1-3).  Alone, the LDS-panel
    // kernel is level or ahead from stage 2 on (stage 3: 85 vs 93 us), but in the 8-stream pipeline the register-resident kernel's
    // 16 KB of LDS per workgroup (against 32-64 KB) lets it share a CU with the pose solver's workgroups: +2.8 % frames/s with the
    // threshold at 300 instead of 1024 (tools/sweep_wino_reg_min.sh); at the 512-channel stage (192 workgroups) the LDS-panel kernel stays
    const bool reg_auto = reg_opt == 0 && (long long)di2p_cdiv(total, 64) * (Cout / 32) >= di2p_opt(DI2P_OPT_WINO_REG_MIN);
    if ((reg_opt >= 2 || reg_auto) && Cin % 4 == 0) {
        const int nw = reg_opt == 3 ? 2 : 4;
        const int n_tb_r = di2p_cdiv(total, nw * 16), n_cb = Cout / 32;
        const int grid = di2p_cdiv(n_tb_r, 8) * 8 * n_cb;
        if (nw == 4) hipLaunchKernelGGL(wino_reg_kernel<4>, dim3(grid), dim3(256), 0, st, x, U, scale, shift, residual, y, Cin, H, W, Cout, TH, TW, (int)total, n_tb_r, n_cb, relu);
        else hipLaunchKernelGGL(wino_reg_kernel<2>, dim3(grid), dim3(128), 0, st, x, U, scale, shift, residual, y, Cin, H, W, Cout, TH, TW, (int)total, n_tb_r, n_cb, relu);
        DI2P_RETURN_LAUNCH();
    }
    const long long kc_opt = di2p_opt(DI2P_OPT_WINO_KC);
    const bool kc4 = kc_opt ? kc_opt == 4 : Cin <= 256;      // measured in the pipeline: K-step 4 wins up to 256 input channels, 8 at 512
    if (cob64) {
        if (!db) DI2P_WINO_LAUNCH(64, false, 8); else if (kc4) DI2P_WINO_LAUNCH(64, true, 4); else DI2P_WINO_LAUNCH(64, true, 8);
    } else {
        if (!db) DI2P_WINO_LAUNCH(32, false, 8); else if (kc4) DI2P_WINO_LAUNCH(32, true, 4); else DI2P_WINO_LAUNCH(32, true, 8);
    }
#undef DI2P_WINO_LAUNCH
    DI2P_RETURN_LAUNCH();
}
