// ResNet-34 image branch kernels (models/resnet.py:56-72,125-216) for gfx950.
//
// conv + BN(eval) + [residual] + [ReLU] as ONE implicit-GEMM kernel on fp32 MFMA:
//   M = Cout, N = B*OH*OW (output pixels of the whole batch, so /32 maps still fill the chip),
//   K = Cin*KH*KW in the weight's own (ci,kh,kw) order.  The B operand is the im2col view of the
//   NCHW input gathered on the fly (never materialised): for a fixed k the 32 lanes of an MFMA
//   operand read 32 consecutive output pixels = consecutive input addresses (stride 1) of one
//   input row, i.e. coalesced 128-B segments straight from the reference's own layout.
#include "mfma_tile.h"

#include <stdlib.h>

namespace {

struct LoaderWtC {
    const float* Wt;  // [K][Cout]
    int K, M;
    __device__ __forceinline__ float load(int k, int m) const { return (k < K && m < M) ? Wt[k * M + m] : 0.0f; }   // K*M < 2^31 (host-checked)
};

struct LoaderIm2col {
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot;
    const float* xb;
    int ih0, iw0;
    bool valid;
    __device__ __forceinline__ void column(int j) {
        valid = j < Ntot;
        const int jj = valid ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        xb = x + (long long)b * Cin * H * W;
        ih0 = oh * stride - pad;
        iw0 = ow * stride - pad;
    }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const {
        k = __builtin_amdgcn_readfirstlane(k);
        if (!valid || k >= K) return 0.0f;
        const int khw = KH * KW;
        const int ci = k / khw, rem = k - ci * khw;
        const int kh = rem / KW, kw = rem - kh * KW;
        const int ih = ih0 + kh, iw = iw0 + kw;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) return 0.0f;
        return xb[((long long)ci * H + ih) * W + iw];
    }
};

// Tap-major K order: k' = (kh*KW + kw)*Cin + ci with Cin % 16 == 0, so one 16-row K-step lies inside ONE filter
// tap: the tap decode, the bounds test and the base address are computed once per K-step per lane, and the
// staged rows are constant-stride (H*W) loads.  (The generic loader above pays two integer divisions per
// staged element, which made the kernel VALU-bound at ~40 TF.)
struct LoaderIm2colTap {
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot;
    const float* xb;
    const float* tile_ptr;   // xb + (ci0*H + ih)*W + iw of the current K-step (nullptr: out of bounds)
    int ih0, iw0, HW;
    int ci0, kh, kw;         // wave-uniform walk over the taps
    bool valid;
    __device__ __forceinline__ void column(int j) {
        valid = j < Ntot;
        const int jj = valid ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        HW = H * W;
        xb = x + (long long)b * Cin * HW;
        ih0 = oh * stride - pad;
        iw0 = ow * stride - pad;
        ci0 = -bk; kh = 0; kw = 0;
        tile_ptr = nullptr;
    }
    int bk;                  // K-step (16 or 32); Cin % bk == 0
    __device__ __forceinline__ void begin_tile(int) {
        ci0 += bk;
        if (ci0 >= Cin) { ci0 = 0; if (++kw == KW) { kw = 0; ++kh; } }
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = valid && kh < KH && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        tile_ptr = ok ? xb + (ci0 * H + ih) * W + iw : nullptr;
    }
    __device__ __forceinline__ float load(int k) const { return tile_ptr ? tile_ptr[(k & (bk - 1)) * HW] : 0.0f; }
};

// Vector stagers (16-byte loads).  Weights: 4 consecutive output channels of one k row (needs Cout % 4 == 0).
struct LoaderWtC4 {
    const float* Wt;
    int K, M;
    __device__ __forceinline__ float4 load4(int k, int m) const {
        if (k < K && m + 3 < M) return *reinterpret_cast<const float4*>(Wt + k * M + m);
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < K) { if (m < M) v.x = Wt[k * M + m]; if (m + 1 < M) v.y = Wt[k * M + m + 1]; if (m + 2 < M) v.z = Wt[k * M + m + 2]; }
        return v;
    }
};

// im2col, tap-major, 4 consecutive output pixels of one output row per lane (needs OW % 4 == 0): for stride 1 they are 4
// consecutive input floats (one dword-aligned 16-byte load; scalar only at the left/right image border), for
// stride 2 four strided scalars.
struct LoaderIm2colTap4 {
    const float* x;
    int Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot;
    const float* xb;
    const float* tile_ptr;
    int ih0, iw0, HW, iw_first;
    int ci0, kh, kw, bk;
    bool valid, full;
    __device__ __forceinline__ void column4(int j) {
        valid = j < Ntot;                      // Ntot % 4 == 0: a group is entirely valid or entirely out
        const int jj = valid ? j : 0;
        const int opix = OH * OW;
        const int b = jj / opix, pix = jj - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        HW = H * W;
        xb = x + (long long)b * Cin * HW;
        ih0 = oh * stride - pad;
        iw0 = ow * stride - pad;
        ci0 = -bk; kh = 0; kw = 0;
        tile_ptr = nullptr; full = false; iw_first = 0;
    }
    __device__ __forceinline__ void begin_tile(int) {
        ci0 += bk;
        if (ci0 >= Cin) { ci0 = 0; if (++kw == KW) { kw = 0; ++kh; } }
        const int ih = ih0 + kh;
        iw_first = iw0 + kw;
        const bool row_ok = valid && kh < KH && (unsigned)ih < (unsigned)H;
        tile_ptr = row_ok ? xb + (ci0 * H + ih) * W : nullptr;            // start of the input row of channel ci0
        full = row_ok && stride == 1 && iw_first >= 0 && iw_first + 3 < W;
    }
    __device__ __forceinline__ float4 load4(int k) const {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (!tile_ptr) return v;
        const float* r = tile_ptr + (k & (bk - 1)) * HW;
        if (full) { const F4u t = *reinterpret_cast<const F4u*>(r + iw_first); v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; return v; }
        const int i0 = iw_first, i1 = iw_first + stride, i2 = iw_first + 2 * stride, i3 = iw_first + 3 * stride;
        if ((unsigned)i0 < (unsigned)W) v.x = r[i0];
        if ((unsigned)i1 < (unsigned)W) v.y = r[i1];
        if ((unsigned)i2 < (unsigned)W) v.z = r[i2];
        if ((unsigned)i3 < (unsigned)W) v.w = r[i3];
        return v;
    }
};

struct EpiConv {
    const float* scale;
    const float* shift;
    const float* residual;
    float* y;
    int Cout, opix, Ntot, relu;
    __device__ __forceinline__ void tile(int mrow0, int j, const f32x16& acc) {
        if (j >= Ntot) return;
        const int b = j / opix, pix = j - b * opix;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < Cout) {
                const long long o = ((long long)b * Cout + m) * opix + pix;
                float v = acc[r] * scale[m] + shift[m];
                if (residual) v += residual[o];
                if (relu) v = fmaxf(v, 0.0f);
                y[o] = v;
            }
        }
    }
};

template <class Cfg, bool TAP>
__global__ __launch_bounds__(Cfg::THREADS) void conv2d_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                               int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride,
                                                               int pad, int Ntot, int relu) {
    extern __shared__ float lds[];
    const int K = Cin * KH * KW;
    LoaderWtC la{Wt, K, Cout};
    EpiConv ep{scale, shift, residual, y, Cout, OH * OW, Ntot, relu};
    if (TAP) {
        LoaderIm2colTap lb;
        lb.x = x; lb.Cin = Cin; lb.H = H; lb.W = W; lb.OH = OH; lb.OW = OW; lb.KH = KH; lb.KW = KW; lb.stride = stride;
        lb.pad = pad; lb.K = K; lb.Ntot = Ntot; lb.bk = Cfg::BK;
        mfma_gemm_block<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
    } else {
        LoaderIm2col lb{x, Cin, H, W, OH, OW, KH, KW, stride, pad, K, Ntot, nullptr, 0, 0, false};
        mfma_gemm_block<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
    }
}

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void conv2d_vec_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                                   int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride,
                                                                   int pad, int Ntot, int relu) {
    extern __shared__ float lds[];
    const int K = Cin * KH * KW;
    LoaderWtC4 la{Wt, K, Cout};
    EpiConv ep{scale, shift, residual, y, Cout, OH * OW, Ntot, relu};
    LoaderIm2colTap4 lb;
    lb.x = x; lb.Cin = Cin; lb.H = H; lb.W = W; lb.OH = OH; lb.OW = OW; lb.KH = KH; lb.KW = KW; lb.stride = stride;
    lb.pad = pad; lb.K = K; lb.Ntot = Ntot; lb.bk = Cfg::BK;
    mfma_gemm_block_vec<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

template <class Cfg>
void launch_conv_vec(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual, float* y,
                     int Cin, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad, int Ntot, int relu,
                     hipStream_t st) {
    const dim3 grid(di2p_cdiv(Ntot, Cfg::BN), di2p_cdiv(Cout, Cfg::BM)), block(Cfg::THREADS);
    hipLaunchKernelGGL(conv2d_vec_kernel<Cfg>, grid, block, Cfg::LDS_FLOATS * sizeof(float), st, x, Wt, scale, shift, residual, y,
                       Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, Ntot, relu);
}

// ----------------------------------------------------------------------------------------------------------------
// Barrier-free variant: every WAVE owns a (TM*32) x (TN*32) output tile and loads its MFMA operands straight from
// global memory in fragment layout -- v_mfma_f32_32x32x2_f32 takes ONE dword per lane per operand (lane l: A[m0+(l&31)]
// of row k+(l>>5), B[n0+(l&31)] of row k+(l>>5)), i.e. each operand load is two perfectly coalesced 128-byte rows, and
// the instruction runs 64 cycles, so there is ample time to stream them through L1/L2 with a register prefetch.
// No LDS staging and no workgroup barrier: the LDS-staged kernels above lose ~45 % of the matrix pipe to barrier
// convoys (4 waves of a workgroup sit on 4 SIMDs, each queued behind other workgroups' waves, once per K-step).
// The 4 waves of a workgroup cover a 2x2 (or 1x4) arrangement of neighbouring tiles so they share operand rows in L1.
template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void conv2d_direct_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ residual, float* __restrict__ y, int Cin,
                                                            int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride,
                                                            int pad, int Ntot, int relu) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = (blockIdx.y * WM + wm) * TM * 32;
    const int n0 = (blockIdx.x * WN + wn) * TN * 32;
    if (m0 >= Cout || n0 >= Ntot) return;          // whole wave out of range (no barriers in this kernel)
    const int HW = H * W, opix = OH * OW;
    // per-lane output columns
    const float* xb[TN];
    int ih0[TN], iw0[TN];
    bool colok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + l31;
        colok[j] = n < Ntot;
        const int nn = colok[j] ? n : 0;
        const int b = nn / opix, pix = nn - b * opix;
        const int oh = pix / OW, ow = pix - oh * OW;
        xb[j] = x + (long long)b * Cin * HW + half * HW;     // this lane's k-row parity is folded into the base
        ih0[j] = oh * stride - pad;
        iw0[j] = ow * stride - pad;
    }
    bool rowok[TM];
    const float* wa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 32 + l31;
        rowok[i] = m < Cout;
        wa[i] = Wt + half * Cout + (rowok[i] ? m : 0);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Software pipeline over the flattened (tap, channel-batch) sequence: the operands of batch s+1 are requested
    // before the MFMAs of batch s are issued, so their L2 latency hides under 4*TM*TN matrix instructions.
    constexpr int U = 4;                       // k-pairs per batch (Cin % 8 == 0 is checked on the host)
    const int cstep = 2 * U;
    const int batches_per_tap = Cin / cstep;
    const int nsteps = KH * KW * batches_per_tap;
    float a0[U][TM], b0[U][TN], a1[U][TM], b1[U][TN];
    int s_kh = 0, s_kw = 0, s_cb = 0;          // loader position (wave-uniform)
    const float* bp[TN];
    auto set_tap = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ih = ih0[j] + s_kh, iw = iw0[j] + s_kw;
            const bool ok = colok[j] && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            bp[j] = ok ? xb[j] + ih * W + iw : nullptr;
        }
    };
    auto fetch = [&](float (&a)[U][TM], float (&b)[U][TN]) {
        const int ci = s_cb * cstep;
        const int wrow = (s_kh * KW + s_kw) * Cin + ci;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[u][i] = rowok[i] ? wa[i][(wrow + 2 * u) * Cout] : 0.0f;
#pragma unroll
            for (int j = 0; j < TN; ++j) b[u][j] = bp[j] ? bp[j][(ci + 2 * u) * HW] : 0.0f;
        }
        if (++s_cb == batches_per_tap) { s_cb = 0; if (++s_kw == KW) { s_kw = 0; ++s_kh; } set_tap(); }
    };
    auto compute = [&](float (&a)[U][TM], float (&b)[U][TN]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    };
    set_tap();
    fetch(a0, b0);
    int st = 0;
    for (; st + 2 <= nsteps; st += 2) {
        if (st + 1 < nsteps) fetch(a1, b1);
        compute(a0, b0);
        if (st + 2 < nsteps) fetch(a0, b0);
        compute(a1, b1);
    }
    if (st < nsteps) compute(a0, b0);
    EpiConv ep{scale, shift, residual, y, Cout, opix, Ntot, relu};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) ep.tile(m0 + i * 32 + 4 * half, n0 + j * 32 + l31, acc[i][j]);
}

template <int TM, int TN, int WM, int WN>
void launch_conv_direct(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual, float* y,
                        int Cin, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad, int Ntot, int relu,
                        hipStream_t st) {
    const dim3 grid(di2p_cdiv(Ntot, WN * TN * 32), di2p_cdiv(Cout, WM * TM * 32));
    hipLaunchKernelGGL((conv2d_direct_kernel<TM, TN, WM, WN>), grid, dim3(256), 0, st, x, Wt, scale, shift, residual, y, Cin, H, W,
                       Cout, OH, OW, KH, KW, stride, pad, Ntot, relu);
}

template <class Cfg>
void launch_conv(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual, float* y,
                 int Cin, int H, int W, int Cout, int OH, int OW, int KH, int KW, int stride, int pad, int Ntot, int relu,
                 int tap_major, hipStream_t st) {
    const dim3 grid(di2p_cdiv(Ntot, Cfg::BN), di2p_cdiv(Cout, Cfg::BM)), block(Cfg::THREADS);
    const size_t lds = Cfg::LDS_FLOATS * sizeof(float);
    if (tap_major)
        hipLaunchKernelGGL((conv2d_kernel<Cfg, true>), grid, block, lds, st, x, Wt, scale, shift, residual, y, Cin, H, W, Cout,
                           OH, OW, KH, KW, stride, pad, Ntot, relu);
    else
        hipLaunchKernelGGL((conv2d_kernel<Cfg, false>), grid, block, lds, st, x, Wt, scale, shift, residual, y, Cin, H, W, Cout,
                           OH, OW, KH, KW, stride, pad, Ntot, relu);
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int OH,
                                                           int OW, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ow = (int)(i % OW);
    const long long t = i / OW;
    const int oh = (int)(t % OH);
    const long long bc = t / OH;
    const float* p = x + bc * H * W;
    float m = -__builtin_inff();
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        const int ih = oh * 2 - 1 + dh;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int iw = ow * 2 - 1 + dw;
            if ((unsigned)iw < (unsigned)W) m = fmaxf(m, p[(long long)ih * W + iw]);
        }
    }
    y[i] = m;
}

// one wavefront per (b,c): mean over HW
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int HW) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* r = x + row * HW;
    float s = 0.0f;
    for (int n = lane; n < HW; n += 64) s += r[n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[row] = s / (float)HW;
}

}  // namespace

using CfgC128x128 = TileCfg<2, 2, 2, 2>;
using CfgC64x128 = TileCfg<2, 2, 1, 2>;
using CfgC64x64 = TileCfg<2, 2, 1, 1>;
using CfgC128x64 = TileCfg<2, 2, 2, 1>;
using CfgC128x128k32 = TileCfg<2, 2, 2, 2, 32>;
using CfgC64x128k32 = TileCfg<2, 2, 1, 2, 32>;
using CfgC64x64k32 = TileCfg<2, 2, 1, 1, 32>;
using CfgC128x64k32 = TileCfg<2, 2, 2, 1, 32>;

extern "C" int di2p_conv2d(const float* x, const float* Wt, const float* scale, const float* shift, const float* residual,
                           float* y, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu,
                           int tap_major, void* stream) {
    DI2P_CHECK_ARG(x && Wt && scale && shift && y, "null pointer");
    DI2P_CHECK_ARG(B >= 0 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "bad size");
    if (B == 0) return 0;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    DI2P_CHECK_ARG(OH >= 1 && OW >= 1, "empty output");
    DI2P_CHECK_ARG(!tap_major || Cin % 16 == 0, "tap-major weights need Cin % 16 == 0");
    DI2P_CHECK_ARG((long long)Cin * H * W < (1ll << 31), "per-image extent must fit 31 bits");
    const long long Ntot_ll = (long long)B * OH * OW;
    DI2P_CHECK_ARG(Ntot_ll < (1ll << 31), "too many output pixels");
    const int Ntot = (int)Ntot_ll;
    hipStream_t st = (hipStream_t)stream;
#define DI2P_CONV(CFG) launch_conv<CFG>(x, Wt, scale, shift, residual, y, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, Ntot, relu, tap_major, st)
    // tile choice: K-step 32 when a tap holds whole 32-channel groups (halves barriers, doubles the prefetch
    // distance); the largest tile that still yields >= ~2 workgroups per CU.  DI2P_CONV_CFG overrides (experiments).
    static int force = -2;
    if (force == -2) { const char* e = getenv("DI2P_CONV_CFG"); force = e ? atoi(e) : -1; }
    const bool k32 = tap_major && Cin % 32 == 0;
    const long long nb64x128 = (long long)di2p_cdiv(Ntot, 128) * di2p_cdiv(Cout, 64);
    const long long nb128x128 = (long long)di2p_cdiv(Ntot, 128) * di2p_cdiv(Cout, 128);
    const long long nb128x64 = (long long)di2p_cdiv(Ntot, 64) * di2p_cdiv(Cout, 128);
    int choice;   // 0: 64x64  1: 64x128  2: 128x64  3: 128x128
    if (Cout >= 128 && nb128x128 >= 512) choice = 3;
    else if (nb64x128 >= 512) choice = 1;
    else if (Cout >= 128 && nb128x64 >= 512) choice = 2;
    else choice = 0;
    if (force >= 0) { choice = force % 10; if (choice >= 2 && Cout < 128) choice = 1; }
    const bool use32 = k32 && (force < 0 || force >= 10);
    // EXPERIMENT, off by default (measured slower): barrier-free direct-to-register kernels
    // (DI2P_CONV_DIRECT=1: 64x64 per wave, 2: 32x64, 3: 32x32, 4: auto)
    static int direct = -1;
    if (direct < 0) { const char* e = getenv("DI2P_CONV_DIRECT"); direct = e ? atoi(e) : 0; }
    if (direct && tap_major && Cin % 8 == 0) {
#define DI2P_CONVD(TM, TN, WM, WN) launch_conv_direct<TM, TN, WM, WN>(x, Wt, scale, shift, residual, y, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, Ntot, relu, st)
        int mode = direct;
        if (mode == 4) {   // enough waves to fill 1024 SIMDs a few times over, biggest tile first
            const long long w64 = (long long)di2p_cdiv(Ntot, 64) * di2p_cdiv(Cout, 64);
            const long long w32x64 = (long long)di2p_cdiv(Ntot, 64) * di2p_cdiv(Cout, 32);
            mode = w64 >= 3072 ? 1 : (w32x64 >= 3072 ? 2 : 3);
        }
        if (mode == 1) { if (Cout >= 128) DI2P_CONVD(2, 2, 2, 2); else DI2P_CONVD(2, 2, 1, 4); }
        else if (mode == 2) DI2P_CONVD(1, 2, 2, 2);
        else DI2P_CONVD(1, 1, 2, 2);
#undef DI2P_CONVD
        DI2P_RETURN_LAUNCH();
    }
    // vector stager: tap-major weights, 32-channel taps, whole 4-pixel groups per output row, 16-byte aligned weights
    static int novec = -1;
    if (novec < 0) { const char* e = getenv("DI2P_CONV_NOVEC"); novec = e ? atoi(e) : 0; }
    const bool vec = !novec && use32 && OW % 4 == 0 && Cout % 4 == 0 && ((uintptr_t)Wt & 15) == 0;
#define DI2P_CONVV(CFG) launch_conv_vec<CFG>(x, Wt, scale, shift, residual, y, Cin, H, W, Cout, OH, OW, KH, KW, stride, pad, Ntot, relu, st)
    if (vec) {
        switch (choice) { case 3: DI2P_CONVV(CfgC128x128k32); break; case 2: DI2P_CONVV(CfgC128x64k32); break;
                          case 1: DI2P_CONVV(CfgC64x128k32); break; default: DI2P_CONVV(CfgC64x64k32); }
    } else if (use32) {
        switch (choice) { case 3: DI2P_CONV(CfgC128x128k32); break; case 2: DI2P_CONV(CfgC128x64k32); break;
                          case 1: DI2P_CONV(CfgC64x128k32); break; default: DI2P_CONV(CfgC64x64k32); }
    } else {
        switch (choice) { case 3: DI2P_CONV(CfgC128x128); break; case 2: DI2P_CONV(CfgC128x64); break;
                          case 1: DI2P_CONV(CfgC64x128); break; default: DI2P_CONV(CfgC64x64); }
    }
#undef DI2P_CONV
#undef DI2P_CONVV
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_maxpool3x3s2(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    DI2P_CHECK_ARG(x && y && B >= 0 && C >= 1 && H >= 1 && W >= 1, "bad args");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * C * OH * OW;
    if (total == 0) return 0;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, OH, OW, total);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_global_avgpool(const float* x, float* y, int B, int C, int HW, void* stream) {
    DI2P_CHECK_ARG(x && y && B >= 0 && C >= 1 && HW >= 1, "bad args");
    const long long rows = (long long)B * C;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(global_avgpool_kernel, dim3(di2p_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, HW);
    DI2P_RETURN_LAUNCH();
}
