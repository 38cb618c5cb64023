// Pointwise contraction kernels (EquivariantLayer / MyConv2d 1x1 + BN(eval) + ReLU) on fp32 MFMA.
//
// Replaces the cuDNN/ATen conv1d/conv2d(1x1)+batch_norm+relu chains of models/layers_pc.py:259-342,
// :110-190 and the torch.cat / expand / gather tensors that feed them (models/networks_pc.py:98,
// models/layers_pc.py:808-813, models/networks_united.py:139-197): concatenation, gather-by-index
// and group-broadcast are done by the B-operand loader, BN/bias/ReLU/max-over-K/interpolated-add by
// the epilogue, so none of those intermediates ever exists in HBM.
#include "mfma_tile.h"

namespace {

struct LoaderWt {  // packed weight [K][M]
    const float* Wt;
    int K, M;
    __device__ __forceinline__ float load(int k, int m) const { return (k < K && m < M) ? Wt[k * M + m] : 0.0f; }   // K*M < 2^31 (host-checked)
};

struct SrcDev {
    const float* ptr[DI2P_MAX_SRC];
    const int* gidx[DI2P_MAX_SRC];
    long long batch_stride[DI2P_MAX_SRC];
    int row_stride[DI2P_MAX_SRC];
    int c_end[DI2P_MAX_SRC];  // exclusive prefix end of each source's channel range
    int mode[DI2P_MAX_SRC];
    int group[DI2P_MAX_SRC];
    int n_src;
};

struct LoaderConcat {
    SrcDev s;
    int b, N, K;
    bool valid;
    const float* base[DI2P_MAX_SRC];  // per-column base pointer (already offset by batch and column)
    __device__ __forceinline__ void column(int n) {
        valid = n < N;
#pragma unroll
        for (int i = 0; i < DI2P_MAX_SRC; ++i) {
            base[i] = nullptr;
            if (i < s.n_src && valid) {
                int off = n;
                if (s.mode[i] == DI2P_SRC_GATHER) off = s.gidx[i][(long long)b * N + n];
                else if (s.mode[i] == DI2P_SRC_GROUP) off = n / s.group[i];
                base[i] = s.ptr[i] + (long long)b * s.batch_stride[i] + off;
            }
        }
    }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const {
        k = __builtin_amdgcn_readfirstlane(k);  // a wave stages one panel row: k is wave-uniform
        if (!valid || k >= K) return 0.0f;
        // within-frame offsets fit 32 bits (checked on the host): no 64-bit multiplies in the hot loader
        if (k < s.c_end[0]) return base[0][k * s.row_stride[0]];
        if (DI2P_MAX_SRC > 1 && k < s.c_end[1]) return base[1][(k - s.c_end[0]) * s.row_stride[1]];
        return base[2][(k - s.c_end[1]) * s.row_stride[2]];
    }
};

struct EpiDev {
    const float* scale;
    const float* shift;
    const float* batch_bias;
    const float* g_table[2];
    const int* g_idx[2];
    const float* g_w[2];
    int g_nodes[2];
    int g_k;
    int relu;
    int group_max;
};

struct EpiPointwise {
    EpiDev e;
    float* Y;
    int b, M, N;
    __device__ __forceinline__ void tile(int mrow0, int n, const f32x16& acc) {
        const bool col_ok = n < N;
        // gathered-add operands of this column (per_point_pn layer 0)
        int gi[2][4];
        float gw[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gi[t][j] = 0;
                gw[t][j] = 0.0f;
                if (e.g_table[t] && col_ok && j < e.g_k) {
                    gi[t][j] = e.g_idx[t][((long long)b * N + n) * e.g_k + j];
                    gw[t][j] = e.g_w[t][((long long)b * N + n) * e.g_k + j];
                }
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            const bool ok = col_ok && m < M;
            float v = acc[r];
            if (ok) {
                if (e.batch_bias) v += e.batch_bias[(long long)b * M + m];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (e.g_table[t]) {
                        const float* g = e.g_table[t] + ((long long)b * M + m) * e.g_nodes[t];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < e.g_k) v += gw[t][j] * g[gi[t][j]];
                    }
                const float sc = e.scale ? e.scale[m] : 1.0f;
                const float sh = e.shift ? e.shift[m] : 0.0f;
                v = v * sc + sh;
                if (e.relu) v = fmaxf(v, 0.0f);
            }
            if (e.group_max > 1) {
                float mx = ok ? v : -__builtin_inff();
                for (int o = 1; o < e.group_max; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                if (ok && (n % e.group_max) == 0) Y[((long long)b * M + m) * (N / e.group_max) + n / e.group_max] = mx;
            } else if (ok) {
                Y[((long long)b * M + m) * N + n] = v;
            }
        }
    }
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void pointwise_gemm_kernel(SrcDev srcs, const float* __restrict__ Wt, float* __restrict__ Y,
                                                                       int M, int K, int N, EpiDev epi) {
    extern __shared__ float lds[];
    LoaderWt la{Wt, K, M};
    LoaderConcat lb;
    lb.s = srcs;
    lb.b = blockIdx.z;
    lb.N = N;
    lb.K = K;
    EpiPointwise ep{epi, Y, (int)blockIdx.z, M, N};
    mfma_gemm_block<Cfg>(lds, la, lb, ep, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

// ---- attention pooling: out[b,c,m] = (1/HW) sum_hw feat[b,c,hw] * score[b,hw,m]
struct LoaderFeat {  // A[k=hw][m=c] = feat[b][c][hw]
    const float* feat;
    int HW, C;
    __device__ __forceinline__ float load(int k, int m) const { return (k < HW && m < C) ? feat[(long long)m * HW + k] : 0.0f; }
};
struct LoaderScore {
    const float* score;  // [HW][Mn] of this batch
    int HW, Mn, n;
    bool valid;
    __device__ __forceinline__ void column(int j) { n = j; valid = j < Mn; }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const { return (valid && k < HW) ? score[(long long)k * Mn + n] : 0.0f; }
};
struct EpiMean {
    float* out;  // [C][Mn] of this batch
    int C, Mn;
    float inv_div;
    __device__ __forceinline__ void tile(int mrow0, int n, const f32x16& acc) {
        if (n >= Mn) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < C) out[(long long)m * Mn + n] = acc[r] / inv_div;
        }
    }
};
template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void attention_pool_kernel(const float* __restrict__ feat, const float* __restrict__ score,
                                                                       float* __restrict__ out, int C, int HW, int Mn) {
    extern __shared__ float lds[];
    const int b = blockIdx.z;
    LoaderFeat la{feat + (long long)b * C * HW, HW, C};
    LoaderScore lb{score + (long long)b * HW * Mn, HW, Mn, 0, false};
    EpiMean ep{out + (long long)b * C * Mn, C, Mn, (float)HW};
    mfma_gemm_block<Cfg>(lds, la, lb, ep, HW, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

__global__ __launch_bounds__(256) void batch_gemv_kernel(const float* __restrict__ Wt, int M, int k0, const float* __restrict__ v,
                                                         int Kv, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* vb = v + (long long)b * Kv;
    float acc = 0.0f;
    for (int k = 0; k < Kv; ++k) acc += Wt[(long long)(k0 + k) * M + m] * vb[k];
    out[(long long)b * M + m] = acc;
}

template <class Cfg>
void launch_pw(const SrcDev& s, const float* Wt, float* Y, int B, int M, int K, int N, const EpiDev& e, hipStream_t st) {
    const dim3 grid(di2p_cdiv(N, Cfg::BN), di2p_cdiv(M, Cfg::BM), B);
    hipLaunchKernelGGL(pointwise_gemm_kernel<Cfg>, grid, dim3(Cfg::THREADS), Cfg::LDS_FLOATS * sizeof(float), st, s, Wt, Y, M, K, N, e);
}

}  // namespace

using Cfg128x128 = TileCfg<2, 2, 2, 2>;
using Cfg64x128 = TileCfg<2, 2, 1, 2>;
using Cfg32x128 = TileCfg<1, 4, 1, 1>;
using Cfg64x64 = TileCfg<2, 2, 1, 1>;

extern "C" int di2p_pointwise_gemm(const di2p_src_t* srcs, int n_src, const float* Wt, float* Y, int B, int M, int K, int N,
                                   const di2p_epilogue_t* epi, void* stream) {
    DI2P_CHECK_ARG(srcs && n_src >= 1 && n_src <= DI2P_MAX_SRC, "1..3 sources");
    DI2P_CHECK_ARG(B >= 0 && M >= 1 && K >= 1 && N >= 0, "bad size");
    DI2P_CHECK_ARG((long long)K * M < (1ll << 31), "weight too large");
    if (B == 0 || N == 0) return 0;
    SrcDev s{};
    int ctot = 0;
    for (int i = 0; i < DI2P_MAX_SRC; ++i) {
        if (i < n_src) {
            DI2P_CHECK_ARG(srcs[i].ptr && srcs[i].channels > 0, "bad source");
            DI2P_CHECK_ARG(srcs[i].mode != DI2P_SRC_GATHER || srcs[i].gidx, "gather source without index");
            DI2P_CHECK_ARG(srcs[i].mode != DI2P_SRC_GROUP || srcs[i].group >= 1, "group source without group");
            DI2P_CHECK_ARG((long long)srcs[i].channels * srcs[i].row_stride < (1ll << 31), "per-frame source extent must fit 31 bits");
            s.ptr[i] = srcs[i].ptr; s.gidx[i] = srcs[i].gidx; s.batch_stride[i] = srcs[i].batch_stride;
            s.row_stride[i] = srcs[i].row_stride; s.mode[i] = srcs[i].mode; s.group[i] = srcs[i].group > 0 ? srcs[i].group : 1;
            ctot += srcs[i].channels;
        }
        s.c_end[i] = ctot;
    }
    s.n_src = n_src;
    DI2P_CHECK_ARG(ctot == K, "source channels do not sum to K");
    EpiDev e{};
    e.group_max = 1;
    if (epi) {
        e.scale = epi->scale; e.shift = epi->shift; e.batch_bias = epi->batch_bias; e.relu = epi->relu;
        e.group_max = epi->group_max > 1 ? epi->group_max : 1;
        for (int t = 0; t < 2; ++t) { e.g_table[t] = epi->g_table[t]; e.g_idx[t] = epi->g_idx[t]; e.g_w[t] = epi->g_w[t]; e.g_nodes[t] = epi->g_nodes[t]; }
        e.g_k = epi->g_k;
        DI2P_CHECK_ARG(e.g_k >= 0 && e.g_k <= 4, "g_k must be <= 4");
    }
    if (e.group_max > 1) {
        const int g = e.group_max;
        DI2P_CHECK_ARG((g & (g - 1)) == 0 && g <= 32 && N % g == 0, "group_max must be a power of two <= 32 dividing N");
    }
    hipStream_t st = (hipStream_t)stream;
    if (M <= 32) launch_pw<Cfg32x128>(s, Wt, Y, B, M, K, N, e, st);
    else if (M <= 64 || (long long)B * di2p_cdiv(N, 128) * di2p_cdiv(M, 128) < 256) launch_pw<Cfg64x128>(s, Wt, Y, B, M, K, N, e, st);
    else launch_pw<Cfg128x128>(s, Wt, Y, B, M, K, N, e, st);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_batch_gemv(const float* Wt, int M, int k0, const float* v, int Kv, float* out, int B, void* stream) {
    DI2P_CHECK_ARG(Wt && v && out && M >= 1 && Kv >= 1 && k0 >= 0 && B >= 0, "bad args");
    if (B == 0) return 0;
    hipLaunchKernelGGL(batch_gemv_kernel, dim3(di2p_cdiv(M, 256), B), dim3(256), 0, (hipStream_t)stream, Wt, M, k0, v, Kv, out);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_attention_pool(const float* feat, const float* score, float* out, int B, int C, int HW, int Mn, void* stream) {
    DI2P_CHECK_ARG(feat && score && out && B >= 0 && C >= 1 && HW >= 1 && Mn >= 1, "bad args");
    if (B == 0) return 0;
    using Cfg = Cfg64x64;
    hipLaunchKernelGGL(attention_pool_kernel<Cfg>, dim3(di2p_cdiv(Mn, Cfg::BN), di2p_cdiv(C, Cfg::BM), B), dim3(Cfg::THREADS),
                       Cfg::LDS_FLOATS * sizeof(float), (hipStream_t)stream, feat, score, out, C, HW, Mn);
    DI2P_RETURN_LAUNCH();
}
