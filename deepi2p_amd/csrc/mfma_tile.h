// fp32 MFMA tile engine shared by the pointwise GEMM and the implicit-GEMM convolution (gfx950).
//
// C[M x N] = A^T[K x M] * B[K x N], both operands "k-major" so that lanes map to consecutive m / n:
//   * A (weights, packed once to [K][M]) and B (activations, or the on-the-fly im2col view) are
//     staged through LDS as [BK][BM] / [BK][BN] panels, double buffered, one barrier per K-step;
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles, A/B operand = one VGPR: lane l holds
//     A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]) reads its operands with conflict-free ds_read_b32
//     (both 32-lane halves read 32 consecutive floats of one panel row);
//   * the next K-step's global loads are issued before the current step's MFMAs (register
//     prefetch), so HBM/L2 latency hides under the 64-cycle matrix instructions;
//   * three entry points: mfma_gemm_block (scalar stager, any shape), mfma_gemm_block_vec (16-byte stager, branch-free
//     K-loop: the production path) and mfma_gemm_block_blds (B panel already in LDS: fused layer chains);
//   * accumulators: TM x TN tiles of 16 VGPRs per wave; C/D layout col = lane&31,
//     row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include <type_traits>
#include "common.h"

// Priority of a wave while it issues an MFMA cluster.  Two waves of a SIMD that contend for the matrix pipe at equal priority drift into
// lockstep (both in the MFMA phase, then both in the load / transform phase with the pipe idle); the wave that enters its cluster first
// keeps the pipe until it is through, the other one runs its VALU / memory phase underneath.  DI2P_MFMA_PRIO=0 builds without it.
#ifndef DI2P_MFMA_PRIO
#define DI2P_MFMA_PRIO 1
#endif
#if DI2P_MFMA_PRIO
#define DI2P_MFMA_BEGIN() __builtin_amdgcn_s_setprio(1)
#define DI2P_MFMA_END() __builtin_amdgcn_s_setprio(0)
#else
#define DI2P_MFMA_BEGIN() ((void)0)
#define DI2P_MFMA_END() ((void)0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM_, int WN_, int TM_, int TN_, int BK_ = 16>
struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = BK_;
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int A_RPP = THREADS / BM, A_PASSES = BK / A_RPP;  // rows per pass / passes
    static constexpr int B_RPP = THREADS / BN, B_PASSES = BK / B_RPP;
    static constexpr int LDS_FLOATS = 2 * BK * (BM + BN);
    static_assert(THREADS % BM == 0 && THREADS % BN == 0, "tile/threads mismatch");
    static_assert(BK % A_RPP == 0 && BK % B_RPP == 0, "BK/passes mismatch");
};

// LoaderA: float load(int k, int m)             (k < K, m < M checked by the loader)
// LoaderB: void  column(int j) ; void begin_tile(int k0) (called once per K-step, in order) ; float load(int k)
// Epi    : void  store(int m, int j, float acc)  -- called for every valid (m, j)
template <class Cfg, class LoaderA, class LoaderB, class Epi>
__device__ __forceinline__ void mfma_gemm_block(float* lds, LoaderA& la, LoaderB& lb, Epi& epi, int K, int m_blk, int j_blk) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    float* As = lds;                 // [2][BK][BM]
    float* Bs = lds + 2 * BK * BM;   // [2][BK][BN]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int a_col = tid % BM, a_row0 = tid / BM;
    const int b_col = tid % BN, b_row0 = tid / BN;
    lb.column(j_blk + b_col);

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float ra[Cfg::A_PASSES], rb[Cfg::B_PASSES];
    const int T = (K + BK - 1) / BK;

    auto gload = [&](int t) {
        const int k0 = t * BK;
        lb.begin_tile(k0);
#pragma unroll
        for (int p = 0; p < Cfg::A_PASSES; ++p) ra[p] = la.load(k0 + a_row0 + p * Cfg::A_RPP, m_blk + a_col);
#pragma unroll
        for (int p = 0; p < Cfg::B_PASSES; ++p) rb[p] = lb.load(k0 + b_row0 + p * Cfg::B_RPP);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < Cfg::A_PASSES; ++p) As[(buf * BK + a_row0 + p * Cfg::A_RPP) * BM + a_col] = ra[p];
#pragma unroll
        for (int p = 0; p < Cfg::B_PASSES; ++p) Bs[(buf * BK + b_row0 + p * Cfg::B_RPP) * BN + b_col] = rb[p];
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) gload(t + 1);
        const float* Ab = As + buf * BK * BM + wm * Cfg::TM * 32 + l31;
        const float* Bb = Bs + buf * BK * BN + wn * Cfg::TN * 32 + l31;
        // all operand reads of the K-step first, then the MFMAs back to back: the LDS latency is paid once per
        // K-step instead of once per k-pair (the compiler otherwise waits lgkmcnt(0) before every MFMA pair)
        float a[BK / 2][Cfg::TM], b[BK / 2][Cfg::TN];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[kk / 2][i] = Ab[(kk + half) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[kk / 2][j] = Bb[(kk + half) * BN + j * 32];
        }
        DI2P_MFMA_BEGIN();
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        DI2P_MFMA_END();
        if (t + 1 < T) lstore(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int jcol = j_blk + (wn * Cfg::TN + j) * 32 + l31;
            const int mrow0 = m_blk + (wm * Cfg::TM + i) * 32 + 4 * half;
            epi.tile(mrow0, jcol, acc[i][j]);
        }
}


// ----------------------------------------------------------------------------------------------------------------
// Vectorised variant: operands are staged with 16-byte global loads and ds_write_b128 (4 consecutive m / n per lane),
// 4x fewer VMEM and LDS-write instructions per MFMA than the scalar stager above.  Loaders provide
//   LoaderA: float4 load4(int k, int m)      (m % 4 == 0); void fix(float4&, int k) zeroes rows k >= K at store time
//   LoaderB: void column4(int j) (j % 4 == 0); void begin_tile(int k0); float4 load4(int k);
//            void fix(float4&, int pass)  -- applied at LDS-store time to the values of the tile loaded after the last
//            begin_tile: loaders keep their loads UNCONDITIONAL (clamped addresses, no control flow in the K-loop)
//            and zero here what must read as zero (convolution padding)
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };   // 16-byte load that only needs dword alignment

// Weights [K][M] with M % 4 == 0: 4 consecutive output channels of one k row, loaded unconditionally from the
// clamped (k, m) and zeroed for k >= K at LDS-store time (rows m >= M are dropped by the epilogue).
struct LoaderWt4 {
    const float* Wt;
    int K, M;
    __device__ __forceinline__ float4 load4(int k, int m) const {
        return *reinterpret_cast<const float4*>(Wt + min(k, K - 1) * M + min(m, M - 4));
    }
    __device__ __forceinline__ void fix(float4& v, int k) const {
        if (k >= K) v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

// What a B loader keeps in registers per staged row between the request and the LDS store: a float4 that fix() patches in place (the
// default), or its own `Raw` type (more loads per row than the four values that go to LDS: the stride-2 window loader of conv.hip) that
// finish() turns into the float4.
template <class L, class = void>
struct StagedOf { using type = float4; };
template <class L>
struct StagedOf<L, std::void_t<typename L::Raw>> { using type = typename L::Raw; };
template <class L, class... Extra>
__device__ __forceinline__ float4 staged_finish(L& lb, float4& v, int pass, const Extra&... extra) {
    lb.fix(v, pass, extra...);
    return v;
}
template <class L, class R, class... Extra>
__device__ __forceinline__ float4 staged_finish(L& lb, R& raw, int pass, const Extra&... extra) {
    return lb.finish(raw, pass, extra...);
}

template <class Cfg, class LoaderA, class LoaderB, class Epi>
__device__ __forceinline__ void mfma_gemm_block_vec(float* lds, LoaderA& la, LoaderB& lb, Epi& epi, int K, int m_blk, int j_blk,
                                                    int t_begin = 0, int t_end = -1) {   // K-steps [t_begin, t_end) (split-K); default: all
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;                       // threads per panel row
    constexpr int A_RPP = Cfg::THREADS / A_TPR, B_RPP = Cfg::THREADS / B_TPR;
    // a panel with fewer float4s than threads (narrow BM at K-step 16) is staged by the first waves only
    constexpr int A_PASSES = BK >= A_RPP ? BK / A_RPP : 1, B_PASSES = BK >= B_RPP ? BK / B_RPP : 1;
    static_assert((BK % A_RPP == 0 || A_RPP % BK == 0) && (BK % B_RPP == 0 || B_RPP % BK == 0), "tile/threads mismatch");
    static_assert(A_RPP * A_TPR == Cfg::THREADS && B_RPP * B_TPR == Cfg::THREADS, "panel rows must divide the workgroup");
    float* As = lds;
    float* Bs = lds + 2 * BK * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_col = (tid % A_TPR) * 4, a_row0 = tid / A_TPR;
    const int b_col = (tid % B_TPR) * 4, b_row0 = tid / B_TPR;
    const bool a_on = A_RPP <= BK || a_row0 < BK, b_on = B_RPP <= BK || b_row0 < BK;      // wave-uniform
    lb.column4(j_blk + b_col);

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[A_PASSES];
    typename StagedOf<LoaderB>::type rb[B_PASSES];
    const int T = (t_end < 0 ? (K + BK - 1) / BK : t_end) - t_begin;      // number of K-steps of this block (>= 1)
    int k_loaded = 0;
    auto gload = [&](int t) {
        const int k0 = (t_begin + t) * BK;
        k_loaded = k0;
        lb.begin_tile(k0);
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) ra[p] = la.load4(k0 + a_row0 + p * A_RPP, m_blk + a_col);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) rb[p] = lb.load4(k0 + b_row0 + p * B_RPP);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            la.fix(ra[p], k_loaded + a_row0 + p * A_RPP);
            if (a_on) *reinterpret_cast<float4*>(&As[(buf * BK + a_row0 + p * A_RPP) * BM + a_col]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const float4 v = staged_finish(lb, rb[p], p);
            if (b_on) *reinterpret_cast<float4*>(&Bs[(buf * BK + b_row0 + p * B_RPP) * BN + b_col]) = v;
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BK * BM + wm * Cfg::TM * 32 + l31;
        const float* Bb = Bs + buf * BK * BN + wn * Cfg::TN * 32 + l31;
        // operand fragments double-buffered in registers: the ds_reads of k-pair kk+1 are in flight under the MFMAs of kk
        float a[2][Cfg::TM], b[2][Cfg::TN];
        auto fread = [&](int kk, int s) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[s][i] = Ab[(kk + half) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[s][j] = Bb[(kk + half) * BN + j * 32];
        };
        DI2P_MFMA_BEGIN();
        fread(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int s = (kk >> 1) & 1;
            if (kk + 2 < BK) fread(kk + 2, s ^ 1);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
        DI2P_MFMA_END();
    };
    // Steady state: branch-free body.  The scheduling fences pin the global loads of step t+1 AHEAD of the MFMA phase of
    // step t and their first use (fix + ds_write) BEHIND it, so the L2/HBM latency hides under the MFMAs; without them
    // hipcc either waits for the loads right away or sinks them below the MFMAs.  The last K-step is peeled (no loads).
    for (int t = 0; t + 1 < T; ++t) {
        const int buf = t & 1;
        gload(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        lstore(buf ^ 1);
        __syncthreads();
    }
    compute((T - 1) & 1);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int jcol = j_blk + (wn * Cfg::TN + j) * 32 + l31;
            const int mrow0 = m_blk + (wm * Cfg::TM + i) * 32 + 4 * half;
            epi.tile(mrow0, jcol, acc[i][j]);
        }
}


// ----------------------------------------------------------------------------------------------------------------
// Depth-2 variant of mfma_gemm_block_vec: the global loads of K-step t+2 are issued before the MFMAs of step t, i.e. two
// K-steps (2 x 16..64 MFMAs per wave) cover the L2/HBM latency instead of one.  Measured reason: a single workgroup per CU
// runs the depth-1 loop at ~30 % MFMA utilisation (45 TF of 147 on the ResNet shapes) -- the latency of the staged loads
// under load (~2 k cycles) is twice the MFMA time of a 64x64 K-step, and only co-resident workgroups hide the rest.
// Same K order, same MFMA sequence: results are bit-identical to the depth-1 engine.  Loaders additionally provide
//   typename Info; Info info() const   -- what fix() needs later about the tile loaded after the last begin_tile()
//   void fix(float4&, int pass, const Info&) const
template <class Cfg, class LoaderA, class LoaderB, class Epi>
__device__ __forceinline__ void mfma_gemm_block_vec2(float* lds, LoaderA& la, LoaderB& lb, Epi& epi, int K, int m_blk, int j_blk,
                                                     int t_begin = 0, int t_end = -1) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;
    constexpr int A_RPP = Cfg::THREADS / A_TPR, B_RPP = Cfg::THREADS / B_TPR;
    constexpr int A_PASSES = BK >= A_RPP ? BK / A_RPP : 1, B_PASSES = BK >= B_RPP ? BK / B_RPP : 1;
    static_assert((BK % A_RPP == 0 || A_RPP % BK == 0) && (BK % B_RPP == 0 || B_RPP % BK == 0), "tile/threads mismatch");
    float* As = lds;
    float* Bs = lds + 2 * BK * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_col = (tid % A_TPR) * 4, a_row0 = tid / A_TPR;
    const int b_col = (tid % B_TPR) * 4, b_row0 = tid / B_TPR;
    const bool a_on = A_RPP <= BK || a_row0 < BK, b_on = B_RPP <= BK || b_row0 < BK;
    lb.column4(j_blk + b_col);

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    struct Stage { float4 ra[A_PASSES]; typename StagedOf<LoaderB>::type rb[B_PASSES]; int k0; typename LoaderB::Info info; };
    Stage s0, s1;
    const int T = (t_end < 0 ? (K + BK - 1) / BK : t_end) - t_begin;
    auto gload = [&](int t, Stage& s) {
        s.k0 = (t_begin + t) * BK;
        lb.begin_tile(s.k0);
        s.info = lb.info();
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) s.ra[p] = la.load4(s.k0 + a_row0 + p * A_RPP, m_blk + a_col);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) s.rb[p] = lb.load4(s.k0 + b_row0 + p * B_RPP);
    };
    auto lstore = [&](int buf, Stage& s) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            la.fix(s.ra[p], s.k0 + a_row0 + p * A_RPP);
            if (a_on) *reinterpret_cast<float4*>(&As[(buf * BK + a_row0 + p * A_RPP) * BM + a_col]) = s.ra[p];
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const float4 v = staged_finish(lb, s.rb[p], p, s.info);
            if (b_on) *reinterpret_cast<float4*>(&Bs[(buf * BK + b_row0 + p * B_RPP) * BN + b_col]) = v;
        }
    };
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BK * BM + wm * Cfg::TM * 32 + l31;
        const float* Bb = Bs + buf * BK * BN + wn * Cfg::TN * 32 + l31;
        float a[2][Cfg::TM], b[2][Cfg::TN];
        auto fread = [&](int kk, int s) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[s][i] = Ab[(kk + half) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[s][j] = Bb[(kk + half) * BN + j * 32];
        };
        DI2P_MFMA_BEGIN();
        fread(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int s = (kk >> 1) & 1;
            if (kk + 2 < BK) fread(kk + 2, s ^ 1);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
        DI2P_MFMA_END();
    };
    // step t computes from LDS buffer t & 1; stage s0 carries the even steps' data, s1 the odd ones (static register sets).
    // (Measured and dropped: letting the LDS store of step t+1 interleave with the MFMAs of step t through a
    //  sched_group_barrier pipeline -- 2 % slower; any instruction between two MFMAs of one accumulator chain costs more
    //  than it hides.  Long uninterrupted MFMA chains (all operand reads first) are slower too: 84-88 vs 99 TF.)
    gload(0, s0);
    if (T > 1) gload(1, s1);
    lstore(0, s0);
    __syncthreads();
    int t = 0;
    for (; t + 2 < T; t += 2) {
        gload(t + 2, s0);                       // s0 (step t) is already in LDS
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        lstore(1, s1);                          // step t + 1, loaded two steps ago
        __syncthreads();
        if (t + 3 < T) gload(t + 3, s1);
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        lstore(0, s0);                          // step t + 2
        __syncthreads();
    }
    // tail: step t is in buffer 0 (t even); step t + 1, if any, waits in s1
    compute(0);
    if (t + 1 < T) {
        lstore(1, s1);
        __syncthreads();
        compute(1);
    }
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int jcol = j_blk + (wn * Cfg::TN + j) * 32 + l31;
            const int mrow0 = m_blk + (wm * Cfg::TM + i) * 32 + 4 * half;
            epi.tile(mrow0, jcol, acc[i][j]);
        }
}


// ----------------------------------------------------------------------------------------------------------------
// Variant for fused layer chains: the B operand is ALREADY in LDS as a [K][ldb] panel (the previous layer's output tile),
// only A (weights [K][M], M % 4 == 0) is staged.  Same K order, same MFMA sequence as mfma_gemm_block_vec, so a layer
// computed here is bit-identical to the same layer computed from global memory.  The accumulators are handed to
// epi.tile() after a workgroup barrier (the epilogue may overwrite the B panel).
template <class Cfg, class LoaderA, class Epi>
__device__ __forceinline__ void mfma_gemm_block_blds(float* lds, LoaderA& la, const float* Bpanel, int ldb, Epi& epi, int K, int m_blk,
                                                     int j_blk) {
    constexpr int BM = Cfg::BM, BK = Cfg::BK;
    constexpr int A_TPR = BM / 4;
    constexpr int A_RPP = Cfg::THREADS / A_TPR;
    constexpr int A_PASSES = BK >= A_RPP ? BK / A_RPP : 1;
    static_assert(BK % A_RPP == 0 || A_RPP % BK == 0, "tile/threads mismatch");
    float* As = lds;       // [2][BK][BM]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int a_col = (tid % A_TPR) * 4, a_row0 = tid / A_TPR;
    const bool a_on = A_RPP <= BK || a_row0 < BK;

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[A_PASSES];
    const int T = (K + BK - 1) / BK;
    int k_loaded = 0;
    auto gload = [&](int t) {
        k_loaded = t * BK;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) ra[p] = la.load4(k_loaded + a_row0 + p * A_RPP, m_blk + a_col);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) {
            la.fix(ra[p], k_loaded + a_row0 + p * A_RPP);
            if (a_on) *reinterpret_cast<float4*>(&As[(buf * BK + a_row0 + p * A_RPP) * BM + a_col]) = ra[p];
        }
    };
    auto compute = [&](int t) {
        const float* Ab = As + (t & 1) * BK * BM + wm * Cfg::TM * 32 + l31;
        const float* Bb = Bpanel + (long long)t * BK * ldb + wn * Cfg::TN * 32 + l31;
        float a[2][Cfg::TM], b[2][Cfg::TN];
        auto fread = [&](int kk, int s) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[s][i] = Ab[(kk + half) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[s][j] = (t * BK + kk + half < K) ? Bb[(kk + half) * ldb + j * 32] : 0.0f;
        };
        DI2P_MFMA_BEGIN();
        fread(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int s = (kk >> 1) & 1;
            if (kk + 2 < BK) fread(kk + 2, s ^ 1);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
        DI2P_MFMA_END();
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t + 1 < T; ++t) {
        gload(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(t);
        __builtin_amdgcn_sched_barrier(0);
        lstore((t & 1) ^ 1);
        __syncthreads();
    }
    compute(T - 1);
    __syncthreads();          // every wave is done with the B panel: the epilogue may overwrite it
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int jcol = j_blk + (wn * Cfg::TN + j) * 32 + l31;
            const int mrow0 = m_blk + (wm * Cfg::TM + i) * 32 + 4 * half;
            epi.tile(mrow0, jcol, acc[i][j]);
        }
}
