// fp32 MFMA tile engine, "k-contiguous panels" (gfx950).  Successor of mfma_tile.h for every shape whose
// operands can be staged in 4x4 blocks (K-step 32, N % 4 == 0).
//
// C[M x N] = A[M x K] * B[K x N].
//   * LDS panels are [row][k] with k contiguous and rows padded to 36 floats:  As[2][BM][36], Bs[2][BN][36].
//     A lane of v_mfma_f32_32x32x2_f32 needs ONE k per instruction (lane l: row l&31, k-slot l>>5).  The k index of
//     a dot product may be visited in any order as long as A and B agree, so MFMA step s (0..15) of a K-step lets
//     lane half h use k = 16*h + s: each lane then reads its 16 operands of a 32-deep K-step as FOUR ds_read_b128
//     (16 consecutive floats of one panel row) instead of sixteen ds_read_b32 -- 4x fewer LDS instructions, and the
//     b128 reads reach the LDS rate from one wave per SIMD (MI355X_MICROARCH.md, LDS table).
//     Bank check (row stride 36 dwords): a ds_read_b128 lane group of 16 rows touches banks 36*i mod 64 (+0..3),
//     all distinct; the two lane halves are separate groups.
//   * A (weights) is packed once as [M][K]: a thread stages 4 consecutive k of one row with one 16-byte load and
//     one ds_write_b128 (8 lanes = one 128-byte panel row).
//   * B (activations / on-the-fly im2col, [K][N] with n contiguous in memory) is staged in 4(k) x 4(n) register
//     blocks: four 16-byte loads along n (rows k..k+3), transposed in registers (free: component selection), four
//     ds_write_b128 into panel rows n..n+3.  Lanes 0..7 of a group hold k-quads 0..7 of the same n-quad, so a
//     ds_write_b128 group writes 32 consecutive dwords (conflict-free) and a global load instruction touches
//     8 rows x 128 contiguous bytes.
//   * register prefetch of K-step t+1 is issued before the MFMAs of step t; one barrier per K-step.
//   * accumulators and the C/D layout are those of mfma_tile.h: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include "mfma_tile.h"

template <int WM_, int WN_, int TM_, int TN_>
struct KcCfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 32, LDK = 36;
    static constexpr int THREADS = WM * WN * 64;
    static constexpr int A_RPP = THREADS / 8, A_PASSES = BM / A_RPP;
    static constexpr int B_QPP = THREADS / 8, B_BLOCKS = (BN / 4) / B_QPP;   // n-quads per pass / 4x4 blocks per thread
    static constexpr int LDS_FLOATS = 2 * (BM + BN) * LDK;
    static_assert(BM % A_RPP == 0 && A_PASSES >= 1, "A panel / threads mismatch");
    static_assert((BN / 4) % B_QPP == 0 && B_BLOCKS >= 1, "B panel / threads mismatch");
};

// Loaders issue UNCONDITIONAL loads from clamped addresses (a load inside an `if` makes hipcc wait vmcnt(0) right
// behind it, which serialises the K-step's global round trips); anything that must read as zero is zeroed by
// fix() at LDS-store time, i.e. after the MFMAs of the previous K-step.
// LoaderA: float4 load4(int m, int k)          4 consecutive k (k % 4 == 0) of row m.  The packed weights are
//                                              zero-padded to K % 32 == 0; rows m >= M may return anything
//                                              finite-or-not (their outputs are discarded by the epilogue)
// LoaderB: void  setup(int blk, int n)         n % 4 == 0: column quad of 4x4 block `blk` of this thread
//          void  begin_tile(int k0)            once per K-step, in order, before the loads of that step
//          float4 load_row(int blk, int k)     4 consecutive n of row k.  Rows k >= K meet zero weights, so they
//                                              only have to be FINITE-or-column-local (clamp k: same columns);
//                                              columns n >= N are discarded by the epilogue
//          void  fix(int blk, float4 (&r)[4])  zero what must be zero (e.g. convolution padding) for the tile
//                                              loaded by the LAST begin_tile
// Epi    : void  tile(int mrow0, int jcol, const f32x16& acc)
template <class Cfg, class LoaderA, class LoaderB, class Epi>
__device__ __forceinline__ void mfma_gemm_block_kc(float* lds, LoaderA& la, LoaderB& lb, Epi& epi, int K, int m_blk, int n_blk) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK, LDK = Cfg::LDK;
    float* As = lds;                    // [2][BM][LDK]
    float* Bs = lds + 2 * BM * LDK;     // [2][BN][LDK]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int kq = tid & 7, rq = tid >> 3;

#pragma unroll
    for (int g = 0; g < Cfg::B_BLOCKS; ++g) lb.setup(g, n_blk + 4 * (rq + g * Cfg::B_QPP));

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[Cfg::A_PASSES], rb[Cfg::B_BLOCKS][4];
    const int T = (K + BK - 1) / BK;
    auto gload = [&](int t) {
        const int k0 = t * BK + 4 * kq;
        lb.begin_tile(t * BK);
#pragma unroll
        for (int p = 0; p < Cfg::A_PASSES; ++p) ra[p] = la.load4(m_blk + rq + p * Cfg::A_RPP, k0);
#pragma unroll
        for (int g = 0; g < Cfg::B_BLOCKS; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) rb[g][r] = lb.load_row(g, k0 + r);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < Cfg::A_PASSES; ++p)
            *reinterpret_cast<float4*>(&As[(buf * BM + rq + p * Cfg::A_RPP) * LDK + 4 * kq]) = ra[p];
#pragma unroll
        for (int g = 0; g < Cfg::B_BLOCKS; ++g) {
            lb.fix(g, rb[g]);
            float* d = &Bs[(buf * BN + 4 * (rq + g * Cfg::B_QPP)) * LDK + 4 * kq];
            *reinterpret_cast<float4*>(d)           = make_float4(rb[g][0].x, rb[g][1].x, rb[g][2].x, rb[g][3].x);
            *reinterpret_cast<float4*>(d + LDK)     = make_float4(rb[g][0].y, rb[g][1].y, rb[g][2].y, rb[g][3].y);
            *reinterpret_cast<float4*>(d + 2 * LDK) = make_float4(rb[g][0].z, rb[g][1].z, rb[g][2].z, rb[g][3].z);
            *reinterpret_cast<float4*>(d + 3 * LDK) = make_float4(rb[g][0].w, rb[g][1].w, rb[g][2].w, rb[g][3].w);
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        // Branch-free loop body (the last step re-loads its own tile, never read).  The scheduling fences pin the
        // global loads AHEAD of the MFMA phase and their first use (transposing moves + ds_write) BEHIND it: left
        // alone, hipcc either parks the moves right behind the loads (with control flow around gload) or sinks the
        // loads to the end of the MFMAs (without), and both expose the full L2/HBM latency every K-step.
        gload(t + 1 < T ? t + 1 : t);
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = As + (buf * BM + wm * Cfg::TM * 32 + l31) * LDK + 16 * half;
        const float* Bb = Bs + (buf * BN + wn * Cfg::TN * 32 + l31) * LDK + 16 * half;
        float4 a[2][Cfg::TM], b[2][Cfg::TN];
        auto fread = [&](int c, int s) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[s][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + 4 * c);
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[s][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + 4 * c);
        };
        fread(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int s = c & 1;
            if (c + 1 < 4) fread(c + 1, s ^ 1);     // next chunk's fragments in flight under this chunk's MFMAs
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i].x, b[s][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i].y, b[s][j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i].z, b[s][j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i].w, b[s][j].w, acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        lstore(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) {
            const int jcol = n_blk + (wn * Cfg::TN + j) * 32 + l31;
            const int mrow0 = m_blk + (wm * Cfg::TM + i) * 32 + 4 * half;
            epi.tile(mrow0, jcol, acc[i][j]);
        }
}
