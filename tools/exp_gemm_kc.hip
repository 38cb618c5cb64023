// Standalone calibration of the fp32-MFMA tile engine (mfma_tile.h, production) and of the experimental k-contiguous
// variant (tools/mfma_tile_kc.h) on dense GEMM shapes with trivial loaders (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deepi2p_amd/csrc -I include -I tools tools/exp_gemm_kc.hip -o tools/bin/exp_gemm_kc
//   tools/bin/exp_gemm_kc M K N [M K N ...]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "mfma_tile_kc.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace {
struct EpiStore {
    float* C; int M, N;
    __device__ __forceinline__ void tile(int mrow0, int j, const f32x16& acc) const {
        if (j >= N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < M) C[(size_t)m * N + j] = acc[r];
        }
    }
};
// old engine loaders: At [K][M], X [K][N]
struct OldA { const float* At; int K, M; __device__ __forceinline__ float4 load4(int k, int m) const {
    return *reinterpret_cast<const float4*>(At + (size_t)min(k, K - 1) * M + min(m, M - 4)); }
    __device__ __forceinline__ void fix(float4& v, int k) const { if (k >= K) v = make_float4(0, 0, 0, 0); } };
struct OldB { const float* X; int K, N; int n; __device__ __forceinline__ void column4(int j) { n = min(j, N - 4); } __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float4 load4(int k) const { return *reinterpret_cast<const float4*>(X + (size_t)min(k, K - 1) * N + n); }
    __device__ __forceinline__ void fix(float4&, int) const {} };
// new engine loaders: A [M][K], X [K][N]
struct NewA { const float* A; int M, K; __device__ __forceinline__ float4 load4(int m, int k) const {   // K % 32 == 0 (zero padded)
    return *reinterpret_cast<const float4*>(A + (size_t)min(m, M - 1) * K + k); } };
template <int NB> struct NewB { const float* X; int K, N; int n[NB];
    __device__ __forceinline__ void setup(int g, int j) { n[g] = j; } __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float4 load_row(int g, int k) const { return *reinterpret_cast<const float4*>(X + (size_t)min(k, K - 1) * N + min(n[g], N - 4)); }
    __device__ __forceinline__ void fix(int, float4 (&)[4]) const {} };

template <class F> float time_ms(F f, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

// Ablation copy of mfma_gemm_block_vec: FLAGS bit0 = skip the global loads + LDS stores inside the loop (MFMA + LDS-read
// loop only, results wrong), bit1 = also skip the barrier, bit2 = skip the LDS operand reads (MFMA issue only).
template <class Cfg, int FLAGS, class LoaderA, class LoaderB, class Epi>
__device__ __forceinline__ void ablate_block(float* lds, LoaderA& la, LoaderB& lb, Epi& epi, int K, int m_blk, int j_blk) {
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;
    constexpr int A_RPP = Cfg::THREADS / A_TPR, B_RPP = Cfg::THREADS / B_TPR;
    constexpr int A_PASSES = BK / A_RPP, B_PASSES = BK / B_RPP;
    float* As = lds;
    float* Bs = lds + 2 * BK * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN, l31 = lane & 31, half = lane >> 5;
    const int a_col = (tid % A_TPR) * 4, a_row0 = tid / A_TPR, b_col = (tid % B_TPR) * 4, b_row0 = tid / B_TPR;
    lb.column4(j_blk + b_col);
    f32x16 acc[Cfg::TM][Cfg::TN];
    for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float4 ra[A_PASSES], rb[B_PASSES];
    const int T = (K + BK - 1) / BK;
    // FLAGS bit3: the A panel goes global -> LDS directly (global_load_lds_dwordx4: 1 KB per wave-instruction, LDS image =
    // wave-uniform base + lane*16, which IS the [k][m] panel when rows are unpadded); no VGPR staging, no ds_write for A.
    constexpr int ROWS_PER_GLDS = 256 / BM;                    // panel rows covered by one 1 KB instruction
    constexpr int GLDS_PER_WAVE = BK / ROWS_PER_GLDS / (Cfg::THREADS / 64);
    auto glds_A = [&](int t, int buf) {
        const int k0 = t * BK;
#pragma unroll
        for (int i = 0; i < GLDS_PER_WAVE; ++i) {
            const int r0 = (wave * GLDS_PER_WAVE + i) * ROWS_PER_GLDS;
            const int row = r0 + lane / (BM / 4), col = (lane % (BM / 4)) * 4;
            const float* src = la.At + (size_t)min(k0 + row, K - 1) * la.M + min(m_blk + col, la.M - 4);
            float* dst = &As[(buf * BK + r0) * BM];            // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto gload = [&](int t) {
        const int k0 = t * BK;
        if (!(FLAGS & 8)) {
#pragma unroll
            for (int p = 0; p < A_PASSES; ++p) ra[p] = la.load4(k0 + a_row0 + p * A_RPP, m_blk + a_col);
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) rb[p] = lb.load4(k0 + b_row0 + p * B_RPP);
    };
    auto lstore = [&](int buf) {
        if (!(FLAGS & 8)) {
#pragma unroll
            for (int p = 0; p < A_PASSES; ++p) *reinterpret_cast<float4*>(&As[(buf * BK + a_row0 + p * A_RPP) * BM + a_col]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) *reinterpret_cast<float4*>(&Bs[(buf * BK + b_row0 + p * B_RPP) * BN + b_col]) = rb[p];
    };
    float a0[Cfg::TM], b0[Cfg::TN];
    for (int i = 0; i < Cfg::TM; ++i) a0[i] = 1.0f + lane; 
    for (int j = 0; j < Cfg::TN; ++j) b0[j] = 2.0f + lane;
    // FLAGS bit4 (16): second accumulator set for the odd k-pairs (no two consecutive MFMAs on the same accumulator; summed at the end)
    // FLAGS bit5 (32): operand fragments read in batches of 4 k-pairs, MFMAs of a batch issued back to back
    f32x16 acc2[Cfg::TM][Cfg::TN];
    for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j) for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BK * BM + wm * Cfg::TM * 32 + l31;
        const float* Bb = Bs + buf * BK * BN + wn * Cfg::TN * 32 + l31;
        if (FLAGS & 192) {                              // 64: 8 k-pairs per batch, 128: the whole K-step; fragments single-buffered,
            constexpr int NB = (FLAGS & 128) ? BK / 2 : 8;     // all reads of a batch, one wait, then its MFMAs back to back
            float a[NB][Cfg::TM], b[NB][Cfg::TN];
#pragma unroll
            for (int kp = 0; kp < BK / 2; kp += NB) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
#pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) a[q][i] = Ab[(2 * (kp + q) + half) * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j) b[q][j] = Bb[(2 * (kp + q) + half) * BN + j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NB; ++q)
#pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                        for (int j = 0; j < Cfg::TN; ++j) {
                            if ((FLAGS & 16) && (q & 1)) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][j], acc2[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if (FLAGS & 32) {
            constexpr int NB = 4;                       // k-pairs per batch
            float a[2][NB][Cfg::TM], b[2][NB][Cfg::TN];
            auto fread = [&](int kp0, int s) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
#pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i) a[s][q][i] = Ab[(2 * (kp0 + q) + half) * BM + i * 32];
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j) b[s][q][j] = Bb[(2 * (kp0 + q) + half) * BN + j * 32];
                }
            };
            fread(0, 0);
#pragma unroll
            for (int kp = 0; kp < BK / 2; kp += NB) {
                const int s = (kp / NB) & 1;
                if (kp + NB < BK / 2) fread(kp + NB, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NB; ++q)
#pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                        for (int j = 0; j < Cfg::TN; ++j) {
                            if ((FLAGS & 16) && (q & 1)) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][q][i], b[s][q][j], acc2[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][q][i], b[s][q][j], acc[i][j], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        float a[2][Cfg::TM], b[2][Cfg::TN];
        auto fread = [&](int kk, int s) {
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) a[s][i] = (FLAGS & 4) ? a0[i] : Ab[(kk + half) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j) b[s][j] = (FLAGS & 4) ? b0[j] : Bb[(kk + half) * BN + j * 32];
        };
        fread(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int s = (kk >> 1) & 1;
            if (kk + 2 < BK) fread(kk + 2, s ^ 1);
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j) {
                    if ((FLAGS & 16) && (kk & 2)) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc2[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
                }
        }
    };
    if (FLAGS & 8) glds_A(0, 0);
    gload(0); lstore(0); __syncthreads();
    for (int t = 0; t + 1 < T; ++t) {
        const int buf = t & 1;
        if (!(FLAGS & 1)) { if (FLAGS & 8) glds_A(t + 1, buf ^ 1); gload(t + 1); }
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        if (!(FLAGS & 1)) lstore(buf ^ 1);
        if (!(FLAGS & 2)) __syncthreads();
    }
    compute((T - 1) & 1);
    if (FLAGS & 16) for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
    for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j)
        epi.tile(m_blk + (wm * Cfg::TM + i) * 32 + 4 * half, j_blk + (wn * Cfg::TN + j) * 32 + l31, acc[i][j]);
}
template <class Cfg, int FLAGS> __global__ __launch_bounds__(Cfg::THREADS) void ablate_kernel(const float* At, const float* X, float* C, int M, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    OldA la{At, K, M}; OldB lb{X, K, N, 0}; EpiStore epi{C, M, N};
    ablate_block<Cfg, FLAGS>(lds, la, lb, epi, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}
template <class Cfg, int FLAGS> void run_ablate(int M, int K, int N, const char* name) {
    float *dAt, *dX, *dC; CK(hipMalloc(&dAt, (size_t)K * M * 4)); CK(hipMalloc(&dX, (size_t)K * N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemset(dAt, 0, (size_t)K * M * 4)); CK(hipMemset(dX, 0, (size_t)K * N * 4));
    if (getenv("EXP_RANDOM")) {       // same kernels on uniform [-1,1) operands: the fp32-MFMA rate is DATA dependent (power / clocks)
        std::vector<float> h((size_t)K * (M > N ? M : N));
        unsigned s = 7; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
        CK(hipMemcpy(dAt, h.data(), (size_t)K * M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    }
    CK(hipFuncSetAttribute((const void*)ablate_kernel<Cfg, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_FLOATS * 4));
    dim3 g((N + Cfg::BN - 1) / Cfg::BN, (M + Cfg::BM - 1) / Cfg::BM);
    float ms = time_ms([&]() { ablate_kernel<Cfg, FLAGS><<<g, Cfg::THREADS, Cfg::LDS_FLOATS * 4>>>(dAt, dX, dC, M, K, N); }, 10);
    CK(hipGetLastError());
    std::vector<float> hc(4096);
    CK(hipMemcpy(hc.data(), dC + (size_t)(M / 2) * N, sizeof(float) * std::min<size_t>(4096, N), hipMemcpyDeviceToHost));
    double checksum = 0; for (float v : hc) checksum += v;
    printf("%-10s M=%d K=%d N=%d  ablation flags %d: %8.3f ms %6.1f TF  (row checksum %.6g)\n", name, M, K, N, FLAGS, ms, 2.0 * M * K * N / ms / 1e9, checksum);
    CK(hipFree(dAt)); CK(hipFree(dX)); CK(hipFree(dC));
}

template <class Cfg> __global__ __launch_bounds__(Cfg::THREADS) void old_kernel(const float* At, const float* X, float* C, int M, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    OldA la{At, K, M}; OldB lb{X, K, N, 0}; EpiStore epi{C, M, N};
    mfma_gemm_block_vec<Cfg>(lds, la, lb, epi, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}
template <class Cfg> __global__ __launch_bounds__(Cfg::THREADS) void new_kernel(const float* A, const float* X, float* C, int M, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    NewA la{A, M, K}; NewB<Cfg::B_BLOCKS> lb; lb.X = X; lb.K = K; lb.N = N; EpiStore epi{C, M, N};
    mfma_gemm_block_kc<Cfg>(lds, la, lb, epi, K, blockIdx.y * Cfg::BM, blockIdx.x * Cfg::BN);
}

double check(const std::vector<float>& A, const std::vector<float>& X, const float* dC, int M, int K, int N) {
    std::vector<float> C((size_t)M * N); CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; unsigned s = 12345;
    for (int t = 0; t < 400; ++t) {
        s = s * 1664525u + 1013904223u; const int m = (s >> 8) % M; s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
        double ref = 0, mag = 0; for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)m * K + k] * X[(size_t)k * N + n]; ref += p; mag += fabs(p); }
        worst = fmax(worst, fabs(ref - C[(size_t)m * N + n]) / (mag + 1e-30));
    }
    return worst;
}
template <class OC, class NC> void run(int M, int K, int N, const char* name) {
    std::vector<float> A((size_t)M * K), At((size_t)K * M), X((size_t)K * N);
    unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { A[(size_t)m * K + k] = rnd(); At[(size_t)k * M + m] = A[(size_t)m * K + k]; }
    for (auto& v : X) v = rnd();
    float *dA, *dAt, *dX, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dAt, At.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    const double flop = 2.0 * M * K * N;
    {
        CK(hipFuncSetAttribute((const void*)old_kernel<OC>, hipFuncAttributeMaxDynamicSharedMemorySize, OC::LDS_FLOATS * 4));
        dim3 g((N + OC::BN - 1) / OC::BN, (M + OC::BM - 1) / OC::BM);
        CK(hipMemset(dC, 0, (size_t)M * N * 4));
        float ms = time_ms([&]() { old_kernel<OC><<<g, OC::THREADS, OC::LDS_FLOATS * 4>>>(dAt, dX, dC, M, K, N); }, 10);
        CK(hipGetLastError());
        printf("%-10s M=%d K=%d N=%d  vec engine [k][m] %8.3f ms %6.1f TF  err %.2e\n", name, M, K, N, ms, flop / ms / 1e9, check(A, X, dC, M, K, N));
    }
    {
        CK(hipFuncSetAttribute((const void*)new_kernel<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, NC::LDS_FLOATS * 4));
        dim3 g((N + NC::BN - 1) / NC::BN, (M + NC::BM - 1) / NC::BM);
        CK(hipMemset(dC, 0, (size_t)M * N * 4));
        float ms = time_ms([&]() { new_kernel<NC><<<g, NC::THREADS, NC::LDS_FLOATS * 4>>>(dA, dX, dC, M, K, N); }, 10);
        CK(hipGetLastError());
        printf("%-10s M=%d K=%d N=%d  kc  engine [m][k] %8.3f ms %6.1f TF  err %.2e\n", name, M, K, N, ms, flop / ms / 1e9, check(A, X, dC, M, K, N));
    }
    CK(hipFree(dA)); CK(hipFree(dAt)); CK(hipFree(dX)); CK(hipFree(dC));
}
}  // namespace

// Persistent variant: gridDim.x workgroups walk the tiles (m fastest) with stride gridDim.x; the first K-step of the NEXT
// tile is requested before the epilogue of the current one, so the per-tile prologue latency hides under the stores.
template <class Cfg>
__global__ __launch_bounds__(Cfg::THREADS) void persist_kernel(const float* At, const float* X, float* C, int M, int K, int N) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = Cfg::BM, BN = Cfg::BN, BK = Cfg::BK;
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;
    constexpr int A_RPP = Cfg::THREADS / A_TPR, B_RPP = Cfg::THREADS / B_TPR;
    constexpr int A_PASSES = BK / A_RPP, B_PASSES = BK / B_RPP;
    float* As = lds;
    float* Bs = lds + 2 * BK * BM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN, l31 = lane & 31, half = lane >> 5;
    const int a_col = (tid % A_TPR) * 4, a_row0 = tid / A_TPR, b_col = (tid % B_TPR) * 4, b_row0 = tid / B_TPR;
    const int gm = (M + BM - 1) / BM, gn = (N + BN - 1) / BN, ntiles = gm * gn;
    const int T = (K + BK - 1) / BK;
    OldA la{At, K, M};
    EpiStore epi{C, M, N};
    float4 ra[A_PASSES], rb[B_PASSES];
    int m_blk = 0, j_blk = 0;
    auto gload = [&](int mb, int jb, int t) {
        const int k0 = t * BK;
        const int n = min(jb + b_col, N - 4);
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) ra[p] = la.load4(k0 + a_row0 + p * A_RPP, mb + a_col);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) rb[p] = *reinterpret_cast<const float4*>(X + (size_t)min(k0 + b_row0 + p * B_RPP, K - 1) * N + n);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p) *reinterpret_cast<float4*>(&As[(buf * BK + a_row0 + p * A_RPP) * BM + a_col]) = ra[p];
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) *reinterpret_cast<float4*>(&Bs[(buf * BK + b_row0 + p * B_RPP) * BN + b_col]) = rb[p];
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    m_blk = (tile % gm) * BM; j_blk = (tile / gm) * BN;
    gload(m_blk, j_blk, 0);
    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[Cfg::TM][Cfg::TN];
        for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
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
            fread(0, 0);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int s = (kk >> 1) & 1;
                if (kk + 2 < BK) fread(kk + 2, s ^ 1);
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
            }
        };
        lstore(0);
        __syncthreads();
        for (int t = 0; t + 1 < T; ++t) {
            gload(m_blk, j_blk, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(t & 1);
            __builtin_amdgcn_sched_barrier(0);
            lstore((t & 1) ^ 1);
            __syncthreads();
        }
        const int cm = m_blk, cj = j_blk;
        const int next = tile + gridDim.x;
        if (next < ntiles) { m_blk = (next % gm) * BM; j_blk = (next / gm) * BN; gload(m_blk, j_blk, 0); }   // in flight under the last MFMAs + epilogue
        __builtin_amdgcn_sched_barrier(0);
        compute((T - 1) & 1);
        for (int i = 0; i < Cfg::TM; ++i) for (int j = 0; j < Cfg::TN; ++j)
            epi.tile(cm + (wm * Cfg::TM + i) * 32 + 4 * half, cj + (wn * Cfg::TN + j) * 32 + l31, acc[i][j]);
        __syncthreads();      // everyone is done reading the panels before the next tile's first store
    }
}
template <class Cfg> void run_persist(int M, int K, int N, const char* name, int blocks_per_cu) {
    std::vector<float> A((size_t)M * K), At((size_t)K * M), X((size_t)K * N);
    unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) { A[(size_t)m * K + k] = rnd(); At[(size_t)k * M + m] = A[(size_t)m * K + k]; }
    for (auto& v : X) v = rnd();
    float *dAt, *dX, *dC; CK(hipMalloc(&dAt, At.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dAt, At.data(), At.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)persist_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_FLOATS * 4));
    CK(hipFuncSetAttribute((const void*)old_kernel<Cfg>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_FLOATS * 4));
    const int ntiles = ((M + Cfg::BM - 1) / Cfg::BM) * ((N + Cfg::BN - 1) / Cfg::BN);
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persist_kernel<Cfg>, Cfg::THREADS, Cfg::LDS_FLOATS * 4));
    if (blocks_per_cu > occ) { printf("(requested %d workgroups per CU, only %d fit: clamped)\n", blocks_per_cu, occ); blocks_per_cu = occ; }
    const int grid = std::min(ntiles, 256 * blocks_per_cu);
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    float ms = time_ms([&]() { persist_kernel<Cfg><<<grid, Cfg::THREADS, Cfg::LDS_FLOATS * 4>>>(dAt, dX, dC, M, K, N); }, 10);
    CK(hipGetLastError());
    const double err = check(A, X, dC, M, K, N);
    dim3 g((N + Cfg::BN - 1) / Cfg::BN, (M + Cfg::BM - 1) / Cfg::BM);
    float ms0 = time_ms([&]() { old_kernel<Cfg><<<g, Cfg::THREADS, Cfg::LDS_FLOATS * 4>>>(dAt, dX, dC, M, K, N); }, 10);
    printf("%-8s M=%d K=%d N=%d  tile-per-workgroup %7.3f ms %6.1f TF | persistent x%d %7.3f ms %6.1f TF  err %.2e\n", name, M, K, N, ms0,
           2.0 * M * K * N / ms0 / 1e9, blocks_per_cu, ms, 2.0 * M * K * N / ms / 1e9, err);
    CK(hipFree(dAt)); CK(hipFree(dX)); CK(hipFree(dC));
}

template <class Cfg> void ablations(int M, int K, int N, const char* name) {
    run_ablate<Cfg, 0>(M, K, N, name); run_ablate<Cfg, 8>(M, K, N, name); run_ablate<Cfg, 1>(M, K, N, name); run_ablate<Cfg, 3>(M, K, N, name); run_ablate<Cfg, 7>(M, K, N, name);
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'a') {       // exp_gemm_kc a : ablation study of the vec engine
        ablations<TileCfg<2, 2, 2, 2, 32>>(4096, 4096, 4096, "128x128");
        ablations<TileCfg<2, 2, 1, 2, 32>>(64, 576, 163840, "64x128");
        ablations<TileCfg<2, 2, 1, 1, 32>>(128, 1152, 40960, "64x64");
        ablations<TileCfg<2, 2, 1, 1, 32>>(256, 2304, 10240, "64x64");
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'v') {       // exp_gemm_kc v : MFMA issue-pattern variants of the K-loop (accumulator chains, read batching)
        auto all = [&](auto cfg, int M, int K, int N, const char* name) {
            using C = decltype(cfg);
            run_ablate<C, 0>(M, K, N, name); run_ablate<C, 32>(M, K, N, name); run_ablate<C, 64>(M, K, N, name); run_ablate<C, 128>(M, K, N, name); run_ablate<C, 144>(M, K, N, name);
        };
        all(TileCfg<2, 2, 1, 1, 32>{}, 128, 1152, 40960, "64x64");
        all(TileCfg<2, 2, 1, 2, 32>{}, 64, 576, 163840, "64x128");
        all(TileCfg<2, 2, 1, 1, 32>{}, 256, 2304, 10240, "64x64");
        all(TileCfg<2, 2, 2, 1, 32>{}, 128, 1152, 40960, "128x64");
        all(TileCfg<2, 2, 2, 2, 32>{}, 128, 1152, 40960, "128x128");
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'q') {       // exp_gemm_kc q : persistent loop with the grid clamped to what is co-resident
        for (int bpc : {1, 2, 3}) {
            run_persist<TileCfg<2, 2, 1, 2, 32>>(64, 576, 163840, "64x128", bpc);
            run_persist<TileCfg<2, 2, 1, 1, 32>>(128, 1152, 40960, "64x64", bpc);
            run_persist<TileCfg<2, 2, 2, 1, 32>>(128, 1152, 40960, "128x64", bpc);
            run_persist<TileCfg<2, 2, 2, 2, 32>>(128, 1152, 40960, "128x128", bpc);
            run_persist<TileCfg<2, 2, 1, 1, 32>>(256, 2304, 10240, "64x64", bpc);
            run_persist<TileCfg<2, 2, 2, 2, 32>>(256, 2304, 10240, "128x128", bpc);
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {       // exp_gemm_kc p : persistent tile loop vs one tile per workgroup
        for (int bpc : {2, 3, 4}) {
            run_persist<TileCfg<2, 2, 1, 2, 32>>(64, 576, 163840, "64x128", bpc);
            run_persist<TileCfg<2, 2, 1, 1, 32>>(128, 1152, 40960, "64x64", bpc);
            run_persist<TileCfg<2, 2, 1, 1, 32>>(256, 2304, 10240, "64x64", bpc);
            run_persist<TileCfg<2, 2, 1, 1, 32>>(512, 4608, 2560, "64x64", bpc);
        }
        run_persist<TileCfg<2, 2, 2, 2, 32>>(4096, 4096, 4096, "128x128", 2);
        return 0;
    }
    std::vector<int> dims;
    for (int i = 1; i < argc; ++i) dims.push_back(atoi(argv[i]));
    if (dims.empty()) dims = {4096, 4096, 4096, 64, 576, 163840, 128, 1152, 40960, 256, 2304, 10240, 512, 4608, 2560, 128, 128, 655360};
    for (size_t i = 0; i + 2 < dims.size(); i += 3) {
        const int M = dims[i], K = dims[i + 1], N = dims[i + 2];
        if (M >= 128) run<TileCfg<2, 2, 2, 2, 32>, KcCfg<2, 2, 2, 2>>(M, K, N, "128x128");
        run<TileCfg<2, 2, 1, 2, 32>, KcCfg<2, 2, 1, 2>>(M, K, N, "64x128");
        run<TileCfg<2, 2, 1, 1, 32>, KcCfg<2, 1, 1, 2>>(M, K, N, "64x64");
        // 8-wave workgroups (second column is the kc engine at the nearest shape it supports; ignore it here)
        if (M >= 128) run<TileCfg<2, 4, 2, 1, 32>, KcCfg<2, 2, 2, 2>>(M, K, N, "128x128w8");
        run<TileCfg<2, 4, 1, 1, 32>, KcCfg<2, 2, 1, 2>>(M, K, N, "64x128w8");
        if (M >= 128) run<TileCfg<4, 2, 1, 2, 32>, KcCfg<2, 2, 2, 2>>(M, K, N, "128x128w8b");
    }
    return 0;
}
