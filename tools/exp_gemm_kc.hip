// Standalone calibration of the fp32-MFMA tile engine (mfma_tile.h, production) and of the experimental k-contiguous
// variant (tools/mfma_tile_kc.h) on dense GEMM shapes with trivial loaders (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I deepi2p_amd/csrc -I include -I tools tools/exp_gemm_kc.hip -o tools/bin/exp_gemm_kc
//   tools/bin/exp_gemm_kc M K N [M K N ...]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

template <class F> float time_ms(F f, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
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

int main(int argc, char** argv) {
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
