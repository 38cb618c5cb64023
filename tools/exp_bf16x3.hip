// Experiment (GPU box): fp32 GEMM through bf16 MFMAs with a three-way operand split ("bf16x3").
//   x = x1 + x2 + x3 exactly (three truncated bf16 terms of 8 significand bits each), and
//   a*b ~= a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1)          six bf16 products, fp32 accumulation,
// dropping terms below 2^-24 |a||b|.  v_mfma_f32_32x32x16_bf16 does K = 16 in 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64:
// six of them are 2.67x the fp32-MFMA rate.  The question this answers: what is left of that after the split (VALU), the 1.5x
// operand bytes (LDS) and the usual staging?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_bf16x3.hip -o tools/bin/exp_bf16x3 && tools/bin/exp_bf16x3
// C[M][N] = A[M][K] * B[K][N]; A (weights) split and packed once on the host as [K/8][M][3] x 8 bf16; B (activations) fp32 in memory,
// split while it is staged into LDS.  Workgroup 128 x 128, 4 waves of 64 x 64 (2 x 2 MFMA tiles), K-step 32.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---------------------------------------------------------------- MFMA issue rates (registers only)
template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const float fa = threadIdx.x * 1e-3f, fb = 1.0f;
    u32x4 ua = {threadIdx.x, 1u, 2u, 3u}, ub = {5u, 6u, 7u, threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------- the split
__device__ __forceinline__ float hi16(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); }
// pack the bf16 (high halves) of two floats: low half <- x0, high half <- x1
__device__ __forceinline__ unsigned pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
// eight floats (consecutive k of one column) -> three 8 x bf16 fragments
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { r1[i] = v[i] - hi16(v[i]); r2[i] = r1[i] - hi16(r1[i]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p1[i] = pack_hi(v[2 * i], v[2 * i + 1]);
        p2[i] = pack_hi(r1[2 * i], r1[2 * i + 1]);
        p3[i] = pack_hi(r2[2 * i], r2[2 * i + 1]);
    }
}

constexpr int BM = 128, BN = 128, BK = 32, KG = BK / 8;          // KG: groups of 8 consecutive k per K-step

// PRODUCTS: 6 = full bf16x3; 3 = a1b1 + a1b2 + a2b1 (16-bit operands, ~2^-16 relative); 1 = plain bf16
// PRESPLIT: B arrives already split ([K/8][N][3] x 8 bf16, what a producer's epilogue could write): staging is a plain copy
template <int PRODUCTS, bool PRESPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const u32x4* __restrict__ Ap, const float* __restrict__ B, float* __restrict__ C,
                                                             int M, int K, int N, const u32x4* __restrict__ Bp = nullptr) {
    __shared__ __attribute__((aligned(16))) u32x4 Bs[2][3][KG][BN];        // 2 x 24 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
    const int T = K / BK;
    // B staging role: column n = tid & 127, k-groups (tid >> 7) and (tid >> 7) + 2
    const int sn = tid & 127, skg = tid >> 7;
    const float* bp = B + (size_t)(skg * 8) * N + n_blk + sn;
    float stage[2][8];
    u32x4 pstage[2][3];
    auto gload = [&](int t) {
        if (PRESPLIT) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 3; ++q) pstage[it][q] = Bp[((size_t)(t * KG + skg + 2 * it) * N + n_blk + sn) * 3 + q];
            return;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) stage[it][i] = bp[((size_t)t * BK + it * 16 + i) * N];
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            u32x4 p1, p2, p3;
            if (PRESPLIT) { p1 = pstage[it][0]; p2 = pstage[it][1]; p3 = pstage[it][2]; }
            else split8(stage[it], p1, p2, p3);
            Bs[buf][0][skg + 2 * it][sn] = p1;
            Bs[buf][1][skg + 2 * it][sn] = p2;
            Bs[buf][2][skg + 2 * it][sn] = p3;
        }
    };
    // A fragments straight from memory: [K/8][M][3] x 16 bytes
    u32x4 af[2][2][3];                                     // [stage][tile i][plane]
    auto aload = [&](int kg_global, int st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4* p = Ap + ((size_t)(kg_global + half) * M + m_blk + wm * 64 + i * 32 + l31) * 3;
#pragma unroll
            for (int q = 0; q < 3; ++q) af[st][i][q] = p[q];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // one k16 sub-step: B fragments from LDS, A fragments of the NEXT sub-step requested first (NEXT = false: the very last one)
    auto substep = [&](int buf, int s, int next_kg, bool has_next) __attribute__((always_inline)) {
        if (has_next) aload(next_kg, s ^ 1);
        u32x4 bf[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) bf[j][q] = Bs[buf][q][2 * s + half][wn * 64 + j * 32 + l31];
        __builtin_amdgcn_s_setprio(1);
        // smallest terms first; four independent accumulators between two MFMAs of one chain
#define PROD(QA, QB)                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s][i][QA]), __builtin_bit_cast(bf16x8, bf[j][QB]), acc[i][j], 0, 0, 0);
        if (PRODUCTS >= 6) { PROD(2, 0) PROD(1, 1) PROD(0, 2) }
        if (PRODUCTS >= 3) { PROD(1, 0) PROD(0, 1) }
        PROD(0, 0)
#undef PROD
        __builtin_amdgcn_s_setprio(0);
    };
    gload(0);
    sstore(0);
    aload(0, 0);
    __syncthreads();
    // steady state without branches: the waitcnt pass merges the counters of both arms of a branch, and an arm without loads turns
    // every wait into vmcnt(0) -- the loads just issued would be waited for at once
    for (int t = 0; t + 1 < T; ++t) {
        const int buf = t & 1;
        gload(t + 1);
        substep(buf, 0, t * KG + 2, true);
        substep(buf, 1, t * KG + 4, true);
        sstore(buf ^ 1);
        __syncthreads();
    }
    substep((T - 1) & 1, 0, (T - 1) * KG + 2, true);
    substep((T - 1) & 1, 1, 0, false);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n_blk + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                C[(size_t)m * N + n] = acc[i][j][r];
            }
        }
}

// fp32-MFMA reference of the same shape (plain LDS-staged 128 x 128 tile, K-step 16) for the accuracy comparison only
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int K, int N) {
    __shared__ float As[16][BM], Bs2[16][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1, m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = tid; e < 16 * 128; e += 256) {
            const int k = e >> 7, c = e & 127;
            As[k][c] = A[(size_t)(m_blk + c) * K + k0 + k];
            Bs2[k][c] = B[(size_t)(k0 + k) * N + n_blk + c];
        }
        __syncthreads();
        for (int kk = 0; kk < 16; kk += 2)
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + half][wm * 64 + i * 32 + l31], Bs2[kk + half][wn * 64 + j * 32 + l31], acc[i][j], 0, 0, 0);
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
        const int n = n_blk + wn * 64 + j * 32 + l31;
        for (int r = 0; r < 16; ++r) C[(size_t)(m_blk + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * N + n] = acc[i][j][r];
    }
}

template <class F> float time_ms(F f, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

uint16_t host_hi16(float x) { uint32_t u; std::memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
float host_trunc(float x) { uint32_t u; std::memcpy(&u, &x, 4); u &= 0xffff0000u; float r; std::memcpy(&r, &u, 4); return r; }

void pack_A(const std::vector<float>& A, int M, int K, std::vector<uint16_t>& Ap) {      // [K/8][M][3][8]
    Ap.assign((size_t)K * M * 3, 0);
    for (int kg = 0; kg < K / 8; ++kg)
        for (int m = 0; m < M; ++m)
            for (int i = 0; i < 8; ++i) {
                const float a = A[(size_t)m * K + kg * 8 + i];
                const float a1 = host_trunc(a), r1 = a - a1, a2 = host_trunc(r1), r2 = r1 - a2;
                uint16_t* d = &Ap[(((size_t)kg * M + m) * 3) * 8];
                d[0 * 8 + i] = host_hi16(a1); d[1 * 8 + i] = host_hi16(a2); d[2 * 8 + i] = host_hi16(r2);
            }
}

void pack_B(const std::vector<float>& B, int K, int N, std::vector<uint16_t>& Bp) {      // [K/8][N][3][8]
    Bp.assign((size_t)K * N * 3, 0);
    for (int kg = 0; kg < K / 8; ++kg)
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < 8; ++i) {
                const float a = B[(size_t)(kg * 8 + i) * N + n];
                const float a1 = host_trunc(a), r1 = a - a1, a2 = host_trunc(r1), r2 = r1 - a2;
                uint16_t* d = &Bp[(((size_t)kg * N + n) * 3) * 8];
                d[0 * 8 + i] = host_hi16(a1); d[1 * 8 + i] = host_hi16(a2); d[2 * 8 + i] = host_hi16(r2);
            }
}

void run_shape(int M, int K, int N, bool check) {
    std::vector<float> A((size_t)M * K), B((size_t)K * N);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    std::vector<uint16_t> Ap;
    pack_A(A, M, K, Ap);
    std::vector<uint16_t> Bp;
    pack_B(B, K, N, Bp);
    uint16_t* dBp; CK(hipMalloc(&dBp, Bp.size() * 2)); CK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
    float *dA, *dB, *dC; uint16_t* dAp;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dAp, Ap.size() * 2));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice));
    dim3 grid(N / BN, M / BM);
    const double fl = 2.0 * M * K * N;
    auto errs = [&](const char* name) {
        std::vector<float> Cc((size_t)M * N);
        CK(hipMemcpy(Cc.data(), dC, Cc.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, ref_max = 0;
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N; n += 13) {
                double r = 0;
                for (int k = 0; k < K; ++k) r += (double)A[(size_t)m * K + k] * B[(size_t)k * N + n];
                emax = std::max(emax, std::fabs(r - Cc[(size_t)m * N + n])); ref_max = std::max(ref_max, std::fabs(r));
            }
        printf("    %-28s max |err| %.3g  (max |C| %.3g, tolerance of the parity tests 3e-6*sqrt(K)*max|C| = %.3g)\n", name, emax, ref_max, 3e-6 * std::sqrt((double)K) * ref_max);
    };
    printf("M=%d K=%d N=%d\n", M, K, N);
    if (check) {
        hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, 0, dA, dB, dC, M, K, N); CK(hipDeviceSynchronize()); errs("fp32 MFMA");
        hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N); CK(hipDeviceSynchronize()); errs("bf16x3 (6 products)");
        hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N); CK(hipDeviceSynchronize()); errs("bf16x2 (3 products)");
        hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N); CK(hipDeviceSynchronize()); errs("bf16 (1 product)");
    }
    const float t6 = time_ms([&] { hipLaunchKernelGGL(gemm_bf16x3_kernel<6>, grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N); }, 10);
    const float t3 = time_ms([&] { hipLaunchKernelGGL(gemm_bf16x3_kernel<3>, grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N); }, 10);
    const float t1 = time_ms([&] { hipLaunchKernelGGL(gemm_bf16x3_kernel<1>, grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N); }, 10);
    const float tp = time_ms([&] { hipLaunchKernelGGL((gemm_bf16x3_kernel<6, true>), grid, dim3(256), 0, 0, (const u32x4*)dAp, dB, dC, M, K, N, (const u32x4*)dBp); }, 10);
    if (check) errs("bf16x3, B split by producer");
    printf("    6 products, B already split in memory: %.3f ms = %.1f TFLOP/s\n", tp, fl / tp / 1e9);
    printf("    6 products %.3f ms = %.1f TFLOP/s (fp32-equivalent) | 3 products %.3f ms = %.1f | 1 product %.3f ms = %.1f\n", t6, fl / t6 / 1e9, t3, fl / t3 / 1e9, t1,
           fl / t1 / 1e9);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dAp)); CK(hipFree(dBp));
}
}  // namespace

int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind) {
        auto f = [&] { if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(1024), dim3(256), 0, 0, out, iters);
                       else hipLaunchKernelGGL(rate_kernel<1>, dim3(1024), dim3(256), 0, 0, out, iters); };
        const float ms = time_ms(f, 5);
        const double flops = 1024.0 * 4 * iters * 4 * (kind == 0 ? 4096.0 : 32768.0);
        printf("MFMA issue only, %s: %.1f TFLOP/s\n", kind == 0 ? "f32 32x32x2" : "bf16 32x32x16", flops / ms / 1e9);
    }
    run_shape(512, 512, 512, true);
    run_shape(4096, 4096, 4096, false);
    run_shape(128, 128, 655360, false);     // a point layer: 128 x 128 weights over 32 x 20480 points
    run_shape(256, 512, 65536, false);      // a kNN-fusion layer
    run_shape(512, 4608, 2560, false);      // a 512-channel 3x3 convolution as a GEMM (K = 9 * 512)
    return 0;
}
