// Segment arg-max ("index_max") for gfx950.
//
// Replaces models/index_max_ext/index_max_cuda.cu:30-62 (one THREAD per batch element walking all
// N points serially, uncoalesced).  Here the row data[b,c,:] is streamed coalesced (16 B per lane),
// and the per-node running (max value, first index) lives in LDS as ONE 64-bit key per node so a
// single ds_max_u64 implements "greater value wins, on equal value the smaller n wins":
//     key = (order-preserving u32 of the float) << 32 | (0xFFFFFFFF - n)
// -0.0 is canonicalised to +0.0 first (the reference's float '>' treats them as equal); NaN and
// values <= -1000 never compete (reference: strict '>' against the -1000 floor).  HBM-bound:
// algorithmic bytes per (b,c) row = 4N (data) + 4N/C-amortised (index) + 4K (out).
#include "common.h"

namespace {

constexpr unsigned long long kInitKey = 0x3B85FFFFFFFFFFFFull;  // ord(-1000.0f)=~0xC47A0000 -> 0x3B85FFFF, low = ~0

__device__ __forceinline__ unsigned long long make_key(float v, unsigned n) {
    const float f = v + 0.0f;  // -0.0 -> +0.0
    const unsigned b = __float_as_uint(f);
    const unsigned ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - n);
}

__device__ __forceinline__ float key_value(unsigned long long key) {
    const unsigned ord = (unsigned)(key >> 32);
    const unsigned b = (ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord;
    return __uint_as_float(b);
}

__device__ __forceinline__ void offer(unsigned long long* state, int k, float v, unsigned n) {
    if (v > -1000.0f) {  // false for NaN and for anything the reference's floor rejects
        const unsigned long long key = make_key(v, n);
        if (key > state[k]) atomicMax(&state[k], key);
    }
}

// grid = (C, B, S): block handles row (b,c), the s-th slice of N.  LDS: K keys.
template <bool kSplit>
__global__ __launch_bounds__(256) void index_max_kernel(const float* __restrict__ data, const int* __restrict__ index,
                                                        int* __restrict__ max_idx, float* __restrict__ max_val,
                                                        const float* __restrict__ mask,
                                                        unsigned long long* __restrict__ ws, int C, int N, int K,
                                                        int slice) {
    extern __shared__ unsigned long long state[];
    const int c = blockIdx.x, b = blockIdx.y, s = blockIdx.z;
    for (int k = threadIdx.x; k < K; k += blockDim.x) state[k] = kInitKey;
    __syncthreads();
    const float* row = data + ((long long)b * C + c) * N;
    const int* idx = index + (long long)b * N;
    const int n0 = s * slice;
    const int n1 = min(N, n0 + slice);
    const bool vec_ok = ((((uintptr_t)row) | ((uintptr_t)idx)) & 15) == 0 && (n0 & 3) == 0;
    if (vec_ok) {
        const int nvec = (n1 - n0) >> 2;
        // four 16-byte loads of data and of index in flight per lane before the first compare (the LDS compare/atomic chain
        // otherwise serialises the stream: one load pair per iteration leaves the HBM pipe half empty)
        constexpr int U = 4;
        int i = threadIdx.x;
        for (; i + (U - 1) * (int)blockDim.x < nvec; i += U * blockDim.x) {
            float4 v[U];
            int4 k4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + 4 * (i + u * (int)blockDim.x);
                v[u] = *reinterpret_cast<const float4*>(row + n);
                k4[u] = *reinterpret_cast<const int4*>(idx + n);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + 4 * (i + u * (int)blockDim.x);
                offer(state, k4[u].x, v[u].x, n);
                offer(state, k4[u].y, v[u].y, n + 1);
                offer(state, k4[u].z, v[u].z, n + 2);
                offer(state, k4[u].w, v[u].w, n + 3);
            }
        }
        for (; i < nvec; i += blockDim.x) {
            const int n = n0 + 4 * i;
            const float4 v = *reinterpret_cast<const float4*>(row + n);
            const int4 k4 = *reinterpret_cast<const int4*>(idx + n);
            offer(state, k4.x, v.x, n);
            offer(state, k4.y, v.y, n + 1);
            offer(state, k4.z, v.z, n + 2);
            offer(state, k4.w, v.w, n + 3);
        }
        for (int m = n0 + 4 * nvec + threadIdx.x; m < n1; m += blockDim.x) offer(state, idx[m], row[m], m);
    } else {
        for (int m = n0 + threadIdx.x; m < n1; m += blockDim.x) offer(state, idx[m], row[m], m);
    }
    __syncthreads();
    const long long obase = ((long long)b * C + c) * K;
    if (kSplit) {
        for (int k = threadIdx.x; k < K; k += blockDim.x)
            if (state[k] != kInitKey) atomicMax(&ws[obase + k], state[k]);
    } else {
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            const unsigned long long key = state[k];
            const bool none = key == kInitKey;
            if (max_idx) max_idx[obase + k] = none ? 0 : (int)(0xFFFFFFFFu - (unsigned)key);
            if (max_val) {
                float v = none ? row[0] : key_value(key);
                const float m = mask ? mask[(long long)b * K + k] : (none ? 0.0f : 1.0f);
                max_val[obase + k] = v * m;
            }
        }
    }
}

__global__ void index_max_init_ws(unsigned long long* ws, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ws[i] = kInitKey;
}

__global__ void index_max_decode(const unsigned long long* __restrict__ ws, const float* __restrict__ data,
                                 const float* __restrict__ mask, int* __restrict__ max_idx,
                                 float* __restrict__ max_val, int C, int N, int K, long long total) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const unsigned long long key = ws[i];
    const bool none = key == kInitKey;
    if (max_idx) max_idx[i] = none ? 0 : (int)(0xFFFFFFFFu - (unsigned)key);
    if (max_val) {
        const long long bc = i / K;
        const int k = (int)(i - bc * K);
        const long long b = bc / C;
        const float v = none ? data[bc * N] : key_value(key);
        const float m = mask ? mask[b * K + k] : (none ? 0.0f : 1.0f);
        max_val[i] = v * m;
    }
}

int launch(const float* data, const int* index, int* max_idx, float* max_val, const float* mask, int B, int C, int N,
           int K, void* workspace, hipStream_t st) {
    if (B == 0 || C == 0 || K == 0) return 0;
    const size_t lds = (size_t)K * sizeof(unsigned long long);
    if (lds > 64 * 1024) { di2p_set_error("index_max: K=%d too large (max 8192)", K); return -1; }
    // split N (u64 atomic merge + init / decode launches) only while the grid has fewer than ~1024 blocks (4 per CU) and every block
    // still streams >= 8 KiB: at B*C = 1024 rows one launch is faster than the split (cold: 20.9 vs 25.9 us at C = 32, B = 32)
    int S = 1;
    const long long rows = (long long)B * C;
    while (rows * S < (long long)di2p_opt(DI2P_OPT_INDEX_MAX_ROWS) && (N / (S * 2)) >= 2048) S *= 2;
    if (S > 1 && workspace == nullptr) S = 1;
    if (S == 1) {
        hipLaunchKernelGGL(index_max_kernel<false>, dim3(C, B, 1), dim3(256), lds, st, data, index, max_idx, max_val,
                           mask, (unsigned long long*)nullptr, C, N, K, N);
    } else {
        const long long total = rows * K;
        int slice = ((N + S - 1) / S + 3) & ~3;
        hipLaunchKernelGGL(index_max_init_ws, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st,
                           (unsigned long long*)workspace, total);
        hipLaunchKernelGGL(index_max_kernel<true>, dim3(C, B, di2p_cdiv(N, slice)), dim3(256), lds, st, data, index,
                           (int*)nullptr, (float*)nullptr, (const float*)nullptr, (unsigned long long*)workspace, C, N,
                           K, slice);
        hipLaunchKernelGGL(index_max_decode, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st,
                           (const unsigned long long*)workspace, data, mask, max_idx, max_val, C, N, K, total);
    }
    return 0;
}

}  // namespace

extern "C" int di2p_index_max_forward(const float* data, const int32_t* index, int32_t* max_idx, int B, int C, int N,
                                      int K, void* workspace, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && C >= 0 && N >= 0 && K >= 0, "negative size");
    if (B == 0 || C == 0 || K == 0) return 0;
    DI2P_CHECK_ARG((data && index && max_idx) || N == 0, "null pointer");
    if (launch(data, index, max_idx, nullptr, nullptr, B, C, N, K, workspace, (hipStream_t)stream)) return -1;
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_index_max_values(const float* data, const int32_t* index, const float* mask, float* max_val,
                                     int32_t* max_idx, int B, int C, int N, int K, void* workspace, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && C >= 0 && N > 0 && K >= 0, "bad size");
    if (B == 0 || C == 0 || K == 0) return 0;
    if (launch(data, index, max_idx, max_val, mask, B, C, N, K, workspace, (hipStream_t)stream)) return -1;
    DI2P_RETURN_LAUNCH();
}
