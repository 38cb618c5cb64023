// Radius neighbour query ("ball_query") for gfx950.
//
// Replaces models/ball_query_ext/ball_query_cuda.cu:11-50 (one THREAD per batch element, serial
// scan).  Here one wavefront owns one (b,m) row of node_to_point_dist and scans it 256 points per
// step with 16-byte loads; hit positions come from 64-bit ballots + popcounts, so output order is
// the reference's ascending-n order.  Early exit once K hits are found; cyclic padding afterwards.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ dist, int* __restrict__ out,
                                                         float radius, int K, long long rows, int N) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* d = dist + row * N;
    int* o = out + row * K;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int found = 0;  // wave-uniform
    const bool vec_ok = (((uintptr_t)d) & 15) == 0;
    for (int base = 0; base < N && found < K; base += 256) {
        float v[4];
        const int n = base + lane * 4;
        if (vec_ok && n + 3 < N) {
            const float4 q = *reinterpret_cast<const float4*>(d + n);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (n + j < N) ? d[n + j] : __builtin_nanf("");
        }
        bool h[4];
        unsigned long long bal[4];
        int before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = v[j] <= radius;  // NaN (padding) never hits
            bal[j] = __ballot(h[j]);
            before += __popcll(bal[j] & lt);
            total += __popcll(bal[j]);
        }
        int pos = found + before;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (h[j]) {
                if (pos < K) o[pos] = n + j;
                ++pos;
            }
        }
        found += total;
    }
    if (found > K) found = K;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (found == 0) {
        for (int i = lane; i < K; i += 64) o[i] = 0;
    } else if (found < K) {
        // slot found+i = slot (i % found): the first `found` slots are final, so every pad slot can be
        // computed independently (the reference's serial copy gives the same periodic extension).
        for (int i = lane; i < K - found; i += 64) o[found + i] = o[i % found];
    }
}

}  // namespace

extern "C" int di2p_ball_query_forward(const float* node_to_point_dist, int32_t* out_idx, float radius, int K, int B,
                                       int M, int N, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && M >= 0 && N >= 0 && K >= 0, "negative size");
    const long long rows = (long long)B * M;
    if (rows == 0 || K == 0) return 0;
    DI2P_CHECK_ARG(node_to_point_dist || N == 0, "null dist");
    DI2P_CHECK_ARG(out_idx, "null out");
    hipLaunchKernelGGL(ball_query_kernel, dim3(di2p_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       node_to_point_dist, out_idx, radius, K, rows, N);
    DI2P_RETURN_LAUNCH();
}
