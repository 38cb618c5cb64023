// On-device random draws of the registration path (gfx950): the restart list of the pose solver and the random
// down-sampling choice of the loader, from a counter-based generator (Philox4x32-10), so that a batch never leaves HBM
// between loading and the pose, and every draw is a pure function of (seed, frame, index): reproducible on any grid.
//
// Replaces  evaluation/registration_lsq.py:163-164   ry_init = init_y_angle + random.gauss(0, ry_sigma),
//                                                    t_init  = [0, 0, random.uniform(-amp, amp)]      (unseeded `random`)
//           data/kitti_pc_img_pose_loader.py:158-171 np.random.choice(n_src, input_pt_num, replace=False)
//           data/kitti_pc_img_pose_loader.py:416-423 np.random.choice(N, node_num * 8, replace=False)  (FPS candidates)
// The reference's generators are host Mersenne Twisters with process-global state; no stream of theirs can be reproduced,
// only the distributions: normal(0, sigma), uniform(-amp, amp), uniform subsets in uniform random order.
#include "common.h"

namespace {

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 ctr, unsigned k0, unsigned k1) {
    constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)M0 * ctr.x, p1 = (unsigned long long)M1 * ctr.z;
        const U4 n{(unsigned)(p1 >> 32) ^ ctr.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ ctr.w ^ k1, (unsigned)p0};
        ctr = n;
        k0 += W0; k1 += W1;
    }
    return ctr;
}

// 53-bit uniform in (0, 1]: never 0, so log() is finite
__device__ __forceinline__ double u53(unsigned hi, unsigned lo) {
    const unsigned long long m = ((unsigned long long)(hi >> 5) << 26) | (unsigned long long)(lo >> 6);
    return ((double)m + 1.0) * (1.0 / 9007199254740992.0);
}

// stream 0: the restart list.  counter = (hypothesis index low, high, stream, 0)
__global__ __launch_bounds__(256) void draw_restarts_kernel(unsigned long long seed, long long n, double sigma, double amp,
                                                            double* __restrict__ ry_noise, double* __restrict__ init_T) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const U4 a = philox4x32_10(U4{(unsigned)i, (unsigned)(i >> 32), 0u, 0u}, (unsigned)seed, (unsigned)(seed >> 32));
    const U4 b = philox4x32_10(U4{(unsigned)i, (unsigned)(i >> 32), 0u, 1u}, (unsigned)seed, (unsigned)(seed >> 32));
    const double u1 = u53(a.x, a.y), u2 = u53(a.z, a.w), u3 = u53(b.x, b.y);
    ry_noise[i] = sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);        // Box-Muller
    init_T[3 * i + 0] = 0.0;
    init_T[3 * i + 1] = 0.0;
    init_T[3 * i + 2] = amp * (2.0 * u3 - 1.0);
}

// Uniform random subset in uniform random order: element i of frame b gets the 64-bit key (philox(b, i) << 32 | i); the
// n_out smallest keys, in key order, are the choice (a random permutation's prefix, like np.random.choice(replace=False)).
// One 1024-thread workgroup per frame sorts the keys (bitonic; LDS chunks of 8192 keys, wide strides in global scratch).
__global__ __launch_bounds__(1024) void random_choice_kernel(unsigned long long seed, int stream_id, int n_src, int P, int n_out,
                                                             unsigned long long* __restrict__ keys_all, int* __restrict__ out) {
    constexpr int CH = 8192;
    __shared__ unsigned long long chunk[CH];
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned long long* keys = keys_all + (long long)b * P;
    for (int n = tid; n < P; n += 1024) {
        unsigned long long k = ~0ull;
        if (n < n_src) {
            const U4 r = philox4x32_10(U4{(unsigned)n, (unsigned)b, (unsigned)stream_id, 2u}, (unsigned)seed, (unsigned)(seed >> 32));
            k = ((unsigned long long)r.x << 32) | (unsigned)n;
        }
        keys[n] = k;
    }
    __syncthreads();
    const int CHe = P < CH ? P : CH;
    const int nchunks = P / CHe;
    auto chunk_stages = [&](int base, int k, int j_first) {
        for (int j = j_first; j > 0; j >>= 1) {
            for (int t = tid; t < CHe / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool up = ((base + i) & k) == 0;
                const unsigned long long a = chunk[i], c = chunk[i + j];
                if ((a > c) == up) { chunk[i] = c; chunk[i + j] = a; }
            }
            __syncthreads();
        }
    };
    for (int c = 0; c < nchunks; ++c) {
        const int base = c * CHe;
        for (int i = tid; i < CHe; i += 1024) chunk[i] = keys[base + i];
        __syncthreads();
        for (int k = 2; k <= CHe; k <<= 1) chunk_stages(base, k, k >> 1);
        for (int i = tid; i < CHe; i += 1024) keys[base + i] = chunk[i];
        __syncthreads();
    }
    for (int k = CHe << 1; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= CHe; j >>= 1) {
            for (int t = tid; t < P / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool up = (i & k) == 0;
                const unsigned long long a = keys[i], c = keys[i + j];
                if ((a > c) == up) { keys[i] = c; keys[i + j] = a; }
            }
            __syncthreads();
        }
        for (int c = 0; c < nchunks; ++c) {
            const int base = c * CHe;
            for (int i = tid; i < CHe; i += 1024) chunk[i] = keys[base + i];
            __syncthreads();
            chunk_stages(base, k, CHe >> 1);
            for (int i = tid; i < CHe; i += 1024) keys[base + i] = chunk[i];
            __syncthreads();
        }
    }
    for (int i = tid; i < n_out; i += 1024) out[(long long)b * n_out + i] = (int)(unsigned)(keys[i] & 0xffffffffull);
}

// stream tag 3: the keep-mask of nn.Dropout (per_point_pn, networks_united.py:57-74; layers_pc.py:300-303,339-340): element i is kept
// with probability 1 - p.  Four elements per Philox block; a pure function of (seed, stream_id, i), so the backward pass can
// re-read or re-draw it.
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned long long seed, int stream_id, float p, long long n, unsigned char* __restrict__ mask) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    const U4 r = philox4x32_10(U4{(unsigned)q, (unsigned)(q >> 32), (unsigned)stream_id, 3u}, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned v[4] = {r.x, r.y, r.z, r.w};
    const double thr = (double)p * 4294967296.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) mask[q * 4 + k] = (double)v[k] >= thr ? 1 : 0;
}

int pow2_at_least(int n) { int p = 64; while (p < n) p <<= 1; return p; }

}  // namespace

extern "C" int di2p_draw_restarts(unsigned long long seed, int F, int R, double ry_sigma, double t_amplitude, double* ry_noise,
                                  double* init_T, void* stream) {
    DI2P_CHECK_ARG(F >= 0 && R >= 0 && ry_noise && init_T, "bad args");
    const long long n = (long long)F * R;
    if (n == 0) return 0;
    hipLaunchKernelGGL(draw_restarts_kernel, dim3(di2p_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, n, ry_sigma, t_amplitude,
                       ry_noise, init_T);
    DI2P_RETURN_LAUNCH();
}

extern "C" long long di2p_random_choice_workspace_bytes(int B, int n_src) {
    if (B < 0 || n_src < 0) return 0;
    return (long long)B * pow2_at_least(n_src) * 8 + 256;
}

extern "C" int di2p_random_choice(unsigned long long seed, int stream_id, int B, int n_src, int n_out, int32_t* idx_out, void* workspace,
                                  void* stream) {
    DI2P_CHECK_ARG(B >= 0 && n_src >= 1 && n_out >= 0 && n_out <= n_src && idx_out && workspace, "bad args (n_out <= n_src)");
    DI2P_CHECK_ARG(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
    if (B == 0 || n_out == 0) return 0;
    hipLaunchKernelGGL(random_choice_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, seed, stream_id, n_src, pow2_at_least(n_src),
                       n_out, (unsigned long long*)workspace, idx_out);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_dropout_mask(unsigned long long seed, int stream_id, float p, long long n, uint8_t* mask, void* stream) {
    DI2P_CHECK_ARG(mask && n >= 0 && p >= 0.0f && p < 1.0f, "bad args (0 <= p < 1)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(di2p_cdiv((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed, stream_id, p, n, (unsigned char*)mask);
    DI2P_RETURN_LAUNCH();
}
