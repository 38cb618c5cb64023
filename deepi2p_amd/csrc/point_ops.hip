// Point <-> node kernels of the SO-Net point branch and the fusion head (gfx950).
//
// The reference materialises B x 3 x N x M broadcast tensors (1 GB each at B=32) to get distances,
// one-hot masks and cluster sums (models/networks_pc.py:61-76) and 4-D gathers for the
// interpolation (models/networks_united.py:76-103).  Here the M node coordinates of a frame sit in
// LDS, one lane owns one query point and keeps its k best (distance, node) pairs in registers, so a
// frame's whole assignment costs one coalesced read of pc and one write of the indices/weights.
#include "common.h"

namespace {

// ----------------------------------------------------------------------------------------------
// k nearest nodes, ascending (distance, node id).  Distance is sqrtf((dx*dx + dy*dy) + dz*dz) with
// separately rounded operations, the same value torch.norm(dim=1) produces for 3 components.
// The scan orders the candidates by the SQUARED distance (sqrtf is monotone: a quarter-rate instruction per node and lane is
// only spent on the k winners) and the node coordinates are wave-uniform reads straight from global memory (scalar loads
// through the constant cache: no LDS staging, no LDS traffic in the loop); the k winners are then re-ordered on the rounded
// distance itself, so that nodes whose distances round to the same float come out lowest-index first, as a sort on d would.
template <int KN>
__global__ __launch_bounds__(256) void knn_nodes_kernel(const float* __restrict__ query, const float* __restrict__ nodes,
                                                        int* __restrict__ idx, float* __restrict__ weights, int Nq,
                                                        int M) {
    const int b = blockIdx.y;
    const float* __restrict__ nb = nodes + (long long)b * 3 * M;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int nc = min(n, Nq - 1);
    const float* q = query + (long long)b * 3 * Nq;
    const float qx = q[nc], qy = q[Nq + nc], qz = q[2 * Nq + nc];
    // The k best candidates, sorted by (d^2, node id), as separate distance and id registers.  Candidates arrive in ascending id, so
    // "closer, then lower id" is the STRICT 32-bit compare d^2 < best d^2 (an equal distance goes behind the one already there, which has
    // the lower id; a NaN distance compares false and is never inserted) and inserting is a branch-free chain of selects.  (Round 3 kept
    // 64-bit keys d^2 << 32 | id: the same selects, but three 64-bit compares per node instead of three 32-bit ones.  The branchy insertion
    // before that compiled to ~72 instructions per node, half of them exec-mask bookkeeping.)
    // The distances are compared as their BIT patterns (unsigned order = float order for d^2 >= +0); the empty-slot pattern 0x7f800001 lies
    // above +inf (an overflowed distance is still inserted, as with the 64-bit keys) and below every quiet NaN (never inserted).
    unsigned kd[KN];
    int ki[KN];
#pragma unroll
    for (int j = 0; j < KN; ++j) { kd[j] = 0x7f800001u; ki[j] = 0x7fffffff; }       // (above +inf, no node)
    auto consider = [&](float nx, float ny, float nz, int m) __attribute__((always_inline)) {
        const float dx = __fsub_rn(qx, nx), dy = __fsub_rn(qy, ny), dz = __fsub_rn(qz, nz);
        const unsigned d = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        bool lt[KN];
#pragma unroll
        for (int j = 0; j < KN; ++j) lt[j] = d < kd[j];
#pragma unroll
        for (int j = KN - 1; j >= 1; --j) {                                             // shift down / insert / keep
            kd[j] = lt[j - 1] ? kd[j - 1] : (lt[j] ? d : kd[j]);
            ki[j] = lt[j - 1] ? ki[j - 1] : (lt[j] ? m : ki[j]);
        }
        kd[0] = lt[0] ? d : kd[0];
        ki[0] = lt[0] ? m : ki[0];
    };
    int m = 0;
    for (; m + 4 <= M; m += 4) {          // uniform addresses: the compiler turns these into s_load_dwordx4
        const float x0 = nb[m], x1 = nb[m + 1], x2 = nb[m + 2], x3 = nb[m + 3];
        const float y0 = nb[M + m], y1 = nb[M + m + 1], y2 = nb[M + m + 2], y3 = nb[M + m + 3];
        const float z0 = nb[2 * M + m], z1 = nb[2 * M + m + 1], z2 = nb[2 * M + m + 2], z3 = nb[2 * M + m + 3];
        consider(x0, y0, z0, m); consider(x1, y1, z1, m + 1); consider(x2, y2, z2, m + 2); consider(x3, y3, z3, m + 3);
    }
    for (; m < M; ++m) consider(nb[m], nb[M + m], nb[2 * M + m], m);
    if (n >= Nq) return;
    float bd[KN];
    int bi[KN];
#pragma unroll
    for (int j = 0; j < KN; ++j) { bd[j] = __fsqrt_rn(__uint_as_float(kd[j])); bi[j] = ki[j]; }
    // equal rounded distances: lowest node id first (insertion sort on (d, id); the squared order is already almost that)
#pragma unroll
    for (int a = 1; a < KN; ++a)
#pragma unroll
        for (int j = a; j > 0; --j) {
            const bool sw = bd[j] == bd[j - 1] && bi[j] < bi[j - 1];
            const int t = bi[j];
            bi[j] = sw ? bi[j - 1] : t;
            bi[j - 1] = sw ? t : bi[j - 1];
        }
    int* o = idx + ((long long)b * Nq + n) * KN;
#pragma unroll
    for (int j = 0; j < KN; ++j) o[j] = bi[j] == 0x7fffffff ? 0 : bi[j];
    if (weights) {
        float sum = bd[0];
#pragma unroll
        for (int j = 1; j < KN; ++j) sum = __fadd_rn(sum, bd[j]);
        float* w = weights + ((long long)b * Nq + n) * KN;
#pragma unroll
        for (int j = 0; j < KN; ++j) w[j] = __fsub_rn(1.0f, __fdiv_rn(bd[j], sum));
    }
}

// Node-level queries (a few hundred per frame: the kNN-fusion layer's k = 16 over cluster means, node_a <- node_b): one
// WAVEFRONT per query instead of one lane.  Each lane holds up to 4 candidate keys (distance bits << 32 | node id: unsigned
// order = ascending (distance, id) for the non-negative distances), a round picks the wave-wide minimum with a butterfly and
// retires it; k rounds give the k nearest in order.  Same distances, same tie rule, same weight arithmetic as the kernel above.
template <int KN>
__global__ __launch_bounds__(256) void knn_nodes_wave_kernel(const float* __restrict__ query, const float* __restrict__ nodes,
                                                             int* __restrict__ idx, float* __restrict__ weights, int Nq, int M) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= Nq) return;                                   // wave-uniform
    const float* nb = nodes + (long long)b * 3 * M;
    const float* q = query + (long long)b * 3 * Nq;
    const float qx = q[n], qy = q[Nq + n], qz = q[2 * Nq + n];
    unsigned long long key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int m = lane + 64 * u;
        key[u] = ~0ull;
        if (m < M) {
            const float dx = __fsub_rn(qx, nb[m]), dy = __fsub_rn(qy, nb[M + m]), dz = __fsub_rn(qz, nb[2 * M + m]);
            const float d = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            key[u] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)m;
            if (d != d) key[u] = ~0ull - 1 - (unsigned)(M - m);       // NaN distances rank last, by id
        }
    }
    float bd[KN];
    int mine_i = 0;
    float mine_d = 0.0f;
#pragma unroll
    for (int j = 0; j < KN; ++j) {
        unsigned long long best = key[0] < key[1] ? key[0] : key[1];
        const unsigned long long b2 = key[2] < key[3] ? key[2] : key[3];
        best = best < b2 ? best : b2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (key[u] == best) key[u] = ~0ull;      // node ids are unique: exactly one lane retires it
        bd[j] = __uint_as_float((unsigned)(best >> 32));
        if (lane == j) { mine_i = best == ~0ull ? 0 : (int)(unsigned)(best & 0xffffffffull); mine_d = bd[j]; }
    }
    if (lane < KN) idx[((long long)b * Nq + n) * KN + lane] = mine_i;
    if (weights) {
        float sum = bd[0];
#pragma unroll
        for (int j = 1; j < KN; ++j) sum = __fadd_rn(sum, bd[j]);
        if (lane < KN) weights[((long long)b * Nq + n) * KN + lane] = __fsub_rn(1.0f, __fdiv_rn(mine_d, sum));
    }
}

// ----------------------------------------------------------------------------------------------
// Cluster sums in 2^-24 m fixed point: integer adds are associative, so the result does not depend
// on the order LDS atomics retire (bit-reproducible run to run), and it is exact for the quantised
// inputs.  One 1024-thread workgroup per frame; the work is tiny (12N bytes).
constexpr double kFix = 16777216.0;

__global__ __launch_bounds__(1024) void cluster_stats_kernel(const float* __restrict__ pc, const int* __restrict__ knn_idx,
                                                             int idx_stride, float* __restrict__ cluster_mean,
                                                             float* __restrict__ mask, int* __restrict__ min_idx, int N,
                                                             int M) {
    extern __shared__ unsigned long long s_acc[];  // [M][4]: x,y,z (two's complement), count
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 4 * M; i += blockDim.x) s_acc[i] = 0ull;
    __syncthreads();
    const float* p = pc + (long long)b * 3 * N;
    const int* ki = knn_idx + (long long)b * N * idx_stride;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int k = ki[(long long)n * idx_stride];
        if (min_idx) min_idx[(long long)b * N + n] = k;
        const long long fx = __double2ll_rn((double)p[n] * kFix);
        const long long fy = __double2ll_rn((double)p[N + n] * kFix);
        const long long fz = __double2ll_rn((double)p[2 * N + n] * kFix);
        atomicAdd(&s_acc[4 * k + 0], (unsigned long long)fx);
        atomicAdd(&s_acc[4 * k + 1], (unsigned long long)fy);
        atomicAdd(&s_acc[4 * k + 2], (unsigned long long)fz);
        atomicAdd(&s_acc[4 * k + 3], 1ull);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const float cnt = (float)s_acc[4 * k + 3];
        const float den = __fadd_rn(cnt, 1e-5f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = (float)((double)(long long)s_acc[4 * k + c] / kFix);
            cluster_mean[((long long)b * 3 + c) * M + k] = __fdiv_rn(s, den);
        }
        if (mask) mask[(long long)b * M + k] = cnt > 0.0f ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(256) void build_point_input_kernel(const float* __restrict__ pc, const float* __restrict__ intensity,
                                                                const float* __restrict__ sn, const float* __restrict__ cluster_mean,
                                                                const int* __restrict__ min_idx, float* __restrict__ centers,
                                                                float* __restrict__ aug, int N, int M) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int k = min_idx[(long long)b * N + n];
    const float* cm = cluster_mean + (long long)b * 3 * M;
    const float* p = pc + (long long)b * 3 * N;
    float* a = aug + (long long)b * 7 * N;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float ctr = cm[c * M + k];
        if (centers) centers[((long long)b * 3 + c) * N + n] = ctr;
        a[c * N + n] = __fsub_rn(p[c * N + n], ctr);
    }
    a[3 * N + n] = intensity[(long long)b * N + n];
    const float* s = sn + (long long)b * 3 * N;
#pragma unroll
    for (int c = 0; c < 3; ++c) a[(4 + c) * N + n] = s[c * N + n];
}

// out[b,c,n] = sum_j w[b,n,j] * feats[b,c,idx[b,n,j]]   (left-to-right sum like torch.sum over k)
__global__ __launch_bounds__(256) void interpolate_kernel(const float* __restrict__ feats, const int* __restrict__ idx,
                                                          const float* __restrict__ weights, float* __restrict__ out, int C,
                                                          int M, int Nq, int k) {
    const int b = blockIdx.z;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Nq) return;
    const int c0 = blockIdx.y * 32;
    const int* ii = idx + ((long long)b * Nq + n) * k;
    const float* ww = weights + ((long long)b * Nq + n) * k;
    for (int c = c0; c < min(C, c0 + 32); ++c) {
        const float* f = feats + ((long long)b * C + c) * M;
        float acc = 0.0f;
        for (int j = 0; j < k; ++j) acc = __fadd_rn(acc, __fmul_rn(ww[j], f[ii[j]]));
        out[((long long)b * C + c) * Nq + n] = acc;
    }
}

// The same sums for k <= 4 with the 32-channel slice of the node table in LDS (32 * M floats): the gathers become LDS reads, the indices and
// weights of a column are read once instead of once per channel.  (Training runs the plain interpolation on 512 and 128 channels of 20480
// points: 370 -> ~100 us for the larger one; inference folds it into the consumer's epilogue and never gets here.)
template <int KN>
__global__ __launch_bounds__(256) void interpolate_lds_kernel(const float* __restrict__ feats, const int* __restrict__ idx,
                                                              const float* __restrict__ weights, float* __restrict__ out, int C, int M, int Nq) {
    extern __shared__ float tab[];                 // [32][M]
    const int b = blockIdx.z, c0 = blockIdx.y * 32, cn = min(32, C - c0);
    const float* f = feats + ((long long)b * C + c0) * M;
    for (int i = threadIdx.x; i < cn * M; i += 256) tab[i] = f[i];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= Nq) return;
    int ii[KN];
    float ww[KN];
#pragma unroll
    for (int j = 0; j < KN; ++j) { ii[j] = idx[((long long)b * Nq + n) * KN + j]; ww[j] = weights[((long long)b * Nq + n) * KN + j]; }
    for (int c = 0; c < cn; ++c) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < KN; ++j) acc = __fadd_rn(acc, __fmul_rn(ww[j], tab[c * M + ii[j]]));
        out[((long long)b * C + c0 + c) * Nq + n] = acc;
    }
}

__global__ void gather_neighbors_kernel(const float* __restrict__ database, const float* __restrict__ query,
                                        const int* __restrict__ idx, float* __restrict__ out, int Md, int Mq, int K) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Mq * K) return;
    const int m = t / K;
    const int src = idx[(long long)b * Mq * K + t];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        out[((long long)b * 3 + c) * Mq * K + t] =
            __fsub_rn(database[((long long)b * 3 + c) * Md + src], query[((long long)b * 3 + c) * Mq + m]);
}

__global__ __launch_bounds__(256) void argmax_channels_kernel(const float* __restrict__ scores, int* __restrict__ out, int C, int N,
                                                              long long batch_stride) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* s = scores + (long long)b * batch_stride;
    float best = s[n];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float v = s[(long long)c * N + n];
        // torch.argmax semantics: NaN ranks above everything and the FIRST NaN / first maximum wins
        if (best == best && (v != v || v > best)) { best = v; bi = c; }
    }
    out[(long long)b * N + n] = bi;
}

// torch.max semantics: NaN propagates (fmaxf would silently drop it and hide a faulty activation)
__device__ __forceinline__ float nanmax(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

// max over the last axis of [rows][N].  A row is handled by G = min(64, pow2 >= N) consecutive lanes, so short rows
// (the K = 16 neighbour axis of the kNN fusion) pack 64/G rows into a wavefront and stay coalesced.
__global__ __launch_bounds__(256) void channel_max_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int N, int G) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / G;                                    // rows per wavefront
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + lane / G;
    const int n0 = lane % G;
    float m = -__builtin_inff();
    if (row < rows) {
        const float* r = x + row * N;
        for (int n = n0; n < N; n += G) m = nanmax(m, r[n]);
    }
    for (int o = G >> 1; o > 0; o >>= 1) m = nanmax(m, __shfl_xor(m, o));
    if (row < rows && n0 == 0) y[row] = m;
}

__global__ void f32_to_f64_kernel(const float* __restrict__ in, double* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (double)in[i];
}

}  // namespace

extern "C" int di2p_knn_nodes(const float* query, const float* nodes, int32_t* idx, float* weights, int B, int Nq, int M,
                              int k, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && Nq >= 0 && M > 0, "bad size");
    DI2P_CHECK_ARG(k >= 1 && k <= 16 && k <= M, "k must be in [1,16] and <= M");
    DI2P_CHECK_ARG(M <= 4096, "M too large for the LDS node table");
    if (B == 0 || Nq == 0) return 0;
    const dim3 grid(di2p_cdiv(Nq, 256), B), block(256);
    hipStream_t st = (hipStream_t)stream;
    // few queries per frame over few nodes: one wavefront per query (the lane-per-query kernel would leave the chip idle)
    const bool per_wave = M <= 256 && Nq <= 1024;
    const dim3 wgrid(di2p_cdiv(Nq, 4), B);
#define DI2P_KNN_CASE(KK) \
    case KK: if (per_wave) hipLaunchKernelGGL(knn_nodes_wave_kernel<KK>, wgrid, block, 0, st, query, nodes, idx, weights, Nq, M); \
             else hipLaunchKernelGGL(knn_nodes_kernel<KK>, grid, block, 0, st, query, nodes, idx, weights, Nq, M); \
             break;
    switch (k) {
        DI2P_KNN_CASE(1) DI2P_KNN_CASE(2) DI2P_KNN_CASE(3) DI2P_KNN_CASE(4) DI2P_KNN_CASE(5) DI2P_KNN_CASE(6)
        DI2P_KNN_CASE(7) DI2P_KNN_CASE(8) DI2P_KNN_CASE(9) DI2P_KNN_CASE(10) DI2P_KNN_CASE(11) DI2P_KNN_CASE(12)
        DI2P_KNN_CASE(13) DI2P_KNN_CASE(14) DI2P_KNN_CASE(15) DI2P_KNN_CASE(16)
    }
#undef DI2P_KNN_CASE
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_cluster_stats(const float* pc, const int32_t* knn_idx, int idx_stride, float* cluster_mean, float* mask,
                                  int32_t* min_idx, int B, int N, int M, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && N >= 0 && M > 0 && idx_stride >= 1, "bad size");
    DI2P_CHECK_ARG(M <= 2048, "M too large");
    if (B == 0) return 0;
    hipLaunchKernelGGL(cluster_stats_kernel, dim3(B), dim3(1024), (size_t)4 * M * sizeof(unsigned long long),
                       (hipStream_t)stream, pc, knn_idx, idx_stride, cluster_mean, mask, min_idx, N, M);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_build_point_input(const float* pc, const float* intensity, const float* sn, const float* cluster_mean,
                                      const int32_t* min_idx, float* pc_centers, float* augmented, int B, int N, int M,
                                      void* stream) {
    DI2P_CHECK_ARG(B >= 0 && N >= 0 && M > 0, "bad size");
    if (B == 0 || N == 0) return 0;
    hipLaunchKernelGGL(build_point_input_kernel, dim3(di2p_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, pc,
                       intensity, sn, cluster_mean, min_idx, pc_centers, augmented, N, M);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_interpolate(const float* feats, const int32_t* idx, const float* weights, float* out, int B, int C, int M,
                                int Nq, int k, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && C >= 0 && M > 0 && Nq >= 0 && k >= 1, "bad size");
    if (B == 0 || C == 0 || Nq == 0) return 0;
    const dim3 grid(di2p_cdiv(Nq, 256), di2p_cdiv(C, 32), B);
    const size_t lds = (size_t)32 * M * sizeof(float);
    if (k <= 4 && lds <= 64 * 1024 && Nq >= 1024) {          // (few columns: staging the table would cost more than its gathers)
        hipStream_t st = (hipStream_t)stream;
        switch (k) {
            case 1: hipLaunchKernelGGL(interpolate_lds_kernel<1>, grid, dim3(256), lds, st, feats, idx, weights, out, C, M, Nq); break;
            case 2: hipLaunchKernelGGL(interpolate_lds_kernel<2>, grid, dim3(256), lds, st, feats, idx, weights, out, C, M, Nq); break;
            case 3: hipLaunchKernelGGL(interpolate_lds_kernel<3>, grid, dim3(256), lds, st, feats, idx, weights, out, C, M, Nq); break;
            default: hipLaunchKernelGGL(interpolate_lds_kernel<4>, grid, dim3(256), lds, st, feats, idx, weights, out, C, M, Nq); break;
        }
        DI2P_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(interpolate_kernel, grid, dim3(256), 0, (hipStream_t)stream, feats, idx, weights, out, C, M, Nq, k);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_gather_neighbors(const float* database, const float* query, const int32_t* idx, float* out, int B, int Md,
                                     int Mq, int K, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && Md > 0 && Mq >= 0 && K >= 1, "bad size");
    if (B == 0 || Mq == 0) return 0;
    hipLaunchKernelGGL(gather_neighbors_kernel, dim3(di2p_cdiv((long long)Mq * K, 256), B), dim3(256), 0,
                       (hipStream_t)stream, database, query, idx, out, Md, Mq, K);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_argmax_channels(const float* scores, int32_t* out, int B, int C, int N, long long batch_stride, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && C >= 1 && N >= 0, "bad size");
    if (B == 0 || N == 0) return 0;
    if (batch_stride <= 0) batch_stride = (long long)C * N;
    hipLaunchKernelGGL(argmax_channels_kernel, dim3(di2p_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, scores, out, C, N, batch_stride);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_channel_max(const float* x, float* y, int B, int C, int N, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && C >= 0 && N >= 1, "bad size");
    const long long rows = (long long)B * C;
    if (rows == 0) return 0;
    int G = 1;
    while (G < N && G < 64) G <<= 1;
    hipLaunchKernelGGL(channel_max_kernel, dim3(di2p_cdiv(rows, 4 * (64 / G))), dim3(256), 0, (hipStream_t)stream, x, y, rows, N, G);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_f32_to_f64(const float* in, double* out, long long n, void* stream) {
    if (n <= 0) return 0;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(f32_to_f64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, n);
    DI2P_RETURN_LAUNCH();
}
