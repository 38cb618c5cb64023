// Batched "inverse camera projection" pose solver for gfx950 -- replaces Ceres + the reference's
// 60-process restart fan-out.
//
// Replaces evaluation/frustum_reg/src/registration.cpp:9-186 (solvePGivenK), the four auto-diff
// functors registration_2d.hpp:35-69,107-129 / registration_3d.hpp:35-68,106-127, and the process
// waves of evaluation/registration_lsq.py:142-186.
//
// One 4-wave WORKGROUP per pose hypothesis (frame = block % F, so a frame's hypotheses share an XCD's L2).  A sweep
// over the frame's records evaluates residuals and ANALYTIC Jacobian rows (the reference differentiates with Jets),
// robustifies with the Cauchy corrector and accumulates J^T J (upper triangle), J^T r and the cost in fp64 registers:
//   * the records are sorted once per call by (label, Morton cell) into 64-point clusters with bounding boxes
//     (prepare_kernel); per sweep a box test against the five frustum planes settles most clusters without touching
//     their points, exactly (cluster_status);
//   * the remaining clusters are classified per point (phase A), the ACTIVE records are compacted into a per-wave LDS
//     queue and evaluated densely (phase B); line-search trials beyond the first are swept cost-only;
//   * wave butterflies + a fixed-order combination of the 4 wave partials give deterministic sums; the
//     Levenberg-Marquardt state machine (damping, box projection, Armijo search, accept/reject, termination tests) lives
//     in LDS and is advanced by thread 0 between sweeps.
// The algorithm statement is the oracle's (oracle/frustum_lm.cpp); DESIGN.md lists it step by step.
#include "common.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>

// default instantiation of the 2-D solver: <min waves per SIMD> x <waves per hypothesis> (override for experiments: -D...).
// 4 x 4: four workgroups of four waves per CU (128 VGPRs; 39.5 KB of LDS each); measured 7.40 vs 7.80 ms per launch and 3679 vs 3555
// frames/s against 3 waves per SIMD once the kernel stopped spilling inside its sweep loops.
#ifndef DI2P_SOLVER_DEFAULT_MINW
#define DI2P_SOLVER_DEFAULT_MINW 4
#endif
#ifndef DI2P_SOLVER_DEFAULT_WPH
#define DI2P_SOLVER_DEFAULT_WPH 4
#endif
#ifndef DI2P_SOLVER_PF
#define DI2P_SOLVER_PF 4          // guard-only clusters walked per straight-line batch of the cluster walk (sizes the per-wave queue as well)
#endif
#ifndef DI2P_SOLVER_PFC
#define DI2P_SOLVER_PFC 2         // clusters CLASSIFIED per straight-line batch (1, 2 or 4): ~2 per label block and round need it since the cache (round 6: 4 -> 2, -2 % vector instructions)
#endif

namespace {

enum { T_MAX_ITER = 0, T_GRADIENT = 1, T_PARAMETER = 2, T_FUNCTION = 3, T_RADIUS = 4, T_INVALID = 5, T_EVAL_FAIL = 6 };

struct Cam { double fx, fy, cx, cy, H1, W1; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int NP> struct Tri { static constexpr int N = NP * (NP + 1) / 2; };
template <int NP> __device__ __forceinline__ constexpr int tri(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

// Rotation of the current iterate and its parameter derivatives, computed once per sweep.
template <int NP>
struct Rot {
    double R[9];        // row-major
    double dR[3][9];    // NP == 6 only: dR/dw_i
};

__device__ __forceinline__ void skew_add(double* M, double s, double ux, double uy, double uz) {  // M += s*[u]x
    M[1] -= s * uz; M[2] += s * uy; M[3] += s * uz; M[5] -= s * ux; M[6] -= s * uy; M[7] += s * ux;
}

template <int NP>
__device__ __forceinline__ void make_rot(const double* x, Rot<NP>& r) {
    if (NP == 4) {
        // AngleAxisRotatePoint with axis (0,theta,0): Ry(theta); first-order branch for theta^2 <= eps
        const double th = x[0];
        double c, s;
        if (th * th > DBL_EPSILON) { sincos(th, &s, &c); } else { c = 1.0; s = th; }      // one argument reduction for both
        r.R[0] = c; r.R[1] = 0; r.R[2] = s; r.R[3] = 0; r.R[4] = 1; r.R[5] = 0; r.R[6] = -s; r.R[7] = 0; r.R[8] = c;
    } else {
        const double wx = x[0], wy = x[1], wz = x[2];
        const double t2 = wx * wx + wy * wy + wz * wz;
        for (int i = 0; i < 9; ++i) { r.R[i] = 0; r.dR[0][i] = r.dR[1][i] = r.dR[2][i] = 0; }
        if (t2 > DBL_EPSILON) {
            const double th = sqrt(t2), ux = wx / th, uy = wy / th, uz = wz / th;
            double c, s;
            sincos(th, &s, &c);
            const double oc = 1.0 - c;
            const double u[3] = {ux, uy, uz};
            r.R[0] = r.R[4] = r.R[8] = c;
            skew_add(r.R, s, ux, uy, uz);
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) r.R[a * 3 + b] += oc * u[a] * u[b];
            for (int i = 0; i < 3; ++i) {
                double du[3];
                for (int j = 0; j < 3; ++j) du[j] = ((i == j ? 1.0 : 0.0) - u[i] * u[j]) / th;
                double* D = r.dR[i];
                D[0] = D[4] = D[8] = -s * u[i];
                skew_add(D, c * u[i], ux, uy, uz);
                skew_add(D, s, du[0], du[1], du[2]);
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) D[a * 3 + b] += s * u[i] * u[a] * u[b] + oc * (du[a] * u[b] + u[a] * du[b]);
            }
        } else {  // pt + w x pt
            r.R[0] = r.R[4] = r.R[8] = 1.0;
            skew_add(r.R, 1.0, wx, wy, wz);
            skew_add(r.dR[0], 1.0, 1, 0, 0);
            skew_add(r.dR[1], 1.0, 0, 1, 0);
            skew_add(r.dR[2], 1.0, 0, 0, 1);
        }
    }
}

// Packed, front-filtered point records: ONE aligned vector load per point per sweep.
//   PT=float : float4 {x, y, z, label bits}            (16 B)
//   PT=double: {double x, y, z; long long label}       (32 B)
// Points whose label is not 0/1 (incl. the -1 the front filter writes) are dropped here once, in stable
// order, so the hot sweeps never see them (registration.cpp:87-125 skips them per residual block).
template <typename PT> struct Rec;
template <> struct __attribute__((aligned(16))) Rec<float> { float x, y, z; int lab; };
template <> struct __attribute__((aligned(16))) Rec<double> { double x, y, z; long long lab; };

// Frame preparation (once per solve call, one 1024-thread workgroup per frame):
//   1. records with label 0/1 are sorted by (label: 1 first, Morton code of the (x,z) ground-plane cell) with an
//      in-workgroup bitonic sort on unique 64-bit keys (key << 32 | point index): deterministic, data-independent
//      network, 64 KB LDS chunks with the wide strides done in the (L2-resident) global scratch;
//   2. consecutive groups of CL = 64 sorted records of one label form a CLUSTER with an axis-aligned bounding box.
// The solver's sweeps test each box against the five planes of the camera frustum of the current iterate and
// classify the points of a cluster individually only when the box touches a plane (see sweep_clusters).
constexpr int CL = 64;
// axis-aligned bounds of a cluster (centre, half extents) in fp32, rounded OUTWARD: the fp32 box contains every point of the cluster
// rxz / r3: the largest ground-plane radius sqrt(x^2 + z^2) and the largest norm of a point of the cluster, rounded UP -- how far a point of
// the cluster can move per radian of iterate rotation (2-D solver: about the y axis; 3-D: any axis), see the classification cache below.
struct __attribute__((aligned(16))) Box { float cx, cy, cz, hx, hy, hz, rxz, r3; };

__device__ __forceinline__ unsigned spread10(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 8)) & 0x00ff00ffu; v = (v | (v << 4)) & 0x0f0f0f0fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// Index of cell (x, y) of a 1024 x 1024 grid along the Hilbert curve.  Unlike the Z-order (Morton) curve it has no jumps: 64 consecutive
// records always form one connected patch of the ground plane, so the cluster boxes are tighter (modelled on the config-2 scene,
// tools/model_cluster_keys.py: 145 instead of 168 of 321 clusters per frame touch a frustum plane).
__device__ __forceinline__ unsigned hilbert10(unsigned x, unsigned y) {
    unsigned d = 0;
#pragma unroll
    for (unsigned s = 512; s > 0; s >>= 1) {
        const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
            const unsigned t = x; x = y; y = t;
        }
    }
    return d;
}

// Wave-wide min / max of four floats at once by DPP (rows of 16 lanes, then row_bcast15 / row_bcast31 into lane 63; v_min / v_max drop NaNs like
// fmin / fmax).  One asm block per call: hipcc expands fminf(v, dpp(v)) into four instructions per join; the four values are interleaved so that
// a value's consecutive joins are three instructions apart (a DPP read needs two wait states after a write of the same register).
#define DI2P_RED4(OP, CTRL)                          \
    OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t" OP " %3, %3, %3 " CTRL "\n\t"
#define DI2P_RED4_ALL(OP)                                                        \
    asm volatile("s_nop 1\n\t"                                                  \
                 DI2P_RED4(OP, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
                 DI2P_RED4(OP, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") \
                 DI2P_RED4(OP, "row_half_mirror row_mask:0xf bank_mask:0xf")     \
                 DI2P_RED4(OP, "row_mirror row_mask:0xf bank_mask:0xf")          \
                 DI2P_RED4(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")        \
                 DI2P_RED4(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf")        \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ void wave_min4_f32(float& a, float& b, float& c, float& d) {
    DI2P_RED4_ALL("v_min_f32_dpp");
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63)); b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
    c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63)); d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 63));
}
__device__ __forceinline__ void wave_max4_f32(float& a, float& b, float& c, float& d) {
    DI2P_RED4_ALL("v_max_f32_dpp");
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63)); b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
    c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63)); d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 63));
}
#undef DI2P_RED4_ALL

template <typename PT>
__global__ __launch_bounds__(1024) void prepare_kernel(const PT* __restrict__ points, const int* __restrict__ labels, int N, int P,
                                                       int NCMAX, unsigned long long* __restrict__ keys_all,
                                                       Rec<PT>* __restrict__ packed, Box* __restrict__ boxes_all,
                                                       int* __restrict__ counts, int force_bitonic, const double* __restrict__ Kmat, double H,
                                                       double W, float* __restrict__ camf_all, const int* __restrict__ only_flagged) {
    // only_flagged != nullptr: this launch is the FALLBACK behind the multi-workgroup preparation (prep_*_kernel below): it runs for the
    // frames whose flag is set (a bucket above 4096 keys: a degenerate scene) and leaves the others alone
    if (only_flagged && only_flagged[blockIdx.x] == 0) return;
    constexpr int CH = 8192;                       // LDS chunk (64 KB)
    __shared__ unsigned long long chunk[CH];
    __shared__ float s_f[4][16];
    __shared__ int s_i[2][16];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PT* px = points + (long long)f * 3 * N;
    const PT* py = px + N;
    const PT* pz = px + 2 * (long long)N;
    const int* lab = labels + (long long)f * N;
    unsigned long long* keys = keys_all + (long long)f * 2 * P;      // two buffers of P keys: scattered / ranked (the bitonic network sorts the first in place)
    unsigned long long* keys2 = keys + P;
    Rec<PT>* out = packed + (long long)f * (N + 2 * CL);      // frame stride: N records + the padding of the two label blocks
    Box* boxes = boxes_all + (long long)f * NCMAX;

    // 1. ground-plane bounding box of the valid points, label counts (fixed-order reductions)
    float mnx = __builtin_inff(), mxx = -__builtin_inff(), mnz = __builtin_inff(), mxz = -__builtin_inff();
    int c1 = 0, c0 = 0;
    for (int n = tid; n < N; n += 1024) {
        const int l = lab[n];
        if (l == 0 || l == 1) {
            const float x = (float)px[n], z = (float)pz[n];
            mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mnz = fminf(mnz, z); mxz = fmaxf(mxz, z);
            c1 += l == 1; c0 += l == 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mnz = fminf(mnz, __shfl_xor(mnz, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
        c1 += __shfl_xor(c1, o); c0 += __shfl_xor(c0, o);
    }
    if (lane == 0) { s_f[0][wave] = mnx; s_f[1][wave] = mxx; s_f[2][wave] = mnz; s_f[3][wave] = mxz; s_i[0][wave] = c1; s_i[1][wave] = c0; }
    __syncthreads();
    mnx = s_f[0][0]; mxx = s_f[1][0]; mnz = s_f[2][0]; mxz = s_f[3][0];
    int n1 = s_i[0][0], n0 = s_i[1][0];
    for (int w = 1; w < 16; ++w) {
        mnx = fminf(mnx, s_f[0][w]); mxx = fmaxf(mxx, s_f[1][w]); mnz = fminf(mnz, s_f[2][w]); mxz = fmaxf(mxz, s_f[3][w]);
        n1 += s_i[0][w]; n0 += s_i[1][w];
    }
    const float ext = fmaxf(mxx - mnx, mxz - mnz);
    const float scale = (ext > 0.0f && ext < __builtin_inff()) ? 1023.0f / ext : 0.0f;

    // 2. sort keys: [label != 1][20-bit Hilbert index of the cell] << 32 | index (unique); points with other labels get ~0 (sort to the end)
    // (branch-free: the three loads are unconditional from a clamped index, so that the 4-way unrolled passes below keep them all in flight)
    auto key_of = [&](int n) -> unsigned long long {
        const int nc = min(n, N - 1);
        const int l = lab[nc];
        const float qx = fminf(fmaxf(((float)px[nc] - mnx) * scale, 0.0f), 1023.0f);
        const float qz = fminf(fmaxf(((float)pz[nc] - mnz) * scale, 0.0f), 1023.0f);
        const unsigned m = hilbert10((unsigned)qx, (unsigned)qz);
        const unsigned long long k = ((unsigned long long)((l == 1 ? 0u : 1u << 20) | m) << 32) | (unsigned)nc;
        return (n < N && (l == 0 || l == 1)) ? k : ~0ull;
    };
    // 3a. FAST sort (round 4): the keys are nearly uniform over their 21 significant bits, so a counting sort on the top 11 bits (label +
    //     a 32 x 32 grid of the Hilbert curve: 2048 buckets, LDS atomics) leaves buckets of ~10-25 keys; every key is then ranked inside its
    //     bucket by counting (one thread per key, the bucket read as broadcasts), which puts it at its final position in the second key buffer.
    //     A bucket above 4096 keys (a degenerate scene) sends the frame to the bitonic network of 3b.  The result is the SAME total order either
    //     way (tests compare the two paths): the preparation takes ~0.1 ms per frame instead of 0.53 (120 passes of a 32768-key network with a
    //     1024-thread barrier each).
    constexpr int NBK = 2048;
    int* cntv = reinterpret_cast<int*>(chunk);            // [NBK] bucket counts, then fill cursors
    int* basev = cntv + NBK;                              // [NBK + 1] exclusive offsets
    __shared__ int s_flag[4];                             // [1]: fall back to the bitonic network
    bool sorted = false;
    if (!force_bitonic) {
        for (int i = tid; i < NBK; i += 1024) cntv[i] = 0;
        if (tid < 4) s_flag[tid] = 0;
        __syncthreads();
        for (int n = tid; n < N; n += 4096) {
            unsigned long long k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k4[u] = key_of(n + 1024 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k4[u] != ~0ull) atomicAdd(&cntv[(int)(k4[u] >> 42)], 1);
        }
        __syncthreads();
        {   // exclusive scan of the 2048 counts: two per thread, wave scan, 16 wave totals
            const int a0 = cntv[2 * tid], a1 = cntv[2 * tid + 1];
            int v = a0 + a1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
            if (lane == 63) s_i[0][wave] = v;
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += s_i[0][w];
            const int excl = woff + v - (a0 + a1);
            basev[2 * tid] = excl; basev[2 * tid + 1] = excl + a0;
            if (tid == 1023) basev[NBK] = excl + a0 + a1;
        }
        __syncthreads();
        for (int i = tid; i < NBK; i += 1024) cntv[i] = 0;
        __syncthreads();
        for (int n = tid; n < N; n += 4096) {
            unsigned long long k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k4[u] = key_of(n + 1024 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k4[u] != ~0ull) { const int bk = (int)(k4[u] >> 42); keys[basev[bk] + atomicAdd(&cntv[bk], 1)] = k4[u]; }
        }
        __syncthreads();
        // rank every key inside its bucket by counting (#{j : key_j < key}; the keys are unique), one THREAD per key: neighbouring threads sit
        // in the same bucket and read the same addresses (broadcast).  The ranked keys go to the second key buffer.  A bucket above 4096 keys
        // (a degenerate scene: everything in a few cells) sends the frame to the bitonic network instead.
        for (int i = tid; i < NBK; i += 1024)
            if (basev[i + 1] - basev[i] > 4096) s_flag[1] = 1;
        __syncthreads();
        if (s_flag[1] == 0) {
            const int nv = basev[NBK];
            for (int p = tid; p < nv; p += 1024) {
                const unsigned long long k = keys[p];
                const int bk = (int)(k >> 42), b0 = basev[bk], c = basev[bk + 1] - b0;
                int rank = 0;
                for (int j = 0; j < c; j += 8) {            // eight loads in flight (clamped; the duplicates do not count)
                    unsigned long long q[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) q[u] = keys[b0 + min(j + u, c - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += (j + u < c && q[u] < k) ? 1 : 0;
                }
                keys2[b0 + rank] = k;
            }
        }
        sorted = s_flag[1] == 0;
        __syncthreads();
    }
    if (!sorted) {      // workgroup-uniform
    for (int n = tid; n < P; n += 1024) keys[n] = key_of(n);
    __syncthreads();

    // 3b. bitonic sort, ascending
    const int CHe = P < CH ? P : CH;
    const int nchunks = P / CHe;
    auto chunk_stages = [&](int base, int k, int j_first) {     // stages j_first, j_first/2, ..., 1 of merge size k on the LDS chunk
        for (int j = j_first; j > 0; j >>= 1) {
            for (int t = tid; t < CHe / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool up = ((base + i) & k) == 0;
                const unsigned long long a = chunk[i], b = chunk[i + j];
                if ((a > b) == up) { chunk[i] = b; chunk[i + j] = a; }
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
                const unsigned long long a = keys[i], b = keys[i + j];
                if ((a > b) == up) { keys[i] = b; keys[i + j] = a; }
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

    }
    // 4. records in sorted order, every label block CLUSTER-ALIGNED: label-1 block [0, n1), padded to nc1 * CL, label-0 block
    //    [nc1 * CL, nc1 * CL + n0), padded to (nc1 + nc0) * CL -- cluster c is always the records [c * CL, c * CL + CL), all of them
    //    addressable (the padding repeats the block's last record; the sweeps mask those lanes out)
    const int nv = n1 + n0;
    const int nc1 = (n1 + CL - 1) / CL, nc0 = (n0 + CL - 1) / CL;
    const unsigned long long* skeys = sorted ? keys2 : keys;
    const int nrec = (nc1 + nc0) * CL;
    for (int i0 = tid; i0 < nrec; i0 += 4096) {          // four records per trip: key loads, then the 16 gathers, all in flight together
        int nn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + 1024 * u, nrec - 1);
            const bool first = i < nc1 * CL;
            const int rank = first ? min(i, n1 - 1) : n1 + min(i - nc1 * CL, n0 - 1);       // position in the sorted key list
            nn[u] = (int)(unsigned)(skeys[rank] & 0xffffffffull);
        }
        Rec<PT> r4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { r4[u].x = px[nn[u]]; r4[u].y = py[nn[u]]; r4[u].z = pz[nn[u]]; r4[u].lab = lab[nn[u]]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 1024 * u < nrec) out[i0 + 1024 * u] = r4[u];
    }
    __syncthreads();

    // 5. bounding boxes: one wavefront REDUCES a cluster (its 64 records), the cluster's eight reduced values are parked in one lane, and after up
    //    to 64 clusters every lane finishes its own box (centre, half extents, radii: ~60 fp64 instructions that used to run on lane 0 once per
    //    cluster, serially)
    const int nct = nc1 + nc0;
    for (int t0 = 0; wave + 16 * t0 < nct; t0 += 64) {
        double mlo[3] = {0, 0, 0}, mhi[3] = {0, 0, 0};
        float mrxz = 0.0f, mr3 = 0.0f;
        bool mnan = false;
        for (int tt = 0; tt < 64 && wave + 16 * (t0 + tt) < nct; ++tt) {
            const int c = wave + 16 * (t0 + tt);
            const int start = c * CL;
            const int end = c < nc1 ? min(start + CL, n1) : min(start + CL, nc1 * CL + n0);
            const bool valid = start + lane < end;
            double x = 0, y = 0, z = 0;
            if (valid) { const Rec<PT> r = out[start + lane]; x = (double)r.x; y = (double)r.y; z = (double)r.z; }
            double lo[3] = {valid ? x : __builtin_inf(), valid ? y : __builtin_inf(), valid ? z : __builtin_inf()};
            double hi[3] = {valid ? x : -__builtin_inf(), valid ? y : -__builtin_inf(), valid ? z : -__builtin_inf()};
            double rxz = valid ? x * x + z * z : 0.0, r3 = valid ? x * x + y * y + z * z : 0.0;
            float frxz, fr3;
            if (sizeof(PT) == 4) {
                // fp32 records: the coordinates ARE floats, so the eight reductions run on fp32 values by DPP (no LDS round trips) and give the
                // very same box -- min / max of floats are exact, and sqrt / scaling / rounding to float are monotone, so the radii may be
                // rounded per lane before the maximum (round 3: 100 ds_bpermute with their latencies per cluster)
                float l0 = (float)lo[0], l1 = (float)lo[1], l2 = (float)lo[2], h0 = (float)hi[0], h1 = (float)hi[1], h2 = (float)hi[2];
                float pad0 = __builtin_inff(), pad1 = 0.0f;
                frxz = (float)(sqrt(rxz) * (1.0 + 1e-6)) + 1e-30f; fr3 = (float)(sqrt(r3) * (1.0 + 1e-6)) + 1e-30f;
                wave_min4_f32(l0, l1, l2, pad0);
                wave_max4_f32(h0, h1, h2, pad1);
                float pad2 = 0.0f, pad3 = 0.0f;
                wave_max4_f32(frxz, fr3, pad2, pad3);
                lo[0] = l0; lo[1] = l1; lo[2] = l2; hi[0] = h0; hi[1] = h1; hi[2] = h2;
            } else {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { rxz = fmax(rxz, __shfl_xor(rxz, o)); r3 = fmax(r3, __shfl_xor(r3, o)); }
                frxz = (float)(sqrt(rxz) * (1.0 + 1e-6)) + 1e-30f; fr3 = (float)(sqrt(r3) * (1.0 + 1e-6)) + 1e-30f;
            }
            // a NaN coordinate must poison the box (fmin/fmax drop it): such clusters are always classified per point
            const bool nan_any = __any(valid && !(x == x && y == y && z == z)) != 0;
            if (lane == tt) {
#pragma unroll
                for (int a = 0; a < 3; ++a) { mlo[a] = lo[a]; mhi[a] = hi[a]; }
                mrxz = frxz; mr3 = fr3; mnan = nan_any;
            }
        }
        const int c = wave + 16 * (t0 + lane);
        if (c < nct) {
            Box bx;
            float* bc = &bx.cx;
            float* bh = &bx.hx;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const double cd = 0.5 * (mlo[a] + mhi[a]);
                const float cf = (float)cd;
                // half extent measured from the ROUNDED centre, grown by 1e-6 (>> 2^-24): centre +- half extent contains lo and hi
                const double hd = fmax(mhi[a] - (double)cf, (double)cf - mlo[a]);
                bc[a] = cf;
                bh[a] = (float)(hd * (1.0 + 1e-6)) + 1e-30f;
            }
            if (mnan) bx.hx = __builtin_nanf("");
            bx.rxz = mrxz;
            bx.r3 = mr3;
            boxes[c] = bx;
        }
    }
    if (tid == 0) {
        counts[4 * f] = n1; counts[4 * f + 1] = n0; counts[4 * f + 2] = nc1; counts[4 * f + 3] = nc0;
        // the frame's NORMALISED frustum-plane coefficients in fp32 (see Pre32 below): they depend on the camera only, so every sweep of every
        // hypothesis reads them back with scalar loads instead of re-deriving them (six conversions, four divisions, eight products per wave and sweep)
        const double* Kf = Kmat + (long long)f * 9;
        float* cf = camf_all + (long long)f * 8;
        const float fx = (float)Kf[0], cx = (float)Kf[2], wcx = (float)((W - 1.0) - Kf[2]), fy = (float)Kf[4], cy = (float)Kf[5], hcy = (float)((H - 1.0) - Kf[5]);
        const float iL = 1.0f / (fabsf(fx) + fabsf(cx)), iR = 1.0f / (fabsf(fx) + fabsf(wcx));
        const float iT = 1.0f / (fabsf(fy) + fabsf(cy)), iB = 1.0f / (fabsf(fy) + fabsf(hcy));
        cf[0] = fx * iL; cf[1] = cx * iL; cf[2] = fx * iR; cf[3] = wcx * iR;
        cf[4] = fy * iT; cf[5] = cy * iT; cf[6] = fy * iB; cf[7] = hcy * iB;
    }
}

// ----------------------------------------------------------------------------------------------
// MULTI-WORKGROUP frame preparation (round 6).  prepare_kernel runs one 1024-thread workgroup per frame: 32 of 256 compute units for 0.26 ms at
// the head of every solve.  The same five steps as five launches, G workgroups per frame each (same keys, same total order, same records and
// boxes, bit for bit -- tests compare the paths):
//   prep_bounds_kernel   partial ground-plane bounds + label counts per slice; clears the frame's flag
//   prep_hist_kernel     keys of the slice (stored), bucket counts of the slice (LDS atomics) -> [frame][slice][bucket]
//   prep_scatter_kernel  bucket offsets = exclusive scan of the summed counts (every workgroup, into LDS); a slice's keys go to
//                        offset[bucket] + (counts of the slices before it) + an LDS cursor: no global atomics
//   prep_rank_kernel     one thread per key: rank inside its bucket by counting -> sorted keys
//   prep_records_kernel  one wavefront per cluster: gathers its 64 records, writes them and reduces the box FROM ITS REGISTERS
// A frame with a bucket above 4096 keys is flagged by the scatter step, skipped by the last two and prepared by prepare_kernel (launched
// behind them with `only_flagged`).
constexpr int PREP_NBK = 2048;        // buckets: the top 11 key bits (label + 32 x 32 grid of the Hilbert curve)
constexpr int PREP_G = 8;             // workgroups per frame of the slice kernels (PREP_NBK % PREP_G == 0)
constexpr int PREP_CPW = 4;           // clusters per wavefront of prep_records_kernel (their keys, then their gathers, in flight together; parked in lanes 0..3, finished lane-parallel)
struct PrepWs { float* partial; int* cntg; int* bases; int* flag; };      // [F][G][8] | [F][G][NBK] bucket counts per slice | [F][NBK + 1] | [F]

// bounds, scale and label counts of a frame from its G partial results (same min / max / sums whatever the slicing: exact operations)
struct PrepFrame { float mnx, mnz, scale; int n1, n0; };
__device__ __forceinline__ PrepFrame prep_frame(const float* __restrict__ partial, int f) {
    float mnx = __builtin_inff(), mxx = -__builtin_inff(), mnz = __builtin_inff(), mxz = -__builtin_inff();
    int n1 = 0, n0 = 0;
    for (int g = 0; g < PREP_G; ++g) {
        const float* q = partial + ((long long)f * PREP_G + g) * 8;
        mnx = fminf(mnx, q[0]); mxx = fmaxf(mxx, q[1]); mnz = fminf(mnz, q[2]); mxz = fmaxf(mxz, q[3]);
        n1 += __builtin_bit_cast(int, q[4]); n0 += __builtin_bit_cast(int, q[5]);
    }
    const float ext = fmaxf(mxx - mnx, mxz - mnz);
    PrepFrame r;
    r.mnx = mnx; r.mnz = mnz; r.scale = (ext > 0.0f && ext < __builtin_inff()) ? 1023.0f / ext : 0.0f; r.n1 = n1; r.n0 = n0;
    return r;
}
__device__ __forceinline__ void prep_slice(int N, int g, int& lo, int& hi) {
    const int S = ((N + PREP_G - 1) / PREP_G + 255) & ~255;
    lo = min(g * S, N); hi = min(lo + S, N);
}

template <typename PT>
__global__ __launch_bounds__(256) void prep_bounds_kernel(const PT* __restrict__ points, const int* __restrict__ labels, int N, PrepWs w) {
    __shared__ float s_f[4][4];
    __shared__ int s_i[2][4];
    const int f = blockIdx.x / PREP_G, g = blockIdx.x % PREP_G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PT* px = points + (long long)f * 3 * N;
    const PT* pz = px + 2 * (long long)N;
    const int* lab = labels + (long long)f * N;
    if (g == 0 && tid == 0) w.flag[f] = 0;
    int lo, hi;
    prep_slice(N, g, lo, hi);
    float mnx = __builtin_inff(), mxx = -__builtin_inff(), mnz = __builtin_inff(), mxz = -__builtin_inff();
    int c1 = 0, c0 = 0;
    for (int n = lo + tid; n < hi; n += 256) {
        const int l = lab[n];
        if (l == 0 || l == 1) {
            const float x = (float)px[n], z = (float)pz[n];
            mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mnz = fminf(mnz, z); mxz = fmaxf(mxz, z);
            c1 += l == 1; c0 += l == 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mnz = fminf(mnz, __shfl_xor(mnz, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
        c1 += __shfl_xor(c1, o); c0 += __shfl_xor(c0, o);
    }
    if (lane == 0) { s_f[0][wave] = mnx; s_f[1][wave] = mxx; s_f[2][wave] = mnz; s_f[3][wave] = mxz; s_i[0][wave] = c1; s_i[1][wave] = c0; }
    __syncthreads();
    if (tid == 0) {
        for (int v = 1; v < 4; ++v) {
            mnx = fminf(mnx, s_f[0][v]); mxx = fmaxf(mxx, s_f[1][v]); mnz = fminf(mnz, s_f[2][v]); mxz = fmaxf(mxz, s_f[3][v]);
            c1 += s_i[0][v]; c0 += s_i[1][v];
        }
        float* q = w.partial + ((long long)f * PREP_G + g) * 8;
        q[0] = mnx; q[1] = mxx; q[2] = mnz; q[3] = mxz; q[4] = __builtin_bit_cast(float, c1); q[5] = __builtin_bit_cast(float, c0);
    }
}

// the sort key of point n (prepare_kernel's key_of: the same expressions, the same bits)
template <typename PT>
__device__ __forceinline__ unsigned long long prep_key(const PT* px, const PT* pz, const int* lab, int n, const PrepFrame& fr) {
    const int l = lab[n];
    const float qx = fminf(fmaxf(((float)px[n] - fr.mnx) * fr.scale, 0.0f), 1023.0f);
    const float qz = fminf(fmaxf(((float)pz[n] - fr.mnz) * fr.scale, 0.0f), 1023.0f);
    const unsigned m = hilbert10((unsigned)qx, (unsigned)qz);
    const unsigned long long k = ((unsigned long long)((l == 1 ? 0u : 1u << 20) | m) << 32) | (unsigned)n;
    return (l == 0 || l == 1) ? k : ~0ull;
}

template <typename PT>
__global__ __launch_bounds__(256) void prep_hist_kernel(const PT* __restrict__ points, const int* __restrict__ labels, int N, int P,
                                                        unsigned long long* __restrict__ keys_all, PrepWs w) {
    __shared__ int cnt[PREP_NBK];
    const int f = blockIdx.x / PREP_G, g = blockIdx.x % PREP_G, tid = threadIdx.x;
    const PT* px = points + (long long)f * 3 * N;
    const PT* pz = px + 2 * (long long)N;
    const int* lab = labels + (long long)f * N;
    unsigned long long* raw = keys_all + (long long)f * 2 * P + P;         // the second key buffer holds the unsorted keys until the ranking
    const PrepFrame fr = prep_frame(w.partial, f);
    for (int i = tid; i < PREP_NBK; i += 256) cnt[i] = 0;
    __syncthreads();
    int lo, hi;
    prep_slice(N, g, lo, hi);
    for (int n = lo + tid; n < hi; n += 256) {
        const unsigned long long k = prep_key(px, pz, lab, n, fr);
        raw[n] = k;
        if (k != ~0ull) atomicAdd(&cnt[(int)(k >> 42)], 1);
    }
    __syncthreads();
    for (int i = tid; i < PREP_NBK; i += 256) w.cntg[((long long)f * PREP_G + g) * PREP_NBK + i] = cnt[i];
}

__global__ __launch_bounds__(1024) void prep_scatter_kernel(int N, int P, unsigned long long* __restrict__ keys_all, PrepWs w) {
    __shared__ int cur[PREP_NBK];           // where the slice's next key of a bucket goes
    __shared__ int s_w[16];
    const int f = blockIdx.x / PREP_G, g = blockIdx.x % PREP_G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long* keys = keys_all + (long long)f * 2 * P;
    const unsigned long long* raw = keys + P;
    {   // exclusive scan of the 2048 summed counts: two buckets per thread, wave scan, 16 wave totals (prepare_kernel's scan)
        const int* cg = w.cntg + (long long)f * PREP_G * PREP_NBK;
        int a0 = 0, a1 = 0, before0 = 0, before1 = 0;       // totals of the two buckets; the keys of the slices before this one
#pragma unroll
        for (int q = 0; q < PREP_G; ++q) {
            const int c0 = cg[q * PREP_NBK + 2 * tid], c1 = cg[q * PREP_NBK + 2 * tid + 1];
            a0 += c0; a1 += c1;
            before0 += q < g ? c0 : 0; before1 += q < g ? c1 : 0;
        }
        int v = a0 + a1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o); if (lane >= o) v += u; }
        if (lane == 63) s_w[wave] = v;
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < wave; ++q) woff += s_w[q];
        const int excl = woff + v - (a0 + a1);
        cur[2 * tid] = excl + before0; cur[2 * tid + 1] = excl + a0 + before1;
        if (g == 0) {
            int* bases = w.bases + (long long)f * (PREP_NBK + 1);
            bases[2 * tid] = excl; bases[2 * tid + 1] = excl + a0;
            if (tid == 1023) bases[PREP_NBK] = excl + a0 + a1;
        }
        if (a0 > 4096 || a1 > 4096) w.flag[f] = 1;            // a degenerate scene: the frame goes to prepare_kernel's bitonic network
    }
    __syncthreads();
    int lo, hi;
    prep_slice(N, g, lo, hi);
    for (int n = lo + tid; n < hi; n += 1024) {
        const unsigned long long k = raw[n];
        if (k != ~0ull) keys[atomicAdd(&cur[(int)(k >> 42)], 1)] = k;
    }
}

__global__ __launch_bounds__(256) void prep_rank_kernel(int P, unsigned long long* __restrict__ keys_all, PrepWs w) {
    const int f = blockIdx.y;
    if (w.flag[f]) return;
    const int* bases = w.bases + (long long)f * (PREP_NBK + 1);
    const unsigned long long* keys = keys_all + (long long)f * 2 * P;
    unsigned long long* keys2 = keys_all + (long long)f * 2 * P + P;
    const int nv = bases[PREP_NBK];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= nv) return;
    // rank inside the bucket by counting (#{j : key_j < key}; the keys are unique): neighbouring threads sit in the same bucket and read the
    // same addresses (broadcast)
    const unsigned long long k = keys[p];
    const int bk = (int)(k >> 42), b0 = bases[bk], c = bases[bk + 1] - b0;
    int rank = 0;
    for (int j = 0; j < c; j += 8) {            // eight loads in flight (clamped; the duplicates do not count)
        unsigned long long q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = keys[b0 + min(j + u, c - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (j + u < c && q[u] < k) ? 1 : 0;
    }
    keys2[b0 + rank] = k;
}

// Records in sorted order (every label block cluster-aligned and padded with copies of its last record) and the cluster boxes: one wavefront
// per cluster holds the cluster's 64 records in its lanes -- written out and reduced from registers; PREP_CPW clusters per wavefront, their
// reduced values parked in lanes 0 .. PREP_CPW - 1, which then finish their own boxes (prepare_kernel steps 4 and 5, the same operations).
template <typename PT>
__global__ __launch_bounds__(256) void prep_records_kernel(const PT* __restrict__ points, const int* __restrict__ labels, int N, int P, int NCMAX,
                                                           const unsigned long long* __restrict__ keys_all, Rec<PT>* __restrict__ packed,
                                                           Box* __restrict__ boxes_all, int* __restrict__ counts, const double* __restrict__ Kmat,
                                                           double H, double W, float* __restrict__ camf_all, PrepWs w) {
    const int f = blockIdx.y;
    if (w.flag[f]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PT* px = points + (long long)f * 3 * N;
    const PT* py = px + N;
    const PT* pz = px + 2 * (long long)N;
    const int* lab = labels + (long long)f * N;
    const unsigned long long* skeys = keys_all + (long long)f * 2 * P + P;
    Rec<PT>* out = packed + (long long)f * (N + 2 * CL);
    Box* boxes = boxes_all + (long long)f * NCMAX;
    const PrepFrame fr = prep_frame(w.partial, f);
    const int n1 = fr.n1, n0 = fr.n0;
    const int nc1 = (n1 + CL - 1) / CL, nc0 = (n0 + CL - 1) / CL, nct = nc1 + nc0;
    if (blockIdx.x == 0 && tid == 0) {
        counts[4 * f] = n1; counts[4 * f + 1] = n0; counts[4 * f + 2] = nc1; counts[4 * f + 3] = nc0;
        const double* Kf = Kmat + (long long)f * 9;
        float* cf = camf_all + (long long)f * 8;
        const float fx = (float)Kf[0], cx = (float)Kf[2], wcx = (float)((W - 1.0) - Kf[2]), fy = (float)Kf[4], cy = (float)Kf[5], hcy = (float)((H - 1.0) - Kf[5]);
        const float iL = 1.0f / (fabsf(fx) + fabsf(cx)), iR = 1.0f / (fabsf(fx) + fabsf(wcx));
        const float iT = 1.0f / (fabsf(fy) + fabsf(cy)), iB = 1.0f / (fabsf(fy) + fabsf(hcy));
        cf[0] = fx * iL; cf[1] = cx * iL; cf[2] = fx * iR; cf[3] = wcx * iR;
        cf[4] = fy * iT; cf[5] = cy * iT; cf[6] = fy * iB; cf[7] = hcy * iB;
    }
    const int c_first = (blockIdx.x * 4 + wave) * PREP_CPW;
    if (c_first >= nct) return;              // wave-uniform
    double mlo[3] = {0, 0, 0}, mhi[3] = {0, 0, 0};
    float mrxz = 0.0f, mr3 = 0.0f;
    bool mnan = false;
    // the PREP_CPW clusters' keys, then their gathers, are requested together (a cluster alone is a chain of two dependent round trips)
    int nn_all[PREP_CPW];
#pragma unroll
    for (int tt = 0; tt < PREP_CPW; ++tt) {
        const int c = min(c_first + tt, nct - 1);
        const int i = c * CL + lane;
        const int rank = c < nc1 ? min(i, n1 - 1) : n1 + min(i - nc1 * CL, n0 - 1);       // position in the sorted key list (padding: the block's last record)
        nn_all[tt] = (int)(unsigned)(skeys[rank] & 0xffffffffull);
    }
    Rec<PT> r_all[PREP_CPW];
#pragma unroll
    for (int tt = 0; tt < PREP_CPW; ++tt) { r_all[tt].x = px[nn_all[tt]]; r_all[tt].y = py[nn_all[tt]]; r_all[tt].z = pz[nn_all[tt]]; r_all[tt].lab = lab[nn_all[tt]]; }
#pragma unroll
    for (int tt = 0; tt < PREP_CPW; ++tt) {
        if (c_first + tt >= nct) break;          // wave-uniform
        const int c = c_first + tt;
        const int i = c * CL + lane;
        const bool first = c < nc1;
        const Rec<PT> r = r_all[tt];
        out[i] = r;
        const int end = first ? n1 : nc1 * CL + n0;
        const bool valid = i < end;
        const double x = valid ? (double)r.x : 0.0, y = valid ? (double)r.y : 0.0, z = valid ? (double)r.z : 0.0;
        double lo[3] = {valid ? x : __builtin_inf(), valid ? y : __builtin_inf(), valid ? z : __builtin_inf()};
        double hi[3] = {valid ? x : -__builtin_inf(), valid ? y : -__builtin_inf(), valid ? z : -__builtin_inf()};
        double rxz = valid ? x * x + z * z : 0.0, r3 = valid ? x * x + y * y + z * z : 0.0;
        float frxz, fr3;
        if (sizeof(PT) == 4) {
            float l0 = (float)lo[0], l1 = (float)lo[1], l2 = (float)lo[2], h0 = (float)hi[0], h1 = (float)hi[1], h2 = (float)hi[2];
            float pad0 = __builtin_inff(), pad1 = 0.0f;
            frxz = (float)(sqrt(rxz) * (1.0 + 1e-6)) + 1e-30f; fr3 = (float)(sqrt(r3) * (1.0 + 1e-6)) + 1e-30f;
            wave_min4_f32(l0, l1, l2, pad0);
            wave_max4_f32(h0, h1, h2, pad1);
            float pad2 = 0.0f, pad3 = 0.0f;
            wave_max4_f32(frxz, fr3, pad2, pad3);
            lo[0] = l0; lo[1] = l1; lo[2] = l2; hi[0] = h0; hi[1] = h1; hi[2] = h2;
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { rxz = fmax(rxz, __shfl_xor(rxz, o)); r3 = fmax(r3, __shfl_xor(r3, o)); }
            frxz = (float)(sqrt(rxz) * (1.0 + 1e-6)) + 1e-30f; fr3 = (float)(sqrt(r3) * (1.0 + 1e-6)) + 1e-30f;
        }
        const bool nan_any = __any(valid && !(x == x && y == y && z == z)) != 0;
        if (lane == tt) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { mlo[a] = lo[a]; mhi[a] = hi[a]; }
            mrxz = frxz; mr3 = fr3; mnan = nan_any;
        }
    }
    const int c = c_first + lane;
    if (lane < PREP_CPW && c < nct) {
        Box bx;
        float* bc = &bx.cx;
        float* bh = &bx.hx;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double cd = 0.5 * (mlo[a] + mhi[a]);
            const float cf = (float)cd;
            const double hd = fmax(mhi[a] - (double)cf, (double)cf - mlo[a]);
            bc[a] = cf;
            bh[a] = (float)(hd * (1.0 + 1e-6)) + 1e-30f;
        }
        if (mnan) bx.hx = __builtin_nanf("");
        bx.rxz = mrxz;
        bx.r3 = mr3;
        boxes[c] = bx;
    }
}

// ----------------------------------------------------------------------------------------------
// One sweep over the frame's records by the 4 waves of a hypothesis' workgroup.
//   phase A (every record, cheap): rotate, translate, one fp64 reciprocal, project, and decide whether the
//     point is ACTIVE (non-zero residual or Jacobian row).  Near a solution ~5 % of the points are.
//   compaction: active record ids go to a per-wave LDS queue (ballot + popcount positions); whenever 64 are
//     queued the wave evaluates them DENSELY (phase B) -- without this, divergence makes every lane pay the
//     log1p + 4x4 normal-equation update for every record, because some lane of 64 is always active.
//   phase B (active records only): residual rows, analytic Jacobian rows, Cauchy corrector, cost and J^T J /
//     J^T r accumulation in registers.
//   reduction: xor-butterfly inside each wave, then the 4 wave partials are combined through LDS in a FIXED
//     order by every thread, so all 256 threads hold bit-identical sums and the LM control flow that
//     follows is workgroup-uniform.
#ifdef DI2P_SOLVER_LMPROF
constexpr int PROF_WORDS = 36;      // variant build: + 8 words of LM sub-stage clocks
#else
constexpr int PROF_WORDS = 28;
#endif      // int64 words per hypothesis of the diagnostics buffer (library version >= 6; 20 in versions 4-5)
constexpr int QCAP = DI2P_SOLVER_PF * 64 + 128;        // per-wave queue capacity (ids); phase B drains it when a batch of the cluster walk (PF clusters) may not fit

constexpr int BOXTEST_WORDS = 16;   // sizeof(BoxAbs) / 4 (the table is fetched as 16-byte LDS reads)
template <int NP, int WPH>   // WPH = waves per hypothesis (workgroup = WPH*64 threads)
struct SweepShared {
    double red[WPH][Tri<NP>::N + NP + 2];  // per wave: {cost product mantissa, g[NP], A[tri], bad flag}
    int red_e[WPH];                        // per wave: exponent of the cost product
    double comb[Tri<NP>::N + NP + 2];      // the WPH partials combined in a fixed order (lane i of wave 0 sums value i)
    int queue[WPH][QCAP];
    // Per-lane running sums {cost mantissa, g[NP], A[tri]} and the cost exponent.  They are only touched by phase B, so they live
    // HERE between drains (one conflict-free 8-byte LDS access per value and lane at the start and at the end of a drain) instead of
    // occupying 2 * (1 + NP + tri) + 1 VGPRs through the cluster walk -- with them in registers the kernel spilled at 3 waves per SIMD.
    double acc[WPH][1 + NP + Tri<NP>::N][64];
    int acc_e[WPH][64];
    alignas(16) float btest[WPH][BOXTEST_WORDS];   // the wave's box-test table of the current sweep (wave-uniform, re-read per cluster round)
    alignas(16) double ring[8][NP];                // iterates of the last RING sweeps (slot = sweep number % RING), see the classification cache
    double rot_cs[2];                              // 2-D solver: cos / sin of the iterate about to be swept, computed ONCE per sweep by the LM lane
};

// v_rcp_f64 + two Newton steps: <= 1 ulp for finite non-zero inputs; 0 / inf / NaN give inf / 0 / NaN-like values that the
// callers' finiteness tests catch.  ~5 instructions instead of the ~15 of the IEEE division sequence.
__device__ __forceinline__ double fast_rcp(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// Division and square root of the LM update's linear solve: reciprocal / reciprocal square root + Newton corrections (<= 1 ulp) instead of
// the IEEE sequences (~25 instructions each with their scaling and fix-up steps) -- 16 of them sit on the critical lane per LM iteration
// (finish + begin 5.8 k -> 4.9 k cycles per sweep).  The HIP-vs-oracle agreement is unchanged hypothesis for hypothesis, iteration
// counts included (tools/solver_oracle_agreement.py, tools/ab_fastdiv.sh); -DDI2P_SOLVER_IEEEDIV restores the IEEE forms.
#ifndef DI2P_SOLVER_IEEEDIV
#define DI2P_SOLVER_FASTDIV 1
#endif
__device__ __forceinline__ double lm_div(double a, double b) {
#ifdef DI2P_SOLVER_FASTDIV
    const double r = fast_rcp(b), q = a * r;
    return fma(fma(-b, q, a), r, q);
#else
    return a / b;
#endif
}
__device__ __forceinline__ double lm_sqrt(double a) {
#ifdef DI2P_SOLVER_FASTDIV
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double v = fma(fma(-g, g, a), h, g);
    return a == 0.0 ? 0.0 : v;              // rsq(0) = inf would turn sqrt(0) into NaN: a zero step / iterate norm must satisfy the parameter tolerance
#else
    return sqrt(a);
#endif
}

template <int NP, typename PT, bool FAST_RCP = false>
__device__ __forceinline__ void project(const Rec<PT>& rc, const Rot<NP>& rot, double tx, double ty, double tz, const Cam& k,
                                        double& X, double& Y, double& Z, double& qx, double& qz, double& p0, double& p1,
                                        double& p2, double& iz, double& pix_x, double& pix_y) {
    X = (double)rc.x; Y = (double)rc.y; Z = (double)rc.z;
    double qy;
    if (NP == 4) {
        qx = rot.R[0] * X + rot.R[2] * Z; qy = Y; qz = rot.R[6] * X + rot.R[8] * Z;
    } else {
        qx = rot.R[0] * X + rot.R[1] * Y + rot.R[2] * Z;
        qy = rot.R[3] * X + rot.R[4] * Y + rot.R[5] * Z;
        qz = rot.R[6] * X + rot.R[7] * Y + rot.R[8] * Z;
    }
    p0 = qx + tx; p1 = qy + ty; p2 = qz + tz;
    iz = FAST_RCP ? fast_rcp(p2) : 1.0 / p2;          // the one reciprocal per record
    pix_x = p0 * k.fx * iz + k.cx;
    pix_y = p1 * k.fy * iz + k.cy;
}

// ln(v) for finite v > 0 without libm: v = m * 2^e with m in [sqrt(1/2), sqrt(2)), ln(m) = 2 atanh(z), z = (m - 1) / (m + 1), |z| <= 0.1716,
// odd series to z^23 (truncation < 2e-19), one reciprocal.  ~30 instructions instead of the ~600 cycles-deep libm log; absolute error
// ~1e-16 * (1 + |e|), far below the 1e-6 relative function tolerance it feeds.  Used once per sweep, on the lane that advances the LM state.
__device__ __forceinline__ double ln_pos(double v) {
    int e = __builtin_amdgcn_frexp_exp(v);
    double m = __builtin_amdgcn_frexp_mant(v);          // [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double z = (m - 1.0) * fast_rcp(m + 1.0), w = z * z;
    double p = 1.0 / 23.0;
    p = fma(p, w, 1.0 / 21.0); p = fma(p, w, 1.0 / 19.0); p = fma(p, w, 1.0 / 17.0); p = fma(p, w, 1.0 / 15.0); p = fma(p, w, 1.0 / 13.0);
    p = fma(p, w, 1.0 / 11.0); p = fma(p, w, 1.0 / 9.0); p = fma(p, w, 1.0 / 7.0); p = fma(p, w, 1.0 / 5.0); p = fma(p, w, 1.0 / 3.0);
    p = fma(p, w, 1.0);
    return fma((double)e, 0.69314718055994530942, 2.0 * z * p);
}

// sum_i log(1 + s_i) of a lane, kept as the PRODUCT prod_i (1 + s_i) = m * 2^e with m in [0.5, 1): one multiply and two
// frexp pairs per block instead of a ~70-instruction fp64 log, one log per lane per sweep at the end; the relative error
// of the product (n * 2^-53) becomes an ABSOLUTE error of the sum, i.e. it is more accurate than adding rounded logs.
struct LogProd {
    double m;
    int e;
    __device__ __forceinline__ void init() { m = 0.5; e = 1; }
    __device__ __forceinline__ void mul(double v) {       // v >= 1 finite (non-finite v poisons m; the caller flags `bad`)
        e += __builtin_amdgcn_frexp_exp(v);
        m *= __builtin_amdgcn_frexp_mant(v);              // in [0.25, 1)
        e += __builtin_amdgcn_frexp_exp(m);
        m = __builtin_amdgcn_frexp_mant(m);
    }
    __device__ __forceinline__ double log_value() const { return log(m) + (double)e * 0.69314718055994530942; }
};

// compile-time list of parameter indices (the non-zero columns of a Jacobian row)
template <int... I> struct ParamSet {
    static constexpr int N = sizeof...(I);
    __device__ static constexpr int at(int u) { constexpr int v[] = {I...}; return v[u]; }
};

// MODE 2: cost, gradient J^T r and normal equations J^T J; MODE 1: cost and gradient only (line-search trials beyond the
// first: Ceres' CUBIC interpolation needs the directional derivative at every trial, 92 % of them are rejected, the accepted
// ones are confirmed by a MODE-2 sweep); MODE 0: cost only.  The residual rows, s, the cost product and the gradient terms
// are computed by the SAME operations in the same order in every mode, so shared outputs are bit-identical across modes.
template <int NP, typename PT, int LAB, int MODE>
__device__ __forceinline__ void eval_active(const Rec<PT>& rc, const Rot<NP>& rot, const double* x, const Cam& k, LogProd& cost,
                                            double* lg, double* lA, bool& bad) {
    constexpr int TOFF = NP == 4 ? 1 : 3;
    double X, Y, Z, qx, qz, p0, p1, p2, iz, pix_x, pix_y;
    project<NP, PT, true>(rc, rot, x[TOFF], x[TOFF + 1], x[TOFF + 2], k, X, Y, Z, qx, qz, p0, p1, p2, iz, pix_x, pix_y);
    const double hw = k.W1 * 0.5, hh = k.H1 * 0.5;
    // residual rows as (value, d/dpix_x, d/dpix_y, d/dp2 direct) -- at most 3 rows
    double rv[3], sx[3], sy[3], sz[3];
    int nr;
    if (LAB == 1) {
        nr = 3;
        const double a0 = -pix_x, b0 = pix_x - k.W1;
        rv[0] = (a0 < 0.0 ? 0.0 : a0) + (b0 < 0.0 ? 0.0 : b0);
        sx[0] = (a0 < 0.0 ? 0.0 : -1.0) + (b0 < 0.0 ? 0.0 : 1.0); sy[0] = 0.0; sz[0] = 0.0;
        const double a1 = -pix_y, b1 = pix_y - k.H1;
        rv[1] = (a1 < 0.0 ? 0.0 : a1) + (b1 < 0.0 ? 0.0 : b1);
        sy[1] = (a1 < 0.0 ? 0.0 : -1.0) + (b1 < 0.0 ? 0.0 : 1.0); sx[1] = 0.0; sz[1] = 0.0;
        const double a2 = -p2;
        rv[2] = (a2 < 0.0 ? 0.0 : a2) * 100.0;
        sz[2] = a2 < 0.0 ? 0.0 : -100.0; sx[2] = 0.0; sy[2] = 0.0;
    } else {
        nr = 1;
        const double ex = pix_x - hw, ey = pix_y - hh;
        const double dx = hw - fabs(ex), dy = hh - fabs(ey);
        rv[0] = dx + dy;                       // only active outside-points get here
        sx[0] = ex < 0.0 ? 1.0 : -1.0;
        sy[0] = ey < 0.0 ? 1.0 : -1.0;
        sz[0] = 0.0;
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < (LAB == 1 ? 3 : 1); ++i) s += rv[i] * rv[i];
    if (!isfinite(s)) bad = true;
    const double s1 = 1.0 + s;
    cost.mul(s1);                                // rho(s) = log(1+s), accumulated as a product
    if (MODE == 0) return;
    const double rho1 = fast_rcp(s1);
    const double ax = k.fx * iz, bx = -k.fx * p0 * iz * iz;   // dpix_x = ax*dp0 + bx*dp2
    const double ay = k.fy * iz, by = -k.fy * p1 * iz * iz;   // dpix_y = ay*dp1 + by*dp2
    if (NP == 4) {
        // 2-D solver: the parameter derivatives of p = Ry(theta) X + t are  dp0 = (d00, 1, 0, 0), dp1 = (0, 0, 1, 0), dp2 = (d20, 0, 0, 1), so every
        // Jacobian row has STRUCTURAL zeros: row "pix_x" lives on parameters {0, 1, 3}, row "pix_y" on {0, 2, 3}, row "p2" on {0, 3}.  Written
        // out, a label-1 block costs 15 instead of 30 normal-equation updates (the generic form multiplies by 0.0 and 1.0: IEEE rules forbid the
        // compiler to drop x * 0.0).  Same values as the generic form whenever the terms are finite (x * 1.0 = x, fma(a, 0.0, y) = y, s + (+-0) = s).
        double d00 = qz, d20 = -qx;                              // d/dtheta of Ry(theta) X
        if (!(x[0] * x[0] > DBL_EPSILON)) { d00 = Z; d20 = -X; }   // first-order branch
        auto add_row = [&](auto idx, const double* J, double r) {
            constexpr int NI = decltype(idx)::N;
            const double wr = rho1 * r;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int a = decltype(idx)::at(u);
                lg[a] += wr * J[a];
                if (MODE == 2) {
                    const double wa = rho1 * J[a];
#pragma unroll
                    for (int v = 0; v <= u; ++v) { const int b = decltype(idx)::at(v); lA[a * (a + 1) / 2 + b] += wa * J[b]; }
                }
            }
        };
        if (LAB == 1) {
            if (!(sx[0] == 0.0)) {
                double J[4];
                J[0] = sx[0] * (ax * d00 + bx * d20); J[1] = sx[0] * ax; J[2] = 0.0; J[3] = sx[0] * bx;
                add_row(ParamSet<0, 1, 3>(), J, rv[0]);
            }
            if (!(sy[1] == 0.0)) {
                double J[4];
                J[0] = sy[1] * (by * d20); J[1] = 0.0; J[2] = sy[1] * ay; J[3] = sy[1] * by;
                add_row(ParamSet<0, 2, 3>(), J, rv[1]);
            }
            if (!(sz[2] == 0.0)) {
                double J[4];
                J[0] = sz[2] * d20; J[1] = 0.0; J[2] = 0.0; J[3] = sz[2];
                add_row(ParamSet<0, 3>(), J, rv[2]);
            }
        } else {
            double J[4];
            J[0] = sx[0] * (ax * d00 + bx * d20) + sy[0] * (by * d20); J[1] = sx[0] * ax; J[2] = sy[0] * ay; J[3] = sx[0] * bx + sy[0] * by;
            add_row(ParamSet<0, 1, 2, 3>(), J, rv[0]);
        }
        return;
    }
    double dp0[NP], dp1[NP], dp2[NP];
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            dp0[i] = rot.dR[i][0] * X + rot.dR[i][1] * Y + rot.dR[i][2] * Z;
            dp1[i] = rot.dR[i][3] * X + rot.dR[i][4] * Y + rot.dR[i][5] * Z;
            dp2[i] = rot.dR[i][6] * X + rot.dR[i][7] * Y + rot.dR[i][8] * Z;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { dp0[TOFF + i] = i == 0 ? 1.0 : 0.0; dp1[TOFF + i] = i == 1 ? 1.0 : 0.0; dp2[TOFF + i] = i == 2 ? 1.0 : 0.0; }
#pragma unroll
    for (int i = 0; i < (LAB == 1 ? 3 : 1); ++i) {
        if (LAB == 1 && sx[i] == 0.0 && sy[i] == 0.0 && sz[i] == 0.0) continue;
        double J[NP];
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            if (LAB == 1) {          // row 0 depends on pix_x only, row 1 on pix_y, row 2 on p2
                J[a] = i == 0 ? sx[0] * (ax * dp0[a] + bx * dp2[a]) : (i == 1 ? sy[1] * (ay * dp1[a] + by * dp2[a]) : sz[2] * dp2[a]);
            } else {
                J[a] = sx[0] * (ax * dp0[a] + bx * dp2[a]) + sy[0] * (ay * dp1[a] + by * dp2[a]);
            }
        }
        const double wr = rho1 * rv[i];
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            lg[a] += wr * J[a];
            if (MODE == 2) {
                const double wa = rho1 * J[a];
#pragma unroll
                for (int b = 0; b <= a; ++b) lA[a * (a + 1) / 2 + b] += wa * J[b];
            }
        }
    }
}

// The five planes of the camera frustum in camera coordinates, f_i(p) = n_i . p, each NORMALISED to |n_i|_1 = 1:
//   pix_x > 0  <=> f_L = (fx*p0 + cx*p2) / (|fx|+|cx|) > 0,  pix_x < W1 <=> f_R = (-fx*p0 + (W1-cx)*p2) / (|fx|+|W1-cx|) > 0   (for p2 > 0; for
//   pix_y > 0  <=> f_T = (fy*p1 + cy*p2) / (|fy|+|cy|) > 0,  pix_y < H1 <=> f_B = (-fy*p1 + (H1-cy)*p2) / (|fy|+|H1-cy|) > 0    p2 < 0 the signs flip
//   p2 > 0     <=> f_Z = p2 > 0.                                                                                              together),
// With unit 1-norms every |f_i(p)| <= |p|_inf, a displacement d of the point changes every f_i by at most |d|_inf, and ONE relative margin
// serves all five planes (min_i (f_i - m S) = min_i f_i - m S: the per-point tests below need one subtraction instead of five fused
// multiply-adds).

// Cluster test, in fp32 with conservative margins.  The axis-aligned box (centre c, half extents h) of a cluster, moved by
// the iterate (R, t), lies strictly on one side of plane i iff |f_i(Rc + t)| > sum_j |(n_i^T R)_j| h_j  (support function of
// the rotated box).  The fp32 evaluation decides only when it clears that bound by
//     1e-5 * support + 4e-6 * (|c|_1 + |h|_1 + |t|_1)
// (>= 7x the worst-case rounding of the fp32 evaluation, see the pre-filter below); anything closer is "undecided" and goes
// to the per-point path, which is always right.  If ALL five planes are decided, every point of the cluster has the sign
// pattern of the centre, none of dx, dy, p2 is zero or non-finite, and the per-point classification is known:
//   label 1: inactive iff all five are positive, else every point is active;
//   label 0: active   iff all five are positive, else every point is inactive (and cannot raise `bad`).
// NaN/inf anywhere fails every comparison and falls back to the per-point path.
// Returns 0: skip, 1: classify per point, 2: all active, 3 (label 0 only): no point is active, guard against exact zeros only, 5: the same
// on the top / bottom / z planes only (below).
struct Pre32;
struct alignas(16) BoxAbs {        // per sweep: |n_i^T R|_j * (1 + 1e-5) of the five planes and |t|_1 (wave-uniform; 16 words, kept in LDS)
    float aL[3], aR[3], aT[3], aB[3], aZ[3], T1;
};
// fp32 PRE-FILTER of the per-point classification (phase A).  The exact test costs ~55 fp64 instructions per 64 points
// (rotation, reciprocal, projection, pixel-form comparisons); most points it is run on are nowhere near a frustum plane.
// Here the five normalised plane functions f_i(p) = n_i . (R x + t) are evaluated in fp32 and compared with the margin
//   m S,  m = 4e-6,  S = |x|_1 + |t|_1,
// >= 7x the worst-case fp32 evaluation error (inputs and normalised coefficients rounded to fp32, <= 5 roundings per coordinate,
// <= 3 per plane: <= 9 * 2^-24 * (|x|_1 + |t|_1)).  A point whose five |f_i| all exceed the margin has, in exact arithmetic,
// pixel coordinates at least 3e-6 * fx away from 0 / W-1 / H-1 and |p2| > 3e-6 * |p|, five orders of magnitude above the
// rounding of the fp64 pixel-form test: its classification is the exact test's and none of dx, dy, p2 is zero or non-finite.
// If ANY lane of a cluster is not certified, the whole cluster takes the exact fp64 path, so the active set -- and every
// sum -- is bit-identical with and without the pre-filter (DI2P_SOLVER_NOPREFILTER=1; tests compare).
constexpr float kPreRel = 4e-6f;
struct Pre32 {
    float R[9], t[3], T1;
    float aL, bL, aR, bR, aT, bT, aB, bB;      // f_L = aL p0 + bL p2, f_R = -aR p0 + bR p2, f_T = aT p1 + bT p2, f_B = -aB p1 + bB p2
};
template <int NP>
__device__ __forceinline__ void make_pre32(const Rot<NP>& rot, double tx, double ty, double tz, const float* camf, Pre32& q) {
#pragma unroll
    for (int i = 0; i < 9; ++i) q.R[i] = (float)rot.R[i];
    q.t[0] = (float)tx; q.t[1] = (float)ty; q.t[2] = (float)tz;
    q.T1 = (fabsf(q.t[0]) + fabsf(q.t[1]) + fabsf(q.t[2])) * 1.000001f;
    // The camera-only part -- the eight normalised plane coefficients: six conversions, four divisions, eight products -- comes from the frame's
    // table (written by the preparation kernels) through SCALAR loads: the table is read through the constant address space (a uniform
    // load from plain global memory becomes a per-lane vector load as soon as the kernel also stores to global memory).  Round 6: -3 % of
    // the kernel's vector instructions, at the head of every sweep where all four waves of a workgroup are in step.
    typedef const float __attribute__((address_space(4))) * ConstF;
    const ConstF cf = (ConstF)camf;
    q.aL = cf[0]; q.bL = cf[1]; q.aR = cf[2]; q.bR = cf[3]; q.aT = cf[4]; q.bT = cf[5]; q.aB = cf[6]; q.bB = cf[7];
    // wave-uniform by construction: R, t, T1 are forced into SGPRs as well (the table is live across the cluster loop)
    float* f = reinterpret_cast<float*>(&q);
#pragma unroll
    for (int i = 0; i < 13; ++i) f[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f[i])));
}
// mS: the margin m S of the point -- round 6: the CLUSTER's, from its box (|x|_1 <= |c|_1 + |h|_1 for every point of it: never smaller than the
// point's own, so more points fall to the exact test and none is certified wrongly), read from the lane that owns the cluster.
// -> act, and the point's SLACK: how far (normalised plane units = metres of displacement of the point, |.|_inf) the point is, beyond the
// margin, from changing its classification -- label 1: |min_i f_i| - m S (an inside point: the nearest plane; an outside point: its most
// negative plane, which keeps it active whatever the others do); label 0: min_i |f_i| - m S (EVERY plane: an exact zero on any of them is an
// evaluation failure in the reference).  slack > 0 <=> the point is certified (act is then the exact test's answer); NaN / inf anywhere
// fails that comparison.
template <int NP, int LAB>
__device__ __forceinline__ void prefilter32(const Pre32& q, float X, float Y, float Z, float mS, bool& act, float& slack) {
    float p0, p1, p2;
    if (NP == 4) {
        p0 = fmaf(q.R[0], X, fmaf(q.R[2], Z, q.t[0])); p1 = Y + q.t[1]; p2 = fmaf(q.R[6], X, fmaf(q.R[8], Z, q.t[2]));
    } else {
        p0 = fmaf(q.R[0], X, fmaf(q.R[1], Y, fmaf(q.R[2], Z, q.t[0])));
        p1 = fmaf(q.R[3], X, fmaf(q.R[4], Y, fmaf(q.R[5], Z, q.t[1])));
        p2 = fmaf(q.R[6], X, fmaf(q.R[7], Y, fmaf(q.R[8], Z, q.t[2])));
    }
    const float fL = fmaf(q.aL, p0, q.bL * p2), fR = fmaf(-q.aR, p0, q.bR * p2);
    const float fT = fmaf(q.aT, p1, q.bT * p2), fB = fmaf(-q.aB, p1, q.bB * p2);
    const float mn = fminf(fminf(fminf(fL, fR), fT), fminf(fB, p2));       // two v_min3_f32
    if (LAB == 1) {
        act = mn < 0.0f;                       // some plane negative -> active (certified iff slack > 0)
        slack = fabsf(mn) - mS;
    } else {
        const float ma = fminf(fminf(fminf(fabsf(fL), fabsf(fR)), fabsf(fT)), fminf(fabsf(fB), fabsf(p2)));
        act = mn > 0.0f;                       // all five positive -> active
        slack = ma - mS;
    }
}

// The guard of a status-5 cluster: slack towards the top / bottom / z planes only (8 instead of 16 instructions for the plane functions).
template <int NP>
__device__ __forceinline__ float guard32_tbz(const Pre32& q, float X, float Y, float Z, float mS) {
    float p1, p2;
    if (NP == 4) {
        p1 = Y + q.t[1]; p2 = fmaf(q.R[6], X, fmaf(q.R[8], Z, q.t[2]));
    } else {
        p1 = fmaf(q.R[3], X, fmaf(q.R[4], Y, fmaf(q.R[5], Z, q.t[1])));
        p2 = fmaf(q.R[6], X, fmaf(q.R[7], Y, fmaf(q.R[8], Z, q.t[2])));
    }
    const float fT = fmaf(q.aT, p1, q.bT * p2), fB = fmaf(-q.aB, p1, q.bB * p2);
    return fminf(fminf(fabsf(fT), fabsf(fB)), fabsf(p2)) - mS;       // one v_min3_f32
}

// The box-test table of an iterate, from the pre-filter's table (same fp32 R, t and normalised plane coefficients).
__device__ __forceinline__ void make_box_abs(const Pre32& p, BoxAbs& q) {
    const float g = 1.0f + 1e-5f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {      // (n^T R)_j = a R0j + b R1j + c R2j
        q.aL[j] = fabsf(p.aL * p.R[j] + p.bL * p.R[6 + j]) * g;
        q.aR[j] = fabsf(-p.aR * p.R[j] + p.bR * p.R[6 + j]) * g;
        q.aT[j] = fabsf(p.aT * p.R[3 + j] + p.bT * p.R[6 + j]) * g;
        q.aB[j] = fabsf(-p.aB * p.R[3 + j] + p.bB * p.R[6 + j]) * g;
        q.aZ[j] = fabsf(p.R[6 + j]) * g;
    }
    q.T1 = fabsf(p.t[0]) + fabsf(p.t[1]) + fabsf(p.t[2]);
}
// Status 5 (label 0): a guard-only cluster whose box clears BOTH the left and the right plane by more than kGuardLrMin
// (normalised plane units = metres).  No point of it can sit on those two planes, and the box's clearance *lr_slack is a lower bound of every
// point's |f_L|, |f_R| minus its margin (the box margin contains the point margin: mS_box >= mS_point, support >= the point's offset): the
// per-point guard evaluates the top / bottom / z planes only and the recorded slack is min(point slack on those three, *lr_slack).
constexpr float kGuardLrMin = 0.25f;
template <int NP, int LAB>
__device__ __forceinline__ int cluster_status(const Box& bx, const Pre32& q, const BoxAbs& ab, float* lr_slack = nullptr) {
    const float mS = kPreRel * (((fabsf(bx.cx) + fabsf(bx.cy)) + (fabsf(bx.cz) + ab.T1)) + ((bx.hx + bx.hy) + bx.hz));
    float p0, p1, p2;
    if (NP == 4) {
        p0 = fmaf(q.R[0], bx.cx, fmaf(q.R[2], bx.cz, q.t[0])); p1 = bx.cy + q.t[1]; p2 = fmaf(q.R[6], bx.cx, fmaf(q.R[8], bx.cz, q.t[2]));
    } else {
        p0 = fmaf(q.R[0], bx.cx, fmaf(q.R[1], bx.cy, fmaf(q.R[2], bx.cz, q.t[0])));
        p1 = fmaf(q.R[3], bx.cx, fmaf(q.R[4], bx.cy, fmaf(q.R[5], bx.cz, q.t[1])));
        p2 = fmaf(q.R[6], bx.cx, fmaf(q.R[7], bx.cy, fmaf(q.R[8], bx.cz, q.t[2])));
    }
    const float fL = fmaf(q.aL, p0, q.bL * p2), fR = fmaf(-q.aR, p0, q.bR * p2);
    const float fT = fmaf(q.aT, p1, q.bT * p2), fB = fmaf(-q.aB, p1, q.bB * p2);
    auto bound = [&](const float* a) { return fmaf(a[0], bx.hx, fmaf(a[1], bx.hy, fmaf(a[2], bx.hz, mS))); };
    const float tL = bound(ab.aL), tR = bound(ab.aR), tT = bound(ab.aT), tB = bound(ab.aB), tZ = bound(ab.aZ);
    // all five decided positive <=> min_i (f_i - t_i) > 0 ; all five decided <=> min_i (|f_i| - t_i) > 0
    const float lo = fminf(fminf(fminf(fL - tL, fR - tR), fminf(fT - tT, fB - tB)), p2 - tZ);
    const float cm = fminf(fminf(fminf(fabsf(fL) - tL, fabsf(fR) - tR), fminf(fabsf(fT) - tT, fabsf(fB) - tB)), fabsf(p2) - tZ);
    const bool inside = lo > 0.0f;
    // The decision table as selects (as nested returns every lane of a round took another path: the wave ran all of them anyway, plus the
    // register copies at their joins -- round 6: 157 -> 133 vector instructions per round).  A NaN fails every comparison -> 1.
    // A decided-negative plane already decides every point: not inside.  Label 1: all active.  Label-0 points of such a cluster may still sit
    // exactly on one of the undecided planes (dx, dy or p2 == 0 is an evaluation failure in the reference): status 3 / 5 = no point can be
    // active, but every point is still checked against exact zeros (fp32 guard, exact test when the guard cannot certify) -- without the
    // ballot / queue work of a real classification.
    const float hi = fminf(fminf(fminf(fL + tL, fR + tR), fminf(fT + tT, fB + tB)), p2 + tZ);
    const bool decided = cm > 0.0f, neg = hi < 0.0f;
    if (LAB == 1) return decided ? (inside ? 0 : 2) : (neg ? 2 : 1);
    const float lrs = fminf(fabsf(fL) - tL, fabsf(fR) - tR);
    *lr_slack = lrs;
    return decided ? (inside ? 2 : 0) : (neg ? (lrs > kGuardLrMin ? 5 : 3) : 1);      // lrs NaN -> 3
}

// Wave-wide minima of FOUR non-negative floats (or +inf) at once, by DPP (no LDS round trip): four joins inside rows of 16 lanes, then
// row_bcast15 / row_bcast31 carry the row results into lane 63, which is read back as a scalar.  Non-negative floats order like their bit
// patterns, so the joins are v_min_u32 with the DPP modifier ON the instruction -- written as one asm block because hipcc expands
// fminf(v, dpp(v)) into mov + mov_dpp + canonicalise + min (24 instructions per value instead of 6).  The four values are interleaved: three
// independent instructions separate a value's consecutive joins, which covers the two wait states a DPP read needs after a VALU write.
__device__ __forceinline__ void wave_min4_nonneg(float& a, float& b, float& c, float& d) {
#define DI2P_MIN4(CTRL)                                \
    "v_min_u32_dpp %0, %0, %0 " CTRL "\n\t"           \
    "v_min_u32_dpp %1, %1, %1 " CTRL "\n\t"           \
    "v_min_u32_dpp %2, %2, %2 " CTRL "\n\t"           \
    "v_min_u32_dpp %3, %3, %3 " CTRL "\n\t"
    asm volatile("s_nop 1\n\t"
                 DI2P_MIN4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 DI2P_MIN4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 DI2P_MIN4("row_half_mirror row_mask:0xf bank_mask:0xf")
                 DI2P_MIN4("row_mirror row_mask:0xf bank_mask:0xf")
                 DI2P_MIN4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DI2P_MIN4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef DI2P_MIN4
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
    c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
    d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), 63));
}

// The same for TWO values and for ONE (batches of the classification walk smaller than four): fewer independent instructions separate a value's
// consecutive joins, so the DPP read-after-write wait states are filled with s_nop.
__device__ __forceinline__ void wave_min2_nonneg(float& a, float& b) {
#define DI2P_MIN2(CTRL) "v_min_u32_dpp %0, %0, %0 " CTRL "\n\t" "v_min_u32_dpp %1, %1, %1 " CTRL "\n\t" "s_nop 0\n\t"
    asm volatile("s_nop 1\n\t"
                 DI2P_MIN2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 DI2P_MIN2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 DI2P_MIN2("row_half_mirror row_mask:0xf bank_mask:0xf")
                 DI2P_MIN2("row_mirror row_mask:0xf bank_mask:0xf")
                 DI2P_MIN2("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DI2P_MIN2("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b));
#undef DI2P_MIN2
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
}
__device__ __forceinline__ void wave_min1_nonneg(float& a) {
#define DI2P_MIN1(CTRL) "v_min_u32_dpp %0, %0, %0 " CTRL "\n\t" "s_nop 1\n\t"
    asm volatile("s_nop 1\n\t"
                 DI2P_MIN1("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 DI2P_MIN1("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 DI2P_MIN1("row_half_mirror row_mask:0xf bank_mask:0xf")
                 DI2P_MIN1("row_mirror row_mask:0xf bank_mask:0xf")
                 DI2P_MIN1("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DI2P_MIN1("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a));
#undef DI2P_MIN1
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
}
template <int N> __device__ __forceinline__ void wave_min_nonneg(float* v) {
    static_assert(N == 1 || N == 2 || N == 4, "batch sizes of the cluster walk");
    if (N == 4) wave_min4_nonneg(v[0], v[1], v[2], v[3]);
    if (N == 2) wave_min2_nonneg(v[0], v[1]);
    if (N == 1) wave_min1_nonneg(v[0]);
}

// CLASSIFICATION CACHE (temporal coherence of the cluster walk).  Between two sweeps of a hypothesis the iterate moves little (line-search
// trials along one step, LM steps that shrink towards the minimum), while a cluster that needs per-point work needs it sweep after sweep
// because a frustum plane passes through its box.  When a cluster IS classified per point, the wave also takes the minimum of the points'
// slacks (prefilter32) and records {active mask, slack, sweep number} per hypothesis and cluster (16 bytes, global memory, L2 resident); the
// iterates of the last RING sweeps stay in LDS.  A later sweep moves every point of the cluster by at most
//     mu = |d theta| * r + |d t|_inf        (r = the cluster's largest ground-plane radius; 3-D solver: |d w|_1 * largest norm)
// against the recorded iterate, and a displacement d changes every normalised plane function by at most |d|_inf.  While
// slack > 1.0001 mu + 1e-6 every point keeps its certified sign pattern with room to spare (>= 1e-6 m beyond the margin, nine orders above the
// rounding of the fp64 test): the recorded mask IS the classification, and no point sits on a plane.  A hit costs the walk one append
// (status 4) or nothing at all (guard-only clusters); the active sequence -- hence every sum -- is bit-identical with the cache off
// (DI2P_SOLVER_NOCACHE=1) and with the cluster test off (tests and tools/fuzz_solver_cull.py compare).  Measured on the oracle's iterate
// traces (tools/model_temporal.py): 70 % of the per-point classifications and 61 % of the guard walks hit.
struct __attribute__((aligned(16))) CacheEnt { unsigned mlo, mhi; float slack; unsigned stamp; };      // stamp = sweep number + 1, 0 = empty
constexpr int RING = 8;            // iterates kept in LDS (power of two)
constexpr int CACHE_PAD = 32;      // per hypothesis: NCMAX + CACHE_PAD entries (each label block is rounded up to a multiple of WPH clusters)

// One label-uniform block of records [recs, recs+cnt) = nc clusters of CL records.  Cluster c belongs to wave
// c % WPH (neighbouring clusters -- which tend to share their status -- spread over the waves).  Per round a lane
// tests one cluster (box test, then the classification cache); the wave then
//   phase I : classifies the clusters that need it -- status 1: fp32 pre-filter (exact fp64 test with the reference's pixel-form
//             conditions for a cluster with an uncertified record) -> active mask + slack into the lane that owns the cluster;
//             status 3: zero guard only -> slack -- and records them in the cache;
//   phase II: appends the active ids of the status 1 / 2 (all active) / 4 (cached mask) clusters to the per-wave LDS queue IN CLUSTER
//             ORDER; the queue is evaluated densely (phase B) 64 records at a time.
// The queue sequence is the same as if every cluster had been classified per point, so the sums are bit-identical to the unculled
// sweep (nocull bit 0 forces status 1 everywhere, bit 1 the exact test, bit 2 switches the cache off: tests compare).
template <int NP, typename PT, int WPH, int LAB, int MODE, bool PROFILE>
__device__ __forceinline__ void sweep_clusters(const Rec<PT>* __restrict__ recs, int cnt, const Box* __restrict__ boxes, int nc,
                                               const Cam& k, const double* x, const Rot<NP>& rot, int nocull,
                                               int* queue, double (*acc)[64], int* acc_e, const Pre32& pre, const float* btest_lds,
                                               CacheEnt* __restrict__ cache, const double* ring, int s_now, bool& bad, int* n_active, long long* tp) {
    constexpr int TOFF = NP == 4 ? 1 : 3;
    const double tx = x[TOFF], ty = x[TOFF + 1], tz = x[TOFF + 2];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the wave index as a SCALAR
    const double hw = k.W1 * 0.5, hh = k.H1 * 0.5;
    int qn = 0;  // wave-uniform
    // A cluster's 64 records: one load per lane at (wave-uniform cluster base) + (lane offset).  Label blocks are cluster-aligned and padded
    // (prepare_kernel), so every lane of every cluster is addressable: no clamping.  Exhausted ring slots (c < 0) re-read cluster 0.
    const char* rec_bytes = reinterpret_cast<const char*>(recs);
    const unsigned lane_off = (unsigned)lane * (unsigned)sizeof(Rec<PT>);
    auto load_rec = [&](int c) {
        const char* base = rec_bytes + (size_t)(unsigned)max(c, 0) * (CL * sizeof(Rec<PT>));      // scalar
        return *reinterpret_cast<const Rec<PT>*>(base + lane_off);
    };
    // Phase B over the queued ids: full rounds of 64 from the FRONT of the queue (position p of the wave's active sequence is
    // always evaluated by lane p % 64, so the per-lane sums do not depend on when the queue is drained); the records of the next
    // round are gathered while the current round is evaluated.  flush: also the last, partial round.
    auto drain = [&](bool flush) {
        const long long td0 = PROFILE ? clock64() : 0;
        const int total = flush ? qn : (qn & ~63);
        __builtin_amdgcn_wave_barrier();          // LDS ops of one wave retire in order; only the compiler must not reorder them
        if (total > 0) {                          // wave-uniform
            int n_cur = queue[lane];
            Rec<PT> r_cur = recs[min(max(n_cur, 0), cnt - 1)];
            // the lane's running sums come out of LDS for the duration of the drain
            LogProd cost;
            double lg[NP], lA[Tri<NP>::N];
            cost.m = acc[0][lane]; cost.e = acc_e[lane];
#pragma unroll
            for (int i = 0; i < NP; ++i) lg[i] = acc[1 + i][lane];
#pragma unroll
            for (int i = 0; i < Tri<NP>::N; ++i) lA[i] = acc[1 + NP + i][lane];
            for (int pos = 0; pos < total; pos += 64) {
                const int n_nxt = queue[min(pos + 64 + lane, QCAP - 1)];
                const Rec<PT> r_nxt = recs[min(max(n_nxt, 0), cnt - 1)];          // unconditional, clamped (the last one is wasted)
                if (pos + lane < total) eval_active<NP, PT, LAB, MODE>(r_cur, rot, x, k, cost, lg, lA, bad);
                if (PROFILE) { n_active[0] += min(64, total - pos); n_active[LAB == 1 ? 9 : 10] += 1; }
                n_cur = n_nxt; r_cur = r_nxt;
            }
            acc[0][lane] = cost.m; acc_e[lane] = cost.e;
#pragma unroll
            for (int i = 0; i < NP; ++i) acc[1 + i][lane] = lg[i];
#pragma unroll
            for (int i = 0; i < Tri<NP>::N; ++i) acc[1 + NP + i][lane] = lA[i];
        }
        const int rem = qn - total;               // < 64 ids stay queued (0 after a flush)
        const int carry = queue[min(total + lane, QCAP - 1)];
        __builtin_amdgcn_wave_barrier();
        if (lane < rem) queue[lane] = carry;
        __builtin_amdgcn_wave_barrier();
        qn = rem;
        if (PROFILE) tp[1] += clock64() - td0;
    };
    // exact classification of one record (the reference's pixel-form conditions)
    auto exact_active = [&](const Rec<PT>& rec, bool valid) {
        double X, Y, Z, qx, qz, p0, p1, p2, iz, pix_x, pix_y;
        project<NP, PT, true>(rec, rot, tx, ty, tz, k, X, Y, Z, qx, qz, p0, p1, p2, iz, pix_x, pix_y);
        if (LAB == 1)
            return valid && (!(-pix_x < 0.0) || !(pix_x - k.W1 < 0.0) || !(-pix_y < 0.0) || !(pix_y - k.H1 < 0.0) || !(-p2 < 0.0));
        const double dx = hw - fabs(pix_x - hw), dy = hh - fabs(pix_y - hh);
        // fmax(v,0)/v is NaN at v == 0 (registration_2d.hpp:53,56,58), and a non-finite pixel poisons the
        // residual: evaluation failure.  One test: dx*dy*p2 is 0 or non-finite exactly in those cases.
        const double chk = dx * dy * p2;
        if (valid && (!(fabs(chk) > 0.0) || !(fabs(chk) < __builtin_inf()))) bad = true;
        return valid && dx > 0.0 && dy > 0.0 && p2 > 0.0;
    };
    const bool use_pre = (nocull & 2) == 0;
    const bool use_cache = (nocull & 5) == 0;            // the cache rides on the cluster test
    nocull &= 1;
    const int mine = (nc - wave + WPH - 1) / WPH;        // clusters wave, wave+WPH, ... < nc
    CacheEnt* cache_w = cache + wave * ((nc + WPH - 1) / WPH);      // the wave's entries are consecutive: lane j <-> its j-th cluster
    constexpr int PF = DI2P_SOLVER_PF;
    // lowest set bit of a wave-uniform mask, cleared; -1 when the mask is empty.  s_ff1_i32_b64 returns -1 for an empty mask by itself and
    // s_bitset0_b64 with index -1 clears bit 63 of an empty mask (a no-op): two scalar instructions instead of the seven of the portable form
    // (add / addc / and for m & (m - 1), ff1, compare, select)
    auto take_bit = [](unsigned long long& m) {
        int b;
        asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(b), "+s"(m));
        return b;
    };
    const int nv_last = cnt - (nc - 1) * CL;                 // valid records of the block's last cluster (scalar), 1..CL
    const unsigned long long tail_mask = nv_last >= CL ? ~0ull : ((1ull << (nv_last & 63)) - 1ull);
    const unsigned tail_lo = (unsigned)tail_mask, tail_hi = (unsigned)(tail_mask >> 32);
    const unsigned queue_addr = (unsigned)(size_t)(__attribute__((address_space(3))) int*)queue;      // LDS byte address of the wave's queue
    for (int j0 = 0; j0 < mine; j0 += 64) {
        const long long ts0 = PROFILE ? clock64() : 0;
        const int j = j0 + lane;
        const int cbase = j0 * WPH + wave;       // cluster of the round's lane 0 (scalar): lane b's cluster is cbase + b * WPH
        auto slot_cluster = [&](int b) {         // (one scalar instruction for four-wave workgroups: hipcc re-associates b * 4 + cbase into add, shift, add)
            int c;
            if (WPH == 4) asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(c) : "s"(b), "s"(cbase) : "scc");
            else c = b * WPH + cbase;
            return c;
        };
        int status = 0;
        unsigned mlo = 0u, mhi = 0u;      // active mask of the lane's cluster (status 4: cached, 2: its valid records, 1: filled in by phase I)
        float slack = 0.0f;               // min slack of the lane's cluster (phase I)
        float lr_slack = 0.0f;            // status 5: the box's clearance of the left / right planes
        bool cached_guard = false;
        // the margin m S of the per-point tests of the lane's cluster, from its box (|x|_1 <= |c|_1 + |h|_1 for every point of it: a margin
        // that is never smaller than the point's own -- more points fall to the exact test, none is certified wrongly); +inf switches the
        // fp32 tests off (solver_noprefilter), NaN (a poisoned box) certifies nothing
        float ms_box = 0.0f;
        if (j < mine) {
            const int c = j * WPH + wave;
            const Box bx = boxes[c];           // (solver_nocull needs it for the margin only)
            ms_box = use_pre ? kPreRel * (((fabsf(bx.cx) + fabsf(bx.cy)) + (fabsf(bx.cz) + pre.T1)) + ((bx.hx + bx.hy) + bx.hz)) : __builtin_inff();
            {   // box test + cache look-up of every lane, as SELECTS (round 6: the nested divergent form ran every path anyway)
                // the table is wave-uniform and only needed here (<= 2 rounds per label block): fetched from LDS per round instead of
                // staying in VGPRs through the cluster walk
                BoxAbs ab;
                asm volatile("" ::: "memory");
                const float4* bt4 = reinterpret_cast<const float4*>(btest_lds);
                float4* dst4 = reinterpret_cast<float4*>(&ab);
#pragma unroll
                for (int i = 0; i < (int)(sizeof(BoxAbs) / 16); ++i) dst4[i] = bt4[i];
                CacheEnt e;                                     // fetched together with the box, whatever the box test will say
                e.mlo = e.mhi = 0u; e.slack = 0.0f; e.stamp = 0u;
                if (use_cache) e = cache_w[j];                  // wave-uniform
                const int st = nocull ? 1 : cluster_status<NP, LAB>(bx, pre, ab, &lr_slack);       // (solver_nocull: the table is stale, the result unused)
                // the cache look-up of every lane, as selects: hit <=> the cluster needs per-point work, its entry is in the ring and its slack
                // covers the motion since the recorded iterate (an empty entry -- stamp 0 -- reads ring slot 7: any iterate, unused)
                const unsigned age = (unsigned)s_now + 1u - e.stamp;               // e.stamp <= s_now + 1
                const double* xr = ring + ((e.stamp - 1u) & (unsigned)(RING - 1)) * NP;
                float mu;
                if (NP == 4) {
                    const double dth = fabs(x[0] - xr[0]);
                    const double dt = fmax(fmax(fabs(x[1] - xr[1]), fabs(x[2] - xr[2])), fabs(x[3] - xr[3]));
                    mu = fmaf((float)dth, bx.rxz, (float)dt);
                } else {
                    const double dth = (fabs(x[0] - xr[0]) + fabs(x[1] - xr[1])) + fabs(x[2] - xr[2]);     // >= |d w|_2
                    const double dt = fmax(fmax(fabs(x[3] - xr[3]), fabs(x[4] - xr[4])), fabs(x[5] - xr[5]));
                    mu = fmaf((float)dth, bx.r3, (float)dt);
                }
                const float need = fmaf(1.0001f, mu, 1e-6f);
                const bool walk = st == 1 || st == 3 || st == 5;
                const bool hit = use_cache & walk & (e.stamp != 0u) & (age < (unsigned)RING) & (e.slack > need);       // NaN fails
                cached_guard = hit & (st != 1);
                status = hit ? (st == 1 ? 4 : 0) : st;
                const bool last = c == nc - 1;
                // active mask: the recorded one (hit), the cluster's valid records (all active: all 64 but for the block's last cluster, whose
                // mask is wave-uniform), else filled in by phase I
                mlo = hit ? e.mlo : (st == 2 ? (last ? tail_lo : ~0u) : 0u);
                mhi = hit ? e.mhi : (st == 2 ? (last ? tail_hi : ~0u) : 0u);
                if (hit && age >= (unsigned)(RING / 2)) {
                    // still valid but about to leave the ring: re-record against THIS iterate with what is left of the slack
                    CacheEnt ne;
                    ne.mlo = e.mlo; ne.mhi = e.mhi; ne.slack = (e.slack - need) * 0.99999f; ne.stamp = (unsigned)s_now + 1u;
                    cache_w[j] = ne;
                }
            }
        }
        const unsigned long long mA = __ballot(status == 1), mB = __ballot(status == 2), mC = __ballot(status == 3), mD = __ballot(status == 4);
        const unsigned long long mE = __ballot(status == 5);
        if (PROFILE) {
            n_active[1] += __popcll(mA); n_active[2] += __popcll(mB); n_active[3] += __popcll(mC | mE);
            n_active[4] += __popcll(mD); n_active[5] += __popcll(__ballot(cached_guard));
            n_active[11] += 1; n_active[8] += __popcll(mA | mB | mD); n_active[12] += __popcll(mE);
            tp[0] += clock64() - ts0;
        }
        // ---- phase I.  The flagged clusters are walked PF AT A TIME in straight-line code (a cluster's 64 records are one 16-byte load per
        // lane): the pre-filters are independent instruction streams and the records of the next PF are in flight meanwhile.
        // (a) zero-guard-only clusters (status 3; status 5: on the top / bottom / z planes only): no point is active; only an exact zero /
        // non-finite value on an undecided plane must be found (sets `bad`).  The cluster's min slack says both whether every record is
        // certified (> 0) and for how long.
        auto guard_walk = [&](auto tbz_tag, unsigned long long mg) {
            constexpr bool TBZ = decltype(tbz_tag)::value;
            int nb[PF];
            Rec<PT> ring_r[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) { nb[u] = take_bit(mg); ring_r[u] = load_rec(slot_cluster(nb[u])); }
            while (nb[0] >= 0) {
                if (PROFILE) n_active[6] += 1;
                float sm[PF];
                int nbp[PF];
                Rec<PT> cur[PF];
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    cur[u] = ring_r[u];
                    nbp[u] = nb[u];
                    nb[u] = take_bit(mg);
                    ring_r[u] = load_rec(slot_cluster(nb[u]));
                    float sl;
                    const float msu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ms_box), nbp[u] & 63));
                    if (TBZ) {
                        sl = guard32_tbz<NP>(pre, (float)cur[u].x, (float)cur[u].y, (float)cur[u].z, msu);
                    } else {
                        bool a;
                        prefilter32<NP, 0>(pre, (float)cur[u].x, (float)cur[u].y, (float)cur[u].z, msu, a, sl);
                    }
                    sl = __builtin_fmaxf(sl, 0.0f);                             // not certified (<= 0, or NaN: v_max returns the other operand) -> 0
                    // no lane mask: the padding lanes of a block's last cluster hold COPIES of its last record (prepare_kernel), so they change
                    // neither the minimum nor the verdict of the exact test; the value of an exhausted slot (nbp < 0) is never used
                    sm[u] = sl;
                }
                wave_min_nonneg<PF>(sm);
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    if (nbp[u] >= 0) {              // wave-uniform
                        if (!(sm[u] > 0.0f)) (void)exact_active(cur[u], true);      // rare: some record the fp32 guard cannot certify -> exact test of this cluster (sets `bad` on a zero)
                        slack = lane == nbp[u] ? sm[u] : slack;                    // into the lane that owns the cluster
                    }
                }
            }
        };
        // Round 6: three of four guard-only clusters of the config-2 workload clear the left / right planes by their box (the band beside the
        // camera, |p2| small): their guard needs the top / bottom / z planes only (28 instead of 36 vector instructions per cluster)
        if (LAB == 0 && mC) guard_walk(std::false_type(), mC);
        if (LAB == 0 && mE) {
            guard_walk(std::true_type(), mE);
            if (status == 5) slack = fminf(slack, lr_slack);       // the box's clearance of the two planes the walk left out (lane-parallel)
        }
        if (mA) {
            // (b) clusters classified per point (status 1), PFC at a time
            constexpr int PFC = DI2P_SOLVER_PFC;
            unsigned long long mo = mA;
            int nb[PFC];
            Rec<PT> ring_r[PFC];
#pragma unroll
            for (int u = 0; u < PFC; ++u) { nb[u] = take_bit(mo); ring_r[u] = load_rec(slot_cluster(nb[u])); }
            while (nb[0] >= 0) {
                if (PROFILE) n_active[7] += 1;
                Rec<PT> cur[PFC];
                int nbp[PFC];
                bool act[PFC];
                float sm[PFC];
#pragma unroll
                for (int u = 0; u < PFC; ++u) {
                    cur[u] = ring_r[u];
                    nbp[u] = nb[u];
                    nb[u] = take_bit(mo);
                    ring_r[u] = load_rec(slot_cluster(nb[u]));
                }
#pragma unroll
                for (int u = 0; u < PFC; ++u) {          // no short circuits: PFC independent, branch-free instruction streams
                    bool a;
                    float sl;
                    const float msu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ms_box), nbp[u] & 63));
                    prefilter32<NP, LAB>(pre, (float)cur[u].x, (float)cur[u].y, (float)cur[u].z, msu, a, sl);
                    sl = __builtin_fmaxf(sl, 0.0f);
                    act[u] = a;           // the padding lanes (copies of the block's last record) are cut from the BALLOT by a scalar mask below
                    sm[u] = sl;
                }
                wave_min_nonneg<PFC>(sm);
#pragma unroll
                for (int u = 0; u < PFC; ++u) {
                    if (nbp[u] >= 0) {              // wave-uniform
                        if (!(sm[u] > 0.0f)) act[u] = exact_active(cur[u], true);        // rare: some record is not certified -> exact test for THAT cluster
                        // the padding lanes of the block's LAST cluster are cut from the ballot (its mask is the block's tail mask: scalar)
                        unsigned long long bal = __ballot(act[u]);
                        if (slot_cluster(nbp[u]) == nc - 1) {          // wave-uniform, once per block at most: the mask is re-derived here rather than kept live
                            const int nvr = cnt - (nc - 1) * CL;
                            bal &= nvr >= CL ? ~0ull : ((1ull << nvr) - 1ull);
                        }
                        // into the lane that owns the cluster
                        mlo = lane == nbp[u] ? (unsigned)bal : mlo;
                        mhi = lane == nbp[u] ? (unsigned)(bal >> 32) : mhi;
                        slack = lane == nbp[u] ? sm[u] : slack;
                    }
                }
            }
        }
        if (use_cache && (status == 1 || status == 3 || status == 5)) {       // what phase I found, one entry per lane
            CacheEnt ne;
            ne.mlo = mlo; ne.mhi = mhi; ne.slack = slack; ne.stamp = (unsigned)s_now + 1u;
            cache_w[j] = ne;
        }
        // ---- phase II: the active ids of the round, in cluster order
        {
            unsigned long long mo = mA | mB | mD;
            while (mo) {
                const int b = take_bit(mo);
                if (qn > QCAP - 64) drain(false);      // queue nearly full: evaluate the full rounds, keep the remainder queued
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)mlo, b), hi = (unsigned)__builtin_amdgcn_readlane((int)mhi, b);
                {   // the store runs under exec = the cluster's active mask (mbcnt does not depend on exec): no per-lane bit test, the address and
                    // the id are (scalar) + (lane term) -- 6 instead of 11 vector instructions per appended cluster
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                    const unsigned addr = (queue_addr + 4u * (unsigned)qn) + 4u * rank;              // (scalar) + 4 * rank
                    const unsigned id = (unsigned)(((j0 + b) * WPH + wave) * CL) + (unsigned)lane;   // (scalar) + lane
                    const unsigned long long m64 = ((unsigned long long)hi << 32) | lo;
                    unsigned long long saved;
                    // (all lanes are active here, so exec & mask = mask: one s_and_saveexec instead of two moves)
                    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                                 : "=&s"(saved) : "s"(m64), "v"(addr), "v"(id) : "memory", "scc");
                    qn += __builtin_popcountll(m64);
                }
            }
        }
    }
    drain(true);
}

// Lane exchange inside rows of 16 lanes by DPP (no LDS round trip): CTRL 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = [2,3,0,1]
// (lane ^ 2), 0x141 = row_half_mirror (lane ^ 7), 0x140 = row_mirror (lane ^ 15).  Applied in this order to a value that is already
// uniform inside the groups joined so far, they combine 2, 4, 8 and 16 lanes.
template <int CTRL> __device__ __forceinline__ int dpp_int(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_double(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = dpp_int<CTRL>((int)b), hi = dpp_int<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Leaves the WPH wave partials {cost, g[NP], A[tri], bad} in sh.red[wave][*]; the caller combines them after a
// barrier.  Records are sorted by label (prepare_kernel): the label-1 block and the label-0 block are swept by two
// specialised loops.
template <int NP, typename PT, int WPH, int MODE, bool PROFILE>
__device__ __forceinline__ void sweep(const Rec<PT>* __restrict__ recs, const Box* __restrict__ boxes, int cnt1, int cnt0, int nc1,
                                     int nc0, const Cam& k, const float* camf, const double* x, int nocull, SweepShared<NP, WPH>& sh, CacheEnt* __restrict__ cache, int s_now,
                                     int* n_active, long long* tp) {
    constexpr int NV = Tri<NP>::N + NP + 2;
    const long long tq0 = PROFILE ? clock64() : 0;
    constexpr int TOFF = NP == 4 ? 1 : 3;
    static_assert(sizeof(BoxAbs) == BOXTEST_WORDS * 4, "box-test table is copied as float4s");
    Rot<NP> rot;
    if (NP == 4) {
        // the rotation of the iterate comes from the LM lane (one fp64 sincos per sweep and workgroup instead of one per wave; same bits)
        const double c = sh.rot_cs[0], s = sh.rot_cs[1];
        rot.R[0] = c; rot.R[1] = 0; rot.R[2] = s; rot.R[3] = 0; rot.R[4] = 1; rot.R[5] = 0; rot.R[6] = -s; rot.R[7] = 0; rot.R[8] = c;
    } else {
        make_rot<NP>(x, rot);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* queue = sh.queue[wave];
    double (*acc)[64] = sh.acc[wave];
    int* acc_e = sh.acc_e[wave];
    {   // the lane's running sums start at zero (cost product: 0.5 * 2^1)
        LogProd c0;
        c0.init();
        acc[0][lane] = c0.m; acc_e[lane] = c0.e;
#pragma unroll
        for (int i = 1; i < 1 + NP + Tri<NP>::N; ++i) acc[i][lane] = 0.0;
    }
    Pre32 pre;        // fp32 table of the iterate (SGPRs), shared by the cluster test and the per-point pre-filter of both label blocks
    make_pre32<NP>(rot, x[TOFF], x[TOFF + 1], x[TOFF + 2], camf, pre);
    if (!(nocull & 1)) {   // box-test table of this iterate: every lane computes the same values, lane 0 stores them
        BoxAbs ab;
        make_box_abs(pre, ab);
        if (lane == 0) {
            const float4* w = reinterpret_cast<const float4*>(&ab);
            float4* d = reinterpret_cast<float4*>(sh.btest[wave]);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(BoxAbs) / 16); ++i) d[i] = w[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    bool bad = false;
    if (PROFILE) tp[2] += clock64() - tq0;          // set-up: rotation, fp32 tables, zeroed sums
    static_assert(RING == 8, "SweepShared::ring holds RING iterates");
    const double* ring = &sh.ring[0][0];
    // the label-0 block's cache entries follow the label-1 block's (each block rounded up to a multiple of WPH clusters)
    sweep_clusters<NP, PT, WPH, 1, MODE, PROFILE>(recs, cnt1, boxes, nc1, k, x, rot, nocull, queue, acc, acc_e, pre, sh.btest[wave], cache, ring, s_now, bad, n_active, tp);
    sweep_clusters<NP, PT, WPH, 0, MODE, PROFILE>(recs + nc1 * CL, cnt0, boxes + nc1, nc0, k, x, rot, nocull, queue, acc, acc_e, pre, sh.btest[wave],
                                                  cache + WPH * ((nc1 + WPH - 1) / WPH), ring, s_now, bad, n_active, tp);
    __builtin_amdgcn_wave_barrier();
    const long long tq1 = PROFILE ? clock64() : 0;
    // Wave totals of the 1 + NP + tri running values, read back from LDS TRANSPOSED: 16 lanes per value (4 values per round), each
    // lane takes 4 consecutive lanes' entries (two 16-byte reads), then four DPP steps join the 16 partial results.  Value 0 is the
    // cost PRODUCT (mantissas multiply, exponents add; the single log is taken by the LM lane after the waves' products are
    // combined), the others are sums.  Fixed association: deterministic, and the same whichever shortcut built the per-lane values.
    // (Replaces a 16-value butterfly over ds_bpermute + one log per lane: 4.1 k cycles per wave and sweep.)
    constexpr int NVAL = 1 + NP + Tri<NP>::N;
    const int grp = lane >> 4, q = lane & 15;
    double* mine = sh.red[wave];
#pragma unroll
    for (int r = 0; r * 4 < NVAL; ++r) {
        const int i = r * 4 + grp;
        const int ic = i < NVAL ? i : NVAL - 1;
        const double4 v = *reinterpret_cast<const double4*>(&acc[ic][q * 4]);
        const bool is_cost = r == 0 && grp == 0;
        double val = is_cost ? (v.x * v.y) * (v.z * v.w) : (v.x + v.y) + (v.z + v.w);
        int ex = 0;
        if (r == 0) { const int4 e4 = *reinterpret_cast<const int4*>(&acc_e[q * 4]); ex = (e4.x + e4.y) + (e4.z + e4.w); }
#define DI2P_JOIN(CTRL) { const double o = dpp_double<CTRL>(val); val = is_cost ? val * o : val + o; if (r == 0) ex += dpp_int<CTRL>(ex); }
        DI2P_JOIN(0xB1) DI2P_JOIN(0x4E) DI2P_JOIN(0x141) DI2P_JOIN(0x140)
#undef DI2P_JOIN
        if (q == 0 && i < NVAL) mine[i] = val;          // >= 2^-64 for the product of 64 mantissas in [0.5, 1): no underflow
        if (r == 0 && lane == 0) sh.red_e[wave] = ex;
    }
    const bool anybad = __any(bad) != 0;        // evaluated by the whole wave (a ballot inside `if (lane == 0)` sees lane 0 only)
    if (lane == 0) mine[NV - 1] = anybad ? 1.0 : 0.0;
    if (PROFILE) tp[3] += clock64() - tq1;      // wave totals
}

// Cholesky solve of M y = rhs IN PLACE: M (lower triangle, row-major) is overwritten by its factor L, y holds rhs on entry and the
// solution on return (forward substitution, then backward substitution, both in place).  Same operations in the same order as the
// oracle's chol_solve; written in place because the LM update runs on one lane of a kernel whose register budget is set by the sweep --
// separate M / L / z arrays pushed it into scratch.
template <int NP>
__device__ __forceinline__ bool chol_solve_inplace(double* M, double* y) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = M[i * (i + 1) / 2 + j];
#pragma unroll
            for (int q = 0; q < j; ++q) s -= M[i * (i + 1) / 2 + q] * M[j * (j + 1) / 2 + q];
            if (i == j) {
                if (!(s > 0.0) || !isfinite(s)) return false;
                M[i * (i + 1) / 2 + i] = lm_sqrt(s);
            } else {
                M[i * (i + 1) / 2 + j] = lm_div(s, M[j * (j + 1) / 2 + j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        double s = y[i];
#pragma unroll
        for (int q = 0; q < i; ++q) s -= M[i * (i + 1) / 2 + q] * y[q];
        y[i] = lm_div(s, M[i * (i + 1) / 2 + i]);
    }
#pragma unroll
    for (int i = NP - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int q = i + 1; q < NP; ++q) s -= M[q * (q + 1) / 2 + i] * y[q];
        y[i] = lm_div(s, M[i * (i + 1) / 2 + i]);
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NP; ++i) ok &= isfinite(y[i]);
    return ok;
}

template <int NP>
__device__ __forceinline__ void plus_proj(const double* x, const double* d, double t, const double* lb, const double* ub, double* out) {
#pragma unroll
    for (int i = 0; i < NP; ++i) out[i] = fmin(fmax(x[i] + t * d[i], lb[i]), ub[i]);
}

template <int NP>
__device__ __forceinline__ double grad_max_norm(const double* x, const double* g, const double* lb, const double* ub) {
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const double pr = fmin(fmax(x[i] - g[i], lb[i]), ub[i]);
        m = fmax(m, fabs(x[i] - pr));
    }
    return m;
}


// ---------------------------------------------------------------------------------------------------------------------
// Ceres' ArmijoLineSearch with CUBIC interpolation (line_search.cc, polynomial.cc), same statement as the oracle's:
// the next step minimises, over [1e-3, 0.6] x current step, the polynomial interpolating value + directional derivative at
// step 0, at the current trial and (when valid) at the previous trial.  Fitted in u = step / current step with the two
// constraints at 0 eliminated; minimiser = best of {interval midpoint, ends, real critical points} (MinimizePolynomial).
// Runs on ONE lane between sweeps: loops are unrolled over compile-time indices so that everything stays in registers.
template <int DEG> __device__ __forceinline__ double poly_eval(const double* c, double u) {
    double v = c[DEG];
#pragma unroll
    for (int i = DEG - 1; i >= 0; --i) v = fma(v, u, c[i]);
    return v;
}

// ---- real critical points by the WHOLE wavefront (wave 0 runs this between sweeps; arguments are wave-uniform).
// Same statement as the oracle's real_roots_in(): the roots of q inside [a,b] are isolated between consecutive roots of q'
// (found the same way one degree down; degree 2 in closed form, as polynomial.cc's FindQuadraticPolynomialRoots), q is
// monotone on each piece, so a sign change pins exactly one root.  The oracle solves a piece by scalar Newton/bisection
// (~25 dependent fp64 divisions); here the 64 lanes sample the piece at 64 points, a ballot finds the sub-interval with the
// sign change (three rounds: 63^3 = 2.5e5 x narrower), and three Newton steps with a reciprocal finish it to rounding.
template <int DEG>
__device__ __forceinline__ double wave_bracket_root(const double* q, double l, double r, double ql) {
    const int lane = threadIdx.x & 63;
    const bool lneg = ql < 0.0;
#pragma unroll 1
    for (int round = 0; round < 3; ++round) {
        const double h = (r - l) * (1.0 / 63.0);
        const double u = fma((double)lane, h, l);
        const double v = poly_eval<DEG>(q, u);
        const unsigned long long same = __ballot(v != 0.0 && (v < 0.0) == lneg);       // monotone piece: a prefix of lanes
        const unsigned long long zero = __ballot(v == 0.0);
        if (zero) return fma((double)__builtin_ctzll(zero), h, l);                       // a sample hit the root exactly
        const int idx = same ? 63 - (int)__builtin_clzll(same) : 0;                      // last lane on the left side
        l = fma((double)idx, h, l);
        r = l + h;
    }
    double dq[DEG];
#pragma unroll
    for (int i = 1; i <= DEG; ++i) dq[i - 1] = i * q[i];
    double x = 0.5 * (l + r);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const double qx = poly_eval<DEG>(q, x), d = poly_eval<DEG - 1>(dq, x);
        const double xn = fma(-qx, fast_rcp(d), x);
        x = (xn >= l && xn <= r) ? xn : x;         // NaN / runaway steps keep the bracketed iterate
    }
    return x;
}

template <int DEG> struct WaveRoots {
    __device__ __forceinline__ static void run(const double* q, double a, double b, bool* has, double* val) {
        if (q[DEG] == 0.0) {                                   // RemoveLeadingZeros (polynomial.cc)
            WaveRoots<DEG - 1>::run(q, a, b, has, val);
            has[DEG - 1] = false;
            return;
        }
        double dq[DEG], vc[DEG - 1];
        bool hc[DEG - 1];
#pragma unroll
        for (int i = 1; i <= DEG; ++i) dq[i - 1] = i * q[i];
        WaveRoots<DEG - 1>::run(dq, a, b, hc, vc);
        double l = a, ql = poly_eval<DEG>(q, a);
#pragma unroll
        for (int i = 0; i < DEG; ++i) {
            has[i] = false;
            const bool last = i == DEG - 1;
            if (last || hc[last ? 0 : i]) {                    // wave-uniform
                const double r = last ? b : vc[last ? 0 : i];
                const double qr = poly_eval<DEG>(q, r);
                if (ql == 0.0) { has[i] = true; val[i] = l; }
                else if (qr != 0.0 && (ql < 0.0) != (qr < 0.0)) { has[i] = true; val[i] = wave_bracket_root<DEG>(q, l, r, ql); }
                l = r; ql = qr;
            }
        }
    }
};
template <> struct WaveRoots<2> {      // closed form, the numerically stable pair of FindQuadraticPolynomialRoots; ascending, inside [a,b]
    __device__ __forceinline__ static void run(const double* q, double a, double b, bool* has, double* val) {
        has[0] = has[1] = false;
        if (q[2] == 0.0) {
            if (q[1] == 0.0) return;
            const double r = -q[0] / q[1];
            if (r >= a && r <= b) { has[0] = true; val[0] = r; }
            return;
        }
        const double D = q[1] * q[1] - 4.0 * q[2] * q[0];
        if (!(D >= 0.0)) return;
        const double sD = sqrt(D);
        const double t = q[1] >= 0.0 ? -q[1] - sD : -q[1] + sD;
        double r0 = t / (2.0 * q[2]), r1 = t != 0.0 ? (2.0 * q[0]) / t : r0;
        if (r0 > r1) { const double tmp = r0; r0 = r1; r1 = tmp; }
        if (r0 >= a && r0 <= b) { has[0] = true; val[0] = r0; }
        if (r1 >= a && r1 <= b && r1 != r0) { has[1] = true; val[1] = r1; }
        if (!has[0] && has[1]) { has[0] = true; val[0] = val[1]; has[1] = false; }
    }
};
template <> struct WaveRoots<1> {
    __device__ __forceinline__ static void run(const double* q, double a, double b, bool* has, double* val) {
        has[0] = false;
        if (q[1] == 0.0) return;
        const double r = -q[0] / q[1];
        if (r >= a && r <= b) { has[0] = true; val[0] = r; }
    }
};

// MinimizePolynomial (polynomial.cc): interval midpoint first, then both ends, then the critical points, strict "<".
// Called by every lane of wave 0 with the same arguments; returns the same value on every lane.
__device__ __forceinline__ double poly_min_on_wave(const double* p, double umin, double umax) {     // p: degree <= 5, ascending
    double best_u = 0.5 * (umin + umax), best_v = poly_eval<5>(p, best_u);
    const double vmin = poly_eval<5>(p, umin);
    if (vmin < best_v) { best_v = vmin; best_u = umin; }
    const double vmax = poly_eval<5>(p, umax);
    if (vmax < best_v) { best_v = vmax; best_u = umax; }
    double dp[5], val[4];
    bool has[4];
#pragma unroll
    for (int i = 1; i <= 5; ++i) dp[i - 1] = i * p[i];
    WaveRoots<4>::run(dp, umin, umax, has, val);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (has[i]) {
            const double v = poly_eval<5>(p, val[i]);
            if (v < best_v) { best_v = v; best_u = val[i]; }
        }
    return best_u;
}

struct LsSample { double x, value, gradient; bool value_ok, grad_ok; };

// LineSearch::InterpolatingPolynomialMinimizingStepSize, CUBIC: p(u) = f0 + g0 xc u + u^2 (r0 + r1 u + r2 u^2 + r3 u^3) with one
// coefficient per valid constraint {cur value, cur gradient, prev value, prev gradient}; missing constraints pin the
// highest coefficients to zero (unit rows), so the 4x4 elimination below always runs on compile-time indices.
// -> true: p[0..5] holds the interpolant in u = step / cur.x, to be minimised over [min_step, max_step] / cur.x ; false: the next
// step is the bisection step (invalid current sample, or a non-finite fit).
__device__ __forceinline__ bool interpolating_fit(double f0, double g0, const LsSample& cur, const LsSample& prev, double* p) {
    if (!cur.value_ok) return false;
    const double xc = cur.x, G0 = g0 * xc;
    const double up = prev.value_ok ? lm_div(prev.x, xc) : 2.0;
    const bool valid[4] = {true, cur.grad_ok, prev.value_ok, prev.value_ok && prev.grad_ok};
    const int m = 1 + (valid[1] ? 1 : 0) + (valid[2] ? 1 : 0) + (valid[3] ? 1 : 0);
    double A[4][5];
    {
        const double us[4] = {1.0, 1.0, up, up};
        const double rhs[4] = {cur.value - f0 - G0, cur.gradient * xc - G0, prev.value - f0 - G0 * up, prev.gradient * xc - G0};
        int pad = m;                                           // next coefficient pinned to zero
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool is_grad = (i & 1) != 0;
            double pw = is_grad ? us[i] : us[i] * us[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double e = is_grad ? (j + 2) * pw : pw;
                A[i][j] = valid[i] ? (j < m ? e : 0.0) : (j == pad ? 1.0 : 0.0);
                pw *= us[i];
            }
            A[i][4] = valid[i] ? rhs[i] : 0.0;
            if (!valid[i]) ++pad;
        }
    }
    // Gaussian elimination with partial pivoting
    bool singular = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        double pv = fabs(A[c][c]);
#pragma unroll
        for (int i = c + 1; i < 4; ++i) if (fabs(A[i][c]) > pv) { pv = fabs(A[i][c]); piv = i; }
        if (pv == 0.0) singular = true;
#pragma unroll
        for (int i = c + 1; i < 4; ++i)
            if (i == piv) {
#pragma unroll
                for (int j = 0; j < 5; ++j) { const double t = A[i][j]; A[i][j] = A[c][j]; A[c][j] = t; }
            }
        const double inv = singular ? 0.0 : lm_div(1.0, A[c][c]);
#pragma unroll
        for (int i = c + 1; i < 4; ++i) {
            const double f = A[i][c] * inv;
#pragma unroll
            for (int j = c; j < 5; ++j) A[i][j] -= f * A[c][j];
        }
    }
    double r[4] = {0.0, 0.0, 0.0, 0.0};
    if (!singular) {
#pragma unroll
        for (int c = 3; c >= 0; --c) {
            double v = A[c][4];
#pragma unroll
            for (int j = c + 1; j < 4; ++j) v -= A[c][j] * r[j];
            r[c] = lm_div(v, A[c][c]);
        }
    }
    p[0] = f0; p[1] = G0; p[2] = r[0]; p[3] = m > 1 ? r[1] : 0.0; p[4] = m > 2 ? r[2] : 0.0; p[5] = m > 3 ? r[3] : 0.0;
    bool fin = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) if (!isfinite(p[j])) fin = false;
    return fin;
}

struct Bounds { double lb[3], ub[3]; };

// Levenberg-Marquardt state of one hypothesis.  Lives ONCE per workgroup in LDS; thread 0 advances it between
// sweeps (a few hundred scalar flops), so the sweep's register budget is not shared with it.
enum { PH_INIT = 0, PH_TRIAL = 1 };
template <int NP>
struct LMState {
    double x[NP], g[NP], A[Tri<NP>::N], S[NP], diag[NP], delta[NP], xe[NP];
    double lb[NP], ub[NP];
    double cost, gmax, radius, decrease, gd, dmax, t, f1, model_change;
    int iter, nsweep, invalid_run, ls_it, phase, reuse_diag, ok1, done, max_iter;
    int want_j;        // what the NEXT sweep must produce: 2 = cost + gradient + normal equations, 1 = cost + gradient
    double g1[NP], A1[Tri<NP>::N];     // sums of the FIRST trial point (t = 1): the candidate when the line search fails
    double prev_t, prev_f, prev_g;     // previous line-search sample (step, cost, directional derivative)
    int prev_vok, prev_gok;
    double poly[6], tn;                // pending interpolant (u = step / t) for the wave-wide minimiser, and the step it returns
    int poly_req, pad_;
    int n_ls_extra, n_ls_late_accept, n_resweep;      // diagnostics: line-search trials beyond the first, accepted ones among them, re-sweeps
};

// Round 6: the LM stages fetch the state they work on from LDS IN ONE BATCH into registers (pinned: every load is issued, then one wait)
// instead of field by field between the arithmetic -- the update runs on one lane, so every LDS round trip it waits for (the copy loops
// alone were 14 read -> wait -> write pairs) is a round trip the whole workgroup waits for.  Same operations on the same values:
// bit-identical; finish + begin 8.5 k -> 7.5 k cycles per iteration (profiles/r06_c6_prepare.txt), no more scratch.
#ifdef DI2P_SOLVER_LMPROF
// variant build: cycles of the LM lane's sub-stages, summed over a hypothesis' iterations (tools/bench_solver.py LMPROF=1 prints them)
//   [0] finish_iteration  [1] begin: fetch + scaled matrix  [2] Cholesky solve  [3] model change  [4] begin: step, projection, stores
//   [5] decide without the fit  [6] interpolating fit  [7] trial_next_decide
__shared__ long long g_lmclk[8];
#define DI2P_LMCLK(i, t0) g_lmclk[i] += clock64() - (t0)
#else
#define DI2P_LMCLK(i, t0) ((void)0)
#endif
template <int N> __device__ __forceinline__ void pin_regs(double* v) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

template <int NP>
__device__ __forceinline__ void lm_begin_iteration(LMState<NP>& st) {
    constexpr int NT = Tri<NP>::N;
    const double kMinDiag = 1e-6, kMaxDiag = 1e32, kMinRadius = 1e-32, kGradTol = 1e-10;
#ifdef DI2P_SOLVER_LMPROF
    long long tb0 = clock64();
#endif
    double S[NP], A[NT], g[NP], diag[NP], sc[3];
#pragma unroll
    for (int a = 0; a < NP; ++a) { S[a] = st.S[a]; g[a] = st.g[a]; diag[a] = st.diag[a]; }
#pragma unroll
    for (int i = 0; i < NT; ++i) A[i] = st.A[i];
    sc[0] = st.gmax; sc[1] = st.radius; sc[2] = st.decrease;
    int iter = st.iter, reuse_diag = st.reuse_diag, invalid_run = st.invalid_run;
    const int max_iter = st.max_iter;
    pin_regs<NP>(S); pin_regs<NP>(g); pin_regs<NP>(diag); pin_regs<NT>(A); pin_regs<3>(sc);
    const double gmax = sc[0];
    double radius = sc[1], decrease = sc[2];
    auto scaled_A = [&](int a, int b) { const int hi = a >= b ? a : b, lo = a >= b ? b : a; return S[hi] * A[hi * (hi + 1) / 2 + lo] * S[lo]; };
    auto write_back = [&]() { st.iter = iter; st.radius = radius; st.decrease = decrease; st.reuse_diag = reuse_diag; st.invalid_run = invalid_run; };
    for (;;) {
        if (iter >= max_iter || gmax <= kGradTol || radius <= kMinRadius) { write_back(); st.done = 1; return; }
        ++iter;
        double M[NT], ds[NP];
#pragma unroll
        for (int a = 0; a < NP; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) M[a * (a + 1) / 2 + b] = scaled_A(a, b);
        if (!reuse_diag) {
#pragma unroll
            for (int a = 0; a < NP; ++a) { diag[a] = fmin(fmax(M[a * (a + 1) / 2 + a], kMinDiag), kMaxDiag); st.diag[a] = diag[a]; }
        }
#pragma unroll
        for (int a = 0; a < NP; ++a) { M[a * (a + 1) / 2 + a] += lm_div(diag[a], radius); ds[a] = -(S[a] * g[a]); }
#ifdef DI2P_SOLVER_LMPROF
        DI2P_LMCLK(1, tb0);
        const long long tc0 = clock64();
#endif
        bool valid = chol_solve_inplace<NP>(M, ds);
#ifdef DI2P_SOLVER_LMPROF
        st.n_resweep += (int)(clock64() - tc0);
        DI2P_LMCLK(2, tc0);
        const long long tm0 = clock64();
#endif
        double model_change = 0.0;
        if (valid) {
            double q = 0.0, l = 0.0;
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                l += ds[a] * (S[a] * g[a]);
#pragma unroll
                for (int b = 0; b < NP; ++b) q += ds[a] * scaled_A(a, b) * ds[b];
            }
            model_change = -(l + 0.5 * q);
            valid = model_change > 0.0;
        }
#ifdef DI2P_SOLVER_LMPROF
        DI2P_LMCLK(3, tm0);
        const long long tp0 = clock64();
#endif
        if (!valid) {
            if (++invalid_run >= 5) { write_back(); st.done = 1; return; }
            radius /= decrease; decrease *= 2.0; reuse_diag = 1;
#ifdef DI2P_SOLVER_LMPROF
            tb0 = clock64();
#endif
            continue;
        }
        invalid_run = 0;
        write_back();
        st.model_change = model_change;
        double x[NP], lb[NP], ub[NP], delta[NP];
#pragma unroll
        for (int a = 0; a < NP; ++a) { x[a] = st.x[a]; lb[a] = st.lb[a]; ub[a] = st.ub[a]; }
        pin_regs<NP>(x); pin_regs<NP>(lb); pin_regs<NP>(ub);
        double gd = 0.0, dmax = 0.0;
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            delta[a] = ds[a] * S[a];
            gd += g[a] * delta[a];
            dmax = fmax(dmax, fabs(delta[a]));
        }
        double xe[NP];
        plus_proj<NP>(x, delta, 1.0, lb, ub, xe);
#pragma unroll
        for (int a = 0; a < NP; ++a) { st.delta[a] = delta[a]; st.xe[a] = xe[a]; }
        st.gd = gd; st.dmax = dmax; st.t = 1.0; st.ls_it = 0;
        st.phase = PH_TRIAL; st.want_j = 2; st.prev_vok = 0; st.prev_gok = 0;
#ifdef DI2P_SOLVER_LMPROF
        DI2P_LMCLK(4, tp0);
#endif
        return;   // needs a sweep at xe
    }
}

// The LM update is a chain of STAGES, each instantiated exactly once in the kernel (lm_decide -> [wave-wide minimiser] ->
// lm_trial_next_decide -> lm_apply = {lm_finish_iteration, lm_begin_iteration}); a stage hands the next one an action code.  (As
// mutually calling inline functions the iteration start was instantiated five times: 13 k instructions, and the register allocator
// spilled inside the sweep loops.)
enum { ACT_NONE = 0, ACT_BEGIN = 1, ACT_FINISH_CUR = 2, ACT_FINISH_FIRST = 3, ACT_TRIAL_NEXT = 4, ACT_POLY = 5 };

// candidate (xe, cand_cost, ge, Ae) against the current iterate: tolerance tests, accept / reject.  -> true: start the next iteration
template <int NP>
__device__ __forceinline__ bool lm_finish_iteration(LMState<NP>& st, double cand_cost, const double* ge, const double* Ae) {
    constexpr int NT = Tri<NP>::N;
    const double kMaxRadius = 1e16, kMinRelDec = 1e-3, kFuncTol = 1e-6, kParamTol = 1e-8;
    double xo[NP], xn[NP], gl[NP], Al[NT], lb[NP], ub[NP], sc[4];
#pragma unroll
    for (int a = 0; a < NP; ++a) { xo[a] = st.x[a]; xn[a] = st.xe[a]; gl[a] = ge[a]; lb[a] = st.lb[a]; ub[a] = st.ub[a]; }
#pragma unroll
    for (int i = 0; i < NT; ++i) Al[i] = Ae[i];
    sc[0] = st.cost; sc[1] = st.model_change; sc[2] = st.radius; sc[3] = st.decrease;
    pin_regs<NP>(xo); pin_regs<NP>(xn); pin_regs<NP>(gl); pin_regs<NT>(Al); pin_regs<NP>(lb); pin_regs<NP>(ub); pin_regs<4>(sc);
    double step_norm = 0.0, x_norm = 0.0;
#pragma unroll
    for (int a = 0; a < NP; ++a) { step_norm += (xo[a] - xn[a]) * (xo[a] - xn[a]); x_norm += xo[a] * xo[a]; }
    step_norm = lm_sqrt(step_norm); x_norm = lm_sqrt(x_norm);
    if (step_norm <= kParamTol * (x_norm + kParamTol)) { st.done = 1; return false; }
    if (fabs(sc[0] - cand_cost) <= kFuncTol * sc[0]) { st.done = 1; return false; }
    const double rel = lm_div(sc[0] - cand_cost, sc[1]);
    if (rel > kMinRelDec) {
#pragma unroll
        for (int a = 0; a < NP; ++a) { st.x[a] = xn[a]; st.g[a] = gl[a]; }
#pragma unroll
        for (int i = 0; i < NT; ++i) st.A[i] = Al[i];
        st.cost = cand_cost;
        st.gmax = grad_max_norm<NP>(xn, gl, lb, ub);
        const double w = 2.0 * rel - 1.0;
        st.radius = fmin(kMaxRadius, lm_div(sc[2], fmax(1.0 / 3.0, 1.0 - w * w * w)));
        st.decrease = 2.0; st.reuse_diag = 0;
    } else {
        st.radius = sc[2] / sc[3]; st.decrease = sc[3] * 2.0; st.reuse_diag = 1;
    }
    return true;
}

// Second half of a failed line-search trial: the next step size st.tn is known.  -> ACT_FINISH_FIRST when the search gives up
// (the candidate is then the first trial point, whose sums were kept), ACT_NONE when the next sweep evaluates the new trial point.
template <int NP>
__device__ __forceinline__ int lm_trial_next_decide(LMState<NP>& st) {
    const double tn = st.tn;
    if (tn * st.dmax < 1e-9) return ACT_FINISH_FIRST;
    st.t = tn;
    plus_proj<NP>(st.x, st.delta, tn, st.lb, st.ub, st.xe);
    return ACT_NONE;
}

// Last stage: finish the iteration with the chosen candidate (the point just swept, or the first trial point) and start the next one.
template <int NP>
__device__ __forceinline__ void lm_apply(LMState<NP>& st, int action, double fe, const double* ge, const double* Ae) {
    bool begin = action == ACT_BEGIN;
    if (action == ACT_FINISH_CUR || action == ACT_FINISH_FIRST) {
#ifdef DI2P_SOLVER_LMPROF
        const long long tf0 = clock64();
#endif
        const bool first = action == ACT_FINISH_FIRST;
        if (first) plus_proj<NP>(st.x, st.delta, 1.0, st.lb, st.ub, st.xe);     // delta stays unscaled: back to the first trial point
        begin = lm_finish_iteration<NP>(st, first ? st.f1 : fe, first ? st.g1 : ge, first ? st.A1 : Ae);
#ifdef DI2P_SOLVER_LMPROF
        DI2P_LMCLK(0, tf0);
#endif
    }
    if (begin) lm_begin_iteration<NP>(st);
}

// Wave 0, all lanes: minimise the pending interpolant (MinimizePolynomial over u in [1e-3 t, 0.6 t] / t).
template <int NP>
__device__ __forceinline__ void lm_poly_wave(LMState<NP>& st) {
    double p[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) p[j] = st.poly[j];
    const double t = st.t;
    const double u = poly_min_on_wave(p, (1e-3 * t) / t, (0.6 * t) / t);
    if ((threadIdx.x & 63) == 0) st.tn = u * t;
}

// First stage, called by thread 0 after every sweep with the combined sums of the point just evaluated (st.xe).  -> action code
template <int NP>
__device__ __forceinline__ int lm_decide(LMState<NP>& st, bool ok, double fe, const double* ge_in, const double* Ae_in) {
    double ge[NP], Ae[Tri<NP>::N];       // the combined sums, fetched in one batch (copied field by field they were read -> wait -> write pairs)
#pragma unroll
    for (int a = 0; a < NP; ++a) ge[a] = ge_in[a];
#pragma unroll
    for (int i = 0; i < Tri<NP>::N; ++i) Ae[i] = Ae_in[i];
    pin_regs<NP>(ge); pin_regs<Tri<NP>::N>(Ae);
    ++st.nsweep;
    if (st.phase == PH_INIT) {
        st.cost = fe;
        if (!ok) { st.done = 1; return ACT_NONE; }
#pragma unroll
        for (int a = 0; a < NP; ++a) { st.g[a] = ge[a]; st.S[a] = 1.0 / (1.0 + sqrt(Ae[a * (a + 1) / 2 + a])); }
#pragma unroll
        for (int i = 0; i < Tri<NP>::N; ++i) st.A[i] = Ae[i];
        st.gmax = grad_max_norm<NP>(st.x, st.g, st.lb, st.ub);
        return ACT_BEGIN;
    }
    // PH_TRIAL: projected Armijo search along delta (Ceres ArmijoLineSearch, CUBIC interpolation: every trial needs its
    // gradient).  Every sweep also carries the normal equations (10 more fma per Jacobian row, ~3 % of a sweep), so whichever
    // trial satisfies Armijo is used at once; the sums of the first trial (t = 1) are kept: it is the candidate when the
    // search fails.
    if (st.ls_it == 0) {
        st.f1 = ok ? fe : DBL_MAX; st.ok1 = ok;
#pragma unroll
        for (int a = 0; a < NP; ++a) st.g1[a] = ge[a];
#pragma unroll
        for (int i = 0; i < Tri<NP>::N; ++i) st.A1[i] = Ae[i];
    } else {
        ++st.n_ls_extra;
    }
    if (ok && fe <= st.cost + 1e-4 * st.gd * st.t) {
        if (st.ls_it > 0) ++st.n_ls_late_accept;
        return ACT_FINISH_CUR;       // every sweep carries its normal equations: an accepted trial needs no second sweep
    }
    if (++st.ls_it >= 20) return ACT_FINISH_FIRST;      // the search gives up
    LsSample cur{st.t, fe, 0.0, ok, false}, prev{st.prev_t, st.prev_f, st.prev_g, st.prev_vok != 0, st.prev_gok != 0};
    if (ok) {
        double gdir = 0.0;
#pragma unroll
        for (int a = 0; a < NP; ++a) gdir += st.delta[a] * ge[a];
        cur.gradient = gdir;
        cur.grad_ok = isfinite(gdir);
    }
    st.prev_t = cur.x; st.prev_f = cur.value; st.prev_g = cur.gradient; st.prev_vok = cur.value_ok; st.prev_gok = cur.grad_ok;
    double p[6];
#ifdef DI2P_SOLVER_LMPROF
    const long long tfit0 = clock64();
    const bool fitted = interpolating_fit(st.cost, st.gd, cur, prev, p);
    DI2P_LMCLK(6, tfit0);
    if (fitted) {
#else
    if (interpolating_fit(st.cost, st.gd, cur, prev, p)) {
#endif
        // the minimiser of the interpolant over [1e-3, 0.6] x t is found by the whole wavefront (lm_poly_wave), then lm_trial_next_decide
#pragma unroll
        for (int j = 0; j < 6; ++j) st.poly[j] = p[j];
        st.poly_req = 1;
        return ACT_POLY;
    }
    st.tn = fmin(fmax(st.t * 0.5, 1e-3 * st.t), 0.6 * st.t);
    return ACT_TRIAL_NEXT;
}

// Kernel arguments as ONE struct: the kernel reads every field through the kernarg segment pointer (scalar loads from constant
// memory at the point of use), so the fields that are only needed before / after the sweep loop (outputs, start values, the
// diagnostics buffers) do not sit in SGPRs through it -- as plain kernel parameters hipcc loaded all 30 of them at entry and kept
// them live, which was a third of the kernel's SGPR spills.
template <typename PT>
struct SolveArgs {
    const Rec<PT>* packed;
    const Box* boxes_all;
    const int* counts;
    const double* Kmat;
    const double* init_y;
    const double* init_T;
    const double* yaw0;
    double* params_out;
    double* cost_out;
    int* iters_out;
    int* sweeps_out;
    long long* prof;
    unsigned long long* state_buf;
    int* pending;
    CacheEnt* cache;           // classification cache: [F * R][NCMAX + CACHE_PAD]
    const float* camf;         // [F][8] normalised frustum-plane coefficients (prepare_kernel)
    double H, W;
    Bounds bnd;
    int NCMAX, nocull, max_iter, F, R, N, budget, resume;
};

template <int NP, typename PT, int MINW, int WPH, bool PROFILE>
__global__ __launch_bounds__(WPH * 64, MINW) void solve_kernel(const SolveArgs<PT> args_by_value) {
    constexpr int TOFF = NP == 4 ? 1 : 3;
    constexpr int NT = Tri<NP>::N;
    constexpr int NV = NT + NP + 2;
    (void)args_by_value;
    typedef const SolveArgs<PT> __attribute__((address_space(4))) * KernArgPtr;
    const KernArgPtr a = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    // 1-D grid, frame = block % F: the dispatcher places block b on XCD b % 8, so (for F % 8 == 0) all the
    // hypotheses of a frame share one XCD's L2 and the frame's records are fetched from HBM once.
    const int F = a->F, R = a->R;
    const int f = blockIdx.x % F;
    const int r = blockIdx.x / F;            // one WPH-wave workgroup per hypothesis
    __shared__ SweepShared<NP, WPH> sh;
    __shared__ LMState<NP> st;
    const Rec<PT>* recs = a->packed + (long long)f * (a->N + 2 * CL);
    const Box* boxes = a->boxes_all + (long long)f * a->NCMAX;
    const int* counts = a->counts;
    const int cnt1 = counts[4 * f], cnt0 = counts[4 * f + 1], nc1 = counts[4 * f + 2], nc0 = counts[4 * f + 3];
    const double* Kf = a->Kmat + (long long)f * 9;
    const float* camf = a->camf + (long long)f * 8;       // the frame's normalised plane coefficients (prepare_kernel)
    const int nocull = a->nocull;
    const long long hr = (long long)f * R + r;
    CacheEnt* cache = a->cache + hr * (a->NCMAX + CACHE_PAD);
    // empty classification cache (also on resume: the wide tier maps clusters to waves differently)
    for (int i = threadIdx.x; i < a->NCMAX + CACHE_PAD; i += WPH * 64) { CacheEnt z; z.mlo = z.mhi = 0u; z.slack = 0.0f; z.stamp = 0u; cache[i] = z; }
    constexpr int ST_WORDS = (int)(sizeof(LMState<NP>) / 8);
    static_assert(sizeof(LMState<NP>) % 8 == 0, "LMState is copied as 8-byte words");
    // Two-tier launch (see launch_solve): the first launch stops a hypothesis after `budget` sweeps and parks its LM state;
    // the second launch (more waves per hypothesis) resumes the parked ones and leaves the finished ones alone.
    if (a->resume) {
        if (a->pending[hr] == 0) return;            // finished in the first tier (workgroup-uniform)
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&st);
        const unsigned long long* src = a->state_buf;
        for (int i = threadIdx.x; i < ST_WORDS; i += WPH * 64) dst[i] = src[hr * ST_WORDS + i];
    } else if (threadIdx.x == 0) {
        for (int i = 0; i < NP; ++i) { st.lb[i] = -DBL_MAX; st.ub[i] = DBL_MAX; }
        for (int i = 0; i < 3; ++i) { st.lb[TOFF + i] = a->bnd.lb[i]; st.ub[TOFF + i] = a->bnd.ub[i]; }
        const double* yaw0 = a->yaw0;
        const double y0 = a->init_y[hr] + (yaw0 ? yaw0[f] : 0.0);
        if (NP == 4) { st.x[0] = y0; } else { st.x[0] = 0.0; st.x[1] = y0; st.x[2] = 0.0; }
        for (int i = 0; i < 3; ++i) st.x[TOFF + i] = a->init_T[hr * 3 + i];
        for (int i = 0; i < NP; ++i) { st.x[i] = fmin(fmax(st.x[i], st.lb[i]), st.ub[i]); st.xe[i] = st.x[i]; sh.ring[0][i] = st.x[i]; }
        st.radius = 1e4; st.decrease = 2.0; st.reuse_diag = 0; st.invalid_run = 0; st.iter = 0; st.nsweep = 0;
        st.phase = PH_INIT; st.done = 0; st.max_iter = a->max_iter; st.cost = 0.0; st.gmax = 0.0;
        st.n_ls_extra = 0; st.n_ls_late_accept = 0; st.n_resweep = 0; st.pad_ = 0; st.want_j = 2; st.poly_req = 0;
#ifdef DI2P_SOLVER_LMPROF
        for (int i = 0; i < 8; ++i) g_lmclk[i] = 0;
#endif
    }
    __syncthreads();
    long long c_sweep = 0, c_wait = 0, c_lm = 0, c_comb = 0;
    // wave 0: [0] phase-B evaluations, [1..3] clusters classified per point / all active / guard-only, [4..5] cache hits {classification, guard},
    // [6..7] straight-line batches of the guard / classification walk, [8] clusters appended by phase II, [9..10] phase-B rounds {label 1, label 0},
    // [11] cluster-test rounds, [12] guard-only clusters walked on the top / bottom / z planes only (status 5; included in [3])
    int n_act[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // sweep numbers of the classification cache count from this launch's first sweep (resume: the cache and the ring start empty)
    const int s_base = a->resume ? st.nsweep : 0;
    if (threadIdx.x == 0) {
        double xn[NP];
        for (int i = 0; i < NP; ++i) { xn[i] = st.xe[i]; sh.ring[0][i] = xn[i]; }
        if (NP == 4) { Rot<NP> r0; make_rot<NP>(xn, r0); sh.rot_cs[0] = r0.R[0]; sh.rot_cs[1] = r0.R[2]; }
    }
    __syncthreads();
    long long tp[4] = {0, 0, 0, 0}; // wave 0, inside the sweep: cluster-test rounds, drains (phase B), set-up, reduction
    long long c_decide = 0, c_poly = 0, c_apply = 0;
    for (;;) {
        double xe[NP];
        asm volatile("" ::: "memory");          // the iterate is READ from LDS here by every thread (nothing carried in registers around the loop)
#pragma unroll
        for (int i = 0; i < NP; ++i) xe[i] = st.xe[i];
        const int s_now = st.nsweep - s_base;
        // The LM state is advanced by wave 0.  (Round 6 measured rotating that wave with the sweep number -- the waves of a workgroup sit on
        // the four SIMDs of a compute unit, so wave 0 of every resident workgroup shares one SIMD: bit-identical, and 1-1.5 % SLOWER both alone
        // and in the pipeline, profiles/r06_c4 / c5: not adopted.)
        const bool lm_wave_here = threadIdx.x < 64;
        const int lm_lane = (int)(threadIdx.x & 63);
        // the camera is re-read (scalar loads, cache hits) at the head of every sweep through a laundered pointer: hoisted out of the sweep
        // loop its twelve dwords were kept in VGPRs and SPILLED (nine scratch reloads per sweep)
        const double* Kq = Kf;
        asm volatile("" : "+s"(Kq));
        const Cam k{Kq[0], Kq[4], Kq[2], Kq[5], a->H - 1.0, a->W - 1.0};
        const long long t0 = PROFILE ? clock64() : 0;
        const float* cfq = camf;
        asm volatile("" : "+s"(cfq));
        sweep<NP, PT, WPH, 2, PROFILE>(recs, boxes, cnt1, cnt0, nc1, nc0, k, cfq, xe, nocull, sh, cache, s_now, n_act, tp);
        const long long t1 = PROFILE ? clock64() : 0;
        __syncthreads();
        const long long t2 = PROFILE ? clock64() : 0;
        if (lm_wave_here) {                     // the LM wave of this sweep: fixed-order combination of the wave partials, one value per lane
            const int i = lm_lane < NV ? lm_lane : NV - 1;
            double part[WPH];
            int ex = 0;
#pragma unroll
            for (int w = 0; w < WPH; ++w) { part[w] = sh.red[w][i]; ex += sh.red_e[w]; }       // all reads in flight together
            double sum = part[0], prod = part[0];
#pragma unroll
            for (int w = 1; w < WPH; ++w) { sum += part[w]; prod *= part[w]; }                 // branch-free: both forms, then a select
            // value 0: the waves' cost products multiply (>= 2^(-64 WPH): no underflow), rho sum = ln(product), halved; evaluated by every
            // lane (no divergent branch; lanes > 0 feed it their harmless sums); a non-finite / non-positive product gives NaN, caught below
            const double cost = 0.5 * (ln_pos(i == 0 ? prod : 1.0) + (double)ex * 0.69314718055994530942);
            double t = i == 0 ? cost : sum;
            // a non-finite Jacobian entry or residual (evaluation failure in the reference) makes a sum non-finite: one vote per sweep
            const bool nonfinite = __any(lm_lane < NV - 1 && !isfinite(t)) != 0;
            if (lm_lane == NV - 1 && nonfinite) t = 1.0;
            if (lm_lane < NV) sh.comb[lm_lane] = t;
        }
        __builtin_amdgcn_wave_barrier();        // same wave: its LDS operations retire in order
        const long long t2b = PROFILE ? clock64() : 0;
        int action = ACT_NONE;
        if (lm_wave_here && lm_lane == 0) {
            const bool ok = sh.comb[NV - 1] == 0.0 && isfinite(sh.comb[0]);
#ifdef DI2P_SOLVER_LMPROF
            const long long td0_ = clock64();
#endif
            action = lm_decide<NP>(st, ok, sh.comb[0], sh.comb + 1, sh.comb + 1 + NP);
#ifdef DI2P_SOLVER_LMPROF
            DI2P_LMCLK(5, td0_);
#endif
        }
        const long long t2c = PROFILE ? clock64() : 0;
        asm volatile("" ::: "memory");          // the interpolant is read back from LDS by all 64 lanes (not forwarded from lane 0's registers)
        if (lm_wave_here) {                     // the LM wave: a failed trial left an interpolant to minimise (wave-uniform branch)
            __builtin_amdgcn_wave_barrier();
            if (st.poly_req) {
                lm_poly_wave<NP>(st);
                __builtin_amdgcn_wave_barrier();
                if (lm_lane == 0) { st.poly_req = 0; action = ACT_TRIAL_NEXT; }
            }
        }
        const long long t2d = PROFILE ? clock64() : 0;
        // the last stage re-reads the combined sums from LDS: kept in registers across the wave-wide minimiser (the compiler merges these
        // loads with lm_decide's) they were spilled -- three scratch reloads with a full wait on the lane everybody waits for, four times per sweep
        asm volatile("" ::: "memory");
        if (lm_wave_here && lm_lane == 0) {
#ifdef DI2P_SOLVER_LMPROF
            const long long ttn0 = clock64();
#endif
            if (action == ACT_TRIAL_NEXT) action = lm_trial_next_decide<NP>(st);
#ifdef DI2P_SOLVER_LMPROF
            DI2P_LMCLK(7, ttn0);
#endif
            lm_apply<NP>(st, action, sh.comb[0], sh.comb + 1, sh.comb + 1 + NP);
            // the iterate of the NEXT sweep enters the ring of the classification cache (slot = its sweep number % RING)
            const int slot = (st.nsweep - s_base) & (RING - 1);
            double xn[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) { xn[i] = st.xe[i]; sh.ring[slot][i] = xn[i]; }
#ifdef DI2P_SOLVER_LMPROF
            const long long tr0 = clock64();
#endif
            if (NP == 4 && !st.done) { Rot<NP> r0; make_rot<NP>(xn, r0); sh.rot_cs[0] = r0.R[0]; sh.rot_cs[1] = r0.R[2]; }
#ifdef DI2P_SOLVER_LMPROF
            st.pad_ += (int)(clock64() - tr0);
#endif
        }
        const long long t3 = PROFILE ? clock64() : 0;
        c_comb += t2b - t2; c_decide += t2c - t2b; c_poly += t2d - t2c; c_apply += t3 - t2d;
        __syncthreads();
        c_sweep += t1 - t0; c_wait += t2 - t1; c_lm += t3 - t2;
        if (st.done) break;
        const int budget = a->budget;
        if (budget > 0 && st.nsweep >= budget) break;       // parked for the wide tier (workgroup-uniform: st is in LDS)
    }
    int* pending = a->pending;
    if (pending) {
        if (threadIdx.x == 0) pending[hr] = st.done ? 0 : 1;
        if (!st.done) {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&st);
            unsigned long long* state_buf = a->state_buf;
            for (int i = threadIdx.x; i < ST_WORDS; i += WPH * 64) state_buf[hr * ST_WORDS + i] = src[i];
        }
    }
    if (PROFILE && threadIdx.x == 0) {   // diagnostics, PROF_WORDS int64 per hypothesis (wave 0's shader-clock cycles and counts; see di2p_solver_set_profile_buffer)
        long long* prof = a->prof + hr * PROF_WORDS;
        long long v[PROF_WORDS] = {c_sweep, c_wait, c_lm, n_act[0], n_act[1], n_act[2], c_comb,
                           (long long)st.n_ls_extra | ((long long)st.n_ls_late_accept << 20) | ((long long)st.n_resweep << 40),
                           c_decide, c_poly, c_apply, n_act[3], tp[0], tp[1], tp[2], tp[3], n_act[4], n_act[5],
                           n_act[6], n_act[7], n_act[8], n_act[9], n_act[10], n_act[11], n_act[12], (long long)st.pad_, 0, 0};
#ifdef DI2P_SOLVER_LMPROF
        for (int i = 0; i < 8; ++i) v[28 + i] = g_lmclk[i];
#endif
        for (int i = 0; i < PROF_WORDS; ++i) prof[i] = (a->resume && i != 7 ? prof[i] : 0) + v[i];
    }
    if (threadIdx.x == 0 && st.done) {
        double* params_out = a->params_out;
        for (int i = 0; i < NP; ++i) params_out[hr * NP + i] = st.x[i];
        a->cost_out[hr] = st.cost;
        a->iters_out[hr] = st.iter;
        int* sweeps_out = a->sweeps_out;
        if (sweeps_out) sweeps_out[hr] = st.nsweep;
    }
}

__device__ __forceinline__ void angle_axis_to_R(const double* w, double* R) {
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (t2 > DBL_EPSILON) {
        const double t = sqrt(t2), wx = w[0] / t, wy = w[1] / t, wz = w[2] / t, c = cos(t), s = sin(t);
        R[0] = c + wx * wx * (1 - c);       R[1] = wx * wy * (1 - c) - wz * s;  R[2] = wy * s + wx * wz * (1 - c);
        R[3] = wz * s + wx * wy * (1 - c);  R[4] = c + wy * wy * (1 - c);       R[5] = -wx * s + wy * wz * (1 - c);
        R[6] = -wy * s + wx * wz * (1 - c); R[7] = wx * s + wy * wz * (1 - c);  R[8] = c + wz * wz * (1 - c);
    } else {
        R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
    }
}

// one wavefront per frame: argmin over R (ties -> lowest r), assemble P
__global__ __launch_bounds__(64) void select_best_kernel(const double* __restrict__ params, const double* __restrict__ cost,
                                                         const int* __restrict__ has_inside, int np, int R, int* __restrict__ best,
                                                         double* __restrict__ P, double* __restrict__ best_cost) {
    const int f = blockIdx.x, lane = threadIdx.x;
    double bc = __builtin_inf();
    int bi = 0x7fffffff;
    for (int r = lane; r < R; r += 64) {
        const double c = cost[(long long)f * R + r];
        if (c < bc || (c == bc && r < bi)) { bc = c; bi = r; }   // NaN never wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double oc = __shfl_xor(bc, o);
        const int oi = __shfl_xor(bi, o);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (lane != 0) return;
    double* Pf = P + (long long)f * 16;
    for (int i = 0; i < 16; ++i) Pf[i] = (i % 5 == 0) ? 1.0 : 0.0;
    if (has_inside && has_inside[f] == 0) {  // registration_lsq.py:329-332
        best[f] = -1;
        best_cost[f] = 1e4;
        return;
    }
    if (bi == 0x7fffffff) bi = 0;
    const double* x = params + ((long long)f * R + bi) * np;
    double w[3] = {0, 0, 0};
    int toff;
    if (np == 4) { w[1] = x[0]; toff = 1; } else { w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; toff = 3; }
    double Rm[9];
    angle_axis_to_R(w, Rm);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Pf[r * 4 + c] = Rm[r * 3 + c]; Pf[r * 4 + 3] = x[toff + r]; }
    best[f] = bi;
    best_cost[f] = cost[(long long)f * R + bi];
}

// registration_lsq.py:196-220.  One 1024-thread workgroup per frame, fixed-order tree reductions.
__global__ __launch_bounds__(1024) void initial_guess_kernel(const double* __restrict__ points, const int* __restrict__ labels,
                                                             double* __restrict__ yaw0, int* __restrict__ labels_out,
                                                             int* __restrict__ has_inside, int N) {
    __shared__ double s_a[1024], s_b[1024], s_c[1024];
    const int f = blockIdx.x, tid = threadIdx.x;
    const double* px = points + (long long)f * 3 * N;
    const double* pz = px + 2 * (long long)N;
    const int* lab = labels + (long long)f * N;
    double sx = 0, sz = 0, cnt = 0;
    for (int n = tid; n < N; n += 1024)
        if (lab[n] == 1) { sx += px[n]; sz += pz[n]; cnt += 1.0; }
    s_a[tid] = sx; s_b[tid] = sz; s_c[tid] = cnt;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { s_a[tid] += s_a[tid + o]; s_b[tid] += s_b[tid + o]; s_c[tid] += s_c[tid + o]; }
        __syncthreads();
    }
    const double count = s_c[0];
    const double mx = s_a[0] / count, mz = s_b[0] / count;
    __syncthreads();
    if (!(count > 0.0)) {
        if (tid == 0) { yaw0[f] = 0.0; has_inside[f] = 0; }
        for (int n = tid; n < N; n += 1024) labels_out[(long long)f * N + n] = lab[n];
        return;
    }
    const double kPi = 3.14159265358979323846;
    double a = fmod(atan2(mz, mx) - kPi / 2 + kPi, 2 * kPi);   // wrap_in_pi (:189-193)
    if (a < 0) a += 2 * kPi;
    a -= kPi;
    const double c = cos(a), s = sin(a);
    double zmin = __builtin_inf();
    for (int n = tid; n < N; n += 1024)
        if (lab[n] == 1) zmin = fmin(zmin, -s * px[n] + c * pz[n]);
    s_a[tid] = zmin;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) s_a[tid] = fmin(s_a[tid], s_a[tid + o]);
        __syncthreads();
    }
    const double thr = s_a[0] - 10.0;
    for (int n = tid; n < N; n += 1024) {
        const double zr = -s * px[n] + c * pz[n];
        labels_out[(long long)f * N + n] = zr > thr ? lab[n] : -1;
    }
    if (tid == 0) { yaw0[f] = a; has_inside[f] = 1; }
}

// Problem::Evaluate (registration.cpp:150-155): loss-corrected residuals in point order, compacted.
template <int NP>
__global__ __launch_bounds__(256) void residuals_kernel(const double* __restrict__ points, const int* __restrict__ labels,
                                                        const double* __restrict__ Kmat, const double* __restrict__ params, double H,
                                                        double W, int N, double* __restrict__ residuals, int* __restrict__ counts,
                                                        double* __restrict__ cost_out) {
    __shared__ int s_scan[256];
    __shared__ double s_cost[256];
    __shared__ int s_base;
    constexpr int TOFF = NP == 4 ? 1 : 3;
    const int f = blockIdx.x, tid = threadIdx.x;
    const double* px = points + (long long)f * 3 * N;
    const int* lab = labels + (long long)f * N;
    const double* Kf = Kmat + (long long)f * 9;
    const Cam k{Kf[0], Kf[4], Kf[2], Kf[5], H - 1.0, W - 1.0};
    const double* x = params + (long long)f * NP;
    double xr[NP];
    for (int i = 0; i < NP; ++i) xr[i] = x[i];
    Rot<NP> rot;
    make_rot<NP>(xr, rot);
    double* out = residuals + (long long)f * 3 * N;
    if (tid == 0) s_base = 0;
    double cost = 0.0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += 256) {
        const int n = n0 + tid;
        int nr = 0;
        double rv[3] = {0, 0, 0};
        if (n < N && (lab[n] == 0 || lab[n] == 1)) {
            const double X = px[n], Y = px[N + n], Z = px[2 * (long long)N + n];
            const double qx = rot.R[0] * X + rot.R[1] * Y + rot.R[2] * Z, qy = rot.R[3] * X + rot.R[4] * Y + rot.R[5] * Z,
                         qz = rot.R[6] * X + rot.R[7] * Y + rot.R[8] * Z;
            const double p0 = qx + xr[TOFF], p1 = qy + xr[TOFF + 1], p2 = qz + xr[TOFF + 2];
            const double pix_x = p0 * k.fx / p2 + k.cx, pix_y = p1 * k.fy / p2 + k.cy;
            if (lab[n] == 1) {
                nr = 3;
                rv[0] = fmax(-pix_x, 0.0) + fmax(pix_x - k.W1, 0.0);
                rv[1] = fmax(-pix_y, 0.0) + fmax(pix_y - k.H1, 0.0);
                rv[2] = fmax(-p2, 0.0) * 100.0;
            } else {
                nr = 1;
                const double dx = k.W1 * 0.5 - fabs(pix_x - k.W1 * 0.5), dy = k.H1 * 0.5 - fabs(pix_y - k.H1 * 0.5);
                rv[0] = (dx + dy) * (fmax(p2, 0.0) / p2) * (fmax(dx, 0.0) / dx) * (fmax(dy, 0.0) / dy);
            }
            const double s = rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2];
            cost += 0.5 * log1p(s);
            const double sq = sqrt(1.0 / (1.0 + s));
            for (int i = 0; i < 3; ++i) rv[i] *= sq;
        }
        s_scan[tid] = nr;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = tid >= o ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int excl = s_scan[tid] - nr + s_base;
        for (int i = 0; i < nr; ++i) out[excl + i] = rv[i];
        __syncthreads();
        if (tid == 255) s_base += s_scan[255];
        __syncthreads();
    }
    s_cost[tid] = cost;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_cost[tid] += s_cost[tid + o];
        __syncthreads();
    }
    if (tid == 0) { counts[f] = s_base; cost_out[f] = s_cost[0]; }
}

static long long* g_prof = nullptr;   // diagnostics hook, see di2p_solver_set_profile_buffer

constexpr size_t kStateBytes = sizeof(LMState<6>) > sizeof(LMState<4>) ? sizeof(LMState<6>) : sizeof(LMState<4>);
struct SolveWs { int P, NCMAX; size_t off_recs, off_boxes, off_keys, off_pending, off_state, off_cache, off_camf, off_partial, off_cntg, off_bases, off_flag, bytes; };
static SolveWs solve_ws_layout(int F, int R, int N) {
    SolveWs w;
    w.P = 64;
    while (w.P < N) w.P <<= 1;
    w.NCMAX = (N + CL - 1) / CL + 2;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    w.off_recs = up((size_t)F * 4 * sizeof(int));
    w.off_boxes = up(w.off_recs + (size_t)F * (N + 2 * CL) * 32);
    w.off_keys = up(w.off_boxes + (size_t)F * w.NCMAX * sizeof(Box));
    w.off_pending = up(w.off_keys + (size_t)F * 2 * w.P * 8);
    w.off_state = up(w.off_pending + (size_t)F * R * sizeof(int));
    w.off_cache = up(w.off_state + (size_t)F * R * kStateBytes);
    w.off_camf = up(w.off_cache + (size_t)F * R * (w.NCMAX + CACHE_PAD) * sizeof(CacheEnt));
    w.off_partial = up(w.off_camf + (size_t)F * 8 * sizeof(float));
    w.off_cntg = up(w.off_partial + (size_t)F * PREP_G * 8 * sizeof(float));
    w.off_bases = up(w.off_cntg + (size_t)F * PREP_G * PREP_NBK * sizeof(int));
    w.off_flag = up(w.off_bases + (size_t)F * (PREP_NBK + 1) * sizeof(int));
    w.bytes = up(w.off_flag + (size_t)F * sizeof(int)) + 256;
    return w;
}

template <typename PT>
int launch_solve(const PT* points, const int* labels, const double* K, const double* init_y, const double* init_T,
                 const double* yaw0, double H, double W, const double* lb, const double* ub, int max_iter, int is_2d, int F,
                 int R, int N, double* params, double* cost, int* iters, int* sweeps, void* workspace, hipStream_t st) {
    Bounds b;
    for (int i = 0; i < 3; ++i) { b.lb[i] = lb[i]; b.ub[i] = ub[i]; }
    // workspace: counts i32[F][4] | records Rec[F][N] | cluster boxes [F][NCMAX] | sort keys u64[F][P] | pending i32[F][R] | parked LM states [F][R] |
    // classification cache [F][R][NCMAX + CACHE_PAD] | normalised plane coefficients f32[F][8] | multi-workgroup preparation: partial bounds
    // f32[F][G][8], bucket counts per slice i32[F][G][2048], bucket offsets i32[F][2049], fallback flag i32[F]
    const SolveWs ws = solve_ws_layout(F, R, N);
    char* base = (char*)workspace;
    int* counts = (int*)base;
    Rec<PT>* packed = (Rec<PT>*)(base + ws.off_recs);
    Box* boxes = (Box*)(base + ws.off_boxes);
    unsigned long long* keys = (unsigned long long*)(base + ws.off_keys);
    float* camf = (float*)(base + ws.off_camf);
    // frame preparation: five multi-workgroup launches + the single-workgroup kernel as the fallback of flagged (degenerate) frames;
    // solver_prep_bitonic = 1: the bitonic network for every frame, solver_prep_single = 1: the round-4 single-workgroup kernel (same results)
    const int prep_bitonic = (int)(di2p_opt(DI2P_OPT_SOLVER_PREP_BITONIC) != 0);
    if (prep_bitonic || di2p_opt(DI2P_OPT_SOLVER_PREP_SINGLE) != 0) {
        hipLaunchKernelGGL(prepare_kernel<PT>, dim3(F), dim3(1024), 0, st, points, labels, N, ws.P, ws.NCMAX, keys, packed, boxes, counts,
                           prep_bitonic, K, H, W, camf, (const int*)nullptr);
    } else {
        static_assert(PREP_NBK % PREP_G == 0 && PREP_NBK == 2048, "the slice kernels zero / scan the histogram in these shapes");
        PrepWs pw;
        pw.partial = (float*)(base + ws.off_partial); pw.cntg = (int*)(base + ws.off_cntg);
        pw.bases = (int*)(base + ws.off_bases); pw.flag = (int*)(base + ws.off_flag);
        hipLaunchKernelGGL(prep_bounds_kernel<PT>, dim3(F * PREP_G), dim3(256), 0, st, points, labels, N, pw);
        hipLaunchKernelGGL(prep_hist_kernel<PT>, dim3(F * PREP_G), dim3(256), 0, st, points, labels, N, ws.P, keys, pw);
        hipLaunchKernelGGL(prep_scatter_kernel, dim3(F * PREP_G), dim3(1024), 0, st, N, ws.P, keys, pw);
        hipLaunchKernelGGL(prep_rank_kernel, dim3(N > 0 ? (N + 255) / 256 : 1, F), dim3(256), 0, st, ws.P, keys, pw);      // (an empty cloud still launches: a grid of 0 is an error)
        hipLaunchKernelGGL(prep_records_kernel<PT>, dim3((ws.NCMAX + 4 * PREP_CPW - 1) / (4 * PREP_CPW), F), dim3(256), 0, st, points, labels, N, ws.P,
                           ws.NCMAX, (const unsigned long long*)keys, packed, boxes, counts, K, H, W, camf, pw);
        hipLaunchKernelGGL(prepare_kernel<PT>, dim3(F), dim3(1024), 0, st, points, labels, N, ws.P, ws.NCMAX, keys, packed, boxes, counts, 0, K, H, W,
                           camf, (const int*)pw.flag);
    }
    // DI2P_SOLVER_CFG=<waves per hypothesis><min waves/SIMD>, e.g. 43 (default); DI2P_SOLVER_NOCULL=1 classifies every
    // cluster per point (the sums are bit-identical by construction: tests compare the two)
    const int cfg = (int)di2p_opt(DI2P_OPT_SOLVER_CFG);
    const int nocull = (di2p_opt(DI2P_OPT_SOLVER_NOCULL) ? 1 : 0) | (di2p_opt(DI2P_OPT_SOLVER_NOPREFILTER) ? 2 : 0) |
                       (di2p_opt(DI2P_OPT_SOLVER_NOCACHE) ? 4 : 0);   // bit 0: no cluster test, bit 1: no fp32 pre-filter, bit 2: no classification cache
    const dim3 grid(R * F);
    // Two tiers against the tail: the sweep counts of the hypotheses are heavy-tailed (median 48, 10 % above 140, max > 200 on the
    // config-2 workload) and a hypothesis is a sequential chain of sweeps, so a lone launch ends with a few long chains on an
    // otherwise idle chip.  Tier 1 (4 waves per hypothesis, 3 workgroups per CU: throughput) parks every hypothesis that needs
    // more than `tier` sweeps; tier 2 resumes the parked ones with 12 waves each (one workgroup per CU: latency).  Which tier a
    // hypothesis ends in depends on its own sweep count only, so results do not depend on the batch.  tier = 0: single launch.
    int* pending = (int*)(base + ws.off_pending);
    unsigned long long* state = (unsigned long long*)(base + ws.off_state);
    const int tier = (int)di2p_opt(DI2P_OPT_SOLVER_TIER_SWEEPS);
    SolveArgs<PT> ka;
    ka.packed = packed; ka.boxes_all = boxes; ka.counts = counts; ka.Kmat = K; ka.init_y = init_y; ka.init_T = init_T; ka.yaw0 = yaw0;
    ka.params_out = params; ka.cost_out = cost; ka.iters_out = iters; ka.sweeps_out = sweeps; ka.prof = g_prof; ka.state_buf = state;
    ka.cache = (CacheEnt*)(base + ws.off_cache);
    ka.camf = camf;
    ka.H = H; ka.W = W; ka.bnd = b; ka.NCMAX = ws.NCMAX; ka.nocull = nocull; ka.max_iter = max_iter; ka.F = F; ka.R = R; ka.N = N;
    // the diagnostics (phase clocks, cluster / evaluation counters) are a separate instantiation: the production kernel carries none of it
#define DI2P_LAUNCH_SOLVE_P(NPV, MW, WP, PEND, BUDGET, RESUME)                                                                   \
    do {                                                                                                                         \
        ka.pending = PEND; ka.budget = BUDGET; ka.resume = RESUME;                                                               \
        /* the instrumented build runs at <= 3 waves per SIMD: at 128 registers its timers push it into scratch and distort the phases */ \
        if (g_prof) hipLaunchKernelGGL((solve_kernel<NPV, PT, (MW > 3 ? 3 : MW), WP, true>), grid, dim3(WP * 64), 0, st, ka);    \
        else hipLaunchKernelGGL((solve_kernel<NPV, PT, MW, WP, false>), grid, dim3(WP * 64), lds_pad, st, ka);                   \
    } while (0)
#define DI2P_LAUNCH_SOLVE(NPV, MW, WP, PEND, BUDGET, RESUME)                                                                     \
    do {                                                                                                                         \
        ka.pending = PEND; ka.budget = BUDGET; ka.resume = RESUME;                                                               \
        hipLaunchKernelGGL((solve_kernel<NPV, PT, MW, WP, false>), grid, dim3(WP * 64), lds_pad, st, ka);                        \
    } while (0)
    // unused dynamic LDS: caps the solver's workgroups per CU (160 KB / (static + pad)) so that registers and LDS stay free for
    // the MFMA kernels of the other streams -- the solver needs VALU issue slots, they need the matrix pipe
    const size_t lds_pad = (size_t)di2p_opt(DI2P_OPT_SOLVER_LDS_PAD);
    int* pend1 = tier > 0 ? pending : nullptr;
    if (is_2d) {
        switch (cfg) {
#ifndef DI2P_SOLVER_MIN_INSTANCES
            // variants kept for measurements (DESIGN.md section 4 lists what each one measured); the diagnostics build exists for the default only
            case 42: DI2P_LAUNCH_SOLVE(4, 2, 4, pend1, tier, 0); break;
            case 43: DI2P_LAUNCH_SOLVE(4, 3, 4, pend1, tier, 0); break;
            case 23: DI2P_LAUNCH_SOLVE(4, 3, 2, pend1, tier, 0); break;
            case 24: DI2P_LAUNCH_SOLVE(4, 4, 2, pend1, tier, 0); break;
            case 14: DI2P_LAUNCH_SOLVE(4, 4, 1, pend1, tier, 0); break;
            case 83: DI2P_LAUNCH_SOLVE(4, 3, 8, pend1, tier, 0); break;
            case 84: DI2P_LAUNCH_SOLVE(4, 4, 8, pend1, tier, 0); break;
#endif
            default: DI2P_LAUNCH_SOLVE_P(4, DI2P_SOLVER_DEFAULT_MINW, DI2P_SOLVER_DEFAULT_WPH, pend1, tier, 0); break;
        }
#ifndef DI2P_SOLVER_MIN_INSTANCES
        if (tier > 0) DI2P_LAUNCH_SOLVE(4, 3, 12, pending, 0, 1);
    } else {
        DI2P_LAUNCH_SOLVE_P(6, 2, 4, pend1, tier, 0);
        if (tier > 0) DI2P_LAUNCH_SOLVE(6, 2, 8, pending, 0, 1);
#endif
    }
#undef DI2P_LAUNCH_SOLVE
#undef DI2P_LAUNCH_SOLVE_P
    return 0;
}

}  // namespace

extern "C" int di2p_solve_batched(const double* points, const int32_t* labels, const double* K, const double* init_y,
                                  const double* init_T, const double* yaw0, double H, double W, const double* lb_host,
                                  const double* ub_host, int max_iter, int is_2d, int F, int R, int N, double* params,
                                  double* cost, int32_t* iters, int32_t* sweeps, void* workspace, void* stream) {
    DI2P_CHECK_ARG(points && labels && K && init_y && init_T && lb_host && ub_host && params && cost && iters && workspace, "null pointer");
    DI2P_CHECK_ARG(F >= 0 && R >= 0 && N >= 0 && max_iter >= 0, "bad size");
    if (F == 0 || R == 0) return 0;
    launch_solve<double>(points, labels, K, init_y, init_T, yaw0, H, W, lb_host, ub_host, max_iter, is_2d, F, R, N, params, cost, iters, sweeps, workspace, (hipStream_t)stream);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_solve_batched_f32(const float* points, const int32_t* labels, const double* K, const double* init_y,
                                      const double* init_T, const double* yaw0, double H, double W, const double* lb_host,
                                      const double* ub_host, int max_iter, int is_2d, int F, int R, int N, double* params,
                                      double* cost, int32_t* iters, int32_t* sweeps, void* workspace, void* stream) {
    DI2P_CHECK_ARG(points && labels && K && init_y && init_T && lb_host && ub_host && params && cost && iters && workspace, "null pointer");
    DI2P_CHECK_ARG(F >= 0 && R >= 0 && N >= 0 && max_iter >= 0, "bad size");
    if (F == 0 || R == 0) return 0;
    launch_solve<float>(points, labels, K, init_y, init_T, yaw0, H, W, lb_host, ub_host, max_iter, is_2d, F, R, N, params, cost, iters, sweeps, workspace, (hipStream_t)stream);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_select_best(const double* params, const double* cost, const int32_t* has_inside, int is_2d, int F, int R,
                                int32_t* best, double* P, double* best_cost, void* stream) {
    DI2P_CHECK_ARG(params && cost && best && P && best_cost && F >= 0 && R >= 1, "bad args");
    if (F == 0) return 0;
    hipLaunchKernelGGL(select_best_kernel, dim3(F), dim3(64), 0, (hipStream_t)stream, params, cost, has_inside, is_2d ? 4 : 6, R, best, P, best_cost);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_initial_guess(const double* points, const int32_t* labels, double* yaw0, int32_t* labels_out,
                                  int32_t* has_inside, int F, int N, void* stream) {
    DI2P_CHECK_ARG(points && labels && yaw0 && labels_out && has_inside && F >= 0 && N >= 0, "bad args");
    if (F == 0) return 0;
    hipLaunchKernelGGL(initial_guess_kernel, dim3(F), dim3(1024), 0, (hipStream_t)stream, points, labels, yaw0, labels_out, has_inside, N);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_solver_residuals(const double* points, const int32_t* labels, const double* K, const double* params, double H,
                                     double W, int is_2d, int F, int N, double* residuals, int32_t* counts, double* cost,
                                     void* stream) {
    DI2P_CHECK_ARG(points && labels && K && params && residuals && counts && cost && F >= 0 && N >= 0, "bad args");
    if (F == 0) return 0;
    if (is_2d)
        hipLaunchKernelGGL(residuals_kernel<4>, dim3(F), dim3(256), 0, (hipStream_t)stream, points, labels, K, params, H, W, N, residuals, counts, cost);
    else
        hipLaunchKernelGGL(residuals_kernel<6>, dim3(F), dim3(256), 0, (hipStream_t)stream, points, labels, K, params, H, W, N, residuals, counts, cost);
    DI2P_RETURN_LAUNCH();
}

extern "C" long long di2p_solve_workspace_bytes(int F, int R, int N) {
    if (F < 0 || N < 0 || R < 0) return 0;
    return (long long)solve_ws_layout(F, R, N).bytes;
}

// Diagnostics: when set to a device buffer of F*R*28 int64 (library version >= 6; 20 in versions 4-5; 16 before), every solve launch (a separate instantiation of the kernel) records per
// hypothesis: [0..2] shader-clock cycles wave 0 spent in {sweep, waiting at the reduction barrier, LM update}, [3] its phase-B
// evaluations, [4..5] its clusters {classified per point, taken as all-active}, [6] cycles combining the wave partials, [7] packed
// line-search counters, [8..10] LM stages {decide, wave-wide interpolant minimiser, finish + begin iteration}, [11] guard-only
// clusters, [12..15] inside the sweep: {cluster-test rounds, drains = phase B, set-up, log + wave reduction}, [16..17] classification-cache
// hits {clusters whose recorded mask was re-used, guard-only clusters skipped} ([4] and [11] count the misses), [18..19] straight-line batches of the
// {guard-only, classification} walk, [20] clusters appended by phase II, [21..22] phase-B rounds of 64 {label 1, label 0}, [23] cluster-test rounds, [24] guard-only
// clusters walked on three planes only (status 5, part of [11]), [25..27] reserved.
extern "C" void di2p_solver_set_profile_buffer(void* buf) { g_prof = (long long*)buf; }
