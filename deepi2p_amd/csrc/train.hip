// Training path of the classifier: the backward kernels that torch.autograd supplies in the reference
// (models/multimodal_classifier.py:213-218 `loss.backward()`) written for gfx950, plus train-mode BatchNorm.
//
//   rc-GEMM          out[row][col] = sum_r A(r,row) * B(r,col) with BOTH operands contiguous along the reduction index r
//                    (the shape of every "gradient with respect to a weight": reduce over points / pixels).  Lanes load along r
//                    (coalesced dwords), the tile is transposed on its way into LDS ([r][row] panels, row pitch 65: the 32 lanes
//                    of a store group hit 32 different banks) and feeds v_mfma_f32_32x32x2_f32 exactly like the forward engine.
//                    The reduction is cut into chunks (one workgroup each, partial tiles in scratch) that a second kernel adds in
//                    chunk order: deterministic, no float atomics.  Instances:
//                      * d W of a 1x1 layer            A = dY[b][m][n],  B = X[b][k][n]                (nn.Conv1d / MyConv2d)
//                      * d W of a convolution          A = dY[b][co][p], B = im2col(X)[b][(ci,ky,kx)][p]
//                      * backward of column gathers    A = dY[b][c][j],  B = sum_k w[b][j][k] [idx[b][j][k] == m]
//                        (torch.gather along the point axis and upsample_by_interpolation, networks_united.py:76-103: the scatter
//                        of a gather written as a product with the sparse selection matrix -- no atomics, any fan-in)
//                      * attention: d feat               A = dOut[b][c][m], B = score[b][hw][m]
//   km-GEMM          out[row][col] = sum_k A[k][row] * B[k][col] per batch (attention: d score)
//   conv dgrad       d X as an implicit GEMM over (co, ky, kx) with the stride folded into the loader
//   BatchNorm        batch statistics in fp64 partial sums (two kernels), normalise + affine (+ residual) (+ ReLU) fused;
//                    backward = one reduction pass (sum g, sum g*xhat) + one elementwise pass
//   arg-max routers  segment max (index_max), max over neighbours / nodes, 3x3/2 max-pool: gradient goes to the saved / recomputed
//                    arg-max (first maximum in scan order, the rule of torch.max / max_pool2d)
#include "common.h"
#include "mfma_tile.h"
#include <type_traits>

namespace {

// =============================================================================================== rc-GEMM
constexpr int RC_BK = 32, RC_BM = 64, RC_LD = 65, RC_PASSES = 8;
constexpr int RC_LDS_FLOATS = 2 * 2 * RC_BK * RC_LD;

// Loader protocol: set_index(p, i) once per pass p (row / column i of the output tile, may be out of range), set_r(r, ok)
// once per K-step (this lane's reduction index; !ok = past the end), load(p) -> value (0 where out of range).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Scalar stager of one operand: lane = (reduction index r_in, 8 rows per pass); any loader.
template <class L>
struct RcStageScalar {
    float v[RC_PASSES];
    int r_in, q0;
    __device__ __forceinline__ void init(int tid, L& l, int blk) {
        r_in = tid & 31; q0 = tid >> 5;
#pragma unroll
        for (int p = 0; p < RC_PASSES; ++p) l.set_index(p, blk + q0 + 8 * p);
    }
    __device__ __forceinline__ void load(L& l, int r0, int r_end) {
        const int r = r0 + r_in;
        const bool ok = r < r_end;
        l.set_r(ok ? r : r_end - 1, ok);
#pragma unroll
        for (int p = 0; p < RC_PASSES; ++p) v[p] = l.load(p);
    }
    __device__ __forceinline__ void store(float* S) const {
#pragma unroll
        for (int p = 0; p < RC_PASSES; ++p) S[r_in * RC_LD + q0 + 8 * p] = v[p];
    }
};

// 16-byte stager for an operand that is plainly strided (value(r, i) = base[i*ld + r]) with 16-byte addressable rows and a
// reduction range that is a multiple of 4: lane = (4 consecutive r, 32 rows per pass): 2 loads instead of 8 per K-step.
struct RcStageVec {
    f32x4 v[2];
    const float* ptr[2];
    bool okp[2];
    int r4, row0;
    template <class L>
    __device__ __forceinline__ void init(int tid, L& l, int blk) {
        r4 = tid & 7; row0 = tid >> 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int i = blk + row0 + 32 * p;
            okp[p] = i < l.n;
            ptr[p] = l.p + (long long)min(i, l.n - 1) * l.ld + r4 * 4;
        }
    }
    template <class L>
    __device__ __forceinline__ void load(L&, int r0, int r_end) {
        const bool ok = r0 + r4 * 4 < r_end;
        const int rc = ok ? r0 : r_end - 32;           // (a clamped, still in-range 16-byte read; zeroed below)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(ptr[p] + (rc < 0 ? 0 : rc));
            v[p] = (ok && okp[p]) ? t : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    __device__ __forceinline__ void store(float* S) const {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) S[(r4 * 4 + i) * RC_LD + row0 + 32 * p] = v[p][i];
    }
};

template <bool AV, bool BV, class LA, class LB>
__device__ __forceinline__ void rc_gemm_tile(float* lds, LA& la, LB& lb, int r_begin, int r_end, int row_blk, int col_blk,
                                             float* __restrict__ out, int rows, int cols) {
    float* As = lds;                          // [2][BK][LD]
    float* Bs = lds + 2 * RC_BK * RC_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    typename std::conditional<AV, RcStageVec, RcStageScalar<LA>>::type sa;
    typename std::conditional<BV, RcStageVec, RcStageScalar<LB>>::type sb;
    sa.init(tid, la, row_blk);
    sb.init(tid, lb, col_blk);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int T = (r_end - r_begin + RC_BK - 1) / RC_BK;
    if (T > 0) {
        sa.load(la, r_begin, r_end); sb.load(lb, r_begin, r_end);
        sa.store(As); sb.store(Bs);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) { sa.load(la, r_begin + (t + 1) * RC_BK, r_end); sb.load(lb, r_begin + (t + 1) * RC_BK, r_end); }
        const float* Ab = As + buf * RC_BK * RC_LD + wm * 32 + l31;
        const float* Bb = Bs + buf * RC_BK * RC_LD + wn * 32 + l31;
        float a[RC_BK / 2], b[RC_BK / 2];
#pragma unroll
        for (int kk = 0; kk < RC_BK; kk += 2) { a[kk / 2] = Ab[(kk + half) * RC_LD]; b[kk / 2] = Bb[(kk + half) * RC_LD]; }
#pragma unroll
        for (int kk = 0; kk < RC_BK / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
        if (t + 1 < T) { sa.store(As + (buf ^ 1) * RC_BK * RC_LD); sb.store(Bs + (buf ^ 1) * RC_BK * RC_LD); }
        __syncthreads();
    }
    const int col = col_blk + wn * 32 + l31;
    if (col < cols) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_blk + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < rows) out[(long long)row * cols + col] = acc[r];
        }
    }
}

// value(r, i) = p[i * ld + r], i < n
struct RcStrided {
    const float* base;
    long long ld, batch_stride;
    int n;
    const float* p;
    long long off[RC_PASSES];
    bool okp[RC_PASSES];
    int r;
    bool okr;
    __device__ __forceinline__ void batch(int z) { p = base + (long long)z * batch_stride; }
    __device__ __forceinline__ void set_index(int ps, int i) { okp[ps] = i < n; off[ps] = (long long)min(i, n - 1) * ld; }
    __device__ __forceinline__ void set_r(int r_, bool ok) { r = r_; okr = ok; }
    __device__ __forceinline__ float load(int ps) const { const float v = p[off[ps] + r]; return (okr && okp[ps]) ? v : 0.0f; }
};

// value(r = output pixel, i = (ci, ky, kx)) = x[b][ci][oy*s - pad + ky][ox*s - pad + kx] or 0
struct RcIm2col {
    const float* base;
    int Cin, H, W, OW, KH, KW, stride, pad, ncols;
    const float* p;
    int ci_off[RC_PASSES], ky[RC_PASSES], kx[RC_PASSES];
    bool okp[RC_PASSES];
    int iy0, ix0;
    bool okr;
    __device__ __forceinline__ void batch(int z) { p = base + (long long)z * Cin * H * W; }
    __device__ __forceinline__ void set_index(int ps, int i) {
        okp[ps] = i < ncols;
        const int ic = min(i, ncols - 1);
        const int ci = ic / (KH * KW), t = ic - ci * KH * KW;
        ci_off[ps] = ci * H * W; ky[ps] = t / KW; kx[ps] = t - (t / KW) * KW;
    }
    __device__ __forceinline__ void set_r(int r, bool ok) { okr = ok; const int oy = r / OW; iy0 = oy * stride - pad; ix0 = (r - oy * OW) * stride - pad; }
    __device__ __forceinline__ float load(int ps) const {
        const int iy = iy0 + ky[ps], ix = ix0 + kx[ps];
        const bool in = okr && okp[ps] && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float v = p[ci_off[ps] + min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)];
        return in ? v : 0.0f;
    }
};

// value(r = column j of the gathered tensor, i = node m) = sum_k w[b][j][k] * [idx[b][j][k] == m]   (w == nullptr: weights 1)
template <int KN>
struct RcSelect {
    const int* idx_base;
    const float* w_base;
    int J, M;
    const int* idx;
    const float* w;
    int m[RC_PASSES];
    int id[KN];
    float wk[KN];
    __device__ __forceinline__ void batch(int z) { idx = idx_base + (long long)z * J * KN; w = w_base ? w_base + (long long)z * J * KN : nullptr; }
    __device__ __forceinline__ void set_index(int ps, int i) { m[ps] = i < M ? i : -1; }
    __device__ __forceinline__ void set_r(int r, bool ok) {
#pragma unroll
        for (int k = 0; k < KN; ++k) { id[k] = ok ? idx[(long long)r * KN + k] : -2; wk[k] = w ? w[(long long)r * KN + k] : 1.0f; }
    }
    __device__ __forceinline__ float load(int ps) const {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < KN; ++k) v += id[k] == m[ps] ? wk[k] : 0.0f;
        return v;
    }
};

template <bool AV, bool BV, class LA, class LB>
__global__ __launch_bounds__(256) void rc_gemm_kernel(LA la, LB lb, int R, int rch, int chunks, float* __restrict__ partial, int rows, int cols) {
    extern __shared__ float lds[];
    const int z = blockIdx.z / chunks, ch = blockIdx.z % chunks;
    la.batch(z); lb.batch(z);
    const int r_begin = ch * rch, r_end = min(R, r_begin + rch);
    rc_gemm_tile<AV, BV>(lds, la, lb, r_begin, r_end, blockIdx.y * RC_BM, blockIdx.x * RC_BM, partial + (long long)blockIdx.z * rows * cols, rows, cols);
}

// 128 x 128 tiles (a wave: 64 x 64 = 2 x 2 matrix tiles) for the plainly strided operand pair with 16-byte rows -- the weight gradients of the
// big point layers.  Those launches are bound by operand traffic (every 64 x 64 tile re-reads its 64 + 64 rows over the whole reduction range:
// dW of the per-point head's first layer, 256 x 736 over 8 x 20480 columns, moved 4 GB for 62 Gflop: 0.86 ms); a 128 x 128 tile reads half of
// that per output element.  Same reduction order per output element for the same chunks.
constexpr int RC2_BM = 128, RC2_LD = 129;
constexpr int RC2_LDS_FLOATS = 2 * 2 * RC_BK * RC2_LD;
struct RcStageVec4 {
    f32x4 v[4];
    const float* ptr[4];
    bool okp[4];
    int r4, row0;
    template <class L>
    __device__ __forceinline__ void init(int tid, L& l, int blk) {
        r4 = tid & 7; row0 = tid >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = blk + row0 + 32 * p;
            okp[p] = i < l.n;
            ptr[p] = l.p + (long long)min(i, l.n - 1) * l.ld + r4 * 4;
        }
    }
    __device__ __forceinline__ void load(int r0, int r_end) {
        const bool ok = r0 + r4 * 4 < r_end;
        const int rc = ok ? r0 : r_end - 32;           // (a clamped, still in-range 16-byte read; zeroed below)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(ptr[p] + (rc < 0 ? 0 : rc));
            v[p] = (ok && okp[p]) ? t : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    __device__ __forceinline__ void store(float* S) const {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) S[(r4 * 4 + i) * RC2_LD + row0 + 32 * p] = v[p][i];
    }
};
__global__ __launch_bounds__(256) void rc_gemm128_kernel(RcStrided la, RcStrided lb, int R, int rch, int chunks, float* __restrict__ partial, int rows, int cols) {
    extern __shared__ float lds[];
    const int z = blockIdx.z / chunks, ch = blockIdx.z % chunks;
    la.batch(z); lb.batch(z);
    const int r_begin = ch * rch, r_end = min(R, r_begin + rch);
    float* out = partial + (long long)blockIdx.z * rows * cols;
    const int row_blk = blockIdx.y * RC2_BM, col_blk = blockIdx.x * RC2_BM;
    float* As = lds;                          // [2][BK][LD]
    float* Bs = lds + 2 * RC_BK * RC2_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
    RcStageVec4 sa, sb;
    sa.init(tid, la, row_blk);
    sb.init(tid, lb, col_blk);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int T = (r_end - r_begin + RC_BK - 1) / RC_BK;
    if (T > 0) {
        sa.load(r_begin, r_end); sb.load(r_begin, r_end);
        sa.store(As); sb.store(Bs);
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) { sa.load(r_begin + (t + 1) * RC_BK, r_end); sb.load(r_begin + (t + 1) * RC_BK, r_end); }
        const float* Ab = As + buf * RC_BK * RC2_LD + wm * 64 + l31;
        const float* Bb = Bs + buf * RC_BK * RC2_LD + wn * 64 + l31;
#pragma unroll
        for (int k0 = 0; k0 < RC_BK; k0 += 8) {          // four k-pairs at a time: 16 operand registers
            float a[4][2], b[4][2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[kk][i] = Ab[(k0 + 2 * kk + half) * RC2_LD + i * 32];
                    b[kk][i] = Bb[(k0 + 2 * kk + half) * RC2_LD + i * 32];
                }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < T) { sa.store(As + (buf ^ 1) * RC_BK * RC2_LD); sb.store(Bs + (buf ^ 1) * RC_BK * RC2_LD); }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col_blk + wn * 64 + j * 32 + l31;
            if (col >= cols) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_blk + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < rows) out[(long long)row * cols + col] = acc[i][j][r];
            }
        }
}

// out[g][e] = alpha * sum_{p < per_group} partial[g * per_group + p][e], in a fixed order
__global__ __launch_bounds__(256) void rc_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int per_group, long long elems,
                                                        long long total, float alpha) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long g = i / elems, e = i - g * elems;
    const float* p = partial + g * per_group * elems + e;
    // eight partial sums over p = k mod 8 (eight loads in flight instead of one dependent chain of up to ~160), combined as a fixed tree:
    // deterministic, independent of launch geometry
    float s[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    int k = 0;
    for (; k + 8 <= per_group; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * elems];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += v[u];
    }
    for (int u = 0; k < per_group; ++k, ++u) s[u] += p[(long long)k * elems];
    out[i] = alpha * (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
}

struct RcPlan { int rch, chunks; long long bytes; };
// the big strided pairs take 128 x 128 tiles (rc_gemm128_kernel)
inline bool rc_big(int rows, int cols) { return rows >= 128 && cols >= 128 && !di2p_opt(DI2P_OPT_RC_TILE64); }
RcPlan rc_plan(int Z, int rows, int cols, int R, int bm = RC_BM) {
    RcPlan pl;
    // enough workgroups to fill the chip, chunks of at least 256 reduction steps
    const long long tiles = (long long)di2p_cdiv(rows, bm) * di2p_cdiv(cols, bm) * Z;
    int chunks = (int)((1024 + tiles - 1) / tiles);
    const int max_chunks = R / 256 > 1 ? R / 256 : 1;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    pl.rch = ((di2p_cdiv(R, chunks) + RC_BK - 1) / RC_BK) * RC_BK;
    pl.chunks = di2p_cdiv(R, pl.rch);
    pl.bytes = (long long)Z * pl.chunks * rows * cols * 4;
    return pl;
}

inline bool rc_vec_ok(const float* base, long long ld, long long batch_stride, int R) {
    return ((uintptr_t)base & 15) == 0 && ld % 4 == 0 && batch_stride % 4 == 0 && R % 4 == 0 && R >= 32;
}

template <class LA, class LB>
int rc_launch(const char* who, LA la, LB lb, int Z, int rows, int cols, int R, float alpha, bool reduce_z, float* out, void* ws, long long ws_bytes,
              hipStream_t st, bool a_vec = false, bool b_vec = false) {
    bool big = false;
    if constexpr (std::is_same<LA, RcStrided>::value && std::is_same<LB, RcStrided>::value) big = a_vec && b_vec && rc_big(rows, cols);
    const RcPlan pl = rc_plan(Z, rows, cols, R, big ? RC2_BM : RC_BM);
    if (!ws || ws_bytes < pl.bytes) { di2p_set_error("%s: workspace too small (%lld bytes needed)", who, pl.bytes); return -1; }
    if ((long long)Z * pl.chunks > 65535) { di2p_set_error("%s: too many reduction chunks", who); return -1; }
    const dim3 grid(di2p_cdiv(cols, big ? RC2_BM : RC_BM), di2p_cdiv(rows, big ? RC2_BM : RC_BM), Z * pl.chunks);
    const size_t lds = RC_LDS_FLOATS * sizeof(float);
    if constexpr (std::is_same<LA, RcStrided>::value && std::is_same<LB, RcStrided>::value) {
        if (big) {
            (void)hipFuncSetAttribute((const void*)rc_gemm128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RC2_LDS_FLOATS * sizeof(float)));
            hipLaunchKernelGGL(rc_gemm128_kernel, grid, dim3(256), RC2_LDS_FLOATS * sizeof(float), st, la, lb, R, pl.rch, pl.chunks, (float*)ws, rows, cols);
        } else if (a_vec && b_vec) hipLaunchKernelGGL((rc_gemm_kernel<true, true, LA, LB>), grid, dim3(256), lds, st, la, lb, R, pl.rch, pl.chunks, (float*)ws, rows, cols);
        else hipLaunchKernelGGL((rc_gemm_kernel<false, false, LA, LB>), grid, dim3(256), lds, st, la, lb, R, pl.rch, pl.chunks, (float*)ws, rows, cols);
    } else if constexpr (std::is_same<LA, RcStrided>::value) {
        if (a_vec) hipLaunchKernelGGL((rc_gemm_kernel<true, false, LA, LB>), grid, dim3(256), lds, st, la, lb, R, pl.rch, pl.chunks, (float*)ws, rows, cols);
        else hipLaunchKernelGGL((rc_gemm_kernel<false, false, LA, LB>), grid, dim3(256), lds, st, la, lb, R, pl.rch, pl.chunks, (float*)ws, rows, cols);
    } else {
        hipLaunchKernelGGL((rc_gemm_kernel<false, false, LA, LB>), grid, dim3(256), lds, st, la, lb, R, pl.rch, pl.chunks, (float*)ws, rows, cols);
    }
    const long long elems = (long long)rows * cols;
    const int per_group = reduce_z ? Z * pl.chunks : pl.chunks;
    const long long total = reduce_z ? elems : elems * Z;
    hipLaunchKernelGGL(rc_reduce_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st, (const float*)ws, out, per_group, elems, total, alpha);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { di2p_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return (int)e; }
    return 0;
}

// =============================================================================================== km-GEMM (per batch, k-major operands)
struct KmA {
    const float* p; int ld, K, M;
    __device__ __forceinline__ float load(int k, int m) const { return (k < K && m < M) ? p[(long long)k * ld + m] : 0.0f; }
};
struct KmB {
    const float* p; int ld, K, N, n; bool valid;
    __device__ __forceinline__ void column(int j) { n = j; valid = j < N; }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const { return (valid && k < K) ? p[(long long)k * ld + n] : 0.0f; }
};
struct KmEpi {
    float* out; int M, N; float alpha;
    __device__ __forceinline__ void tile(int mrow0, int n, const f32x16& acc) {
        if (n >= N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < M) out[(long long)m * N + n] = alpha * acc[r];
        }
    }
};
using KmCfg = TileCfg<2, 2, 1, 1, 16>;
__global__ __launch_bounds__(KmCfg::THREADS) void bmm_km_kernel(const float* __restrict__ A, int lda, long long a_bs, const float* __restrict__ Bm, int ldb,
                                                                long long b_bs, float* __restrict__ out, int rows, int cols, int K, float alpha) {
    extern __shared__ float lds[];
    const int z = blockIdx.z;
    KmA la{A + z * a_bs, lda, K, rows};
    KmB lb{Bm + z * b_bs, ldb, K, cols, 0, false};
    KmEpi ep{out + (long long)z * rows * cols, rows, cols, alpha};
    mfma_gemm_block<KmCfg>(lds, la, lb, ep, K, blockIdx.y * KmCfg::BM, blockIdx.x * KmCfg::BN);
}

// =============================================================================================== convolution: d input
// dX[b][ci][iy][ix] = sum_{co,ky,kx} W[co][ci][ky][kx] * dY[b][co][(iy+pad-ky)/s][(ix+pad-kx)/s]   (where divisible and in range)
struct DgradA {   // A(k = (co,ky,kx), m = ci)
    const float* Wg; int Cin, KH, KW, K;
    __device__ __forceinline__ float load(int k, int m) const {
        if (k >= K || m >= Cin) return 0.0f;
        const int co = k / (KH * KW), t = k - co * KH * KW;
        return Wg[((long long)co * Cin + m) * KH * KW + t];
    }
};
struct DgradB {   // B(k, column = input pixel)
    const float* dy; int OH, OW, H, W, KH, KW, stride, pad, K;
    int iy, ix; bool valid;
    __device__ __forceinline__ void column(int j) { valid = j < H * W; const int jc = valid ? j : 0; iy = jc / W; ix = jc - iy * W; }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const {
        if (!valid || k >= K) return 0.0f;
        const int co = k / (KH * KW), t = k - co * KH * KW, ky = t / KW, kx = t - ky * KW;
        const int ty = iy + pad - ky, tx = ix + pad - kx;
        if (ty < 0 || tx < 0) return 0.0f;
        const int oy = ty / stride, ox = tx / stride;
        if (oy * stride != ty || ox * stride != tx || oy >= OH || ox >= OW) return 0.0f;
        return dy[((long long)co * OH + oy) * OW + ox];
    }
};
__global__ __launch_bounds__(KmCfg::THREADS) void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ Wg, float* __restrict__ dx,
                                                                    int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
    extern __shared__ float lds[];
    const int b = blockIdx.z, K = Cout * KH * KW;
    DgradA la{Wg, Cin, KH, KW, K};
    DgradB lb{dy + (long long)b * Cout * OH * OW, OH, OW, H, W, KH, KW, stride, pad, K, 0, 0, false};
    KmEpi ep{dx + (long long)b * Cin * H * W, Cin, H * W, 1.0f};
    mfma_gemm_block<KmCfg>(lds, la, lb, ep, K, blockIdx.y * KmCfg::BM, blockIdx.x * KmCfg::BN);
}

// Stride 2: an input pixel only meets the taps of ITS parity -- ky = (iy + pad) mod 2 + 2 ty, kx likewise: 1, 2, 2 or 4 of a 3 x 3 filter's nine,
// one or none of a 1 x 1 filter's -- and the dense kernel above multiplies the other 3/4 (or more) of its K range by zeros it first computes with
// divisions.  Here a workgroup owns ONE parity class (blockIdx.z = frame * 4 + class): its columns are that class's pixels, its K range the
// class's taps in the dense kernel's order (co, ky, kx ascending), so every output sums the same non-zero products in the same order -- the
// fp32 matrix instruction is an exact fma chain (tools/probe_mfma_rounding.hip) and a zero product changes nothing in one: bit-identical to
// the dense kernel on finite operands (up to the sign of a zero), 1/4 of its matrix work.  A class without taps (1 x 1 / stride 2: three of
// four) writes zeros.
struct DgradS2 {
    const float* Wg; const float* dy;
    int Cin, KH, KW, pad, OH, OW, W;
    int ky0, kx0, ny, nx, K;          // this class's taps and K = Cout * ny * nx
    int Wc, ncol, a, b;               // the class's pixel grid: columns = (qy, qx), iy = 2 qy + a, ix = 2 qx + b
    int sh_t, sh_x;                   // log2(ny * nx), log2(nx) when both are powers of two (every class of a 3 x 3 or 1 x 1 filter), else -1
    int iy, ix; bool valid;
    __device__ __forceinline__ void split(int k, int& co, int& ty, int& tx) const {
        if (sh_t >= 0) { co = k >> sh_t; const int t = k & ((1 << sh_t) - 1); ty = t >> sh_x; tx = t & ((1 << sh_x) - 1); }
        else { const int nt = ny * nx; co = k / nt; const int t = k - co * nt; ty = t / nx; tx = t - ty * nx; }
    }
    __device__ __forceinline__ float load(int k, int m) const {            // A(k, m = ci)
        if (k >= K || m >= Cin) return 0.0f;
        int co, ty, tx;
        split(k, co, ty, tx);
        return Wg[(((long long)co * Cin + m) * KH + ky0 + 2 * ty) * KW + kx0 + 2 * tx];
    }
    __device__ __forceinline__ void column(int j) { valid = j < ncol; const int jc = valid ? j : 0, qy = jc / Wc; iy = 2 * qy + a; ix = 2 * (jc - qy * Wc) + b; }
    __device__ __forceinline__ void begin_tile(int) {}
    __device__ __forceinline__ float load(int k) const {                   // B(k, this thread's column)
        if (!valid || k >= K) return 0.0f;
        int co, ty, tx;
        split(k, co, ty, tx);
        const int oy = (iy + pad - ky0 - 2 * ty) >> 1, ox = (ix + pad - kx0 - 2 * tx) >> 1;      // exact: the tap has the pixel's parity
        if (oy < 0 || ox < 0 || oy >= OH || ox >= OW) return 0.0f;
        return dy[((long long)co * OH + oy) * OW + ox];
    }
};
struct DgradS2Epi {
    float* out; int M, HW, W, Wc, ncol, a, b;
    __device__ __forceinline__ void tile(int mrow0, int n, const f32x16& acc) {
        if (n >= ncol) return;
        const int qy = n / Wc, pix = (2 * qy + a) * W + 2 * (n - qy * Wc) + b;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mrow0 + (r & 3) + 8 * (r >> 2);
            if (m < M) out[(long long)m * HW + pix] = acc[r];
        }
    }
};
__global__ __launch_bounds__(KmCfg::THREADS) void conv_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ Wg, float* __restrict__ dx,
                                                                       int Cin, int H, int W, int Cout, int KH, int KW, int pad, int OH, int OW) {
    extern __shared__ float lds[];
    const int bz = blockIdx.z >> 2, cls = blockIdx.z & 3, a = cls >> 1, b = cls & 1;
    const int Hc = (H - a + 1) >> 1, Wc = (W - b + 1) >> 1, ncol = Hc * Wc;
    if ((int)blockIdx.x * KmCfg::BN >= ncol) return;                        // (the grid is sized for the largest class)
    const int ky0 = (a + pad) & 1, kx0 = (b + pad) & 1;
    const int ny = ky0 < KH ? (KH - ky0 + 1) >> 1 : 0, nx = kx0 < KW ? (KW - kx0 + 1) >> 1 : 0;
    const int nx1 = nx > 0 ? nx : 1, nt = ny * nx1;
    const bool pow2 = nt > 0 && (nt & (nt - 1)) == 0 && (nx1 & (nx1 - 1)) == 0;
    DgradS2 l{Wg, dy + (long long)bz * Cout * OH * OW, Cin, KH, KW, pad, OH, OW, W, ky0, kx0, ny, nx1, Cout * ny * nx, Wc, ncol, a, b,
              pow2 ? 31 - __builtin_clz(nt) : -1, pow2 ? 31 - __builtin_clz(nx1) : 0, 0, 0, false};
    DgradS2Epi ep{dx + (long long)bz * Cin * H * W, Cin, H * W, W, Wc, ncol, a, b};
    mfma_gemm_block<KmCfg>(lds, l, l, ep, l.K, blockIdx.y * KmCfg::BM, blockIdx.x * KmCfg::BN);
}

// =============================================================================================== per-channel reductions (BatchNorm, bias)
// x[B][C][N]; grid (C, B * SN): block (c, s) reduces n in [chunk*per, ...) of row (b, c); fp64 partial pairs
template <int MODE>   // 0: (sum x, sum x^2)   1: (sum g, sum g*xhat) with g = dy * [y > 0 if relu]
__global__ __launch_bounds__(256) void channel_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, int relu, int C, int N,
                                                             int SN, int per, double* __restrict__ partial) {
    __shared__ double sh[2][4];
    const int c = blockIdx.x, s = blockIdx.y, b = s / SN, ch = s - b * SN;
    const long long row = ((long long)b * C + c) * N;
    const int n0 = ch * per, n1 = min(N, n0 + per);
    double a0 = 0.0, a1 = 0.0;
    float mu = 0.0f, is = 0.0f;
    if (MODE == 1) { mu = mean[c]; is = invstd[c]; }
    for (int n = n0 + threadIdx.x; n < n1; n += 256) {
        if (MODE == 0) {
            const double v = x[row + n];
            a0 += v; a1 += v * v;
        } else {
            float g = dy[row + n];
            if (relu && !(y[row + n] > 0.0f)) g = 0.0f;
            const float xh = (x[row + n] - mu) * is;
            a0 += g; a1 += (double)g * xh;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = a0; sh[1][wave] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[((long long)c * gridDim.y + s) * 2 + 0] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        partial[((long long)c * gridDim.y + s) * 2 + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

// MODE 0: batch statistics -> save_mean, save_invstd, running stats (unbiased variance, torch.nn.BatchNorm semantics)
// MODE 1: -> dgamma = sum g*xhat, dbeta = sum g, and the two means the elementwise pass needs (sums[c] = {sum g, sum g*xhat} / count)
// MODE 2: -> out0 = sum x (bias gradient)
template <int MODE>
__global__ __launch_bounds__(64) void channel_finalize_kernel(const double* __restrict__ partial, int S, int C, double count, float eps, float momentum,
                                                              float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ run_mean,
                                                              float* __restrict__ run_var, float* __restrict__ sums) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double a0 = 0.0, a1 = 0.0;
    for (int s = 0; s < S; ++s) { a0 += partial[((long long)c * S + s) * 2]; a1 += partial[((long long)c * S + s) * 2 + 1]; }
    if (MODE == 0) {
        const double mu = a0 / count;
        double var = a1 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        out0[c] = (float)mu;
        out1[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
        }
    } else if (MODE == 1) {
        out0[c] = (float)a1;          // dgamma
        out1[c] = (float)a0;          // dbeta
        sums[2 * c] = (float)(a0 / count);
        sums[2 * c + 1] = (float)(a1 / count);
    } else {
        out0[c] = (float)a0;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ res,
                                                       float* __restrict__ y, int relu, int C, int N, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / N) % C);
    float v = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
    if (res) v += res[i];
    if (relu) v = fmaxf(v, 0.0f);
    y[i] = v;
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g*xhat)),  dres = g
__global__ __launch_bounds__(256) void bn_backward_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ sums, int relu,
                                                                float* __restrict__ dx, float* __restrict__ dres, int C, int N, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / N) % C);
    float g = dy[i];
    if (relu && !(y[i] > 0.0f)) g = 0.0f;
    const float xh = (x[i] - mean[c]) * invstd[c];
    dx[i] = gamma[c] * invstd[c] * (g - sums[2 * c] - xh * sums[2 * c + 1]);
    if (dres) dres[i] = g;
}

// Round 6: finalize + elementwise pass as ONE launch per direction (55 BatchNorm layers x 2 directions x a 64-thread finalize launch per step:
// 0.5 ms of kernels and as much again in launch gaps).  Grid (C, S) like the reduction: block (c, s) first forms the channel's statistics from
// the S fp64 partial pairs ON ONE THREAD IN THE FINALIZE KERNEL'S ORDER (the same sums, the same bits, whichever block forms them; block s == 0
// also writes them out / updates the running buffers), then streams its chunk of row (b, c) with the elementwise kernel's expression.
__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ res, float* __restrict__ y, const double* __restrict__ partial,
                                                                int S, double count, float eps, float momentum, float* __restrict__ save_mean,
                                                                float* __restrict__ save_invstd, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                int relu, int C, int N, int SN, int per) {
    __shared__ float s_stat[2];
    const int c = blockIdx.x, s = blockIdx.y, b = s / SN, ch = s - b * SN;
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int q = 0; q < S; ++q) { a0 += partial[((long long)c * S + q) * 2]; a1 += partial[((long long)c * S + q) * 2 + 1]; }
        const double mu = a0 / count;
        double var = a1 / count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float fm = (float)mu, fi = (float)(1.0 / sqrt(var + (double)eps));
        s_stat[0] = fm; s_stat[1] = fi;
        if (s == 0) {
            save_mean[c] = fm;
            save_invstd[c] = fi;
            if (run_mean) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
                run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
            }
        }
    }
    __syncthreads();
    const float mean = s_stat[0], invstd = s_stat[1], g = gamma[c], be = beta[c];
    const long long row = ((long long)b * C + c) * N;
    const int n0 = ch * per, n1 = min(N, n0 + per);
    for (int n = n0 + threadIdx.x; n < n1; n += 256) {
        float v = (x[row + n] - mean) * invstd * g + be;
        if (res) v += res[row + n];
        if (relu) v = fmaxf(v, 0.0f);
        y[row + n] = v;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
                                                                   const float* __restrict__ gamma, const float* __restrict__ mean_,
                                                                   const float* __restrict__ invstd_, const double* __restrict__ partial, int S, double count,
                                                                   int relu, float* __restrict__ dx, float* __restrict__ dres, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int C, int N, int SN, int per) {
    __shared__ float s_sum[2];
    const int c = blockIdx.x, s = blockIdx.y, b = s / SN, ch = s - b * SN;
    if (threadIdx.x == 0) {
        double a0 = 0.0, a1 = 0.0;
        for (int q = 0; q < S; ++q) { a0 += partial[((long long)c * S + q) * 2]; a1 += partial[((long long)c * S + q) * 2 + 1]; }
        s_sum[0] = (float)(a0 / count);
        s_sum[1] = (float)(a1 / count);
        if (s == 0) { dgamma[c] = (float)a1; dbeta[c] = (float)a0; }
    }
    __syncthreads();
    const float m0 = s_sum[0], m1 = s_sum[1], mean = mean_[c], invstd = invstd_[c], gm = gamma[c];
    const long long row = ((long long)b * C + c) * N;
    const int n0 = ch * per, n1 = min(N, n0 + per);
    for (int n = n0 + threadIdx.x; n < n1; n += 256) {
        float g = dy[row + n];
        if (relu && !(y[row + n] > 0.0f)) g = 0.0f;
        const float xh = (x[row + n] - mean) * invstd;
        dx[row + n] = gm * invstd * (g - m0 - xh * m1);
        if (dres) dres[row + n] = g;
    }
}

struct RedPlan { int SN, per, S; long long bytes; };
RedPlan red_plan(int B, int C, int N) {
    RedPlan p;
    p.SN = N / 4096; if (p.SN < 1) p.SN = 1; if (p.SN > 16) p.SN = 16;
    p.per = di2p_cdiv(N, p.SN);
    p.SN = di2p_cdiv(N, p.per);
    p.S = B * p.SN;
    p.bytes = (long long)C * p.S * 16 + (long long)C * 8;
    return p;
}

// =============================================================================================== arg-max routers
// 3x3 / stride 2 / pad 1 max-pool backward: the gradient of an output goes to the FIRST maximum of its window in (ky, kx)
// scan order (max_pool2d keeps `val > maxval`).  GATHER form: one thread per INPUT pixel visits the <= 4 windows that cover it
// (oy in {ceil((iy-1)/2) .. floor((iy+1)/2)}, likewise ox), recomputes each window's first arg-max and adds the gradients of the
// windows it wins in a fixed (oy, ox) order -- deterministic, no float atomics, no memset of dx.
__global__ __launch_bounds__(256) void maxpool_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                               int OH, int OW, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ix = (int)(i % W), iy = (int)((i / W) % H);
    const long long plane = i / ((long long)W * H);
    const float* xp = x + plane * H * W;
    const float* dyp = dy + plane * OH * OW;
    float g = 0.0f;
    for (int oy = max(iy / 2, 0); oy <= min((iy + 1) / 2, OH - 1); ++oy) {          // windows with oy*2-1 <= iy <= oy*2+1
        for (int ox = max(ix / 2, 0); ox <= min((ix + 1) / 2, OW - 1); ++ox) {
            float best = 0.0f;
            int arg = -1;
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = oy * 2 - 1 + ky;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = ox * 2 - 1 + kx;
                    if (xx < 0 || xx >= W) continue;
                    const float v = xp[yy * W + xx];
                    if (arg < 0 || v > best) { best = v; arg = yy * W + xx; }
                }
            }
            if (arg == iy * W + ix) g += dyp[oy * OW + ox];
        }
    }
    dx[i] = g;
}

// The same gradient with the windows' arg-maxima computed ONCE: a workgroup owns MPB_R output rows of one plane -- it stages the 2 R + 3 input
// rows their windows and their neighbours' touch in LDS, finds the first maximum of the (R + 1) x OW windows that reach its 2 R input rows
// (nine LDS reads each instead of 36 memory reads per input pixel), and every input pixel then adds the gradients of the <= 4 windows it won,
// in the same (oy, ox) order: the same bits.
constexpr int MPB_R = 4;
__global__ __launch_bounds__(256) void maxpool_backward_lds_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int H, int W,
                                                                   int OH, int OW) {
    extern __shared__ float mp_lds[];
    float* xs = mp_lds;                                   // [2 R + 3][W]: input rows 2 oy0 - 1 ..
    int* arg = reinterpret_cast<int*>(mp_lds + (2 * MPB_R + 3) * W);     // [R + 1][OW]: linear index (iy * W + ix) of the window's first maximum
    const int oy0 = blockIdx.x * MPB_R;
    const long long plane = blockIdx.y;
    const float* xp = x + plane * H * W;
    const float* dyp = dy + plane * OH * OW;
    const int y_lo = 2 * oy0 - 1;
    for (int i = threadIdx.x; i < (2 * MPB_R + 3) * W; i += 256) {
        const int r = i / W, yy = y_lo + r;
        xs[i] = (yy >= 0 && yy < H) ? xp[yy * W + (i - r * W)] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (MPB_R + 1) * OW; i += 256) {
        const int wr = i / OW, oy = oy0 + wr, ox = i - wr * OW;
        float best = 0.0f;
        int a = -1;
        if (oy < OH) {
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = oy * 2 - 1 + ky;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = ox * 2 - 1 + kx;
                    if (xx < 0 || xx >= W) continue;
                    const float v = xs[(yy - y_lo) * W + xx];
                    if (a < 0 || v > best) { best = v; a = yy * W + xx; }
                }
            }
        }
        arg[i] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * MPB_R * W; i += 256) {
        const int r = i / W, iy = 2 * oy0 + r, ix = i - r * W;
        if (iy >= H) break;
        float g = 0.0f;
        for (int oy = max(iy / 2, 0); oy <= min((iy + 1) / 2, OH - 1); ++oy)
            for (int ox = max(ix / 2, 0); ox <= min((ix + 1) / 2, OW - 1); ++ox)
                if (arg[(oy - oy0) * OW + ox] == iy * W + ix) g += dyp[oy * OW + ox];
        dx[plane * H * W + iy * W + ix] = g;
    }
}

// dX[b][c][max_idx[b][c][m]] += dV[b][c][m] * mask[b][m]   (index_max + gather + mask of networks_pc.py:88-93,101-104)
__global__ __launch_bounds__(256) void segment_max_backward_kernel(const float* __restrict__ dv, const int* __restrict__ max_idx, const float* __restrict__ mask,
                                                                   float* __restrict__ dx, int C, int N, int M, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int m = (int)(i % M);
    const long long bc = i / M;
    const int b = (int)(bc / C);
    const float w = mask ? mask[(long long)b * M + m] : 1.0f;
    if (w == 0.0f) return;
    const int n = max_idx[i];
    if (n >= 0 && n < N) atomicAdd(dx + bc * N + n, dv[i] * w);
}

// y[row] = max_k x[row][k] with the first arg-max (torch.max(dim) rule on ties: lowest index)
__global__ __launch_bounds__(256) void group_max_forward_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ arg, int K, long long rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const float* p = x + i * K;
    float best = p[0];
    int a = 0;
    for (int k = 1; k < K; ++k) {
        const float v = p[k];
        if (v > best || (v != v && best == best)) { best = v; a = k; }
    }
    y[i] = best;
    arg[i] = a;
}
__global__ __launch_bounds__(256) void group_max_backward_kernel(const float* __restrict__ dy, const int* __restrict__ arg, float* __restrict__ dx, int K, long long rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    dx[i * K + arg[i]] = dy[i];
}

__global__ __launch_bounds__(256) void apply_mask_kernel(const float* __restrict__ x, const unsigned char* __restrict__ mask, float scale, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = mask[i] ? x[i] * scale : 0.0f;
}
// eight elements per thread (two 16-byte loads, one 8-byte mask load, two 16-byte stores); n % 8 == 0, 16-byte aligned x / y, 8-byte aligned mask
__global__ __launch_bounds__(256) void apply_mask8_kernel(const float4* __restrict__ x, const uint2* __restrict__ mask, float scale, float4* __restrict__ y, long long n8) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = x[2 * i], b = x[2 * i + 1];
    const uint2 m = mask[i];
    y[2 * i] = make_float4((m.x & 0xffu) ? a.x * scale : 0.0f, (m.x & 0xff00u) ? a.y * scale : 0.0f, (m.x & 0xff0000u) ? a.z * scale : 0.0f,
                           (m.x & 0xff000000u) ? a.w * scale : 0.0f);
    y[2 * i + 1] = make_float4((m.y & 0xffu) ? b.x * scale : 0.0f, (m.y & 0xff00u) ? b.y * scale : 0.0f, (m.y & 0xff0000u) ? b.z * scale : 0.0f,
                               (m.y & 0xff000000u) ? b.w * scale : 0.0f);
}

}  // namespace

// =============================================================================================== C ABI
extern "C" long long di2p_channel_reduce_workspace_bytes(int B, int C, int N) {
    if (B < 1 || C < 1 || N < 1) return 0;
    return red_plan(B, C, N).bytes;
}

extern "C" int di2p_bn_train_forward(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* save_mean,
                                     float* save_invstd, float* running_mean, float* running_var, float momentum, float eps, int relu, int B, int C,
                                     int N, void* workspace, void* stream) {
    DI2P_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && workspace, "null pointer");
    DI2P_CHECK_ARG(B >= 1 && C >= 1 && N >= 1 && (long long)B * N >= 2, "batch statistics need at least two values per channel");
    DI2P_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running_mean / running_var go together");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan p = red_plan(B, C, N);
    double* part = (double*)workspace;
    hipLaunchKernelGGL((channel_reduce_kernel<0>), dim3(C, p.S), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0, C, N, p.SN, p.per, part);
    if (di2p_opt(DI2P_OPT_BN_UNFUSED)) {          // rounds 2-5: finalize and the elementwise pass as two launches (same results)
        hipLaunchKernelGGL((channel_finalize_kernel<0>), dim3(di2p_cdiv(C, 64)), dim3(64), 0, st, (const double*)part, p.S, C, (double)B * N, eps, momentum,
                           save_mean, save_invstd, running_mean, running_var, (float*)nullptr);
        const long long total = (long long)B * C * N;
        hipLaunchKernelGGL(bn_apply_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st, x, gamma, beta, (const float*)save_mean, (const float*)save_invstd,
                           residual, y, relu, C, N, total);
    } else {
        hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(C, p.S), dim3(256), 0, st, x, gamma, beta, residual, y, (const double*)part, p.S, (double)B * N, eps,
                           momentum, save_mean, save_invstd, running_mean, running_var, relu, C, N, p.SN, p.per);
    }
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_bn_train_backward(const float* x, const float* y, const float* dy, const float* gamma, const float* save_mean,
                                      const float* save_invstd, int relu, float* dx, float* dresidual, float* dgamma, float* dbeta, int B, int C, int N,
                                      void* workspace, void* stream) {
    DI2P_CHECK_ARG(x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta && workspace, "null pointer");
    DI2P_CHECK_ARG(!relu || y, "relu needs the forward output");
    DI2P_CHECK_ARG(B >= 1 && C >= 1 && N >= 1, "bad size");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan p = red_plan(B, C, N);
    double* part = (double*)workspace;
    float* sums = (float*)((char*)workspace + (long long)C * p.S * 16);
    hipLaunchKernelGGL((channel_reduce_kernel<1>), dim3(C, p.S), dim3(256), 0, st, x, dy, y, save_mean, save_invstd, relu, C, N, p.SN, p.per, part);
    if (di2p_opt(DI2P_OPT_BN_UNFUSED)) {
        hipLaunchKernelGGL((channel_finalize_kernel<1>), dim3(di2p_cdiv(C, 64)), dim3(64), 0, st, (const double*)part, p.S, C, (double)B * N, 0.0f, 0.0f, dgamma,
                           dbeta, (float*)nullptr, (float*)nullptr, sums);
        const long long total = (long long)B * C * N;
        hipLaunchKernelGGL(bn_backward_apply_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st, x, dy, y, gamma, save_mean, save_invstd, (const float*)sums,
                           relu, dx, dresidual, C, N, total);
    } else {
        hipLaunchKernelGGL(bn_finalize_backward_kernel, dim3(C, p.S), dim3(256), 0, st, x, dy, y, gamma, save_mean, save_invstd, (const double*)part, p.S,
                           (double)B * N, relu, dx, dresidual, dgamma, dbeta, C, N, p.SN, p.per);
    }
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_channel_sum(const float* x, float* out, int B, int C, int N, void* workspace, void* stream) {
    DI2P_CHECK_ARG(x && out && workspace && B >= 1 && C >= 1 && N >= 1, "bad args");
    hipStream_t st = (hipStream_t)stream;
    const RedPlan p = red_plan(B, C, N);
    double* part = (double*)workspace;
    hipLaunchKernelGGL((channel_reduce_kernel<0>), dim3(C, p.S), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0, C, N, p.SN, p.per, part);
    hipLaunchKernelGGL((channel_finalize_kernel<2>), dim3(di2p_cdiv(C, 64)), dim3(64), 0, st, (const double*)part, p.S, C, 1.0, 0.0f, 0.0f, out, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr);
    DI2P_RETURN_LAUNCH();
}

extern "C" long long di2p_bmm_rc_workspace_bytes(int Z, int rows, int cols, int R) {
    if (Z < 1 || rows < 1 || cols < 1 || R < 1) return 0;
    const long long b64 = rc_plan(Z, rows, cols, R).bytes, b128 = rc_plan(Z, rows, cols, R, RC2_BM).bytes;      // (whichever kernel the call takes)
    return b64 > b128 ? b64 : b128;
}

extern "C" int di2p_bmm_rc(const float* A, long long lda, long long a_batch_stride, const float* Bm, long long ldb, long long b_batch_stride, float* out,
                           int Z, int rows, int cols, int R, float alpha, int reduce_z, void* workspace, long long workspace_bytes, void* stream) {
    DI2P_CHECK_ARG(A && Bm && out && Z >= 1 && rows >= 1 && cols >= 1 && R >= 1, "bad args");
    RcStrided la{}; la.base = A; la.ld = lda; la.batch_stride = a_batch_stride; la.n = rows;
    RcStrided lb{}; lb.base = Bm; lb.ld = ldb; lb.batch_stride = b_batch_stride; lb.n = cols;
    return rc_launch(__func__, la, lb, Z, rows, cols, R, alpha, reduce_z != 0, out, workspace, workspace_bytes, (hipStream_t)stream,
                     rc_vec_ok(A, lda, a_batch_stride, R), rc_vec_ok(Bm, ldb, b_batch_stride, R));
}

extern "C" int di2p_bmm_km(const float* A, int lda, long long a_batch_stride, const float* Bm, int ldb, long long b_batch_stride, float* out, int Z,
                           int rows, int cols, int K, float alpha, void* stream) {
    DI2P_CHECK_ARG(A && Bm && out && Z >= 1 && rows >= 1 && cols >= 1 && K >= 1 && Z <= 65535, "bad args");
    hipLaunchKernelGGL(bmm_km_kernel, dim3(di2p_cdiv(cols, KmCfg::BN), di2p_cdiv(rows, KmCfg::BM), Z), dim3(KmCfg::THREADS), KmCfg::LDS_FLOATS * sizeof(float),
                       (hipStream_t)stream, A, lda, a_batch_stride, Bm, ldb, b_batch_stride, out, rows, cols, K, alpha);
    DI2P_RETURN_LAUNCH();
}

extern "C" long long di2p_gather_backward_workspace_bytes(int B, int C, int J, int M) {
    if (B < 1 || C < 1 || J < 1 || M < 1) return 0;
    return rc_plan(B, C, M, J).bytes;
}

extern "C" int di2p_gather_backward(const float* dy, const int32_t* idx, const float* weights, int k, float* dfeats, int B, int C, int J, int M,
                                    void* workspace, long long workspace_bytes, void* stream) {
    DI2P_CHECK_ARG(dy && idx && dfeats && B >= 1 && C >= 1 && J >= 1 && M >= 1, "bad args");
    DI2P_CHECK_ARG(k == 1 || k == 3, "k must be 1 (gather) or 3 (3-NN interpolation)");
    RcStrided la{}; la.base = dy; la.ld = J; la.batch_stride = (long long)C * J; la.n = C;
    if (k == 1) {
        RcSelect<1> lb{}; lb.idx_base = idx; lb.w_base = weights; lb.J = J; lb.M = M;
        return rc_launch(__func__, la, lb, B, C, M, J, 1.0f, false, dfeats, workspace, workspace_bytes, (hipStream_t)stream);
    }
    RcSelect<3> lb{}; lb.idx_base = idx; lb.w_base = weights; lb.J = J; lb.M = M;
    return rc_launch(__func__, la, lb, B, C, M, J, 1.0f, false, dfeats, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" long long di2p_conv2d_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad) {
    if (B < 1 || Cin < 1 || Cout < 1 || KH < 1 || KW < 1 || stride < 1) return 0;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (OH < 1 || OW < 1) return 0;
    return rc_plan(B, Cout, Cin * KH * KW, OH * OW).bytes;
}

extern "C" int di2p_conv2d_wgrad(const float* x, const float* dy, float* dW, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                 void* workspace, long long workspace_bytes, void* stream) {
    DI2P_CHECK_ARG(x && dy && dW && B >= 1 && Cin >= 1 && Cout >= 1 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "bad args");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    DI2P_CHECK_ARG(OH >= 1 && OW >= 1, "empty output");
    RcStrided la{}; la.base = dy; la.ld = (long long)OH * OW; la.batch_stride = (long long)Cout * OH * OW; la.n = Cout;
    RcIm2col lb{}; lb.base = x; lb.Cin = Cin; lb.H = H; lb.W = W; lb.OW = OW; lb.KH = KH; lb.KW = KW; lb.stride = stride; lb.pad = pad; lb.ncols = Cin * KH * KW;
    return rc_launch(__func__, la, lb, B, Cout, Cin * KH * KW, OH * OW, 1.0f, true, dW, workspace, workspace_bytes, (hipStream_t)stream,
                     rc_vec_ok(dy, (long long)OH * OW, (long long)Cout * OH * OW, OH * OW));
}

extern "C" int di2p_conv2d_dgrad(const float* dy, const float* Wgt, float* dx, int B, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                 void* stream) {
    DI2P_CHECK_ARG(dy && Wgt && dx && B >= 1 && B <= 65535 && Cin >= 1 && Cout >= 1 && KH >= 1 && KW >= 1 && stride >= 1 && pad >= 0, "bad args");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    DI2P_CHECK_ARG(OH >= 1 && OW >= 1, "empty output");
    if (stride == 2 && B <= 16383 && !di2p_opt(DI2P_OPT_CONV_DGRAD_DENSE)) {          // one parity class of input pixels per workgroup: 1/4 of the matrix work
        const int ncol = ((H + 1) / 2) * ((W + 1) / 2);
        hipLaunchKernelGGL(conv_dgrad_s2_kernel, dim3(di2p_cdiv(ncol, KmCfg::BN), di2p_cdiv(Cin, KmCfg::BM), B * 4), dim3(KmCfg::THREADS),
                           KmCfg::LDS_FLOATS * sizeof(float), (hipStream_t)stream, dy, Wgt, dx, Cin, H, W, Cout, KH, KW, pad, OH, OW);
        DI2P_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(conv_dgrad_kernel, dim3(di2p_cdiv((long long)H * W, KmCfg::BN), di2p_cdiv(Cin, KmCfg::BM), B), dim3(KmCfg::THREADS),
                       KmCfg::LDS_FLOATS * sizeof(float), (hipStream_t)stream, dy, Wgt, dx, Cin, H, W, Cout, KH, KW, stride, pad, OH, OW);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int B, int C, int H, int W, void* stream) {
    DI2P_CHECK_ARG(x && dy && dx && B >= 1 && C >= 1 && H >= 1 && W >= 1, "bad args");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * C * H * W;
    const size_t lds = ((size_t)(2 * MPB_R + 3) * W + (size_t)(MPB_R + 1) * OW) * 4;
    if (lds <= 64 * 1024 && (long long)B * C <= 65535) {
        hipLaunchKernelGGL(maxpool_backward_lds_kernel, dim3(di2p_cdiv(OH, MPB_R), B * C), dim3(256), lds, st, x, dy, dx, H, W, OH, OW);
        DI2P_RETURN_LAUNCH();
    }
    hipLaunchKernelGGL(maxpool_backward_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st, x, dy, dx, H, W, OH, OW, total);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_segment_max_backward(const float* dvalues, const int32_t* max_idx, const float* mask, float* dx, int B, int C, int N, int M,
                                         void* stream) {
    DI2P_CHECK_ARG(dvalues && max_idx && dx && B >= 1 && C >= 1 && N >= 1 && M >= 1, "bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dx, 0, (size_t)B * C * N * 4, st) != hipSuccess) { di2p_set_error("%s: memset failed", __func__); return -1; }
    const long long total = (long long)B * C * M;
    hipLaunchKernelGGL(segment_max_backward_kernel, dim3(di2p_cdiv(total, 256)), dim3(256), 0, st, dvalues, max_idx, mask, dx, C, N, M, total);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_group_max_forward(const float* x, float* y, int32_t* arg, long long rows, int K, void* stream) {
    DI2P_CHECK_ARG(x && y && arg && rows >= 1 && K >= 1, "bad args");
    hipLaunchKernelGGL(group_max_forward_kernel, dim3(di2p_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, y, arg, K, rows);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_group_max_backward(const float* dy, const int32_t* arg, float* dx, long long rows, int K, void* stream) {
    DI2P_CHECK_ARG(dy && arg && dx && rows >= 1 && K >= 1, "bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dx, 0, (size_t)rows * K * 4, st) != hipSuccess) { di2p_set_error("%s: memset failed", __func__); return -1; }
    hipLaunchKernelGGL(group_max_backward_kernel, dim3(di2p_cdiv(rows, 256)), dim3(256), 0, st, dy, arg, dx, K, rows);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_apply_mask(const float* x, const uint8_t* mask, float scale, float* y, long long n, void* stream) {
    DI2P_CHECK_ARG(x && mask && y && n >= 0, "bad args");
    if (n == 0) return 0;
    if (n % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && ((uintptr_t)mask & 7) == 0)
        hipLaunchKernelGGL(apply_mask8_kernel, dim3(di2p_cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (const uint2*)mask, scale, (float4*)y, n / 8);
    else
        hipLaunchKernelGGL(apply_mask_kernel, dim3(di2p_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (const unsigned char*)mask, scale, y, n);
    DI2P_RETURN_LAUNCH();
}
