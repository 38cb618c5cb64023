// "Next" rows of the hot path: what feeds the classifier and what consumes its labels.
//
//  * farthest point sampling of the SO-Net nodes -- replaces the per-sample numpy loop of
//    data/kitti_helper.py:224-243 (FarthestSampler.sample, called at data/kitti_pc_img_pose_loader.py:416-423):
//    128 argmax sweeps over 1024 points, twice per frame, in DataLoader worker processes.  Here one workgroup per
//    (frame, node set) keeps the running min-distances in registers; numpy semantics are kept exactly: fp64
//    distances ((dx*dx + dy*dy) + dz*dz, the sum over axis 0), first-occurrence argmax, np.minimum update.
//  * index gather for the random down-sampling (data/kitti_pc_img_pose_loader.py:158-171): the RNG stays on the
//    host as an explicit index list, like the solver's restart list.
//  * ground-truth label projection, accuracies and the 7 x N "pc_label" hand-off record of
//    evaluation/visualize_and_save_data.py:100-147,174-186 (consumed by evaluation/registration_lsq.py:291-302),
//    kept in HBM instead of going through .npy files.
#include "common.h"

namespace {

constexpr int FPS_THREADS = 256;
constexpr int FPS_MAX_PER_THREAD = 16;   // M <= 4096

__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ pts, const int* __restrict__ init_idx,
                                                          int* __restrict__ idx_out, float* __restrict__ nodes_out, int M, int k) {
    __shared__ double s_val[FPS_THREADS];
    __shared__ int s_idx[FPS_THREADS];
    __shared__ double s_p[3];
    __shared__ int s_sel;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* px = pts + (long long)b * 3 * M;
    double X[FPS_MAX_PER_THREAD], Y[FPS_MAX_PER_THREAD], Z[FPS_MAX_PER_THREAD], D[FPS_MAX_PER_THREAD];
#pragma unroll
    for (int j = 0; j < FPS_MAX_PER_THREAD; ++j) {
        const int m = tid + j * FPS_THREADS;
        X[j] = m < M ? (double)px[m] : 0.0;
        Y[j] = m < M ? (double)px[M + m] : 0.0;
        Z[j] = m < M ? (double)px[2 * (long long)M + m] : 0.0;
        D[j] = __builtin_inf();
    }
    int sel = init_idx ? init_idx[b] : 0;
    for (int i = 0; i < k; ++i) {
        if (tid == 0) {
            s_p[0] = (double)px[sel]; s_p[1] = (double)px[M + sel]; s_p[2] = (double)px[2 * (long long)M + sel];
            idx_out[(long long)b * k + i] = sel;
            if (nodes_out) {
                for (int c = 0; c < 3; ++c) nodes_out[((long long)b * 3 + c) * k + i] = px[(long long)c * M + sel];
            }
        }
        __syncthreads();
        if (i + 1 == k) break;
        const double p0 = s_p[0], p1 = s_p[1], p2 = s_p[2];
        double best = -1.0;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < FPS_MAX_PER_THREAD; ++j) {
            const int m = tid + j * FPS_THREADS;
            if (m < M) {
                const double dx = p0 - X[j], dy = p1 - Y[j], dz = p2 - Z[j];
                const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                D[j] = fmin(D[j], d);                        // first round: distances = calc_distances(p0, pts)
                if (D[j] > best || (D[j] == best && m < bi)) { best = D[j]; bi = m; }
            }
        }
        s_val[tid] = best; s_idx[tid] = bi;
        __syncthreads();
        for (int o = FPS_THREADS / 2; o > 0; o >>= 1) {
            if (tid < o) {
                const double v = s_val[tid + o];
                const int vi = s_idx[tid + o];
                if (v > s_val[tid] || (v == s_val[tid] && vi < s_idx[tid])) { s_val[tid] = v; s_idx[tid] = vi; }
            }
            __syncthreads();
        }
        if (tid == 0) s_sel = s_idx[0];
        __syncthreads();
        sel = s_sel;
    }
}

__global__ void gather_points_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, int C,
                                     int Nsrc, int Nout) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Nout) return;
    out[((long long)b * C + c) * Nout + n] = src[((long long)b * C + c) * Nsrc + idx[(long long)b * Nout + n]];
}

// visualize_and_save_data.py:100-115,138-139: fp32 like the reference's torch.matmul on float tensors
__global__ void project_labels_kernel(const float* __restrict__ pc, const float* __restrict__ P, int p_rows, const float* __restrict__ K,
                                      float H, float W, float scale, int W_fine, int* __restrict__ coarse, int* __restrict__ fine,
                                      float* __restrict__ pxpy, int N) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* p = pc + (long long)b * 3 * N;
    const float* Pb = P + (long long)b * p_rows * 4;
    const float* Kb = K + (long long)b * 9;
    const float x = p[n], y = p[N + n], z = p[2 * (long long)N + n];
    float cam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        cam[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Pb[r * 4], x), __fmul_rn(Pb[r * 4 + 1], y)), __fmul_rn(Pb[r * 4 + 2], z)), Pb[r * 4 + 3]);
    float kp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        kp[r] = __fadd_rn(__fadd_rn(__fmul_rn(Kb[r * 3], cam[0]), __fmul_rn(Kb[r * 3 + 1], cam[1])), __fmul_rn(Kb[r * 3 + 2], cam[2]));
    const float u = __fdiv_rn(kp[0], kp[2]), v = __fdiv_rn(kp[1], kp[2]);
    const bool inside = u >= 0.0f && u <= W - 1.0f && v >= 0.0f && v <= H - 1.0f && cam[2] > 0.1f;
    coarse[(long long)b * N + n] = inside ? 1 : 0;
    if (fine) fine[(long long)b * N + n] = (int)floorf(__fdiv_rn(u, scale)) + (int)floorf(__fdiv_rn(v, scale)) * W_fine;
    if (pxpy) { pxpy[((long long)b * 2) * N + n] = u; pxpy[((long long)b * 2 + 1) * N + n] = v; }
}

// one workgroup per frame: coarse accuracy = mean(pred == gt); fine accuracy = mean over gt-inside points
__global__ __launch_bounds__(256) void label_accuracy_kernel(const int* __restrict__ cp, const int* __restrict__ cg, const int* __restrict__ fp,
                                                             const int* __restrict__ fg, float* __restrict__ out, int N) {
    __shared__ int s_a[256], s_b[256], s_c[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    int ok_c = 0, n_in = 0, ok_f = 0;
    for (int n = tid; n < N; n += 256) {
        const long long i = (long long)b * N + n;
        ok_c += cp[i] == cg[i];
        if (cg[i] == 1) { ++n_in; if (fp && fg) ok_f += fp[i] == fg[i]; }
    }
    s_a[tid] = ok_c; s_b[tid] = n_in; s_c[tid] = ok_f;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_a[tid] += s_a[tid + o]; s_b[tid] += s_b[tid + o]; s_c[tid] += s_c[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[b * 2] = (float)((double)s_a[0] / (double)N);
        out[b * 2 + 1] = s_b[0] > 0 ? (float)((double)s_c[0] / (double)s_b[0]) : __builtin_nanf("");   // np.mean of empty = nan
    }
}

__global__ void pack_pc_label_kernel(const float* __restrict__ pc, const int* __restrict__ cp, const int* __restrict__ cg,
                                     const int* __restrict__ fp, const int* __restrict__ fg, double* __restrict__ out, int N) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double* o = out + (long long)b * 7 * N;
    const long long i = (long long)b * N + n;
    for (int c = 0; c < 3; ++c) o[(long long)c * N + n] = (double)pc[((long long)b * 3 + c) * N + n];
    o[3ll * N + n] = (double)cp[i];
    o[4ll * N + n] = (double)cg[i];
    o[5ll * N + n] = (double)(fp ? fp[i] : cp[i]);   // coarse-only models store the coarse prediction twice (:97-98)
    o[6ll * N + n] = (double)(fg ? fg[i] : 0);
}

}  // namespace

extern "C" int di2p_farthest_point_sampling(const float* pts, const int32_t* init_idx, int32_t* idx_out, float* nodes_out, int B,
                                            int M, int k, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && M >= 1 && k >= 1, "bad size");
    DI2P_CHECK_ARG(M <= FPS_THREADS * FPS_MAX_PER_THREAD, "M too large (max 4096 candidates)");
    if (B == 0) return 0;
    DI2P_CHECK_ARG(pts && idx_out, "null pointer");
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(FPS_THREADS), 0, (hipStream_t)stream, pts, init_idx, idx_out, nodes_out, M, k);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_gather_points(const float* src, const int32_t* idx, float* out, int B, int C, int Nsrc, int Nout, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && C >= 1 && Nsrc >= 1 && Nout >= 0, "bad size");
    if (B == 0 || Nout == 0) return 0;
    DI2P_CHECK_ARG(src && idx && out, "null pointer");
    hipLaunchKernelGGL(gather_points_kernel, dim3(di2p_cdiv(Nout, 256), C, B), dim3(256), 0, (hipStream_t)stream, src, idx, out, C, Nsrc, Nout);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_project_labels(const float* pc, const float* P, int p_rows, const float* K, float H, float W, float fine_scale,
                                   int32_t* coarse, int32_t* fine, float* pxpy, int B, int N, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && N >= 0 && (p_rows == 3 || p_rows == 4) && fine_scale > 0, "bad args");
    if (B == 0 || N == 0) return 0;
    DI2P_CHECK_ARG(pc && P && K && coarse, "null pointer");
    const int W_fine = (int)lroundf(W / fine_scale);
    hipLaunchKernelGGL(project_labels_kernel, dim3(di2p_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, pc, P, p_rows, K, H, W,
                       fine_scale, W_fine, coarse, fine, pxpy, N);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_label_accuracy(const int32_t* coarse_pred, const int32_t* coarse_gt, const int32_t* fine_pred,
                                   const int32_t* fine_gt, float* out, int B, int N, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && N >= 1, "bad size");
    if (B == 0) return 0;
    DI2P_CHECK_ARG(coarse_pred && coarse_gt && out, "null pointer");
    hipLaunchKernelGGL(label_accuracy_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, coarse_pred, coarse_gt, fine_pred, fine_gt, out, N);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_pack_pc_label(const float* pc, const int32_t* coarse_pred, const int32_t* coarse_gt, const int32_t* fine_pred,
                                  const int32_t* fine_gt, double* out, int B, int N, void* stream) {
    DI2P_CHECK_ARG(B >= 0 && N >= 0, "bad size");
    if (B == 0 || N == 0) return 0;
    DI2P_CHECK_ARG(pc && coarse_pred && coarse_gt && out, "null pointer");
    hipLaunchKernelGGL(pack_pc_label_kernel, dim3(di2p_cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, pc, coarse_pred, coarse_gt,
                       fine_pred, fine_gt, out, N);
    DI2P_RETURN_LAUNCH();
}
