// Training-side head of the classifier for gfx950 (first slice): the two losses of
// models/multimodal_classifier.py:189-191 and their gradients with respect to the logits, in one pass over the scores.
//   coarse: models/focal_loss.py:55-112  FocalLoss(alpha=0.5, gamma=2, reduction='mean') on [B,2,N] scores, times coarse_loss_alpha
//           p = softmax(x) + 1e-6 (FocalLoss.eps, :159,165) ; h = one_hot(label) + 1e-6 ; loss = mean_{b,n} sum_c h_c * (-alpha (1 - p_c)^gamma log p_c)
//   fine:   nn.CrossEntropyLoss (mean) over the points with coarse label 1 only (:169-190), L classes
// plus the accuracies of :195-201.  Replaces the softmax / one_hot / gather / sort tensors and autograd's backward through them:
// d loss / d scores is written directly (zero rows for points outside the image in the fine head).
// Arithmetic in fp32 like torch (softmax as exp(x - max) / sum), the batch means accumulated in fp64 in a fixed order
// (per-workgroup partials, then one ordered pass): deterministic.
#include "common.h"

#include <math.h>

namespace {

constexpr int LOSS_T = 256;

__device__ __forceinline__ void block_sum5(double* v, double (*sh)[LOSS_T]) {      // fixed-order tree over the workgroup
    const int tid = threadIdx.x;
    for (int a = 0; a < 5; ++a) sh[a][tid] = v[a];
    __syncthreads();
    for (int o = LOSS_T / 2; o > 0; o >>= 1) {
        if (tid < o) for (int a = 0; a < 5; ++a) sh[a][tid] += sh[a][tid + o];
        __syncthreads();
    }
    for (int a = 0; a < 5; ++a) v[a] = sh[a][0];
}

// per-point loss terms -> partials[block][5] = {sum focal, sum fine CE, inside count, coarse correct, fine correct}
__global__ __launch_bounds__(LOSS_T) void loss_forward_kernel(const float* __restrict__ coarse, const float* __restrict__ fine,
                                                              const int* __restrict__ clab, const int* __restrict__ flab, int N, int L,
                                                              float alpha, float gamma, double* __restrict__ partials) {
    __shared__ double sh[5][LOSS_T];
    const int b = blockIdx.y, n = blockIdx.x * LOSS_T + threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    if (n < N) {
        const float x0 = coarse[((long long)b * 2) * N + n], x1 = coarse[((long long)b * 2 + 1) * N + n];
        const int lab = clab[(long long)b * N + n];
        const float mx = fmaxf(x0, x1);
        const float e0 = expf(x0 - mx), e1 = expf(x1 - mx), inv = 1.0f / (e0 + e1);
        const float p0 = e0 * inv + 1e-6f, p1 = e1 * inv + 1e-6f;
        const float f0 = -alpha * powf(-p0 + 1.0f, gamma) * logf(p0), f1 = -alpha * powf(-p1 + 1.0f, gamma) * logf(p1);
        const float h0 = (lab == 0 ? 1.0f : 0.0f) + 1e-6f, h1 = (lab == 1 ? 1.0f : 0.0f) + 1e-6f;
        acc[0] = (double)(h0 * f0 + h1 * f1);
        acc[3] = ((x1 > x0) ? 1 : 0) == lab ? 1.0 : 0.0;          // torch.max: first maximum wins -> class 0 on a tie
        if (fine && lab == 1) {
            const float* fp = fine + (long long)b * L * N + n;
            float m = fp[0];
            int am = 0;
            for (int c = 1; c < L; ++c) { const float v = fp[(long long)c * N]; if (v > m) { m = v; am = c; } }
            float s = 0.0f;
            for (int c = 0; c < L; ++c) s += expf(fp[(long long)c * N] - m);
            const int fl = flab[(long long)b * N + n];
            acc[1] = (double)(logf(s) - (fp[(long long)fl * N] - m));      // -log_softmax[label]
            acc[2] = 1.0;
            acc[4] = am == fl ? 1.0 : 0.0;
        }
    }
    block_sum5(acc, sh);
    if (threadIdx.x == 0) {
        double* o = partials + ((long long)b * gridDim.x + blockIdx.x) * 5;
        for (int a = 0; a < 5; ++a) o[a] = acc[a];
    }
}

// one workgroup: ordered sum of the partials -> out[8] = {loss, coarse loss, fine loss, coarse acc, fine acc, inside count, 0, 0}
__global__ __launch_bounds__(LOSS_T) void loss_reduce_kernel(const double* __restrict__ partials, int nblocks, long long BN,
                                                             float coarse_loss_alpha, int has_fine, double* __restrict__ out) {
    __shared__ double sh[5][LOSS_T];
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += LOSS_T) for (int a = 0; a < 5; ++a) acc[a] += partials[(long long)i * 5 + a];
    block_sum5(acc, sh);
    if (threadIdx.x == 0) {
        const double cl = acc[0] / (double)BN * (double)coarse_loss_alpha;
        const double fl = has_fine ? acc[1] / acc[2] : 0.0;           // 0/0 = NaN when no point is inside, as torch's mean of nothing
        out[0] = cl + fl; out[1] = cl; out[2] = fl; out[3] = acc[3] / (double)BN; out[4] = has_fine ? acc[4] / acc[2] : 0.0;
        out[5] = acc[2]; out[6] = 0.0; out[7] = 0.0;
    }
}

__global__ __launch_bounds__(LOSS_T) void loss_backward_kernel(const float* __restrict__ coarse, const float* __restrict__ fine,
                                                               const int* __restrict__ clab, const int* __restrict__ flab, int N, int L,
                                                               float alpha, float gamma, float coarse_scale, const double* __restrict__ out,
                                                               float* __restrict__ d_coarse, float* __restrict__ d_fine) {
    const int b = blockIdx.y, n = blockIdx.x * LOSS_T + threadIdx.x;
    if (n >= N) return;
    const float x0 = coarse[((long long)b * 2) * N + n], x1 = coarse[((long long)b * 2 + 1) * N + n];
    const int lab = clab[(long long)b * N + n];
    const float mx = fmaxf(x0, x1);
    const float e0 = expf(x0 - mx), e1 = expf(x1 - mx), inv = 1.0f / (e0 + e1);
    const float s0 = e0 * inv, s1 = e1 * inv;
    const float p[2] = {s0 + 1e-6f, s1 + 1e-6f};
    float g[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float h = (lab == c ? 1.0f : 0.0f) + 1e-6f;
        const float om = -p[c] + 1.0f;
        // d/dp [ -alpha (1-p)^gamma log p ] = alpha ( gamma (1-p)^(gamma-1) log p - (1-p)^gamma / p )
        g[c] = h * alpha * (gamma * powf(om, gamma - 1.0f) * logf(p[c]) - powf(om, gamma) / p[c]);
    }
    const float dot = g[0] * s0 + g[1] * s1;
    d_coarse[((long long)b * 2) * N + n] = coarse_scale * s0 * (g[0] - dot);
    d_coarse[((long long)b * 2 + 1) * N + n] = coarse_scale * s1 * (g[1] - dot);
    if (fine) {
        const float* fp = fine + (long long)b * L * N + n;
        float* dp = d_fine + (long long)b * L * N + n;
        if (lab == 1) {
            float m = fp[0];
            for (int c = 1; c < L; ++c) m = fmaxf(m, fp[(long long)c * N]);
            float s = 0.0f;
            for (int c = 0; c < L; ++c) s += expf(fp[(long long)c * N] - m);
            const float is = 1.0f / s, in = (float)(1.0 / out[5]);
            const int fl = flab[(long long)b * N + n];
            for (int c = 0; c < L; ++c) dp[(long long)c * N] = (expf(fp[(long long)c * N] - m) * is - (c == fl ? 1.0f : 0.0f)) * in;
        } else {
            for (int c = 0; c < L; ++c) dp[(long long)c * N] = 0.0f;
        }
    }
}

// torch.optim.Adam step (betas, eps, weight_decay = 0 as the reference builds it, multimodal_classifier.py:44-47) on a flat buffer
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}

}  // namespace

extern "C" long long di2p_classifier_loss_workspace_bytes(int B, int N) {
    if (B < 0 || N < 0) return 0;
    return (long long)B * di2p_cdiv(N, LOSS_T) * 5 * 8 + 256;
}

extern "C" int di2p_classifier_loss(const float* coarse, const float* fine, const int32_t* coarse_labels, const int32_t* fine_labels,
                                    int B, int N, int L, float alpha, float gamma, float coarse_loss_alpha, double* out8,
                                    float* d_coarse, float* d_fine, void* workspace, void* stream) {
    DI2P_CHECK_ARG(coarse && coarse_labels && out8 && workspace && B >= 0 && N >= 1, "bad args");
    DI2P_CHECK_ARG(!fine || (fine_labels && L >= 1), "fine scores need fine labels and L >= 1");
    DI2P_CHECK_ARG(!d_fine || fine, "d_fine without fine scores");
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(di2p_cdiv(N, LOSS_T), B);
    double* partials = (double*)workspace;
    hipLaunchKernelGGL(loss_forward_kernel, grid, dim3(LOSS_T), 0, st, coarse, fine, coarse_labels, fine_labels, N, L, alpha, gamma, partials);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(LOSS_T), 0, st, partials, (int)(grid.x * grid.y), (long long)B * N,
                       coarse_loss_alpha, fine ? 1 : 0, out8);
    if (d_coarse)
        hipLaunchKernelGGL(loss_backward_kernel, grid, dim3(LOSS_T), 0, st, coarse, d_fine ? fine : (const float*)nullptr, coarse_labels,
                           fine_labels, N, L, alpha, gamma, coarse_loss_alpha / (float)((long long)B * N), out8, d_coarse, d_fine);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr,
                              float beta1, float beta2, float eps, void* stream) {
    DI2P_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "bad args");
    if (n == 0) return 0;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(di2p_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, bc1, bc2);
    DI2P_RETURN_LAUNCH();
}
