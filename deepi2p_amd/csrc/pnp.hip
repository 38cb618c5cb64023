// PnP-RANSAC back end of BASELINE config 3 for gfx950 -- replaces cv2.solvePnPRansac(EPNP, 500 iterations,
// reprojectionError 0.6) as called by evaluation/registration_pnp.py:95-148 (solve_PnP), one CPU call per frame.
//
// OpenCV is not in the reference tree nor in this image and its RANSAC RNG is internal: PARITY UNPINNED (SURVEY.md
// 8c).  The algorithm is therefore stated explicitly (and mirrored by oracle/pnp_np.py):
//   1. correspondences (registration_pnp.py:97-110): points with coarse label 1; pixel = (fine % W_f, fine / W_f) in the
//      1/32-scaled image, K scaled likewise (camera_matrix_scaling, :58-61).  Compacted once per frame.
//   2. hypotheses: for each of `iters` samples of 6 correspondences (an explicit input table, like the solver's
//      restart list) a normalised DLT: 11 of the 12 linear equations in normalised image coordinates, null vector by
//      Gaussian elimination with partial pivoting, scale from |r3| = 1, sign from positive depth, rotation
//      orthonormalised by Newton polar iteration.  One thread per hypothesis.
//   3. scoring: one WAVEFRONT per hypothesis strides over the frame's correspondences and counts reprojection
//      errors < reproj_err (and positive depth).
//   4. per frame: best hypothesis (most inliers, ties -> lowest sample id), then locally optimised: up to
//      `refine_rounds` rounds of {inlier set of the current model -> `refine_iters` Gauss-Newton steps on the 6-DoF
//      reprojection error over it (left-multiplicative rotation update)}, a round kept only if it loses no inliers.
//   5. acceptance as the reference: success (>= 6 inliers) and |t| < 14.14, else identity / outlier ratio 1 (:134-140).
#include "common.h"
#include "epnp.h"

#include <float.h>
#include <math.h>

namespace {

struct __attribute__((aligned(16))) Corr { float x, y, z, u, v, pad0, pad1, pad2; };

__global__ __launch_bounds__(256) void pnp_pack_kernel(const float* __restrict__ pc, const int* __restrict__ coarse,
                                                       const int* __restrict__ fine, const float* __restrict__ pixels, int N, int W_fine,
                                                       Corr* __restrict__ out, int* __restrict__ counts) {
    __shared__ int s_scan[256];
    __shared__ int s_base;
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* p = pc + (long long)f * 3 * N;
    Corr* o = out + (long long)f * N;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += 256) {
        const int n = n0 + tid;
        const int keep = (n < N && coarse[(long long)f * N + n] == 1) ? 1 : 0;
        s_scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int v = tid >= off ? s_scan[tid - off] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        if (keep) {
            Corr c;
            c.x = p[n]; c.y = p[N + n]; c.z = p[2 * (long long)N + n];
            if (pixels) {            // explicit 2-D observations (tests / other front ends)
                c.u = pixels[((long long)f * 2) * N + n]; c.v = pixels[((long long)f * 2 + 1) * N + n];
            } else {
                const int fl = fine[(long long)f * N + n];
                const int py = (int)floorf((float)fl / (float)W_fine);      // np.floor(fine / W) (:108)
                c.u = (float)(fl - py * W_fine); c.v = (float)py;
            }
            c.pad0 = c.pad1 = c.pad2 = 0.f;
            o[s_base + s_scan[tid] - 1] = c;
        }
        __syncthreads();
        if (tid == 255) s_base += s_scan[255];
        __syncthreads();
    }
    if (tid == 0) counts[f] = s_base;
}

__device__ __forceinline__ bool inv3(const double* M, double* inv) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(fabs(det) > 1e-300) || !isfinite(det)) return false;
    const double id = 1.0 / det;
    inv[0] = (e * i - f * h) * id; inv[1] = (c * h - b * i) * id; inv[2] = (b * f - c * e) * id;
    inv[3] = (f * g - d * i) * id; inv[4] = (a * i - c * g) * id; inv[5] = (c * d - a * f) * id;
    inv[6] = (d * h - e * g) * id; inv[7] = (b * g - a * h) * id; inv[8] = (a * e - b * d) * id;
    return true;
}

// hypothesis layout: R (row-major 9), t (3), valid flag as double (12 + 1)
constexpr int HYP = 13;

__global__ __launch_bounds__(64) void pnp_hypotheses_kernel(const Corr* __restrict__ corr, const int* __restrict__ counts,
                                                            const double* __restrict__ Kmat, const int* __restrict__ samples, int N,
                                                            int iters, double* __restrict__ hyp) {
    const int f = blockIdx.y;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= iters) return;
    double* h = hyp + ((long long)f * iters + it) * HYP;
    h[12] = 0.0;
    const int cnt = counts[f];
    if (cnt < 6) return;
    const Corr* c = corr + (long long)f * N;
    const double* K = Kmat + (long long)f * 9;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double X[6][3], xh[6], yh[6];
    double cen[3] = {0, 0, 0};
    for (int j = 0; j < 6; ++j) {
        int s = samples[((long long)f * iters + it) * 6 + j] % cnt;
        if (s < 0) s += cnt;
        const Corr q = c[s];
        X[j][0] = q.x; X[j][1] = q.y; X[j][2] = q.z;
        xh[j] = ((double)q.u - cx) / fx; yh[j] = ((double)q.v - cy) / fy;
        for (int a = 0; a < 3; ++a) cen[a] += X[j][a] / 6.0;
    }
    double sc = 0.0;
    for (int j = 0; j < 6; ++j) {
        double d2 = 0.0;
        for (int a = 0; a < 3; ++a) { X[j][a] -= cen[a]; d2 += X[j][a] * X[j][a]; }
        sc += sqrt(d2) / 6.0;
    }
    if (!(sc > 1e-9)) return;
    for (int j = 0; j < 6; ++j) for (int a = 0; a < 3; ++a) X[j][a] /= sc;
    // 11 x 12 system A p = 0 (rows 2j, 2j+1 of point j; the 12th row is dropped)
    double A[11][12];
    for (int j = 0; j < 6; ++j) {
        const double x = X[j][0], y = X[j][1], z = X[j][2];
        const int r0 = 2 * j, r1 = 2 * j + 1;
        const double row0[12] = {x, y, z, 1, 0, 0, 0, 0, -xh[j] * x, -xh[j] * y, -xh[j] * z, -xh[j]};
        const double row1[12] = {0, 0, 0, 0, x, y, z, 1, -yh[j] * x, -yh[j] * y, -yh[j] * z, -yh[j]};
        for (int a = 0; a < 12; ++a) { A[r0][a] = row0[a]; if (r1 < 11) A[r1][a] = row1[a]; }
    }
    // reduced row echelon form with partial pivoting; the single pivot-free column gives the null vector
    int piv_col[11];
    int row = 0, free_col = -1;
    for (int col = 0; col < 12 && row < 11; ++col) {
        int best = row;
        double bv = fabs(A[row][col]);
        for (int r = row + 1; r < 11; ++r) if (fabs(A[r][col]) > bv) { bv = fabs(A[r][col]); best = r; }
        if (bv < 1e-10) { if (free_col >= 0) return; free_col = col; continue; }   // rank deficient by more than one: degenerate
        if (best != row) for (int a = 0; a < 12; ++a) { const double t = A[row][a]; A[row][a] = A[best][a]; A[best][a] = t; }
        const double ip = 1.0 / A[row][col];
        for (int a = 0; a < 12; ++a) A[row][a] *= ip;
        for (int r = 0; r < 11; ++r) if (r != row) { const double m = A[r][col]; if (m != 0.0) for (int a = 0; a < 12; ++a) A[r][a] -= m * A[row][a]; }
        piv_col[row] = col;
        ++row;
    }
    if (row < 11) return;
    if (free_col < 0) free_col = 11;
    double p[12];
    for (int a = 0; a < 12; ++a) p[a] = 0.0;
    p[free_col] = 1.0;
    for (int r = 0; r < 11; ++r) p[piv_col[r]] = -A[r][free_col];
    // un-normalise: P = P' * T with T = [I/sc | -cen/sc]
    double Pm[3][4];
    for (int r = 0; r < 3; ++r) {
        for (int a = 0; a < 3; ++a) Pm[r][a] = p[r * 4 + a] / sc;
        Pm[r][3] = p[r * 4 + 3] - (p[r * 4] * cen[0] + p[r * 4 + 1] * cen[1] + p[r * 4 + 2] * cen[2]) / sc;
    }
    double n3 = sqrt(Pm[2][0] * Pm[2][0] + Pm[2][1] * Pm[2][1] + Pm[2][2] * Pm[2][2]);
    if (!(n3 > 1e-300) || !isfinite(n3)) return;
    const double x0 = X[0][0] * sc + cen[0], y0 = X[0][1] * sc + cen[1], z0 = X[0][2] * sc + cen[2];
    double s = 1.0 / n3;
    if ((Pm[2][0] * x0 + Pm[2][1] * y0 + Pm[2][2] * z0 + Pm[2][3]) * s < 0.0) s = -s;   // positive depth
    double R[9], t[3];
    for (int r = 0; r < 3; ++r) { for (int a = 0; a < 3; ++a) R[r * 3 + a] = Pm[r][a] * s; t[r] = Pm[r][3] * s; }
    // nearest rotation: Newton iteration R <- (R + R^-T)/2
    for (int k = 0; k < 8; ++k) {
        double inv[9];
        if (!inv3(R, inv)) return;
        for (int r = 0; r < 3; ++r) for (int a = 0; a < 3; ++a) R[r * 3 + a] = 0.5 * (R[r * 3 + a] + inv[a * 3 + r]);
    }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (!(det > 0.5)) return;
    for (int a = 0; a < 9; ++a) { if (!isfinite(R[a])) return; h[a] = R[a]; }
    for (int a = 0; a < 3; ++a) { if (!isfinite(t[a])) return; h[9 + a] = t[a]; }
    h[12] = 1.0;
}

__device__ __forceinline__ bool reproj_inlier(const Corr& q, const double* R, const double* t, double fx, double fy, double cx,
                                              double cy, double thr2) {
    const double X = q.x, Y = q.y, Z = q.z;
    const double p0 = R[0] * X + R[1] * Y + R[2] * Z + t[0], p1 = R[3] * X + R[4] * Y + R[5] * Z + t[1],
                 p2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    if (!(p2 > 1e-9)) return false;
    const double du = fx * p0 / p2 + cx - (double)q.u, dv = fy * p1 / p2 + cy - (double)q.v;
    return du * du + dv * dv < thr2;
}

// one wavefront per hypothesis
__global__ __launch_bounds__(256) void pnp_score_kernel(const Corr* __restrict__ corr, const int* __restrict__ counts,
                                                        const double* __restrict__ Kmat, const double* __restrict__ hyp, int N, int iters,
                                                        double thr2, int* __restrict__ inliers) {
    const int f = blockIdx.y;
    const int it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= iters) return;
    const int lane = threadIdx.x & 63;
    const double* h = hyp + ((long long)f * iters + it) * HYP;
    int cntin = 0;
    if (h[12] != 0.0) {
        const double* K = Kmat + (long long)f * 9;
        const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
        double R[9], t[3];
        for (int a = 0; a < 9; ++a) R[a] = h[a];
        for (int a = 0; a < 3; ++a) t[a] = h[9 + a];
        const Corr* c = corr + (long long)f * N;
        const int cnt = counts[f];
        for (int n = lane; n < cnt; n += 64) cntin += reproj_inlier(c[n], R, t, fx, fy, cx, cy, thr2) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cntin += __shfl_xor(cntin, o);
    if (lane == 0) inliers[(long long)f * iters + it] = h[12] != 0.0 ? cntin : -1;
}

__device__ __forceinline__ void rodrigues(const double* w, double* R) {
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (t2 > DBL_EPSILON) {
        const double t = sqrt(t2), x = w[0] / t, y = w[1] / t, z = w[2] / t, c = cos(t), s = sin(t), oc = 1 - c;
        R[0] = c + x * x * oc; R[1] = x * y * oc - z * s; R[2] = y * s + x * z * oc;
        R[3] = z * s + x * y * oc; R[4] = c + y * y * oc; R[5] = -x * s + y * z * oc;
        R[6] = -y * s + x * z * oc; R[7] = x * s + y * z * oc; R[8] = c + z * z * oc;
    } else {
        R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
    }
}

// one workgroup per frame: best hypothesis, Gauss-Newton refinement on its inliers, acceptance test
__global__ __launch_bounds__(256) void pnp_select_refine_kernel(const Corr* __restrict__ corr, const int* __restrict__ counts,
                                                                const double* __restrict__ Kmat, const double* __restrict__ hyp,
                                                                const int* __restrict__ inliers, int N, int iters, double thr2,
                                                                int refine_rounds, int refine_iters, double* __restrict__ P_out,
                                                                double* __restrict__ outlier_ratio, int* __restrict__ n_inliers,
                                                                int* __restrict__ best_out, unsigned char* __restrict__ mask) {
    __shared__ double s_red[256];
    __shared__ int s_i[256], s_j[256];
    __shared__ double s_R[9], s_t[3], s_sum[28];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int cnt = counts[f];
    const Corr* c = corr + (long long)f * N;
    const double* K = Kmat + (long long)f * 9;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    unsigned char* m = mask + (long long)f * N;
    // --- argmax inliers, ties -> lowest sample id
    int bi = 0x7fffffff, bn = -1;
    for (int it = tid; it < iters; it += 256) {
        const int v = inliers[(long long)f * iters + it];
        if (v > bn || (v == bn && it < bi)) { bn = v; bi = it; }
    }
    s_i[tid] = bn; s_j[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { if (s_i[tid + o] > s_i[tid] || (s_i[tid + o] == s_i[tid] && s_j[tid + o] < s_j[tid])) { s_i[tid] = s_i[tid + o]; s_j[tid] = s_j[tid + o]; } }
        __syncthreads();
    }
    bn = s_i[0]; bi = s_j[0];
    double* Pf = P_out + (long long)f * 16;
    if (bn < 6 || cnt < 6) {   // no model: identity, outlier ratio 1 (registration_pnp.py:138-146)
        if (tid == 0) {
            for (int a = 0; a < 16; ++a) Pf[a] = (a % 5 == 0) ? 1.0 : 0.0;
            outlier_ratio[f] = 1.0; n_inliers[f] = 0; best_out[f] = -1;
        }
        return;
    }
    if (tid == 0) {
        const double* h = hyp + ((long long)f * iters + bi) * HYP;
        for (int a = 0; a < 9; ++a) s_R[a] = h[a];
        for (int a = 0; a < 3; ++a) s_t[a] = h[9 + a];
    }
    __syncthreads();
    // --- locally optimised RANSAC: rounds of {inliers of the current model -> Gauss-Newton on the 6-DoF reprojection
    //     error over those inliers}; a round is kept only if it does not lose inliers, otherwise the loop stops.
    double Rc[9], tc[3];                 // current accepted model (identical in every thread)
    for (int a = 0; a < 9; ++a) Rc[a] = s_R[a];
    for (int a = 0; a < 3; ++a) tc[a] = s_t[a];
    int nin = bn;
    for (int round = 0; round < refine_rounds; ++round) {
        for (int n = tid; n < cnt; n += 256) m[n] = reproj_inlier(c[n], Rc, tc, fx, fy, cx, cy, thr2) ? 1 : 0;
        if (tid == 0) { for (int a = 0; a < 9; ++a) s_R[a] = Rc[a]; for (int a = 0; a < 3; ++a) s_t[a] = tc[a]; }
        __syncthreads();
        for (int gi = 0; gi < refine_iters; ++gi) {
            double R[9], t[3];
            for (int a = 0; a < 9; ++a) R[a] = s_R[a];
            for (int a = 0; a < 3; ++a) t[a] = s_t[a];
            double acc[28];
            for (int a = 0; a < 28; ++a) acc[a] = 0.0;
            for (int n = tid; n < cnt; n += 256) {
                if (!m[n]) continue;
                const Corr q = c[n];
                const double X = q.x, Y = q.y, Z = q.z;
                const double q0 = R[0] * X + R[1] * Y + R[2] * Z, q1 = R[3] * X + R[4] * Y + R[5] * Z, q2 = R[6] * X + R[7] * Y + R[8] * Z;
                const double p0 = q0 + t[0], p1 = q1 + t[1], p2 = q2 + t[2];
                const double iz = 1.0 / p2;
                const double ru = fx * p0 * iz + cx - (double)q.u, rv = fy * p1 * iz + cy - (double)q.v;
                // dp/d(dw) = -[q]x (left perturbation R <- exp(dw) R), dp/dt = I
                const double dp[3][6] = {{0, q2, -q1, 1, 0, 0}, {-q2, 0, q0, 0, 1, 0}, {q1, -q0, 0, 0, 0, 1}};
                double Ju[6], Jv[6];
                for (int a = 0; a < 6; ++a) {
                    Ju[a] = fx * iz * dp[0][a] - fx * p0 * iz * iz * dp[2][a];
                    Jv[a] = fy * iz * dp[1][a] - fy * p1 * iz * iz * dp[2][a];
                }
                int k = 0;
                for (int a = 0; a < 6; ++a) for (int b2 = 0; b2 <= a; ++b2) acc[k++] += Ju[a] * Ju[b2] + Jv[a] * Jv[b2];
                for (int a = 0; a < 6; ++a) acc[21 + a] += Ju[a] * ru + Jv[a] * rv;
                acc[27] += ru * ru + rv * rv;
            }
            for (int a = 0; a < 28; ++a) {          // fixed-order tree reduction per component
                s_red[tid] = acc[a];
                __syncthreads();
                for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_red[tid] += s_red[tid + o]; __syncthreads(); }
                if (tid == 0) s_sum[a] = s_red[0];
                __syncthreads();
            }
            if (tid == 0) {
                double L[21], z[6], d[6];
                bool ok = true;
                for (int i = 0; i < 6 && ok; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double sv = s_sum[i * (i + 1) / 2 + j] + (i == j ? 1e-9 * (1.0 + s_sum[i * (i + 1) / 2 + i]) : 0.0);
                        for (int q = 0; q < j; ++q) sv -= L[i * (i + 1) / 2 + q] * L[j * (j + 1) / 2 + q];
                        if (i == j) { if (!(sv > 0.0)) { ok = false; break; } L[i * (i + 1) / 2 + i] = sqrt(sv); }
                        else L[i * (i + 1) / 2 + j] = sv / L[j * (j + 1) / 2 + j];
                    }
                if (ok) {
                    for (int i = 0; i < 6; ++i) { double sv = -s_sum[21 + i]; for (int q = 0; q < i; ++q) sv -= L[i * (i + 1) / 2 + q] * z[q]; z[i] = sv / L[i * (i + 1) / 2 + i]; }
                    for (int i = 5; i >= 0; --i) { double sv = z[i]; for (int q = i + 1; q < 6; ++q) sv -= L[q * (q + 1) / 2 + i] * d[q]; d[i] = sv / L[i * (i + 1) / 2 + i]; }
                    double dR[9], Rn[9];
                    rodrigues(d, dR);
                    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rn[r * 3 + cc] = dR[r * 3] * s_R[cc] + dR[r * 3 + 1] * s_R[3 + cc] + dR[r * 3 + 2] * s_R[6 + cc];
                    bool fin = true;
                    for (int a = 0; a < 9; ++a) fin &= isfinite(Rn[a]);
                    for (int a = 0; a < 3; ++a) fin &= isfinite(d[3 + a]);
                    if (fin) { for (int a = 0; a < 9; ++a) s_R[a] = Rn[a]; for (int a = 0; a < 3; ++a) s_t[a] += d[3 + a]; }
                }
            }
            __syncthreads();
        }
        double R1[9], t1[3];
        for (int a = 0; a < 9; ++a) R1[a] = s_R[a];
        for (int a = 0; a < 3; ++a) t1[a] = s_t[a];
        int c1 = 0;
        for (int n = tid; n < cnt; n += 256) c1 += reproj_inlier(c[n], R1, t1, fx, fy, cx, cy, thr2) ? 1 : 0;
        s_i[tid] = c1;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_i[tid] += s_i[tid + o]; __syncthreads(); }
        const int c1tot = s_i[0];
        __syncthreads();
        if (c1tot < nin) break;                       // (uniform) refinement lost inliers: keep the current model
        for (int a = 0; a < 9; ++a) Rc[a] = R1[a];
        for (int a = 0; a < 3; ++a) tc[a] = t1[a];
        nin = c1tot;
    }
    if (tid == 0) {
        const double tn = sqrt(tc[0] * tc[0] + tc[1] * tc[1] + tc[2] * tc[2]);
        for (int a = 0; a < 16; ++a) Pf[a] = (a % 5 == 0) ? 1.0 : 0.0;
        if (tn < 14.14) {
            for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) Pf[r * 4 + cc] = Rc[r * 3 + cc]; Pf[r * 4 + 3] = tc[r]; }
            outlier_ratio[f] = 1.0 - (double)nin / (double)cnt;
        } else {
            outlier_ratio[f] = 1.0;
        }
        n_inliers[f] = nin; best_out[f] = bi;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// EPnP variant (the estimator the reference actually asks for: cv2.solvePnPRansac(flags=SOLVEPNP_EPNP)):
//   hypotheses: EPnP on a minimal sample of 5 correspondences (OpenCV's model_points for EPNP; 4 when the frame has only 4 --
//               OpenCV switches to P3P there -- with a Gauss-Newton polish of the pose, EPnP alone being ambiguous on 4 points);
//   scoring:    squared reprojection error <= threshold^2 (no depth test, as OpenCV's computeError);
//   result:     the model with the most inliers, then ONE EPnP re-fit on all its inliers; the inlier count reported is the
//               winning model's; success iff it has at least the minimal-sample size; |t| < 14.14 as the reference.
struct ThreadPoints {           // <= 5 correspondences of one sample, one thread
    double X[5][3], uv[5][2];
    int m;
    __device__ int count() const { return m; }
    template <class F> __device__ void for_each(F f) const { for (int i = 0; i < m; ++i) f(X[i][0], X[i][1], X[i][2], uv[i][0], uv[i][1], i == 0); }
    __device__ double reduce(double v) const { return v; }
};
struct WavePoints {             // the masked correspondences of a frame, lanes of one wavefront stride over them
    const Corr* c;
    const unsigned char* mask;
    int cnt, first;             // first = index of the first masked correspondence
    __device__ int count() const { return cnt; }
    template <class F> __device__ void for_each(F f) const {
        for (int n = threadIdx.x & 63; n < cnt; n += 64)
            if (mask[n]) { const Corr q = c[n]; f((double)q.x, (double)q.y, (double)q.z, (double)q.u, (double)q.v, n == first); }
    }
    __device__ double reduce(double v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    }
};

// Gauss-Newton polish of a pose on a point set (left-multiplicative rotation update), `iters` steps
template <class PS>
__device__ inline void gn_polish(PS& ps, const epnp::Cam4& k, double* R, double* t, int iters) {
    for (int it = 0; it < iters; ++it) {
        double acc[27];
        for (int a = 0; a < 27; ++a) acc[a] = 0.0;
        ps.for_each([&](double X, double Y, double Z, double u, double v, bool) {
            const double q0 = R[0] * X + R[1] * Y + R[2] * Z, q1 = R[3] * X + R[4] * Y + R[5] * Z, q2 = R[6] * X + R[7] * Y + R[8] * Z;
            const double p0 = q0 + t[0], p1 = q1 + t[1], p2 = q2 + t[2];
            const double iz = 1.0 / p2;
            const double ru = k.fu * p0 * iz + k.uc - u, rv = k.fv * p1 * iz + k.vc - v;
            const double dp[3][6] = {{0, q2, -q1, 1, 0, 0}, {-q2, 0, q0, 0, 1, 0}, {q1, -q0, 0, 0, 0, 1}};
            double Ju[6], Jv[6];
            for (int a = 0; a < 6; ++a) {
                Ju[a] = k.fu * iz * dp[0][a] - k.fu * p0 * iz * iz * dp[2][a];
                Jv[a] = k.fv * iz * dp[1][a] - k.fv * p1 * iz * iz * dp[2][a];
            }
            int idx = 0;
            for (int a = 0; a < 6; ++a) for (int b2 = 0; b2 <= a; ++b2) acc[idx++] += Ju[a] * Ju[b2] + Jv[a] * Jv[b2];
            for (int a = 0; a < 6; ++a) acc[21 + a] += Ju[a] * ru + Jv[a] * rv;
        });
        for (int a = 0; a < 27; ++a) acc[a] = ps.reduce(acc[a]);
        double L[21], z[6], d[6];
        bool ok = true;
        for (int i = 0; i < 6 && ok; ++i)
            for (int j = 0; j <= i; ++j) {
                double sv = acc[i * (i + 1) / 2 + j] + (i == j ? 1e-9 * (1.0 + acc[i * (i + 1) / 2 + i]) : 0.0);
                for (int q = 0; q < j; ++q) sv -= L[i * (i + 1) / 2 + q] * L[j * (j + 1) / 2 + q];
                if (i == j) { if (!(sv > 0.0)) { ok = false; break; } L[i * (i + 1) / 2 + i] = sqrt(sv); }
                else L[i * (i + 1) / 2 + j] = sv / L[j * (j + 1) / 2 + j];
            }
        if (!ok) return;
        for (int i = 0; i < 6; ++i) { double sv = -acc[21 + i]; for (int q = 0; q < i; ++q) sv -= L[i * (i + 1) / 2 + q] * z[q]; z[i] = sv / L[i * (i + 1) / 2 + i]; }
        for (int i = 5; i >= 0; --i) { double sv = z[i]; for (int q = i + 1; q < 6; ++q) sv -= L[q * (q + 1) / 2 + i] * d[q]; d[i] = sv / L[i * (i + 1) / 2 + i]; }
        double dR[9], Rn[9];
        rodrigues(d, dR);
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rn[r * 3 + cc] = dR[r * 3] * R[cc] + dR[r * 3 + 1] * R[3 + cc] + dR[r * 3 + 2] * R[6 + cc];
        bool fin = true;
        for (int a = 0; a < 9; ++a) fin = fin && isfinite(Rn[a]);
        for (int a = 0; a < 3; ++a) fin = fin && isfinite(d[3 + a]);
        if (!fin) return;
        for (int a = 0; a < 9; ++a) R[a] = Rn[a];
        for (int a = 0; a < 3; ++a) t[a] += d[3 + a];
    }
}

__global__ __launch_bounds__(64) void pnp_hypotheses_epnp_kernel(const Corr* __restrict__ corr, const int* __restrict__ counts,
                                                                 const double* __restrict__ Kmat, const int* __restrict__ samples, int N,
                                                                 int iters, double* __restrict__ hyp) {
    const int f = blockIdx.y;
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= iters) return;
    double* h = hyp + ((long long)f * iters + it) * HYP;
    h[12] = 0.0;
    const int cnt = counts[f];
    if (cnt < 4) return;
    const Corr* c = corr + (long long)f * N;
    const double* K = Kmat + (long long)f * 9;
    const epnp::Cam4 k{K[0], K[4], K[2], K[5]};
    ThreadPoints ps;
    ps.m = cnt >= 5 ? 5 : 4;
    int ids[5];
    for (int j = 0; j < ps.m; ++j) {
        int s = samples[((long long)f * iters + it) * 6 + j] % cnt;
        if (s < 0) s += cnt;
        for (int q = 0; q < j; ++q) if (ids[q] == s) return;           // a sample with a repeated correspondence is skipped
        ids[j] = s;
        const Corr q = c[s];
        ps.X[j][0] = q.x; ps.X[j][1] = q.y; ps.X[j][2] = q.z; ps.uv[j][0] = q.u; ps.uv[j][1] = q.v;
    }
    epnp::Pose pose;
    if (ps.m == 4) epnp::solve(ps, k, pose, [&](double* R, double* t) { gn_polish(ps, k, R, t, 15); });
    else epnp::solve(ps, k, pose);
    if (!pose.ok) return;
    for (int a = 0; a < 9; ++a) { if (!isfinite(pose.R[a])) return; h[a] = pose.R[a]; }
    for (int a = 0; a < 3; ++a) { if (!isfinite(pose.t[a])) return; h[9 + a] = pose.t[a]; }
    h[12] = 1.0;
}

__device__ __forceinline__ bool reproj_inlier_cv(const Corr& q, const double* R, const double* t, double fx, double fy, double cx,
                                                 double cy, double thr2) {
    const double X = q.x, Y = q.y, Z = q.z;
    const double p0 = R[0] * X + R[1] * Y + R[2] * Z + t[0], p1 = R[3] * X + R[4] * Y + R[5] * Z + t[1],
                 p2 = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double du = fx * p0 / p2 + cx - (double)q.u, dv = fy * p1 / p2 + cy - (double)q.v;
    return du * du + dv * dv <= thr2;            // false for NaN
}

__global__ __launch_bounds__(256) void pnp_score_cv_kernel(const Corr* __restrict__ corr, const int* __restrict__ counts,
                                                           const double* __restrict__ Kmat, const double* __restrict__ hyp, int N, int iters,
                                                           double thr2, int* __restrict__ inliers) {
    const int f = blockIdx.y;
    const int it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= iters) return;
    const int lane = threadIdx.x & 63;
    const double* h = hyp + ((long long)f * iters + it) * HYP;
    int cntin = 0;
    if (h[12] != 0.0) {
        const double* K = Kmat + (long long)f * 9;
        double R[9], t[3];
        for (int a = 0; a < 9; ++a) R[a] = h[a];
        for (int a = 0; a < 3; ++a) t[a] = h[9 + a];
        const Corr* c = corr + (long long)f * N;
        const int cnt = counts[f];
        for (int n = lane; n < cnt; n += 64) cntin += reproj_inlier_cv(c[n], R, t, K[0], K[4], K[2], K[5], thr2) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cntin += __shfl_xor(cntin, o);
    if (lane == 0) inliers[(long long)f * iters + it] = h[12] != 0.0 ? cntin : -1;
}

// one workgroup per frame: best hypothesis, its inlier mask, EPnP re-fit on the inliers (wave 0), acceptance
__global__ __launch_bounds__(256) void pnp_select_refit_kernel(const Corr* __restrict__ corr, const int* __restrict__ counts,
                                                               const double* __restrict__ Kmat, const double* __restrict__ hyp,
                                                               const int* __restrict__ inliers, int N, int iters, double thr2,
                                                               double* __restrict__ P_out, double* __restrict__ outlier_ratio,
                                                               int* __restrict__ n_inliers, int* __restrict__ best_out,
                                                               unsigned char* __restrict__ mask) {
    __shared__ int s_i[256], s_j[256];
    __shared__ int s_first;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int cnt = counts[f];
    const Corr* c = corr + (long long)f * N;
    const double* K = Kmat + (long long)f * 9;
    unsigned char* m = mask + (long long)f * N;
    int bi = 0x7fffffff, bn = -1;
    for (int it = tid; it < iters; it += 256) {
        const int v = inliers[(long long)f * iters + it];
        if (v > bn || (v == bn && it < bi)) { bn = v; bi = it; }
    }
    s_i[tid] = bn; s_j[tid] = bi;
    if (tid == 0) s_first = 0x7fffffff;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { if (s_i[tid + o] > s_i[tid] || (s_i[tid + o] == s_i[tid] && s_j[tid + o] < s_j[tid])) { s_i[tid] = s_i[tid + o]; s_j[tid] = s_j[tid + o]; } }
        __syncthreads();
    }
    bn = s_i[0]; bi = s_j[0];
    double* Pf = P_out + (long long)f * 16;
    const int msize = cnt >= 5 ? 5 : 4;
    if (cnt < 4 || bn < msize) {
        if (tid == 0) {
            for (int a = 0; a < 16; ++a) Pf[a] = (a % 5 == 0) ? 1.0 : 0.0;
            outlier_ratio[f] = 1.0; n_inliers[f] = 0; best_out[f] = -1;
        }
        return;
    }
    const double* h = hyp + ((long long)f * iters + bi) * HYP;
    double R[9], t[3];
    for (int a = 0; a < 9; ++a) R[a] = h[a];
    for (int a = 0; a < 3; ++a) t[a] = h[9 + a];
    for (int n = tid; n < cnt; n += 256) {
        const bool in = reproj_inlier_cv(c[n], R, t, K[0], K[4], K[2], K[5], thr2);
        m[n] = in ? 1 : 0;
        if (in) atomicMin(&s_first, n);
    }
    __syncthreads();
    if (tid < 64) {              // wave 0: EPnP over all inliers of the winning model
        WavePoints ps{c, m, cnt, s_first};
        const epnp::Cam4 k{K[0], K[4], K[2], K[5]};
        epnp::Pose pose;
        epnp::solve(ps, k, pose);
        if (tid == 0) {
            const double* Ro = pose.ok ? pose.R : R;
            const double* to = pose.ok ? pose.t : t;
            const double tn = sqrt(to[0] * to[0] + to[1] * to[1] + to[2] * to[2]);
            for (int a = 0; a < 16; ++a) Pf[a] = (a % 5 == 0) ? 1.0 : 0.0;
            if (tn < 14.14) {
                for (int r = 0; r < 3; ++r) { for (int cc = 0; cc < 3; ++cc) Pf[r * 4 + cc] = Ro[r * 3 + cc]; Pf[r * 4 + 3] = to[r]; }
                outlier_ratio[f] = 1.0 - (double)bn / (double)cnt;
            } else {
                outlier_ratio[f] = 1.0;
            }
            n_inliers[f] = bn; best_out[f] = bi;
        }
    }
}

}  // namespace

extern "C" long long di2p_pnp_workspace_bytes(int F, int N, int iters) {
    return 256 + (long long)F * 4 + 256 + (long long)F * N * 32 + (long long)F * iters * (HYP * 8 + 4) + (long long)F * N + 1024;
}

// The correspondence list alone -- exactly what solve_PnP hands to cv2.solvePnPRansac (evaluation/registration_pnp.py:97-110,125-127):
// points[n_corr[f]][3] = pc[:, coarse == 1].T and pixels[n_corr[f]][2] = (fine - floor(fine / W) * W, floor(fine / W)) in point order.
// corr: f32 [F][N][8] records {x, y, z, u, v, 0, 0, 0}, the first n_corr[f] of each frame valid.
extern "C" int di2p_pnp_pack(const float* pc, const int32_t* coarse, const int32_t* fine, const float* pixels, int W_fine, int F, int N,
                             float* corr, int32_t* n_corr, void* stream) {
    DI2P_CHECK_ARG(F >= 0 && N >= 1 && W_fine >= 1, "bad size");
    if (F == 0) return 0;
    DI2P_CHECK_ARG(pc && coarse && (fine || pixels) && corr && n_corr, "null pointer");
    hipLaunchKernelGGL(pnp_pack_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, pc, coarse, fine, pixels, N, W_fine, (Corr*)corr, n_corr);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_pnp_ransac(const float* pc, const int32_t* coarse, const int32_t* fine, const float* pixels,
                               const double* K_scaled, int W_fine,
                               const int32_t* samples, int iters, double reproj_err, int refine_rounds, int refine_iters, int F, int N,
                               double* P_out,
                               double* outlier_ratio, int32_t* n_inliers, int32_t* n_corr, int32_t* best, void* workspace,
                               void* stream) {
    DI2P_CHECK_ARG(F >= 0 && N >= 1 && iters >= 1 && W_fine >= 1 && reproj_err > 0 && refine_iters >= 0 && refine_rounds >= 0, "bad size");
    if (F == 0) return 0;
    DI2P_CHECK_ARG(pc && coarse && (fine || pixels) && K_scaled && samples && P_out && outlier_ratio && n_inliers && n_corr && best && workspace, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    Corr* corr = (Corr*)w;                                      w += (((size_t)F * N * sizeof(Corr)) + 255) & ~(size_t)255;
    double* hyp = (double*)w;                                   w += (((size_t)F * iters * HYP * 8) + 255) & ~(size_t)255;
    int* inl = (int*)w;                                         w += (((size_t)F * iters * 4) + 255) & ~(size_t)255;
    unsigned char* mask = (unsigned char*)w;
    hipLaunchKernelGGL(pnp_pack_kernel, dim3(F), dim3(256), 0, st, pc, coarse, fine, pixels, N, W_fine, corr, n_corr);
    hipLaunchKernelGGL(pnp_hypotheses_kernel, dim3(di2p_cdiv(iters, 64), F), dim3(64), 0, st, corr, n_corr, K_scaled, samples, N, iters, hyp);
    hipLaunchKernelGGL(pnp_score_kernel, dim3(di2p_cdiv(iters, 4), F), dim3(256), 0, st, corr, n_corr, K_scaled, hyp, N, iters,
                       reproj_err * reproj_err, inl);
    hipLaunchKernelGGL(pnp_select_refine_kernel, dim3(F), dim3(256), 0, st, corr, n_corr, K_scaled, hyp, inl, N, iters,
                       reproj_err * reproj_err, refine_rounds, refine_iters, P_out, outlier_ratio, n_inliers, best, mask);
    DI2P_RETURN_LAUNCH();
}

extern "C" int di2p_pnp_ransac_epnp(const float* pc, const int32_t* coarse, const int32_t* fine, const float* pixels,
                                    const double* K_scaled, int W_fine, const int32_t* samples, int iters, double reproj_err, int F,
                                    int N, double* P_out, double* outlier_ratio, int32_t* n_inliers, int32_t* n_corr, int32_t* best,
                                    void* workspace, void* stream) {
    DI2P_CHECK_ARG(F >= 0 && N >= 1 && iters >= 1 && W_fine >= 1 && reproj_err > 0, "bad size");
    if (F == 0) return 0;
    DI2P_CHECK_ARG(pc && coarse && (fine || pixels) && K_scaled && samples && P_out && outlier_ratio && n_inliers && n_corr && best && workspace, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    Corr* corr = (Corr*)w;                                      w += (((size_t)F * N * sizeof(Corr)) + 255) & ~(size_t)255;
    double* hyp = (double*)w;                                   w += (((size_t)F * iters * HYP * 8) + 255) & ~(size_t)255;
    int* inl = (int*)w;                                         w += (((size_t)F * iters * 4) + 255) & ~(size_t)255;
    unsigned char* mask = (unsigned char*)w;
    hipLaunchKernelGGL(pnp_pack_kernel, dim3(F), dim3(256), 0, st, pc, coarse, fine, pixels, N, W_fine, corr, n_corr);
    hipLaunchKernelGGL(pnp_hypotheses_epnp_kernel, dim3(di2p_cdiv(iters, 64), F), dim3(64), 0, st, corr, n_corr, K_scaled, samples, N, iters, hyp);
    hipLaunchKernelGGL(pnp_score_cv_kernel, dim3(di2p_cdiv(iters, 4), F), dim3(256), 0, st, corr, n_corr, K_scaled, hyp, N, iters,
                       reproj_err * reproj_err, inl);
    hipLaunchKernelGGL(pnp_select_refit_kernel, dim3(F), dim3(256), 0, st, corr, n_corr, K_scaled, hyp, inl, N, iters,
                       reproj_err * reproj_err, P_out, outlier_ratio, n_inliers, best, mask);
    DI2P_RETURN_LAUNCH();
}
