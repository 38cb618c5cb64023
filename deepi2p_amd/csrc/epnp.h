// EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) for gfx950, arranged as OpenCV's calib3d epnp.cpp and as the numpy
// restatement oracle/epnp_np.py (statement in its header): control points from the principal directions, barycentric
// coordinates, the 12 x 12 matrix M^T M, its four smallest eigenvectors, the 6 x 10 distance system, three linearisations
// refined by five Gauss-Newton steps, absolute orientation, smallest mean reprojection error.
//
// The routine is written once over a "point set" policy: every pass over the points is a for_each + a reduction, so the
// same code runs with ONE THREAD per minimal sample (RANSAC hypotheses: 4-5 points in registers, reduce = identity) and with
// ONE WAVEFRONT per frame (the re-fit on all inliers: lanes stride over the inliers, reduce = wave butterfly sum).
#pragma once
#include "common.h"

#include <float.h>
#include <math.h>

namespace epnp {

struct Cam4 { double fu, fv, uc, vc; };
struct Pose { double R[9], t[3], err; bool ok; };

// symmetric N x N eigen-decomposition by cyclic Jacobi; A is destroyed (diagonal = eigenvalues), V columns = eigenvectors
template <int N>
__device__ inline void jacobi_eig(double* A, double* V) {
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < N; ++i) { diag += A[i * N + i] * A[i * N + i]; for (int j = i + 1; j < N; ++j) off += A[i * N + j] * A[i * N + j]; }
        if (!(off > 1e-30 * diag) || !isfinite(off)) break;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p * N + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq; A[k * N + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk; A[q * N + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq; V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
}

// least squares min |A x - b| for a 6 x K system by Householder QR (the normal equations square the condition number, and
// the Gauss-Newton systems of the beta refinement are badly scaled: one dominant beta, three small ones); a column whose
// diagonal entry of R vanishes (a beta that does not matter) gets x = 0
template <int K>
__device__ inline bool lstsq6(const double (*Ain)[K], const double* bin, double* x) {
    double A[6][K], b[6];
    for (int r = 0; r < 6; ++r) { for (int c = 0; c < K; ++c) A[r][c] = Ain[r][c]; b[r] = bin[r]; }
    double rmax = 0.0;
    for (int c = 0; c < K; ++c) {
        double nrm = 0.0;
        for (int r = c; r < 6; ++r) nrm += A[r][c] * A[r][c];
        nrm = sqrt(nrm);
        if (nrm > 0.0) {
            const double alpha = A[c][c] > 0.0 ? -nrm : nrm;
            double v[6];
            for (int r = 0; r < 6; ++r) v[r] = r < c ? 0.0 : A[r][c];
            v[c] -= alpha;
            double vn = 0.0;
            for (int r = c; r < 6; ++r) vn += v[r] * v[r];
            if (vn > 0.0) {
                for (int j = c; j < K; ++j) {
                    double d = 0.0;
                    for (int r = c; r < 6; ++r) d += v[r] * A[r][j];
                    const double f = 2.0 * d / vn;
                    for (int r = c; r < 6; ++r) A[r][j] -= f * v[r];
                }
                double d = 0.0;
                for (int r = c; r < 6; ++r) d += v[r] * b[r];
                const double f = 2.0 * d / vn;
                for (int r = c; r < 6; ++r) b[r] -= f * v[r];
            }
        }
        rmax = fmax(rmax, fabs(A[c][c]));
    }
    for (int c = K - 1; c >= 0; --c) {
        double v = b[c];
        for (int j = c + 1; j < K; ++j) v -= A[c][j] * x[j];
        x[c] = fabs(A[c][c]) > 1e-14 * rmax ? v / A[c][c] : 0.0;
    }
    for (int i = 0; i < K; ++i) if (!isfinite(x[i])) return false;
    return true;
}

__device__ inline bool inv3x3(const double* M, double* inv) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(fabs(det) > 1e-300) || !isfinite(det)) return false;
    const double id = 1.0 / det;
    inv[0] = (e * i - f * h) * id; inv[1] = (c * h - b * i) * id; inv[2] = (b * f - c * e) * id;
    inv[3] = (f * g - d * i) * id; inv[4] = (a * i - c * g) * id; inv[5] = (c * d - a * f) * id;
    inv[6] = (d * h - e * g) * id; inv[7] = (b * g - a * h) * id; inv[8] = (a * e - b * d) * id;
    return true;
}

// PS: int count(); template<F> void for_each(F f) with f(double X, Y, Z, u, v, bool first); double reduce(double) (sum over
// the set, same value returned to every participant).
struct NoPolish { __device__ void operator()(double*, double*) const {} };

// polish(R, t): applied to each of the three candidate poses before their reprojection errors are compared (used for 4-point
// sets, where EPnP's linearisations are ambiguous and a Gauss-Newton descent from each candidate separates them)
template <class PS, class Polish = NoPolish>
__device__ inline void solve(PS& ps, const Cam4& k, Pose& out, Polish polish = Polish()) {
    out.ok = false;
    out.err = DBL_MAX;
    // ---- pass 1: centroid and scatter -> control points (centroid + principal directions scaled by the standard deviations)
    double s1[3] = {0, 0, 0}, n_loc = 0.0;
    ps.for_each([&](double X, double Y, double Z, double, double, bool) { s1[0] += X; s1[1] += Y; s1[2] += Z; n_loc += 1.0; });
    const double n = ps.reduce(n_loc);
    if (!(n >= 4.0)) return;
    double c0[3];
    for (int a = 0; a < 3; ++a) c0[a] = ps.reduce(s1[a]) / n;
    double sc[6] = {0, 0, 0, 0, 0, 0};
    ps.for_each([&](double X, double Y, double Z, double, double, bool) {
        const double dx = X - c0[0], dy = Y - c0[1], dz = Z - c0[2];
        sc[0] += dx * dx; sc[1] += dx * dy; sc[2] += dx * dz; sc[3] += dy * dy; sc[4] += dy * dz; sc[5] += dz * dz;
    });
    for (int a = 0; a < 6; ++a) sc[a] = ps.reduce(sc[a]);
    double C3[9] = {sc[0], sc[1], sc[2], sc[1], sc[3], sc[4], sc[2], sc[4], sc[5]}, U3[9];
    jacobi_eig<3>(C3, U3);
    double cw[4][3];
    for (int a = 0; a < 3; ++a) cw[0][a] = c0[a];
    {   // directions in descending eigenvalue order, as OpenCV's SVD delivers them
        int ord[3] = {0, 1, 2};
        for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) if (C3[ord[j] * 3 + ord[j]] > C3[ord[i] * 3 + ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
        for (int i = 0; i < 3; ++i) {
            const double kk = sqrt(fmax(C3[ord[i] * 3 + ord[i]], 0.0) / n);
            // sign convention of a principal direction: its largest component is positive (with noisy correspondences EPnP is
            // not invariant to mirroring a control point, and an eigen-solver's sign is arbitrary)
            int big = 0;
            for (int a = 1; a < 3; ++a) if (fabs(U3[a * 3 + ord[i]]) > fabs(U3[big * 3 + ord[i]])) big = a;
            const double sg = U3[big * 3 + ord[i]] < 0.0 ? -1.0 : 1.0;
            for (int a = 0; a < 3; ++a) cw[i + 1][a] = c0[a] + sg * kk * U3[a * 3 + ord[i]];
        }
    }
    double CC[9], CCi[9];            // columns = c_j - c_0
    for (int a = 0; a < 3; ++a) for (int j = 0; j < 3; ++j) CC[a * 3 + j] = cw[j + 1][a] - cw[0][a];
    if (!inv3x3(CC, CCi)) return;
    // ---- pass 2: barycentric coordinates -> M^T M (upper triangle), sum a, sum a x^T, sum x
    double mtm[78], sa[4] = {0, 0, 0, 0}, saw[12], a_first[4] = {0, 0, 0, 0};
    for (int i = 0; i < 78; ++i) mtm[i] = 0.0;
    for (int i = 0; i < 12; ++i) saw[i] = 0.0;
    auto alphas_of = [&](double X, double Y, double Z, double* al) {
        const double dx = X - cw[0][0], dy = Y - cw[0][1], dz = Z - cw[0][2];
        al[1] = CCi[0] * dx + CCi[1] * dy + CCi[2] * dz;
        al[2] = CCi[3] * dx + CCi[4] * dy + CCi[5] * dz;
        al[3] = CCi[6] * dx + CCi[7] * dy + CCi[8] * dz;
        al[0] = 1.0 - al[1] - al[2] - al[3];
    };
    ps.for_each([&](double X, double Y, double Z, double u, double v, bool first) {
        double al[4];
        alphas_of(X, Y, Z, al);
        double r0[12], r1[12];
        for (int j = 0; j < 4; ++j) {
            r0[3 * j] = al[j] * k.fu; r0[3 * j + 1] = 0.0; r0[3 * j + 2] = al[j] * (k.uc - u);
            r1[3 * j] = 0.0; r1[3 * j + 1] = al[j] * k.fv; r1[3 * j + 2] = al[j] * (k.vc - v);
        }
        int idx = 0;
        for (int a = 0; a < 12; ++a) for (int b = a; b < 12; ++b) mtm[idx++] += r0[a] * r0[b] + r1[a] * r1[b];
        const double pw[3] = {X, Y, Z};
        for (int j = 0; j < 4; ++j) { sa[j] += al[j]; for (int a = 0; a < 3; ++a) saw[j * 3 + a] += al[j] * pw[a]; }
        if (first) for (int j = 0; j < 4; ++j) a_first[j] = al[j];
    });
    for (int i = 0; i < 78; ++i) mtm[i] = ps.reduce(mtm[i]);
    for (int j = 0; j < 4; ++j) { sa[j] = ps.reduce(sa[j]); a_first[j] = ps.reduce(a_first[j]); }
    for (int i = 0; i < 12; ++i) saw[i] = ps.reduce(saw[i]);
    double A12[144], V12[144];
    {
        int idx = 0;
        for (int a = 0; a < 12; ++a) for (int b = a; b < 12; ++b) { A12[a * 12 + b] = mtm[idx]; A12[b * 12 + a] = mtm[idx]; ++idx; }
    }
    jacobi_eig<12>(A12, V12);
    int ord[4];                       // the four smallest eigenvalues, ascending
    {
        bool used[12];
        for (int i = 0; i < 12; ++i) used[i] = false;
        for (int s = 0; s < 4; ++s) {
            int b = -1;
            for (int i = 0; i < 12; ++i) if (!used[i] && (b < 0 || A12[i * 12 + i] < A12[b * 12 + b])) b = i;
            ord[s] = b; used[b] = true;
        }
    }
    double vv[4][12];
    for (int s = 0; s < 4; ++s) for (int a = 0; a < 12; ++a) vv[s][a] = V12[a * 12 + ord[s]];
    // ---- L (6 x 10) and rho
    const int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
    double L[6][10], rho[6];
    for (int p = 0; p < 6; ++p) {
        double d[4][3];
        for (int i = 0; i < 4; ++i) for (int a = 0; a < 3; ++a) d[i][a] = vv[i][3 * PA[p] + a] - vv[i][3 * PB[p] + a];
        auto dot = [&](int i, int j) { return d[i][0] * d[j][0] + d[i][1] * d[j][1] + d[i][2] * d[j][2]; };
        L[p][0] = dot(0, 0); L[p][1] = 2 * dot(0, 1); L[p][2] = dot(1, 1); L[p][3] = 2 * dot(0, 2); L[p][4] = 2 * dot(1, 2);
        L[p][5] = dot(2, 2); L[p][6] = 2 * dot(0, 3); L[p][7] = 2 * dot(1, 3); L[p][8] = 2 * dot(2, 3); L[p][9] = dot(3, 3);
        double r2 = 0.0;
        for (int a = 0; a < 3; ++a) { const double t = cw[PA[p]][a] - cw[PB[p]][a]; r2 += t * t; }
        rho[p] = r2;
    }
    const double pw0[3] = {c0[0], c0[1], c0[2]};
    // ---- the three linearisations
    Pose cand[3];
    for (int N = 1; N <= 3; ++N) {
        Pose& cd = cand[N - 1];
        cd.ok = false; cd.err = DBL_MAX;
        double be[4] = {0, 0, 0, 0};
        if (N == 1) {
            double A[6][4], x[4];
            for (int r = 0; r < 6; ++r) { A[r][0] = L[r][0]; A[r][1] = L[r][1]; A[r][2] = L[r][3]; A[r][3] = L[r][6]; }
            if (!lstsq6<4>(A, rho, x)) continue;
            const double s = sqrt(fabs(x[0])), sg = x[0] < 0 ? -1.0 : 1.0;
            be[0] = s; be[1] = sg * x[1] / s; be[2] = sg * x[2] / s; be[3] = sg * x[3] / s;
        } else if (N == 2) {
            double A[6][3], x[3];
            for (int r = 0; r < 6; ++r) { A[r][0] = L[r][0]; A[r][1] = L[r][1]; A[r][2] = L[r][2]; }
            if (!lstsq6<3>(A, rho, x)) continue;
            if (x[0] < 0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; } else { be[0] = sqrt(x[0]); be[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) be[0] = -be[0];
        } else {
            double A[6][5], x[5];
            for (int r = 0; r < 6; ++r) for (int c = 0; c < 5; ++c) A[r][c] = L[r][c];
            if (!lstsq6<5>(A, rho, x)) continue;
            if (x[0] < 0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; } else { be[0] = sqrt(x[0]); be[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) be[0] = -be[0];
            be[2] = x[3] / be[0];
        }
        bool fin = true;
        for (int it = 0; it < 5 && fin; ++it) {              // Gauss-Newton on the ten-term distance equations
            double A[6][4], r[6], x[4];
            for (int p = 0; p < 6; ++p) {
                const double* l = L[p];
                A[p][0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
                A[p][1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
                A[p][2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
                A[p][3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
                r[p] = rho[p] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] +
                                 l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
            }
            if (!lstsq6<4>(A, r, x)) { fin = false; break; }
            for (int i = 0; i < 4; ++i) be[i] += x[i];
        }
        for (int i = 0; i < 4; ++i) fin = fin && isfinite(be[i]);
        if (!fin) continue;
        // control points in the camera frame, sign from the first point's depth, absolute orientation from the moments
        double cc[4][3];
        for (int j = 0; j < 4; ++j) for (int a = 0; a < 3; ++a) cc[j][a] = be[0] * vv[0][3 * j + a] + be[1] * vv[1][3 * j + a] + be[2] * vv[2][3 * j + a] + be[3] * vv[3][3 * j + a];
        double z1 = 0.0;
        for (int j = 0; j < 4; ++j) z1 += a_first[j] * cc[j][2];
        if (z1 < 0.0) for (int j = 0; j < 4; ++j) for (int a = 0; a < 3; ++a) cc[j][a] = -cc[j][a];
        double pc0[3] = {0, 0, 0}, ABt[9];
        for (int a = 0; a < 3; ++a) { for (int j = 0; j < 4; ++j) pc0[a] += sa[j] * cc[j][a]; pc0[a] /= n; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
            double s = 0.0;
            for (int j = 0; j < 4; ++j) s += cc[j][a] * (saw[j * 3 + b] - sa[j] * pw0[b]);
            ABt[a * 3 + b] = s;
        }
        // R = U V^T of ABt: orthogonal polar factor by Newton iteration (det(R) = sign det(ABt); OpenCV then flips the third row)
        double R[9];
        double nrm = 0.0;
        for (int a = 0; a < 9; ++a) nrm += ABt[a] * ABt[a];
        nrm = sqrt(nrm);
        if (!(nrm > 1e-300)) continue;
        for (int a = 0; a < 9; ++a) R[a] = ABt[a] / nrm;
        bool okp = true;
        for (int itn = 0; itn < 40; ++itn) {
            double inv[9];
            if (!inv3x3(R, inv)) { okp = false; break; }
            double dlt = 0.0;
            for (int r2 = 0; r2 < 3; ++r2) for (int a = 0; a < 3; ++a) {
                const double nv = 0.5 * (R[r2 * 3 + a] + inv[a * 3 + r2]);
                dlt = fmax(dlt, fabs(nv - R[r2 * 3 + a]));
                R[r2 * 3 + a] = nv;
            }
            if (dlt < 1e-15) break;
        }
        if (!okp) continue;
        const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (det < 0.0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
        for (int a = 0; a < 9; ++a) { cd.R[a] = R[a]; fin = fin && isfinite(R[a]); }
        for (int a = 0; a < 3; ++a) { cd.t[a] = pc0[a] - (R[a * 3] * pw0[0] + R[a * 3 + 1] * pw0[1] + R[a * 3 + 2] * pw0[2]); fin = fin && isfinite(cd.t[a]); }
        cd.ok = fin;
        if (fin) {
            polish(cd.R, cd.t);
            for (int a = 0; a < 9; ++a) cd.ok = cd.ok && isfinite(cd.R[a]);
            for (int a = 0; a < 3; ++a) cd.ok = cd.ok && isfinite(cd.t[a]);
        }
    }
    // ---- pass 3: mean reprojection error of the candidates
    double e[3] = {0, 0, 0};
    ps.for_each([&](double X, double Y, double Z, double u, double v, bool) {
        for (int c = 0; c < 3; ++c) {
            if (!cand[c].ok) continue;
            const double* R = cand[c].R;
            const double p0 = R[0] * X + R[1] * Y + R[2] * Z + cand[c].t[0], p1 = R[3] * X + R[4] * Y + R[5] * Z + cand[c].t[1],
                         p2 = R[6] * X + R[7] * Y + R[8] * Z + cand[c].t[2];
            const double iz = 1.0 / p2;
            const double du = u - (k.uc + k.fu * p0 * iz), dv = v - (k.vc + k.fv * p1 * iz);
            e[c] += sqrt(du * du + dv * dv);
        }
    });
    int best = -1;
    for (int c = 0; c < 3; ++c) {
        e[c] = ps.reduce(e[c]) / n;
        if (cand[c].ok && isfinite(e[c]) && (best < 0 || e[c] < e[best])) best = c;
    }
    if (best < 0) return;
    out = cand[best];
    out.err = e[best];
    out.ok = true;
}

}  // namespace epnp
