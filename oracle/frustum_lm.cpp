// Oracle: CPU restatement of the frustum_reg "inverse camera projection" solver.
//
// TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Also timed as the CPU baseline
// (cpu_baseline.kind == "port") by bench.py.
//
// Restates, with file:line relative to /root/reference/evaluation/frustum_reg/src:
//   solvePGivenK                     registration.cpp:9-186
//   GivenKInsideImgError2D / 3D      registration_2d.hpp:107-129 / registration_3d.hpp:106-127
//   GivenKOutsideImgError2D / 3D     registration_2d.hpp:35-69  / registration_3d.hpp:35-68
// and the third-party pieces the reference calls but does not vendor:
//   Ceres Solver (find_package(Ceres REQUIRED), CMakeLists.txt:9, version UNPINNED, era 1.14-2.0):
//     AutoDiffCostFunction (forward-mode duals, here `Dual<NP>`), CauchyLoss(1.0) with the
//     Corrector degenerate case (rho'' < 0  =>  residual and Jacobian scaled by sqrt(rho')),
//     cost = 1/2 sum_blocks rho(|r_block|^2), trust-region Levenberg-Marquardt with Ceres'
//     documented defaults (initial radius 1e4, Jacobi column scaling 1/(1+|J_col|) fixed at the
//     first iterate, LM diagonal clamp [1e-6,1e32], min_relative_decrease 1e-3, radius update
//     r/max(1/3, 1-(2q-1)^3) on success and r/2,4,8.. on failure, function/gradient/parameter
//     tolerances 1e-6/1e-10/1e-8, box bounds by projection inside Plus() + projected-gradient
//     test, Armijo line search along the projected step for bounded problems, at most 5
//     consecutive invalid steps), ceres::AngleAxisRotatePoint (rotation.h), and
//     Problem::Evaluate returning loss-corrected residuals.
//
// PARITY UNPINNED: Ceres/Eigen are absent from the reference tree and from this image, the
// reference holds no golden vector for this path (evaluation/test_frustum_solver.py has a
// stale signature and no asserts), so iterate-level equality with Ceres cannot be checked.
// The Armijo search follows Ceres' defaults (line_search.cc ArmijoLineSearch::DoSearch, polynomial.cc):
// line_search_interpolation_type = CUBIC, i.e. every trial point is evaluated WITH its gradient, the
// next step minimises the polynomial that interpolates value + directional derivative at step 0, at
// the current trial and (from the third trial on) at the previous trial -- a cubic, then a quintic --
// over [1e-3, 0.6] x current step (MinimizePolynomial: interval midpoint, both ends, critical points),
// an invalid trial (non-finite cost or Jacobian) halves the step, at most 20 trials, minimum step 1e-9.
// Remaining deviations from Ceres, all equal in exact arithmetic:
//   * linear solve: Cholesky of the 4x4/6x6 normal equations instead of DENSE_QR of the stacked Jacobian;
//   * interpolating polynomial: fitted in the normalised variable u = step / current step with the two
//     constraints at step 0 eliminated analytically (Ceres: FullPivLU of the raw Vandermonde system);
//   * its minimiser: real critical points isolated by bracketing (Ceres: real parts of the eigenvalues of
//     the companion matrix of the derivative; real parts of complex roots can never win the strict
//     comparison of MinimizePolynomial because the minimum over a closed interval is attained at an end
//     point or a real critical point, which are all candidates).
// What pins this oracle instead (tests/test_oracle_solver.py):
// residual known-answers from an independent numpy evaluation, dual-number Jacobians vs central
// finite differences, cost == 1/2 sum log(1+|r|^2), synthetic recover-the-pose runs, and a
// solution-level cross-check against scipy.optimize.least_squares on the corrected residuals.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// Diagnostics (tools/model_temporal.py): when set, eval() appends every point it is called at with gradients -- the sequence of
// trial points a solve visits, which the HIP kernel sweeps in the same order.
thread_local std::vector<double>* g_eval_trace = nullptr;

template <int NP>
struct Dual {
    double a;
    double v[NP];
    Dual() : a(0.0) { for (int i = 0; i < NP; ++i) v[i] = 0.0; }
    explicit Dual(double x) : a(x) { for (int i = 0; i < NP; ++i) v[i] = 0.0; }
    Dual(double x, int k) : a(x) { for (int i = 0; i < NP; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int NP> Dual<NP> operator+(const Dual<NP>& x, const Dual<NP>& y) { Dual<NP> r; r.a = x.a + y.a; for (int i = 0; i < NP; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int NP> Dual<NP> operator-(const Dual<NP>& x, const Dual<NP>& y) { Dual<NP> r; r.a = x.a - y.a; for (int i = 0; i < NP; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int NP> Dual<NP> operator-(const Dual<NP>& x) { Dual<NP> r; r.a = -x.a; for (int i = 0; i < NP; ++i) r.v[i] = -x.v[i]; return r; }
template <int NP> Dual<NP> operator*(const Dual<NP>& x, const Dual<NP>& y) { Dual<NP> r; r.a = x.a * y.a; for (int i = 0; i < NP; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int NP> Dual<NP> operator/(const Dual<NP>& f, const Dual<NP>& g) {
    Dual<NP> r; const double gi = 1.0 / g.a; const double q = f.a * gi; r.a = q;
    for (int i = 0; i < NP; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi; return r;
}
template <int NP> Dual<NP> operator*(const Dual<NP>& x, double s) { Dual<NP> r; r.a = x.a * s; for (int i = 0; i < NP; ++i) r.v[i] = x.v[i] * s; return r; }
template <int NP> Dual<NP> dsqrt(const Dual<NP>& x) { Dual<NP> r; r.a = std::sqrt(x.a); const double d = 0.5 / r.a; for (int i = 0; i < NP; ++i) r.v[i] = x.v[i] * d; return r; }
template <int NP> Dual<NP> dsin(const Dual<NP>& x) { Dual<NP> r; r.a = std::sin(x.a); const double c = std::cos(x.a); for (int i = 0; i < NP; ++i) r.v[i] = x.v[i] * c; return r; }
template <int NP> Dual<NP> dcos(const Dual<NP>& x) { Dual<NP> r; r.a = std::cos(x.a); const double s = -std::sin(x.a); for (int i = 0; i < NP; ++i) r.v[i] = x.v[i] * s; return r; }
template <int NP> Dual<NP> dabs(const Dual<NP>& x) { return x.a < 0.0 ? -x : x; }          // sign(0) = +
template <int NP> Dual<NP> dfmax(const Dual<NP>& x, const Dual<NP>& y) { return x.a < y.a ? y : x; }  // tie -> x

// ceres::AngleAxisRotatePoint (rotation.h): Rodrigues for theta^2 > eps, first order otherwise.
template <int NP>
void angle_axis_rotate(const Dual<NP> w[3], const double pt[3], Dual<NP> out[3]) {
    const Dual<NP> theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    Dual<NP> p[3] = {Dual<NP>(pt[0]), Dual<NP>(pt[1]), Dual<NP>(pt[2])};
    if (theta2.a > DBL_EPSILON) {
        const Dual<NP> theta = dsqrt(theta2);
        const Dual<NP> c = dcos(theta), s = dsin(theta);
        const Dual<NP> ti = Dual<NP>(1.0) / theta;
        const Dual<NP> u[3] = {w[0] * ti, w[1] * ti, w[2] * ti};
        const Dual<NP> ucp[3] = {u[1] * p[2] - u[2] * p[1], u[2] * p[0] - u[0] * p[2], u[0] * p[1] - u[1] * p[0]};
        const Dual<NP> tmp = (u[0] * p[0] + u[1] * p[1] + u[2] * p[2]) * (Dual<NP>(1.0) - c);
        for (int i = 0; i < 3; ++i) out[i] = p[i] * c + ucp[i] * s + u[i] * tmp;
    } else {
        const Dual<NP> wcp[3] = {w[1] * p[2] - w[2] * p[1], w[2] * p[0] - w[0] * p[2], w[0] * p[1] - w[1] * p[0]};
        for (int i = 0; i < 3; ++i) out[i] = p[i] + wcp[i];
    }
}

struct Camera { double fx, fy, cx, cy, H1, W1; };  // H1 = H-1, W1 = W-1 (registration.cpp:21-22)

// Shared front half of all four functors: p = R x + t, pixel = K p.
template <int NP>
void project(const Dual<NP>* cam, const double pt[3], const Camera& k, Dual<NP> p[3], Dual<NP>& px, Dual<NP>& py) {
    Dual<NP> w[3];
    int toff;
    if (NP == 4) { w[0] = Dual<NP>(0.0); w[1] = cam[0]; w[2] = Dual<NP>(0.0); toff = 1; }   // registration_2d.hpp:40-41
    else { w[0] = cam[0]; w[1] = cam[1]; w[2] = cam[2]; toff = 3; }                            // registration_3d.hpp:39
    angle_axis_rotate<NP>(w, pt, p);
    for (int i = 0; i < 3; ++i) p[i] = p[i] + cam[toff + i];
    px = p[0] * k.fx / p[2] + Dual<NP>(k.cx);
    py = p[1] * k.fy / p[2] + Dual<NP>(k.cy);
}

template <int NP>
void inside_residual(const Dual<NP>* cam, const double pt[3], const Camera& k, Dual<NP> r[3]) {
    Dual<NP> p[3], px, py;
    project<NP>(cam, pt, k, p, px, py);
    const Dual<NP> zero(0.0);
    r[0] = dfmax(-px, zero) + dfmax(px - Dual<NP>(k.W1), zero);      // registration_2d.hpp:118
    r[1] = dfmax(-py, zero) + dfmax(py - Dual<NP>(k.H1), zero);      // :119
    r[2] = dfmax(-p[2], zero) * 100.0;                               // :122
}

template <int NP>
void outside_residual(const Dual<NP>* cam, const double pt[3], const Camera& k, Dual<NP> r[1]) {
    Dual<NP> p[3], px, py;
    project<NP>(cam, pt, k, p, px, py);
    const Dual<NP> zero(0.0);
    const Dual<NP> dx = Dual<NP>(k.W1 * 0.5) - dabs(px - Dual<NP>(k.W1 * 0.5));   // registration_2d.hpp:52
    const Dual<NP> ix = dfmax(dx, zero) / dx;                                      // :53 (NaN at dx == 0)
    const Dual<NP> dy = Dual<NP>(k.H1 * 0.5) - dabs(py - Dual<NP>(k.H1 * 0.5));   // :55
    const Dual<NP> iy = dfmax(dy, zero) / dy;                                      // :56
    const Dual<NP> fr = dfmax(p[2], zero) / p[2];                                  // :58
    r[0] = (dx + dy) * fr * ix * iy;                                               // :60-62
}

template <int NP>
struct Problem {
    const double* pts;   // 3 x N, row-major (numpy (3,N) C-order == what pybind hands Eigen)
    const int* labels;
    int N;
    Camera k;
    double lb[NP], ub[NP];

    void plus(const double* x, const double* d, double* out) const {     // ParameterBlock::Plus + box projection
        for (int i = 0; i < NP; ++i) out[i] = std::min(std::max(x[i] + d[i], lb[i]), ub[i]);
    }

    // cost = 1/2 sum rho(s); optionally g = J^T r and A = J^T J of the loss-corrected system.
    // Returns false when anything is non-finite (Ceres: evaluation failure).
    bool eval(const double* x, double* cost, double* g, double* A, std::vector<double>* res_out) const {
        Dual<NP> cam[NP];
        for (int i = 0; i < NP; ++i) cam[i] = Dual<NP>(x[i], i);
        if (g_eval_trace && g) for (int i = 0; i < NP; ++i) g_eval_trace->push_back(x[i]);
        double c = 0.0;
        if (g) { std::memset(g, 0, sizeof(double) * NP); std::memset(A, 0, sizeof(double) * NP * NP); }
        bool ok = true;
        for (int n = 0; n < N; ++n) {
            const int lab = labels[n];
            if (lab != 0 && lab != 1) continue;                         // registration.cpp:87-125
            const double pt[3] = {pts[n], pts[N + n], pts[2 * N + n]};
            Dual<NP> r[3];
            const int nr = lab == 1 ? 3 : 1;
            if (lab == 1) inside_residual<NP>(cam, pt, k, r); else outside_residual<NP>(cam, pt, k, r);
            double s = 0.0;
            for (int i = 0; i < nr; ++i) s += r[i].a * r[i].a;
            c += 0.5 * std::log1p(s);                                   // CauchyLoss(1): rho(s) = log(1+s)
            const double rho1 = 1.0 / (1.0 + s);
            const double sq = std::sqrt(rho1);                          // Corrector, rho'' < 0 branch
            if (!std::isfinite(s)) ok = false;
            if (res_out) for (int i = 0; i < nr; ++i) res_out->push_back(r[i].a * sq);
            if (g) {
                for (int i = 0; i < nr; ++i) {
                    for (int a = 0; a < NP; ++a) {
                        if (!std::isfinite(r[i].v[a])) ok = false;
                        g[a] += rho1 * r[i].v[a] * r[i].a;
                        for (int b = 0; b <= a; ++b) A[a * NP + b] += rho1 * r[i].v[a] * r[i].v[b];
                    }
                }
            }
        }
        if (g) for (int a = 0; a < NP; ++a) for (int b = a + 1; b < NP; ++b) A[a * NP + b] = A[b * NP + a];
        *cost = c;
        return ok && std::isfinite(c);
    }
};

// Cholesky solve of M y = rhs (M symmetric NPxNP).  false if not positive definite / non-finite.
template <int NP>
bool chol_solve(const double* M, const double* rhs, double* y) {
    double L[NP * NP] = {0};
    for (int i = 0; i < NP; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = M[i * NP + j];
            for (int q = 0; q < j; ++q) s -= L[i * NP + q] * L[j * NP + q];
            if (i == j) { if (!(s > 0.0) || !std::isfinite(s)) return false; L[i * NP + i] = std::sqrt(s); }
            else L[i * NP + j] = s / L[j * NP + j];
        }
    }
    double z[NP];
    for (int i = 0; i < NP; ++i) { double s = rhs[i]; for (int q = 0; q < i; ++q) s -= L[i * NP + q] * z[q]; z[i] = s / L[i * NP + i]; }
    for (int i = NP - 1; i >= 0; --i) { double s = z[i]; for (int q = i + 1; q < NP; ++q) s -= L[q * NP + i] * y[q]; y[i] = s / L[i * NP + i]; }
    for (int i = 0; i < NP; ++i) if (!std::isfinite(y[i])) return false;
    return true;
}


// ---------------------------------------------------------------------------------------------------
// Ceres polynomial.cc / line_search.cc restated (see header for the three exact-arithmetic-equal deviations).
struct Sample { double x, value, gradient; bool value_ok, grad_ok; };

inline double poly_eval(const double* c, int deg, double u) {     // c[0] + c[1] u + ... (Horner)
    double v = c[deg];
    for (int i = deg - 1; i >= 0; --i) v = v * u + c[i];
    return v;
}

// root of the monotone piece of q on [a,b] with q(a), q(b) of opposite sign: safeguarded Newton / bisection
inline double bracket_root(const double* q, int deg, double a, double b, double qa, double qb) {
    double dq[6];
    for (int i = 1; i <= deg; ++i) dq[i - 1] = i * q[i];
    double x = 0.5 * (a + b);
    for (int it = 0; it < 200; ++it) {
        const double qx = poly_eval(q, deg, x);
        if (qx == 0.0) return x;
        if ((qx < 0.0) == (qa < 0.0)) { a = x; qa = qx; } else { b = x; qb = qx; }
        const double d = poly_eval(dq, deg - 1, x);
        double xn = x - qx / d;
        if (!(xn > a && xn < b)) xn = 0.5 * (a + b);
        if (xn == x || b - a <= 4e-16 * std::fabs(x)) return xn;
        x = xn;
    }
    return x;
}

// all real roots of q (degree <= 4, ascending coefficients) inside [a,b], ascending, via the roots of q'
inline int real_roots_in(const double* q, int deg, double a, double b, double* roots) {
    while (deg > 0 && q[deg] == 0.0) --deg;                          // RemoveLeadingZeros
    if (deg == 0) return 0;
    if (deg == 1) { const double r = -q[0] / q[1]; if (r >= a && r <= b) { roots[0] = r; return 1; } return 0; }
    double dq[5], crit[4];
    for (int i = 1; i <= deg; ++i) dq[i - 1] = i * q[i];
    const int nc = real_roots_in(dq, deg - 1, a, b, crit);
    double knots[6];
    int nk = 0;
    knots[nk++] = a;
    for (int i = 0; i < nc; ++i) knots[nk++] = crit[i];
    knots[nk++] = b;
    int n = 0;
    for (int i = 0; i + 1 < nk; ++i) {
        const double l = knots[i], r = knots[i + 1];
        const double ql = poly_eval(q, deg, l), qr = poly_eval(q, deg, r);
        if (ql == 0.0) { if (n == 0 || roots[n - 1] != l) roots[n++] = l; }
        if (ql != 0.0 && qr != 0.0 && (ql < 0.0) != (qr < 0.0)) roots[n++] = bracket_root(q, deg, l, r, ql, qr);
        if (i + 2 == nk && qr == 0.0) roots[n++] = r;
    }
    return n;
}

// LineSearch::InterpolatingPolynomialMinimizingStepSize for CUBIC: lowerbound = (0, f0, g0) always valid.
inline double interpolating_step(double f0, double g0, const Sample& cur, const Sample& prev, double min_step, double max_step) {
    if (!cur.value_ok) return std::min(std::max(cur.x * 0.5, min_step), max_step);
    // polynomial p(u) = f0 + g0 xc u + u^2 r(u), u = x / xc; r has one coefficient per remaining constraint
    const double xc = cur.x, G0 = g0 * xc;
    double M[4][5];
    int m = 0;
    auto row_value = [&](double u, double val) {       // u^2 r(u) = val - f0 - G0 u
        double pw = u * u;
        for (int j = 0; j < 4; ++j) { M[m][j] = pw; pw *= u; }
        M[m][4] = val - f0 - G0 * u; ++m;
    };
    auto row_grad = [&](double u, double gu) {         // d/du [u^2 r(u)] = gu - G0,  column j: (j+2) u^(j+1)
        double pw = u;
        for (int j = 0; j < 4; ++j) { M[m][j] = (j + 2) * pw; pw *= u; }
        M[m][4] = gu - G0; ++m;
    };
    row_value(1.0, cur.value);
    if (cur.grad_ok) row_grad(1.0, cur.gradient * xc);
    if (prev.value_ok) {
        const double up = prev.x / xc;
        row_value(up, prev.value);
        if (prev.grad_ok) row_grad(up, prev.gradient * xc);
    }
    // m x m solve, Gaussian elimination with partial pivoting (columns 0..m-1 are the unknowns r_0..r_{m-1})
    double rc[4] = {0, 0, 0, 0};
    bool singular = false;
    for (int c = 0; c < m; ++c) {
        int piv = c;
        for (int r = c + 1; r < m; ++r) if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
        if (M[piv][c] == 0.0) { singular = true; break; }
        if (piv != c) for (int j = 0; j < 5; ++j) std::swap(M[piv][j], M[c][j]);
        for (int r = c + 1; r < m; ++r) {
            const double f = M[r][c] / M[c][c];
            for (int j = c; j < m; ++j) M[r][j] -= f * M[c][j];
            M[r][4] -= f * M[c][4];
        }
    }
    if (!singular)
        for (int c = m - 1; c >= 0; --c) {
            double v = M[c][4];
            for (int j = c + 1; j < m; ++j) v -= M[c][j] * rc[j];
            rc[c] = v / M[c][c];
        }
    double p[6] = {f0, G0, 0, 0, 0, 0};
    const int deg = m + 1;
    for (int j = 0; j < m; ++j) p[j + 2] = rc[j];
    for (int j = 0; j <= deg; ++j) if (!std::isfinite(p[j])) return std::min(std::max(cur.x * 0.5, min_step), max_step);
    // MinimizePolynomial over u in [umin, umax]
    const double umin = min_step / xc, umax = max_step / xc;
    double best_u = 0.5 * (umin + umax), best_v = poly_eval(p, deg, best_u);
    const double vmin = poly_eval(p, deg, umin);
    if (vmin < best_v) { best_v = vmin; best_u = umin; }
    const double vmax = poly_eval(p, deg, umax);
    if (vmax < best_v) { best_v = vmax; best_u = umax; }
    double dp[5], roots[4];
    for (int i = 1; i <= deg; ++i) dp[i - 1] = i * p[i];
    const int nr = real_roots_in(dp, deg - 1, umin, umax, roots);
    for (int i = 0; i < nr; ++i) {
        const double v = poly_eval(p, deg, roots[i]);
        if (v < best_v) { best_v = v; best_u = roots[i]; }
    }
    return best_u * xc;
}

enum Term { T_MAX_ITER = 0, T_GRADIENT = 1, T_PARAMETER = 2, T_FUNCTION = 3, T_RADIUS = 4, T_INVALID = 5, T_EVAL_FAIL = 6 };

template <int NP>
void minimize(const Problem<NP>& P, double* x, int max_iter, int* iters_out, int* term_out, int* n_eval_out) {
    const double kMinDiag = 1e-6, kMaxDiag = 1e32, kMaxRadius = 1e16, kMinRadius = 1e-32;
    const double kMinRelDec = 1e-3, kFuncTol = 1e-6, kGradTol = 1e-10, kParamTol = 1e-8;
    double zero[NP] = {0}, tmp[NP];
    P.plus(x, zero, tmp);                                                // project the start onto the box
    std::memcpy(x, tmp, sizeof(tmp));
    double cost, g[NP], A[NP * NP];
    int n_eval = 1;
    int iter = 0, term = T_MAX_ITER;
    if (!P.eval(x, &cost, g, A, nullptr)) { *iters_out = 0; *term_out = T_EVAL_FAIL; *n_eval_out = n_eval; return; }
    double S[NP];                                                        // Jacobi scaling, fixed at iteration 0
    for (int i = 0; i < NP; ++i) S[i] = 1.0 / (1.0 + std::sqrt(A[i * NP + i]));
    auto grad_max_norm = [&](const double* xx, const double* gg) {       // projected gradient, inf-norm
        double neg[NP], pr[NP], m = 0.0;
        for (int i = 0; i < NP; ++i) neg[i] = -gg[i];
        P.plus(xx, neg, pr);
        for (int i = 0; i < NP; ++i) m = std::max(m, std::fabs(xx[i] - pr[i]));
        return m;
    };
    double gmax = grad_max_norm(x, g);
    double radius = 1e4, decrease = 2.0;
    bool reuse_diag = false;
    double diag[NP];
    int invalid_run = 0;
    for (;;) {
        if (iter >= max_iter) { term = T_MAX_ITER; break; }
        if (gmax <= kGradTol) { term = T_GRADIENT; break; }
        if (radius <= kMinRadius) { term = T_RADIUS; break; }
        ++iter;
        // --- LevenbergMarquardtStrategy::ComputeStep on the column-scaled system
        double As[NP * NP], gs[NP], M[NP * NP], rhs[NP], ds[NP];
        for (int a = 0; a < NP; ++a) { gs[a] = S[a] * g[a]; for (int b = 0; b < NP; ++b) As[a * NP + b] = S[a] * A[a * NP + b] * S[b]; }
        if (!reuse_diag) for (int a = 0; a < NP; ++a) diag[a] = std::min(std::max(As[a * NP + a], kMinDiag), kMaxDiag);
        std::memcpy(M, As, sizeof(M));
        for (int a = 0; a < NP; ++a) { M[a * NP + a] += diag[a] / radius; rhs[a] = -gs[a]; }
        bool valid = chol_solve<NP>(M, rhs, ds);
        double model_change = 0.0;
        if (valid) {
            double q = 0.0, l = 0.0;
            for (int a = 0; a < NP; ++a) { l += ds[a] * gs[a]; for (int b = 0; b < NP; ++b) q += ds[a] * As[a * NP + b] * ds[b]; }
            model_change = -(l + 0.5 * q);
            valid = model_change > 0.0;
        }
        if (!valid) {
            if (++invalid_run >= 5) { term = T_INVALID; break; }
            radius /= decrease; decrease *= 2.0; reuse_diag = true;
            continue;
        }
        invalid_run = 0;
        double delta[NP];
        for (int a = 0; a < NP; ++a) delta[a] = ds[a] * S[a];
        // --- projected Armijo line search (bounded problem), Ceres defaults: CUBIC interpolation, so every trial is
        //     evaluated with its gradient (TrustRegionMinimizer::DoLineSearch -> ArmijoLineSearch::DoSearch)
        {
            double gd = 0.0, dmax = 0.0;
            for (int a = 0; a < NP; ++a) { gd += g[a] * delta[a]; dmax = std::max(dmax, std::fabs(delta[a])); }
            auto sample_at = [&](double tt) {
                Sample sm{tt, 0.0, 0.0, false, false};
                double d2[NP], xc2[NP], gt[NP], At[NP * NP];
                for (int a = 0; a < NP; ++a) d2[a] = tt * delta[a];
                P.plus(x, d2, xc2);
                ++n_eval;
                if (!P.eval(xc2, &sm.value, gt, At, nullptr)) return sm;     // non-finite cost OR Jacobian: invalid sample
                sm.value_ok = true;
                double gdir = 0.0;
                for (int a = 0; a < NP; ++a) gdir += delta[a] * gt[a];
                sm.gradient = gdir;
                sm.grad_ok = std::isfinite(gdir);
                return sm;
            };
            Sample prev{0.0, 0.0, 0.0, false, false};
            Sample cur = sample_at(1.0);
            int ls_it = 0; bool success = false;
            for (;;) {
                if (cur.value_ok && cur.value <= cost + 1e-4 * gd * cur.x) { success = true; break; }
                if (++ls_it >= 20) break;
                const double tn = interpolating_step(cost, gd, cur, prev, 1e-3 * cur.x, 0.6 * cur.x);
                if (tn * dmax < 1e-9) break;
                prev = cur;
                cur = sample_at(tn);
            }
            if (success && cur.x != 1.0) for (int a = 0; a < NP; ++a) delta[a] *= cur.x;
        }
        double xc[NP], cand_cost;
        P.plus(x, delta, xc);
        ++n_eval;
        if (!P.eval(xc, &cand_cost, nullptr, nullptr, nullptr)) cand_cost = DBL_MAX;
        double step_norm = 0.0, x_norm = 0.0;
        for (int a = 0; a < NP; ++a) { step_norm += (x[a] - xc[a]) * (x[a] - xc[a]); x_norm += x[a] * x[a]; }
        step_norm = std::sqrt(step_norm); x_norm = std::sqrt(x_norm);
        if (step_norm <= kParamTol * (x_norm + kParamTol)) { term = T_PARAMETER; break; }
        if (std::fabs(cost - cand_cost) <= kFuncTol * cost) { term = T_FUNCTION; break; }
        const double rel = (cost - cand_cost) / model_change;
        if (rel > kMinRelDec) {
            std::memcpy(x, xc, sizeof(xc));
            ++n_eval;
            if (!P.eval(x, &cost, g, A, nullptr)) { term = T_EVAL_FAIL; break; }
            gmax = grad_max_norm(x, g);
            radius = std::min(kMaxRadius, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            decrease = 2.0; reuse_diag = false;
        } else {
            radius /= decrease; decrease *= 2.0; reuse_diag = true;
        }
    }
    *iters_out = iter; *term_out = term; *n_eval_out = n_eval;
}

void angle_axis_to_R(const double w[3], double R[9]) {     // ceres::AngleAxisToRotationMatrix, row-major out
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (t2 > DBL_EPSILON) {
        const double t = std::sqrt(t2), wx = w[0] / t, wy = w[1] / t, wz = w[2] / t, c = std::cos(t), s = std::sin(t);
        R[0] = c + wx * wx * (1 - c);      R[1] = wx * wy * (1 - c) - wz * s; R[2] = wy * s + wx * wz * (1 - c);
        R[3] = wz * s + wx * wy * (1 - c); R[4] = c + wy * wy * (1 - c);      R[5] = -wx * s + wy * wz * (1 - c);
        R[6] = -wy * s + wx * wz * (1 - c); R[7] = wx * s + wy * wz * (1 - c); R[8] = c + wz * wz * (1 - c);
    } else {
        R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
    }
}

template <int NP>
int solve_impl(const double* pts, const int* labels, int N, const double K[9], double init_y, const double init_T[3],
               double H, double W, const double lb[3], const double ub[3], int max_iter,
               double P_out[16], double* final_cost, double* residuals, int* n_res, int* iters, int* term, double* params_out) {
    Problem<NP> P;
    P.pts = pts; P.labels = labels; P.N = N;
    P.k = Camera{K[0], K[4], K[2], K[5], H - 1.0, W - 1.0};          // registration.cpp:21-22,79-82
    const int toff = NP == 4 ? 1 : 3;
    for (int i = 0; i < NP; ++i) { P.lb[i] = -DBL_MAX; P.ub[i] = DBL_MAX; }
    for (int i = 0; i < 3; ++i) { P.lb[toff + i] = lb[i]; P.ub[toff + i] = ub[i]; }   // :128-135
    double x[NP];
    if (NP == 4) { x[0] = init_y; } else { x[0] = 0.0; x[1] = init_y; x[2] = 0.0; }   // :34-50
    for (int i = 0; i < 3; ++i) x[toff + i] = init_T[i];
    int it = 0, tm = 0, ne = 0;
    minimize<NP>(P, x, max_iter, &it, &tm, &ne);
    std::vector<double> res;
    double c = 0.0;
    P.eval(x, &c, nullptr, nullptr, residuals ? &res : nullptr);       // Problem::Evaluate (:150-155)
    if (residuals) std::memcpy(residuals, res.data(), sizeof(double) * res.size());
    if (n_res) { int cnt = 0; for (int n = 0; n < N; ++n) cnt += labels[n] == 1 ? 3 : (labels[n] == 0 ? 1 : 0); *n_res = cnt; }
    double w[3] = {0, 0, 0};
    if (NP == 4) w[1] = x[0]; else { w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; }
    double R[9];
    angle_axis_to_R(w, R);
    for (int i = 0; i < 16; ++i) P_out[i] = 0.0;
    for (int r = 0; r < 3; ++r) { for (int cidx = 0; cidx < 3; ++cidx) P_out[r * 4 + cidx] = R[r * 3 + cidx]; P_out[r * 4 + 3] = x[toff + r]; }
    P_out[15] = 1.0;
    *final_cost = c;
    if (iters) *iters = it;
    if (term) *term = tm;
    if (params_out) for (int i = 0; i < NP; ++i) params_out[i] = x[i];
    return ne;
}

}  // namespace

extern "C" {

// One solvePGivenK call.  points: 3 x N row-major f64; labels i32[N]; K row-major 3x3.
// residuals may be NULL; otherwise must hold 3*N_in + N_out doubles.  Returns #cost evaluations.
int oracle_solve_p_given_k(const double* points, const int* labels, int N, const double* K, double init_y_angle,
                           const double* init_T, double H, double W, const double* lb, const double* ub, int max_iter,
                           int is_2d, double* P_out, double* final_cost, double* residuals, int* n_res, int* iters,
                           int* term, double* params_out) {
    if (is_2d) return solve_impl<4>(points, labels, N, K, init_y_angle, init_T, H, W, lb, ub, max_iter, P_out, final_cost, residuals, n_res, iters, term, params_out);
    return solve_impl<6>(points, labels, N, K, init_y_angle, init_T, H, W, lb, ub, max_iter, P_out, final_cost, residuals, n_res, iters, term, params_out);
}

// Diagnostics: the same solve, also returning the points every cost + gradient evaluation was made at (trace: cap x (4|6) doubles).
// Returns the number of evaluations (may exceed cap; only the first cap are stored).
int oracle_solve_trace(const double* points, const int* labels, int N, const double* K, double init_y_angle, const double* init_T,
                       double H, double W, const double* lb, const double* ub, int max_iter, int is_2d, double* final_cost,
                       double* trace, int cap) {
    std::vector<double> tr;
    g_eval_trace = &tr;
    double P[16];
    oracle_solve_p_given_k(points, labels, N, K, init_y_angle, init_T, H, W, lb, ub, max_iter, is_2d, P, final_cost, nullptr, nullptr,
                           nullptr, nullptr, nullptr);
    g_eval_trace = nullptr;
    const int np = is_2d ? 4 : 6, n = (int)(tr.size() / np);
    std::memcpy(trace, tr.data(), sizeof(double) * np * std::min(n, cap));
    return n;
}

// R independent restarts of one frame spread over nthreads std::threads (mirrors the reference's
// waves of OS processes, evaluation/registration_lsq.py:142-186).  init_T: R x 3.
void oracle_solve_restarts(const double* points, const int* labels, int N, const double* K, int R,
                           const double* init_y_angles, const double* init_Ts, double H, double W, const double* lb,
                           const double* ub, int max_iter, int is_2d, int nthreads, double* P_out /*R x 16*/,
                           double* costs /*R*/, int* iters /*R*/, int* terms /*R*/, double* params /*R x (4|6)*/) {
    const int np = is_2d ? 4 : 6;
    auto work = [&](int t) {
        for (int r = t; r < R; r += nthreads)
            oracle_solve_p_given_k(points, labels, N, K, init_y_angles[r], init_Ts + 3 * r, H, W, lb, ub, max_iter, is_2d,
                                   P_out + 16 * r, costs + r, nullptr, nullptr, iters + r, terms + r, params + np * r);
    };
    if (nthreads <= 1) { work(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
}

// Residuals (UNcorrected), Jacobian (row-major n_res x np, may be NULL), corrected cost at given params.
int oracle_residuals(const double* points, const int* labels, int N, const double* K, double H, double W, int is_2d,
                     const double* params, double* r_out, double* J_out, double* cost_out) {
    Camera k{K[0], K[4], K[2], K[5], H - 1.0, W - 1.0};
    int row = 0; double c = 0.0;
    for (int n = 0; n < N; ++n) {
        const int lab = labels[n];
        if (lab != 0 && lab != 1) continue;
        const double pt[3] = {points[n], points[N + n], points[2 * N + n]};
        const int nr = lab == 1 ? 3 : 1;
        double s = 0.0;
        if (is_2d) {
            Dual<4> cam[4], r[3]; for (int i = 0; i < 4; ++i) cam[i] = Dual<4>(params[i], i);
            if (lab == 1) inside_residual<4>(cam, pt, k, r); else outside_residual<4>(cam, pt, k, r);
            for (int i = 0; i < nr; ++i) { r_out[row + i] = r[i].a; s += r[i].a * r[i].a; if (J_out) for (int a = 0; a < 4; ++a) J_out[(row + i) * 4 + a] = r[i].v[a]; }
        } else {
            Dual<6> cam[6], r[3]; for (int i = 0; i < 6; ++i) cam[i] = Dual<6>(params[i], i);
            if (lab == 1) inside_residual<6>(cam, pt, k, r); else outside_residual<6>(cam, pt, k, r);
            for (int i = 0; i < nr; ++i) { r_out[row + i] = r[i].a; s += r[i].a * r[i].a; if (J_out) for (int a = 0; a < 6; ++a) J_out[(row + i) * 6 + a] = r[i].v[a]; }
        }
        c += 0.5 * std::log1p(s);
        row += nr;
    }
    if (cost_out) *cost_out = c;
    return row;
}

}  // extern "C"
