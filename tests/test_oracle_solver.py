"""CPU: pins for the solver oracle (oracle/frustum_lm.cpp).  The reference holds no golden vector for this path
and Ceres is absent (PARITY UNPINNED, see the oracle header); the oracle is pinned by independent evaluations:
residual known-answers (numpy), Jacobians (central differences), the cost definition, synthetic pose recovery,
a solution-level cross-check against scipy.optimize.least_squares, and the golden vectors of the four small
driver functions generated from the reference's own function bodies."""
import math

import numpy as np
import pytest

from deepi2p_amd import synthetic
from oracle import frustum_lm as flm

H, W = 160, 512
LB, UB = [-5, -0.1, -10], [5, 0.1, 10]


def _np_residuals(pts, lab, K, params, is_2d):
    """Independent restatement of registration_2d.hpp:35-69,107-129 / registration_3d.hpp in numpy."""
    from scipy.spatial.transform import Rotation
    if is_2d:
        R = Rotation.from_rotvec([0, params[0], 0]).as_matrix()
        t = params[1:4]
    else:
        R = Rotation.from_rotvec(params[0:3]).as_matrix()
        t = params[3:6]
    p = R @ pts + t[:, None]
    px = K[0, 0] * p[0] / p[2] + K[0, 2]
    py = K[1, 1] * p[1] / p[2] + K[1, 2]
    H1, W1 = H - 1.0, W - 1.0
    out = []
    for i in range(pts.shape[1]):
        if lab[i] == 1:
            out += [max(-px[i], 0) + max(px[i] - W1, 0), max(-py[i], 0) + max(py[i] - H1, 0), max(-p[2, i], 0) * 100.0]
        elif lab[i] == 0:
            dx = W1 / 2 - abs(px[i] - W1 / 2)
            dy = H1 / 2 - abs(py[i] - H1 / 2)
            out.append((dx + dy) if (dx > 0 and dy > 0 and p[2, i] > 0) else 0.0)
    return np.array(out)


@pytest.mark.parametrize("is_2d", [True, False])
def test_residuals_known_answers_and_jacobian(is_2d):
    rng = np.random.default_rng(0)
    f = synthetic.make_frame(rng, N=400, H=H, W=W, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"].copy()
    lab[::13] = 7
    params = np.array([0.4, 0.7, -0.03, 1.5]) if is_2d else np.array([0.05, 0.4, -0.03, 0.7, -0.03, 1.5])
    r, J, cost = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, is_2d, params)
    np.testing.assert_allclose(r, _np_residuals(pts, lab, f["K"], params, is_2d), rtol=1e-10, atol=1e-9)
    # cost = 1/2 sum_blocks log(1 + |r_block|^2)   (CauchyLoss(1), per residual BLOCK)
    sizes = np.where(lab[(lab == 0) | (lab == 1)] == 1, 3, 1)
    s = np.add.reduceat(r * r, np.r_[0, np.cumsum(sizes)[:-1]])
    assert abs(cost - 0.5 * np.log1p(s).sum()) < 1e-9 * cost
    # dual-number Jacobian vs central differences (away from the kinks: compare only rows that are smooth there)
    eps = 1e-6
    Jn = np.zeros_like(J)
    for a in range(params.size):
        d = np.zeros_like(params)
        d[a] = eps
        rp, _, _ = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, is_2d, params + d)
        rm, _, _ = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, is_2d, params - d)
        Jn[:, a] = (rp - rm) / (2 * eps)
    smooth = np.abs(Jn - J).max(axis=1) < 1e-3 * (1 + np.abs(J).max(axis=1))
    assert smooth.mean() > 0.995                       # a handful of rows straddle an indicator edge
    np.testing.assert_allclose(J[smooth], Jn[smooth], rtol=1e-5, atol=1e-5)


def test_recovers_pose_from_exact_labels():
    rng = np.random.default_rng(1)
    for _ in range(2):
        f = synthetic.make_frame(rng, N=3000, H=H, W=W, flip=0.0, with_image=False)
        pts, lab = f["pc"].astype(np.float64), f["labels"]
        _, y0, pcf, labf = flm.get_initial_guess(pts, lab)
        ys, Ts = flm.draw_restarts(rng, 24, y0, 10 * math.pi / 180, 10)
        P, cost, best, info = flm.solve_P_random_perturb(pcf, labf, f["K"], H, W, ys, Ts, LB, UB, True, 500, nthreads=8)
        t, r = flm.get_P_diff(P, f["P_gt"])
        assert t < 2.0 and r < 5.0                    # the reference's success rule
        assert np.all(info["iters"] <= 500) and best == int(np.argmin(info["cost"]))
        tr = P[:3, 3]
        assert np.all(tr >= np.array(LB) - 1e-12) and np.all(tr <= np.array(UB) + 1e-12)


def test_solution_level_crosscheck_with_scipy():
    """Solution-level cross-check against an independent bounded trust-region solver (scipy 'trf') fed the
    loss-corrected residuals.  The objective is discontinuous, so two solvers need not stop at the same stationary
    point (a single start routinely stalls on a few "outside" points whose residual can only vanish by jumping
    across the image border -- that is why the reference runs 60 restarts).  What must hold from a start in the
    right basin: (a) the cost definitions agree at any point, (b) with exact labels both solvers cut the cost at least 4x,
    (c) with noisy labels the oracle's minimum is no worse than 1.25x scipy's, and the oracle's pose stays within
    the reference's success rule of the ground truth."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(2)
    for flip in (0.0, 0.02):
        f = synthetic.make_frame(rng, N=1500, H=H, W=W, flip=flip, with_image=False)
        pts, lab = f["pc"].astype(np.float64), f["labels"]
        x0 = np.array([f["yaw_gt"] + 0.03, f["t_gt"][0] + 0.3, 0.0, f["t_gt"][2] - 0.4])
        sizes = np.where(lab == 1, 3, 1)

        def corrected(p):
            r, _, _ = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, True, p)
            s = np.add.reduceat(r * r, np.r_[0, np.cumsum(sizes)[:-1]])
            return np.sqrt(np.repeat(np.log1p(s), sizes) / np.maximum(np.repeat(s, sizes), 1e-300)) * r   # |.|^2 sums to rho

        _, _, c0 = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, True, x0)
        assert abs(c0 - 0.5 * np.sum(corrected(x0) ** 2)) <= 1e-9 * max(c0, 1.0)                 # (a)
        sol = least_squares(corrected, x0, bounds=([-np.inf] + LB, [np.inf] + UB), method="trf", xtol=1e-12, ftol=1e-12)
        c_scipy = 0.5 * np.sum(corrected(sol.x) ** 2)
        P, cost, res, info = flm.solvePGivenK(pts, lab, f["K"], x0[0], x0[1:], H, W, LB, UB, 500, False, True, return_info=True)
        if flip == 0.0:
            assert cost <= 0.25 * c0 and c_scipy <= 0.25 * c0                                     # (b)
        else:
            assert cost <= c0 and cost <= 1.25 * c_scipy + 1e-9                                  # (c)
        t, r = flm.get_P_diff(P, f["P_gt"])
        assert t < 2.0 and r < 5.0


def test_solvePGivenK_contract():
    rng = np.random.default_rng(3)
    f = synthetic.make_frame(rng, N=600, H=H, W=W, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    P, cost, res = flm.solvePGivenK(pts, lab, f["K"], 0.2, np.zeros(3), H, W, LB, UB, 50, False, True)
    assert P.shape == (4, 4) and np.allclose(P[3], [0, 0, 0, 1]) and abs(np.linalg.det(P[:3, :3]) - 1) < 1e-12
    assert res.shape[0] == 3 * int((lab == 1).sum()) + int((lab == 0).sum())
    P6, c6, _ = flm.solvePGivenK(pts, lab, f["K"], 0.2, np.zeros(3), H, W, LB, UB, 50, False, False)
    assert abs(np.linalg.det(P6[:3, :3]) - 1) < 1e-12
    with pytest.raises(IndexError):
        flm.solvePGivenK(pts, lab, f["K"], 0.2, np.zeros(3), H, W, [0, 0], UB, 5, False, True)
    # max_iter = 0 returns the (projected) start
    P0, _, _, info = flm.solvePGivenK(pts, lab, f["K"], 0.2, np.array([9.0, 0, 0]), H, W, LB, UB, 0, False, True, return_info=True)
    assert info["iters"] == 0 and P0[0, 3] == 5.0


def test_driver_functions_vs_reference_golden(golden):
    g = golden("lsq_driver_golden.npz")
    np.testing.assert_allclose([flm.wrap_in_pi(float(x)) for x in g["wrap_in"]], g["wrap_out"], atol=1e-15)
    for a, R in zip(g["a2r_in"], g["a2r_out"]):
        np.testing.assert_allclose(flm.angles2rotation_matrix(a), R, atol=1e-15)
    for i in range(3):
        P, y, pcf, labf = flm.get_initial_guess(g["ig%d_pc" % i], g["ig%d_lab" % i])
        assert abs(y - float(g["ig%d_y" % i])) < 1e-14
        np.testing.assert_allclose(P, g["ig%d_P" % i], atol=1e-14)
        np.testing.assert_array_equal(pcf, g["ig%d_pcf" % i])
        np.testing.assert_array_equal(labf, g["ig%d_labf" % i])
        t, r = flm.get_P_diff(g["pd%d_A" % i], g["pd%d_B" % i])
        assert abs(t - float(g["pd%d_t" % i])) < 1e-12 and abs(r - float(g["pd%d_r" % i])) < 1e-9
        m = flm.get_inside_img_mask(g["ig%d_pc" % i], g["pd%d_A" % i], g["im%d_K" % i], 160, 512)
        np.testing.assert_array_equal(m, g["im%d_mask" % i])
        m2 = synthetic.inside_mask(g["ig%d_pc" % i], g["pd%d_A" % i], g["im%d_K" % i], 160, 512)   # product-side generator
        np.testing.assert_array_equal(m2, g["im%d_mask" % i])


def test_fullsize_crosscheck_with_scipy_best_of_starts():
    """Config-2 size (N = 20480, 5 % label flips), 6 starts near the ground truth, oracle vs scipy 'trf' on the loss-corrected
    residuals.  The objective is discontinuous (a label-0 point entering the image jumps from 0 to dx+dy), so single runs of
    two different trust-region codes stall up to a few per cent apart; what the registration pipeline consumes is the
    MINIMUM over the restarts, and that -- plus mutual stationarity -- is what can be asserted tightly:
      (a) best-of-starts costs agree within 2 % (measured: 1e-4) and the two best poses within 0.25 m / 0.01 rad;
      (b) restarting the oracle from scipy's solution never raises the cost and lowers it by < 2 %, and vice versa:
          each solver accepts the other's answer as (near-)stationary."""
    from scipy.optimize import least_squares
    rng = np.random.default_rng(11)
    f = synthetic.make_frame(rng, N=20480, H=H, W=W, flip=0.05, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    sizes = np.where(lab == 1, 3, 1)
    starts = np.cumsum(sizes)[:-1]

    def corrected(p):
        r, _, _ = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, True, p)
        s = np.add.reduceat(r * r, np.r_[0, starts])
        return np.sqrt(np.repeat(np.log1p(s), sizes) / np.maximum(np.repeat(s, sizes), 1e-300)) * r

    def scipy_solve(x0):
        x0 = np.concatenate(([x0[0]], np.clip(x0[1:], LB, UB)))
        sol = least_squares(corrected, x0, bounds=([-np.inf] + LB, [np.inf] + UB), method="trf", xtol=1e-12, ftol=1e-12, gtol=1e-12)
        return sol.x, 0.5 * np.sum(corrected(sol.x) ** 2)

    best_o, best_s = (np.inf, None), (np.inf, None)
    for i in range(6):
        x0 = np.array([f["yaw_gt"] + rng.normal(0, 0.03), f["t_gt"][0] + rng.normal(0, 0.3), 0.0, f["t_gt"][2] + rng.normal(0, 0.4)])
        x0[1:] = np.clip(x0[1:], LB, UB)
        xs, cs = scipy_solve(x0)
        _, co, _, info = flm.solvePGivenK(pts, lab, f["K"], x0[0], x0[1:], H, W, LB, UB, 500, False, True, return_info=True)
        xo = info["params"]
        if i < 3:                                                                                   # (b) on three of the starts
            _, co2, _, _ = flm.solvePGivenK(pts, lab, f["K"], xs[0], xs[1:], H, W, LB, UB, 500, False, True, return_info=True)
            assert co2 <= cs * (1 + 1e-9) and co2 >= 0.98 * cs, (cs, co2)
            _, cs2 = scipy_solve(xo)
            assert cs2 <= co * (1 + 1e-9) and cs2 >= 0.98 * co, (co, cs2)
        if co < best_o[0]:
            best_o = (co, xo)
        if cs < best_s[0]:
            best_s = (cs, xs)
    assert max(best_o[0], best_s[0]) <= 1.02 * min(best_o[0], best_s[0]), (best_o[0], best_s[0])   # (a)
    assert np.linalg.norm(best_o[1][1:] - best_s[1][1:]) < 0.25 and abs(best_o[1][0] - best_s[1][0]) < 0.01
