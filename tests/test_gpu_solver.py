"""GPU parity of the batched pose solver against the oracle (oracle/frustum_lm.cpp) on identical hypotheses.

The solver path is PARITY-UNPINNED w.r.t. Ceres (see oracle header); what is checked here is HIP == oracle:
  * residuals / cost at given parameters: 1e-9 relative (pure fp64 formula evaluation);
  * per-hypothesis solutions: the objective is discontinuous, so an LM trajectory can fork on a 1-ulp
    difference in a reduction.  The bar is what is MEASURED on these fixed seeds: EVERY hypothesis agrees to |dt| <= 1e-3 m,
    |dR| <= 1e-3 rad (with equal iteration counts) except the ones enumerated in KNOWN_FORKS, and the best-of-R cost agrees to
    1e-6 relative (stated tolerance).  A regression that makes one more hypothesis fork fails the test."""
import math

import numpy as np
import pytest
import torch

from deepi2p_amd import synthetic
from oracle import frustum_lm as flm

pytestmark = pytest.mark.gpu
H, W = 160, 512
LB, UB = [-5, -0.1, -10], [5, 0.1, 10]


def _frame(seed, N, flip=0.05):
    rng = np.random.default_rng(seed)
    f = synthetic.make_frame(rng, N=N, H=H, W=W, flip=flip, with_image=False)
    return f, rng


@pytest.mark.parametrize("is_2d", [True, False])
def test_residuals_match_oracle(dev, is_2d):
    from deepi2p_amd import ops
    f, rng = _frame(1, 3000)
    pts = f["pc"].astype(np.float64)
    lab = f["labels"].copy()
    lab[::17] = 2                                                   # labels outside {0,1} are skipped
    params = np.array([0.3, 1.0, 0.05, -2.0]) if is_2d else np.array([0.02, 0.3, -0.01, 1.0, 0.05, -2.0])
    r, J, cost = flm.residuals_and_jacobian(pts, lab, f["K"], H, W, is_2d, params)
    s = np.add.reduceat(r * r, np.r_[0, np.cumsum(np.where(lab[(lab == 0) | (lab == 1)] == 1, 3, 1))[:-1]])
    corrected = r * np.repeat(np.sqrt(1 / (1 + s)), np.where(lab[(lab == 0) | (lab == 1)] == 1, 3, 1))
    res, counts, c = ops.solver_residuals(torch.from_numpy(pts).to(dev).unsqueeze(0), torch.from_numpy(lab).to(dev).unsqueeze(0),
                                          torch.from_numpy(f["K"]).to(dev).unsqueeze(0),
                                          torch.from_numpy(params).to(dev).unsqueeze(0), H, W, is_2d)
    n = int(counts.item())
    assert n == r.shape[0]
    np.testing.assert_allclose(res[0, :n].cpu().numpy(), corrected, rtol=1e-9, atol=1e-9)
    assert abs(float(c.item()) - cost) <= 1e-9 * abs(cost)


def test_initial_guess_matches_oracle(dev):
    from deepi2p_amd import registration
    f, _ = _frame(2, 5000, flip=0.0)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    P0, y0, pcf, labf = flm.get_initial_guess(pts, lab)
    P1, y1, pcf1, labf1 = registration.get_initial_guess(pts, lab)
    assert abs(y0 - y1) < 1e-12
    np.testing.assert_array_equal(pcf, pcf1)
    np.testing.assert_array_equal(labf, labf1)
    np.testing.assert_allclose(P0, P1, atol=1e-12)


# hypotheses (by index) whose HIP trajectory ends in another local minimum than the oracle's on these fixed seeds -- the two
# implementations differ by an ulp per residual and by summation order, and the objective jumps where a point crosses a frustum plane.
# Measured on MI355X (tools/solver_oracle_agreement.py prints the same lists); everything not listed must agree.
KNOWN_FORKS = {
}


def _check_forks(key, ok):
    bad = sorted(np.nonzero(~ok)[0].tolist())
    assert bad == KNOWN_FORKS.get(key, []), "hypotheses that disagree with the oracle on case %r: %s (expected %s)" % (key, bad, KNOWN_FORKS.get(key, []))


def _agreement(po, pg, is_2d):
    toff = 1 if is_2d else 3
    dt = np.linalg.norm(po[:, toff:] - pg[:, toff:], axis=1)
    dr = np.linalg.norm(po[:, :toff] - pg[:, :toff], axis=1)
    return (dt <= 1e-3) & (dr <= 1e-3)


@pytest.mark.parametrize("is_2d,N,R", [(True, 4096, 24), (False, 2048, 12)])
def test_solver_matches_oracle_per_hypothesis(dev, is_2d, N, R):
    from deepi2p_amd import registration
    f, rng = _frame(3 if is_2d else 4, N)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    _, y0, pcf, labf = flm.get_initial_guess(pts, lab)
    ys, Ts = flm.draw_restarts(rng, R, y0, 10 * math.pi / 180, 10)
    Po, co, it_o, term_o, par_o = flm.solve_restarts(pcf, labf, f["K"], ys, Ts, H, W, LB, UB, 500, is_2d, nthreads=8)
    Pg, cg, best, par_g, it_g = registration.solvePGivenK_batched(pcf, labf, f["K"], ys, Ts, H, W, LB, UB, 500, is_2d, return_all=True)
    ok = _agreement(par_o, par_g, is_2d)
    _check_forks(("per_hypothesis", is_2d, N, R), ok)
    np.testing.assert_allclose(cg[ok], co[ok], rtol=1e-6)
    assert abs(cg.min() - co.min()) <= 1e-6 * co.min()
    np.testing.assert_allclose(Pg[ok], Po[ok], atol=2e-3)
    np.testing.assert_array_equal(it_g[ok], it_o[ok])
    # bounds respected
    toff = 1 if is_2d else 3
    assert np.all(par_g[:, toff:] >= np.array(LB) - 1e-12) and np.all(par_g[:, toff:] <= np.array(UB) + 1e-12)


@pytest.mark.parametrize("is_2d,use_f32", [(True, True), (True, False), (False, False)])
def test_cluster_culling_is_exact(dev, is_2d, use_f32):
    """The frustum-plane test of whole 64-point clusters must never change a result: with DI2P_SOLVER_NOCULL=1 every
    cluster is classified point by point, and params / cost / iteration and sweep counts have to be BIT-identical.
    The frame mixes a synthetic scan with adversarial points: exactly on the frustum planes of the initial poses,
    on the camera plane, duplicated, far away, and labels outside {0,1}."""
    import os
    from deepi2p_amd import ops
    f, rng = _frame(21, 6000)
    pts, lab = f["pc"].astype(np.float64).copy(), f["labels"].astype(np.int32).copy()
    K = f["K"]
    n_adv = 600
    z = rng.uniform(0.5, 60, n_adv)
    side = rng.integers(0, 5, n_adv)
    x = np.where(side == 0, (0 - K[0, 2]) * z / K[0, 0], np.where(side == 1, (W - 1 - K[0, 2]) * z / K[0, 0], rng.uniform(-40, 40, n_adv)))
    y = np.where(side == 2, (0 - K[1, 2]) * z / K[1, 1], np.where(side == 3, (H - 1 - K[1, 2]) * z / K[1, 1], rng.uniform(-2, 2, n_adv)))
    z = np.where(side == 4, 0.0, z)
    pts[:, :n_adv] = np.stack((x, y, z))
    pts[:, n_adv:n_adv + 50] = pts[:, n_adv + 50:n_adv + 100]           # duplicates
    pts[:, n_adv + 100:n_adv + 110] *= 1e3                              # far outliers (huge cluster radii)
    lab[:n_adv] = rng.integers(0, 2, n_adv)
    lab[::37] = 2
    if use_f32:
        pts = pts.astype(np.float32).astype(np.float64)
    R = 16
    ys = np.concatenate(([0.0, 0.0], rng.normal(0, 0.3, R - 2)))
    Ts = np.concatenate((np.zeros((2, 3)), rng.uniform(-3, 3, (R - 2, 3))))      # hypotheses 0/1: points sit ON the planes
    Ts[:, 1] = np.clip(Ts[:, 1], -0.1, 0.1)
    tp = torch.from_numpy(pts.astype(np.float32) if use_f32 else pts).to(dev).unsqueeze(0)
    args = (tp, torch.from_numpy(lab).to(dev).unsqueeze(0), torch.from_numpy(K).to(dev).view(1, 3, 3),
            torch.from_numpy(ys).to(dev).view(1, R), torch.from_numpy(Ts).to(dev).view(1, R, 3), H, W, LB, UB, 60, is_2d)

    def run():
        sweeps = torch.zeros((1, R), dtype=torch.int32, device=dev)
        params, cost, iters = ops.solve_batched(*args, sweeps=sweeps)
        return params.cpu().numpy(), cost.cpu().numpy(), iters.cpu().numpy(), sweeps.cpu().numpy()
    a = run()
    from deepi2p_amd import _lib
    with _lib.option("solver_nocull", 1):
        b = run()
    for u, v in zip(a, b):
        assert u.tobytes() == v.tobytes()        # bit-identical, NaN-safe
    # the fp32 pre-filter of the per-point classification certifies or defers to the exact test: same bits without it,
    # also with every cluster forced through the per-point path (the planted on-plane points must all be deferred)
    with _lib.option("solver_noprefilter", 1):
        c = run()
        with _lib.option("solver_nocull", 1):
            d = run()
    for u, v, w in zip(a, c, d):
        assert u.tobytes() == v.tobytes() == w.tobytes()
    # the classification cache re-uses a cluster's recorded active mask only while the iterate's motion is provably below the
    # recorded slack: same bits with it switched off (a was computed with it on)
    with _lib.option("solver_nocache", 1):
        e = run()
    for u, v in zip(a, e):
        assert u.tobytes() == v.tobytes()
    assert np.isfinite(a[1]).sum() >= R - 4      # the planted on-plane points may fail hypotheses 0/1 only


def test_two_tier_launch_agrees_with_single_launch(dev):
    """solver_tier_sweeps > 0: hypotheses that need more sweeps than the budget are parked by the first launch and resumed by a wide
    (12-wave) one, which starts from an EMPTY classification cache and combines its wave partials in another order: same iteration and
    sweep counts, parameters and costs to rounding (not bit-identical)."""
    from deepi2p_amd import ops, _lib
    f, rng = _frame(5, 6000)
    R = 24
    _, y0, pcf, labf = flm.get_initial_guess(f["pc"].astype(np.float64), f["labels"])
    ys, Ts = flm.draw_restarts(rng, R, y0, 10 * math.pi / 180, 10)
    args = (torch.from_numpy(np.ascontiguousarray(pcf)).to(dev).unsqueeze(0), torch.from_numpy(np.ascontiguousarray(labf.astype(np.int32))).to(dev).unsqueeze(0),
            torch.from_numpy(np.ascontiguousarray(f["K"])).to(dev).view(1, 3, 3), torch.from_numpy(np.ascontiguousarray(ys)).to(dev).view(1, R),
            torch.from_numpy(np.ascontiguousarray(Ts)).to(dev).view(1, R, 3),
            H, W, LB, UB, 500, True)

    def run():
        sweeps = torch.zeros((1, R), dtype=torch.int32, device=dev)
        params, cost, iters = ops.solve_batched(*args, sweeps=sweeps)
        return params.cpu().numpy()[0], cost.cpu().numpy()[0], iters.cpu().numpy()[0], sweeps.cpu().numpy()[0]
    a = run()
    assert a[3].max() > 16                      # some hypotheses will be parked below
    for tier in (8, 16):
        with _lib.option("solver_tier_sweeps", tier):
            b = run()
        ok = _agreement(a[0], b[0], True)
        # a parked hypothesis may end in a neighbouring minimum (the wide tier adds its wave partials in another order): the ones that do
        # are ENUMERATED in KNOWN_FORKS; one more is a failure
        _check_forks(("two_tier", tier), ok)
        np.testing.assert_allclose(b[1][ok], a[1][ok], rtol=1e-6)
        assert abs(b[1].min() - a[1].min()) <= 1e-6 * a[1].min()


@pytest.mark.parametrize("seed,is_2d,N", [(101, True, 4096), (102, True, 20480), (103, False, 3000), (104, False, 9000)])
def test_classification_cache_is_bit_identical_in_both_tiers(dev, seed, is_2d, N):
    """Round-4 advisor finding: the classification cache (recorded masks re-used while the iterate's motion stays below the recorded slack;
    constants 1.0001 mu + 1e-6, the 0.99999 re-record factor, RING / 2 refresh) was compared with `solver_nocache` on a few fixed seeds of the
    2-D path in the narrow tier only.  Here: random scenes, the 2-D and the 3-D path, and BOTH tiers (the wide tier resumes parked hypotheses
    from an empty cache) -- each tier with the cache on against the SAME tier with it off, bit for bit, and the cache must actually have been
    hit (profile counters of the PROFILE instantiation)."""
    from deepi2p_amd import ops, _lib
    rng = np.random.default_rng(seed)
    f = synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.05, with_image=False)
    _, y0, pcf, labf = flm.get_initial_guess(f["pc"].astype(np.float64), f["labels"])
    R = 16
    ys, Ts = flm.draw_restarts(rng, R, y0, 10 * math.pi / 180, 10)
    pts = torch.from_numpy(np.ascontiguousarray(pcf.astype(np.float32))).to(dev).unsqueeze(0)
    args = (pts, torch.from_numpy(np.ascontiguousarray(labf.astype(np.int32))).to(dev).unsqueeze(0),
            torch.from_numpy(np.ascontiguousarray(f["K"])).to(dev).view(1, 3, 3), torch.from_numpy(np.ascontiguousarray(ys)).to(dev).view(1, R),
            torch.from_numpy(np.ascontiguousarray(Ts)).to(dev).view(1, R, 3), H, W, LB, UB, 500, is_2d)

    def run(profile=False):
        sweeps = torch.zeros((1, R), dtype=torch.int32, device=dev)
        prof = torch.zeros((R, 28), dtype=torch.int64, device=dev) if profile else None      # 28 words per hypothesis since library version 6
        if profile:
            _lib.load().di2p_solver_set_profile_buffer(prof.data_ptr())
        try:
            params, cost, iters = ops.solve_batched(*args, sweeps=sweeps)
            torch.cuda.synchronize()
        finally:
            if profile:
                _lib.load().di2p_solver_set_profile_buffer(None)
        out = (params.cpu().numpy(), cost.cpu().numpy(), iters.cpu().numpy(), sweeps.cpu().numpy())
        return out + ((prof.cpu().numpy(),) if profile else ())
    for tier in (0, 8):
        with _lib.option("solver_tier_sweeps", tier):
            on = run()
            with _lib.option("solver_nocache", 1):
                off = run()
            for u, v in zip(on, off):
                assert u.tobytes() == v.tobytes(), (tier, is_2d)
            prof = run(profile=True)[4]
        # columns 16 / 17: clusters whose recorded mask was re-used / whose guard walk was skipped (wave 0, summed over the sweeps)
        assert prof[:, 16].sum() > 0 and prof[:, 17].sum() > 0, (tier, prof[:, 16].sum(), prof[:, 17].sum())
        assert on[3].max() > 8                      # tier 8 did park hypotheses


def test_solvePGivenK_drop_in(dev):
    """Reference call signature and return triple (registration.cpp:190-206)."""
    from deepi2p_amd import FrustumRegistration
    f, rng = _frame(5, 2048)
    pts, lab = f["pc"].astype(np.float64), f["labels"].astype(np.int64)      # int64 labels are accepted (pybind casts)
    P, cost, res = FrustumRegistration.solvePGivenK(points=pts, labels=lab, K=f["K"], init_y_angle=f["yaw_gt"] + 0.1,
                                                     init_T=np.array([0.0, 0.0, 1.0]), H=H, W=W,
                                                     t_xyz_lower_bound=LB, t_xyz_upper_bound=UB, max_iter=500,
                                                     is_debug=False, is_2d=True)
    Po, co, ro = flm.solvePGivenK(pts, lab, f["K"], f["yaw_gt"] + 0.1, np.array([0.0, 0.0, 1.0]), H, W, LB, UB, 500, False, True)
    assert P.shape == (4, 4) and res.shape == (3 * int((lab == 1).sum()) + int((lab == 0).sum()),)
    # cost definition: 1/2 sum_blocks log(1+s); with corrected residuals r~ = r/sqrt(1+s): log(1+s) = -log(1-|r~|^2)
    sizes = np.where(lab[(lab == 0) | (lab == 1)] == 1, 3, 1)
    st = np.add.reduceat(res * res, np.r_[0, np.cumsum(sizes)[:-1]])
    assert abs(cost - 0.5 * np.sum(-np.log1p(-st))) <= 1e-9 * cost
    assert abs(cost - co) <= 1e-6 * co
    np.testing.assert_allclose(P, Po, atol=2e-3)
    with pytest.raises(IndexError):
        FrustumRegistration.solvePGivenK(pts, lab, f["K"], 0.0, np.zeros(3), H, W, [0, 0], UB, 10, False, True)


def test_pipeline_recovers_pose_full_size(dev):
    """BASELINE config-2 solver shape: N=20480, R=60, exact labels -> every frame within the reference's
    success rule (RTE < 2 m, RRE < 5 deg; registration_result_analysis.py:37-47) and cost ~ 0."""
    from deepi2p_amd.registration import RegistrationPipeline
    F = 4
    rng = np.random.default_rng(7)
    frames = [synthetic.make_frame(rng, N=20480, H=H, W=W, flip=0.0, with_image=False) for _ in range(F)]
    pc = torch.from_numpy(np.stack([f["pc"] for f in frames])).to(dev)
    lab = torch.from_numpy(np.stack([f["labels"] for f in frames])).to(dev)
    K = torch.from_numpy(np.stack([f["K"] for f in frames])).to(dev)
    pipe = RegistrationPipeline(H, W, R=60, seed=1)
    out = pipe(pc, lab, K, pipe.draw(F, dev))
    P = out["P"].cpu().numpy()
    for i, f in enumerate(frames):
        t, r = flm.get_P_diff(P[i], f["P_gt"])
        assert t < 2.0 and r < 5.0, (i, t, r)
    assert torch.all(out["iters"] <= 500) and torch.all(out["best"] >= 0)
    # argmin rule: best == lowest index among minimal costs
    costs = out["costs"].cpu().numpy()
    np.testing.assert_array_equal(out["best"].cpu().numpy(), costs.argmin(axis=1))


def test_pipeline_no_inside_points(dev):
    """registration_lsq.py:329-332: no point predicted inside -> identity pose, cost 1e4."""
    from deepi2p_amd.registration import RegistrationPipeline
    f, _ = _frame(9, 1024)
    pc = torch.from_numpy(f["pc"]).to(dev).unsqueeze(0)
    lab = torch.zeros(1, 1024, dtype=torch.int32, device=dev)
    K = torch.from_numpy(f["K"]).to(dev).unsqueeze(0)
    pipe = RegistrationPipeline(H, W, R=8, seed=0)
    out = pipe(pc, lab, K, pipe.draw(1, dev))
    assert float(out["cost"][0]) == 1e4 and int(out["best"][0]) == -1
    np.testing.assert_array_equal(out["P"][0].cpu().numpy(), np.eye(4))


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 130, 777])
def test_small_and_ragged_frames(dev, N):
    """Frame preparation (bitonic sort padded to a power of two >= 64, partial clusters, empty label blocks) and the
    cluster walk at sizes around the 64-point cluster: same costs as the oracle on identical hypotheses."""
    from deepi2p_amd import registration
    f, rng = _frame(100 + N, max(N, 8))
    pts, lab = f["pc"][:, :N].astype(np.float64), f["labels"][:N].copy()
    if N >= 63:
        lab[::9] = 2                                   # ignored labels
    if N == 64:
        lab[:] = 0                                     # no label-1 block at all
    R = 6
    ys = rng.normal(f["yaw_gt"], 0.2, R)
    Ts = np.stack((np.zeros(R), np.zeros(R), rng.uniform(-5, 5, R)), axis=1)
    Po, co, it_o, term_o, par_o = flm.solve_restarts(pts, lab, f["K"], ys, Ts, H, W, LB, UB, 100, True, nthreads=2)
    Pg, cg, best, par_g, it_g = registration.solvePGivenK_batched(pts, lab, f["K"], ys, Ts, H, W, LB, UB, 100, True, return_all=True)
    assert np.all(np.isfinite(cg)) and np.all(cg >= 0)
    ok = _agreement(par_o, par_g, True)
    # tiny problems are rank-deficient / flat: require agreement of the COST wherever the iterates agree, and of the best cost
    np.testing.assert_allclose(cg[ok], co[ok], rtol=1e-6, atol=1e-12)
    assert abs(cg.min() - co.min()) <= 1e-6 * max(co.min(), 1e-9) + 1e-12
    _check_forks(("ragged", N), ok)


@pytest.mark.parametrize("seed,is_2d,N,HW,flip", [(31, True, 8192, (384, 640), 0.1), (32, False, 3000, (64, 128), 0.02),
                                                  (33, True, 20480, (160, 512), 0.05)])
def test_solver_matches_oracle_other_cameras(dev, seed, is_2d, N, HW, flip):
    """Same parity statement as above on other image sizes / intrinsics / label noise (the cluster shortcut and the
    cost-only line-search sweeps must not move any iterate: EVERY hypothesis agrees -- KNOWN_FORKS is empty --, costs and the best cost
    to 1e-6, equal iteration counts)."""
    from deepi2p_amd import registration
    h, w = HW
    rng = np.random.default_rng(seed)
    f = synthetic.make_frame(rng, N=N, H=h, W=w, flip=flip, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    _, y0, pcf, labf = flm.get_initial_guess(pts, lab)
    R = 12
    ys, Ts = flm.draw_restarts(rng, R, y0, 10 * math.pi / 180, 10)
    Po, co, it_o, term_o, par_o = flm.solve_restarts(pcf, labf, f["K"], ys, Ts, h, w, LB, UB, 500, is_2d, nthreads=8)
    Pg, cg, best, par_g, it_g = registration.solvePGivenK_batched(pcf, labf, f["K"], ys, Ts, h, w, LB, UB, 500, is_2d, return_all=True)
    ok = _agreement(par_o, par_g, is_2d)
    _check_forks(("other_cameras", seed), ok)
    np.testing.assert_allclose(cg[ok], co[ok], rtol=1e-6)
    assert abs(cg.min() - co.min()) <= 1e-6 * co.min()
    np.testing.assert_array_equal(it_g[ok], it_o[ok])            # same number of LM iterations where the iterates agree


@pytest.mark.parametrize("scene", ["scan", "patchy", "collapsed", "duplicates", "bucket_under_4096", "bucket_over_4096"])
@pytest.mark.parametrize("N", [777, 20480])
def test_frame_preparation_sort_paths_agree(dev, scene, N):
    """Frame preparation sorts the records by (label, Hilbert cell, index).  The shipped path is a counting sort into (label, cell) buckets
    followed by a ranking of every key inside its bucket, ONE THREAD PER KEY -- round 6: as five launches of several workgroups per frame
    (solver.hip prep_*_kernel), round 4: one workgroup per frame (prepare_kernel step 3a, knob solver_prep_single); a frame with a bucket above
    4096 keys goes to the bitonic network instead.  Either way the order must be the SAME as the bitonic network's (knob solver_prep_bitonic):
    params, costs, iteration and sweep counts bit-identical.  Scenes: a scan (small buckets), patchy (dense patches: large buckets), collapsed
    (a far outlier stretches the grid so that everything shares a few cells: the fallback), duplicates (equal coordinates, unique keys by
    index), and one bucket just under / just over the 4096-key limit (a 1 m patch of equally labelled points inside one 5 m cell)."""
    from deepi2p_amd import _lib, ops
    f, rng = _frame(900 + N, N)
    pts, lab = f["pc"].astype(np.float32).copy(), f["labels"].astype(np.int32).copy()
    if scene == "patchy":
        m = N // 3
        pts[0, :m] = 10.0 + rng.uniform(0, 4.0, m); pts[2, :m] = 20.0 + rng.uniform(0, 4.0, m)          # one 4 m patch holds a third of the points
    elif scene == "collapsed":
        pts[:, 0] = [4e4, 0.0, 4e4]
    elif scene == "duplicates":
        pts[:, N // 2:] = pts[:, :N - N // 2]
    elif scene.startswith("bucket_"):
        m = min(N - 1, 4040 if scene == "bucket_under_4096" else 4130)
        pts[0, :m] = 11.0 + rng.uniform(0, 1.0, m); pts[2, :m] = 21.0 + rng.uniform(0, 1.0, m)
        lab[:m] = 1
    R = 6
    ys = rng.normal(f["yaw_gt"], 0.2, R)
    Ts = np.stack((np.zeros(R), np.zeros(R), rng.uniform(-5, 5, R)), axis=1)
    args = (torch.from_numpy(pts).to(dev).unsqueeze(0), torch.from_numpy(lab).to(dev).unsqueeze(0), torch.from_numpy(f["K"]).to(dev).view(1, 3, 3),
            torch.from_numpy(ys).to(dev).view(1, R), torch.from_numpy(Ts).to(dev).view(1, R, 3), H, W, LB, UB, 40, True)

    def run():
        sweeps = torch.zeros((1, R), dtype=torch.int32, device=dev)
        params, cost, iters = ops.solve_batched(*args, sweeps=sweeps)
        return [t.cpu().numpy().tobytes() for t in (params, cost, iters, sweeps)]
    a = run()                                    # round 6: five multi-workgroup launches (+ the single-workgroup kernel for flagged frames)
    with _lib.option("solver_prep_bitonic", 1):
        b = run()
    assert a == b
    with _lib.option("solver_prep_single", 1):   # the round-4 single-workgroup kernel (counting sort + ranking)
        c = run()
    assert a == c
