"""PnP back end of BASELINE config 3 (evaluation/registration_pnp.py:95-148).  PARITY UNPINNED w.r.t. OpenCV (absent;
internal RNG): the oracle (oracle/pnp_np.py) is pinned by pose recovery, the HIP kernel is compared with the oracle on
identical RANSAC draws.  Tolerances: exact correspondences -> pose within 1e-6; HIP vs oracle -> per-hypothesis inlier
counts equal for >= 99 % of the hypotheses (a count can differ when a reprojection error sits on the 0.6 px threshold),
final pose within 1e-6 m / 1e-6 rad when both select the same model."""
import numpy as np
import pytest

from deepi2p_amd import synthetic
from oracle import pnp_np

H, W, SCALE = 160, 512, 32


def _frame(seed, N=8192, exact=True):
    rng = np.random.default_rng(seed)
    f = synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.0, with_image=False)
    K_fine = f["K"] / SCALE
    K_fine[2, 2] = 1.0
    cam = f["P_gt"][:3, :3] @ f["pc"].astype(np.float64) + f["P_gt"][:3, 3:4]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K_fine[0, 0] * cam[0] / cam[2] + K_fine[0, 2]
        v = K_fine[1, 1] * cam[1] / cam[2] + K_fine[1, 2]
    coarse = f["labels_gt"].copy()
    Wf = W // SCALE
    fine = (np.floor(u).astype(np.int64) + np.floor(v).astype(np.int64) * Wf).astype(np.int32)
    fine[coarse == 0] = 0
    pixels = np.stack((u, v)).astype(np.float32)
    pixels[:, coarse == 0] = 0
    samples = rng.integers(0, 2 ** 30, size=(200, 6)).astype(np.int32)
    return f, K_fine, coarse.astype(np.int32), fine, pixels, samples, Wf


def _pose_err(P, P_gt):
    from scipy.spatial.transform import Rotation
    D = np.linalg.inv(P) @ P_gt
    return np.linalg.norm(D[:3, 3]), np.linalg.norm(Rotation.from_matrix(D[:3, :3]).as_rotvec())


def test_oracle_recovers_pose_from_exact_correspondences():
    f, K_fine, coarse, fine, pixels, samples, Wf = _frame(1)
    P, ratio, nin, cnt, best, counts = pnp_np.pnp_ransac(f["pc"], coarse, fine, K_fine, Wf, samples, pixels=pixels)
    t, r = _pose_err(P, f["P_gt"])
    assert t < 1e-3 and r < 1e-4 and ratio < 0.01 and cnt == int(coarse.sum())     # f32 pixels: ~1e-5 px noise


@pytest.mark.parametrize("seed,outliers", [(2, 0.0), (3, 0.0), (5, 0.2), (6, 0.2)])
def test_oracle_cell_quantised_correspondences_and_outliers(seed, outliers):
    """The reference's own front end: the observation is the top-left corner of the predicted 32x32 cell of a 16x5
    cell image, inlier threshold 0.6 cells.  That convention carries a systematic half-cell bias (~2.6 deg per axis at
    fx = 359 px), so what can be asserted is: translation inside the reference's 2 m success bound, summed Euler
    error below 8 deg, most correspondences explained -- with and without 20 % wrong cells."""
    f, K_fine, coarse, fine, pixels, samples, Wf = _frame(seed)
    rng = np.random.default_rng(0)
    bad = (rng.random(fine.shape) < outliers) & (coarse == 1)
    fine = np.where(bad, rng.integers(0, Wf * (H // SCALE), fine.shape), fine).astype(np.int32)
    rs = np.random.default_rng(1).integers(0, 2 ** 30, size=(500, 6)).astype(np.int32)
    P, ratio, nin, cnt, best, counts = pnp_np.pnp_ransac(f["pc"], coarse, fine, K_fine, Wf, rs)
    from oracle import frustum_lm as flm
    t, r = flm.get_P_diff(P, f["P_gt"])
    assert t < 2.0 and r < 8.0 and ratio < 0.35 + outliers


def test_oracle_rejections():
    f, K_fine, coarse, fine, pixels, samples, Wf = _frame(3, N=512)
    none = np.zeros_like(coarse)
    P, ratio, *_ = pnp_np.pnp_ransac(f["pc"], none, fine, K_fine, Wf, samples)
    assert np.array_equal(P, np.eye(4)) and ratio == 1.0                            # < 6 correspondences
    far = f["pc"].copy()
    far[2] += 100.0                                                                 # solution has |t| >> 14.14 -> rejected (:134)
    Pg = f["P_gt"].copy()
    Pg[2, 3] -= 100.0
    P2, ratio2, *_ = pnp_np.pnp_ransac(far, coarse, fine, K_fine, Wf, samples, pixels=pixels)
    assert np.array_equal(P2, np.eye(4)) and ratio2 == 1.0


@pytest.mark.gpu
def test_hip_matches_oracle_on_identical_draws(dev):
    import torch
    from deepi2p_amd import registration_pnp as rp
    frames = [_frame(10 + i) for i in range(3)]
    pc = torch.from_numpy(np.stack([fr[0]["pc"] for fr in frames])).to(dev)
    K = torch.from_numpy(np.stack([fr[1] for fr in frames])).to(dev)
    co = torch.from_numpy(np.stack([fr[2] for fr in frames])).to(dev)
    fi = torch.from_numpy(np.stack([fr[3] for fr in frames])).to(dev)
    px = torch.from_numpy(np.stack([fr[4] for fr in frames])).to(dev)
    sm = torch.from_numpy(np.stack([fr[5] for fr in frames])).to(dev)
    Wf = frames[0][6]
    for use_pixels in (True, False):
        out = rp.pnp_ransac(pc, co, fi, K, Wf, sm, pixels=px if use_pixels else None, method="dlt_lo")      # vs the DLT oracle
        for i, fr in enumerate(frames):
            P, ratio, nin, cnt, best, counts = pnp_np.pnp_ransac(fr[0]["pc"], fr[2], fr[3], fr[1], Wf, fr[5],
                                                                 pixels=fr[4] if use_pixels else None)
            assert int(out["n_corr"][i]) == cnt
            Pg = out["P"][i].cpu().numpy()
            if int(out["best"][i]) == best:
                t, r = _pose_err(Pg, P)
                assert t < 1e-6 and r < 1e-6, (use_pixels, i, t, r)
                assert abs(int(out["n_inliers"][i]) - nin) <= max(2, nin // 200)
            t, r = _pose_err(Pg, fr[0]["P_gt"])
            if use_pixels:
                assert t < 1e-3 and r < 1e-4


@pytest.mark.gpu
def test_hip_solve_PnP_drop_in(dev):
    """Reference signature: solve_PnP(pc, coarse, fine, K, H, W, 1/32, 500, method) -> (P, outlier_ratio)."""
    from deepi2p_amd import registration_pnp as rp
    f, K_fine, coarse, fine, pixels, samples, Wf = _frame(20)
    P, ratio = rp.solve_PnP(f["pc"], coarse, fine, f["K"], H, W, 1.0 / SCALE, 500, method=None, rng=np.random.default_rng(0))
    from oracle import frustum_lm as flm
    t, r = flm.get_P_diff(P, f["P_gt"])
    assert P.shape == (4, 4) and t < 2.0 and r < 8.0 and 0.0 <= ratio < 1.0     # half-cell bias, see the oracle test
    P0, r0 = rp.solve_PnP(f["pc"], np.zeros_like(coarse), fine, f["K"], H, W, 1.0 / SCALE, 50)
    assert np.array_equal(P0, np.eye(4)) and r0 == 1
    np.testing.assert_allclose(rp.camera_matrix_scaling(f["K"], 1 / 32)[:2], f["K"][:2] / 32)


# ---------------------------------------------------------------------------------------------- EPnP (the reference's estimator)
def _exact_set(rng, n, K):
    from scipy.spatial.transform import Rotation
    R = Rotation.from_rotvec(rng.normal(0, 0.5, 3)).as_matrix()
    t = np.array([rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-2, 2)])
    pc = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 3, n), rng.uniform(4, 40, n)])
    X = R.T @ (pc - t[:, None])
    uv = np.stack([K[0, 0] * pc[0] / pc[2] + K[0, 2], K[1, 1] * pc[1] / pc[2] + K[1, 2]])
    P = np.eye(4)
    P[:3, :3], P[:3, 3] = R, t
    return X, uv, P


def test_epnp_oracle_recovers_pose_exact():
    """The EPnP restatement on exact correspondences: n = 5 (RANSAC's minimal sample), 6, 50, 2000 -> pose to rounding."""
    from oracle import epnp_np
    rng = np.random.default_rng(0)
    K = np.array([[350.0, 0, 256], [0, 350.0, 80], [0, 0, 1]])
    for n in (5, 6, 50, 2000):
        for _ in range(10):
            X, uv, P = _exact_set(rng, n, K)
            R, t, err = epnp_np.epnp(X, uv, K)
            Pe = np.eye(4)
            Pe[:3, :3], Pe[:3, 3] = R, t
            dt, dr = _pose_err(Pe, P)
            assert dt < 1e-8 and dr < 1e-9 and err < 1e-8, (n, dt, dr, err)


def test_epnp_ransac_oracle_with_outliers():
    from oracle import epnp_np
    rng = np.random.default_rng(1)
    K = np.array([[350.0, 0, 256], [0, 350.0, 80], [0, 0, 1]]) / SCALE
    K[2, 2] = 1.0
    X, uv, P = _exact_set(rng, 800, K)
    bad = rng.random(800) < 0.3
    uv[:, bad] = rng.uniform(0, 16, (2, int(bad.sum())))
    samples = rng.integers(0, 2 ** 30, size=(300, 6)).astype(np.int32)
    R, t, mask, best, counts = epnp_np.epnp_ransac(X, uv, K, samples, reproj_err=0.6)
    Pe = np.eye(4)
    Pe[:3, :3], Pe[:3, 3] = R, t
    dt, dr = _pose_err(Pe, P)
    assert dt < 1e-6 and dr < 1e-7 and mask.sum() >= (~bad).sum() and best >= 0
    # fewer than four correspondences: nothing to estimate
    assert epnp_np.epnp_ransac(X[:, :3], uv[:, :3], K, samples)[0] is None


@pytest.mark.gpu
def test_hip_epnp_ransac_matches_oracle(dev):
    """HIP EPnP RANSAC vs the restatement on identical draws: per-hypothesis inlier counts (>= 99 % equal), the same winner, the
    re-fitted pose to 1e-6; exact correspondences with 30 % gross outliers -> ground truth recovered."""
    import torch
    from deepi2p_amd import registration_pnp as rp
    from oracle import epnp_np
    rng = np.random.default_rng(4)
    F, N = 3, 4096
    K = np.array([[350.0, 0, 256], [0, 350.0, 80], [0, 0, 1]]) / SCALE
    K[2, 2] = 1.0
    pcs, pxs, cos, Ps = [], [], [], []
    for f in range(F):
        X, uv, P = _exact_set(rng, N, K)
        co = (rng.random(N) < 0.25).astype(np.int32)
        bad = (rng.random(N) < 0.3) & (co == 1)
        uv[:, bad] = rng.uniform(0, 16, (2, int(bad.sum())))
        pcs.append(X.astype(np.float32)); pxs.append(uv.astype(np.float32)); cos.append(co); Ps.append(P)
    samples = rng.integers(0, 2 ** 30, size=(F, 200, 6)).astype(np.int32)
    pc, px, co = [torch.from_numpy(np.stack(a)).to(dev) for a in (pcs, pxs, cos)]
    Kt = torch.from_numpy(np.stack([K] * F)).to(dev)
    fi = torch.zeros((F, N), dtype=torch.int32, device=dev)
    out = rp.pnp_ransac(pc, co, fi, Kt, 16, torch.from_numpy(samples).to(dev), pixels=px, method="epnp")
    out2 = rp.pnp_ransac(pc, co, fi, Kt, 16, torch.from_numpy(samples).to(dev), pixels=px, method="epnp")
    assert torch.equal(out["P"], out2["P"])
    for f in range(F):
        m = cos[f] == 1
        X, uv = pcs[f][:, m].astype(np.float64), pxs[f][:, m].astype(np.float64)
        R, t, mask, best, counts = epnp_np.epnp_ransac(X, uv, K, samples[f], reproj_err=0.6)
        assert int(out["n_corr"][f]) == X.shape[1]
        assert int(out["best"][f]) == best and abs(int(out["n_inliers"][f]) - int(mask.sum())) <= 2
        Po = np.eye(4)
        Po[:3, :3], Po[:3, 3] = R, t
        Pg = out["P"][f].cpu().numpy()
        dt, dr = _pose_err(Pg, Po)
        # identical inlier sets -> the re-fit agrees to rounding; a correspondence sitting on the 0.6 threshold may flip (f32
        # observations), and an EPnP re-fit without iterative refinement then moves by millimetres
        same = int(out["n_inliers"][f]) == int(mask.sum())
        assert (dt < 1e-6 and dr < 1e-6) if same else (dt < 2e-2 and dr < 2e-3), (f, same, dt, dr)
        dt, dr = _pose_err(Pg, Ps[f])
        assert dt < 5e-2 and dr < 5e-3                                # f32 observations, EPnP without an iterative refinement


@pytest.mark.gpu
def test_hip_epnp_minimal_counts(dev):
    """Frames with 5 and with 4 correspondences are solved (the reference accepts >= 4, registration_pnp.py:123; the DLT variant
    needs 6); 3 correspondences -> identity, outlier ratio 1."""
    import torch
    from deepi2p_amd import registration_pnp as rp
    rng = np.random.default_rng(9)
    K = np.array([[350.0, 0, 256], [0, 350.0, 80], [0, 0, 1]]) / SCALE
    K[2, 2] = 1.0
    N = 64
    ok5 = ok4 = 0
    trials = 12
    for trial in range(trials):
        X, uv, P = _exact_set(rng, N, K)
        for cnt in (5, 4, 3):
            co = np.zeros(N, np.int32)
            co[:cnt] = 1
            samples = rng.integers(0, 2 ** 30, size=(1, 64, 6)).astype(np.int32)
            out = rp.pnp_ransac(torch.from_numpy(X.astype(np.float32)).to(dev).unsqueeze(0), torch.from_numpy(co).to(dev).unsqueeze(0),
                                torch.zeros((1, N), dtype=torch.int32, device=dev), torch.from_numpy(K).to(dev).unsqueeze(0), 16,
                                torch.from_numpy(samples).to(dev), pixels=torch.from_numpy(uv.astype(np.float32)).to(dev).unsqueeze(0),
                                method="epnp")
            Pg = out["P"][0].cpu().numpy()
            if cnt == 3:
                assert np.array_equal(Pg, np.eye(4)) and float(out["outlier_ratio"][0]) == 1.0
                continue
            dt, dr = _pose_err(Pg, P)
            good = dt < 1e-2 and dr < 1e-3
            ok5 += good and cnt == 5
            ok4 += good and cnt == 4
    assert ok5 >= trials - 2, ok5               # five points determine the pose (f32 observations: a near-degenerate sample may miss 1 cm)
    # four points: EPnP + Gauss-Newton polish (OpenCV uses P3P here).  The problem has up to four solutions; with the 0.6-cell
    # threshold several of them keep all four points as inliers and the lowest sample id wins, so only a majority is required
    assert ok4 >= trials * 0.5, ok4
