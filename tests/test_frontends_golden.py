"""Front ends of the path pinned by the REFERENCE ITSELF: its own method / function / script bodies were extracted with ``ast`` and run
against stub collaborators by tests/golden/make_golden.py (make_front_ends); only inputs and outputs are stored.

  forward_pass_golden.npz   MMClassifer.foraward_pass (models/multimodal_classifier.py:119-212): label projection, fine labels, loss assembly, accuracies
  eval_script_golden.npz    the per-batch loop body of evaluation/visualize_and_save_data.py:81-187: GT labels, accuracies, the saved pc_label / K / P
  pnp_frontend_golden.npz   solve_PnP + camera_matrix_scaling (evaluation/registration_pnp.py:58-61,95-148) with a recording cv2
  lsq_restart_golden.npz    solve_P_random_perturb + solver_wrapper (evaluation/registration_lsq.py:127-186) with a recording FrustumRegistration

CPU tests: the oracle restatements (oracle/prep_np.py, losses_torch.py, pnp_np.py, frustum_lm.py) and the host logic against these fixtures.
GPU tests: the HIP kernels (csrc/prep.hip, loss.hip, pnp.hip) and the drop-in front ends against the same fixtures, through the C ABI."""
import math
import random

import numpy as np
import pytest
import torch

from oracle import frustum_lm as flm
from oracle import losses_torch
from oracle import pnp_np, prep_np

FP_CASES = (0, 1)
EV_CASES = (0, 1)
PNP_CASES = range(6)
RP_CASES = range(4)


def _near_border(pxpy, z, H, W, tol):
    """points whose label can legitimately differ between two fp32 evaluation orders: within tol px of an image border or tol of z = 0.1"""
    return ((np.abs(pxpy[:, 0]) < tol) | (np.abs(pxpy[:, 0] - (W - 1)) < tol) | (np.abs(pxpy[:, 1]) < tol) | (np.abs(pxpy[:, 1] - (H - 1)) < tol)
            | (np.abs(z - 0.1) < tol * 1e-2))


def _near_cell_edge(pxpy, scale, tol=1e-4):
    q = pxpy / scale
    return (np.abs(q - np.round(q)) < tol).any(axis=1)


# ------------------------------------------------------------------------------------------------ CPU: oracle vs reference fixtures
@pytest.mark.parametrize("ci", FP_CASES)
def test_oracle_labels_losses_accuracy_vs_forward_pass(golden, ci):
    g = golden("forward_pass_golden.npz")
    k = "fp%d_" % ci
    H, W, scale = [int(v) for v in g[k + "HW"]]
    pc, P, K = g[k + "pc"], g[k + "P"], g[k + "K"]
    coarse, fine, pxpy = prep_np.project_labels(pc, P, K, H, W, scale)
    ref_px = g[k + "KP_pc_pxpy"]
    fin = np.isfinite(ref_px).all(axis=1) & (np.abs(ref_px) < 1e4).all(axis=1)          # far-away pixels (z ~ 0) are ill-conditioned and never inside
    np.testing.assert_allclose(pxpy.transpose(0, 2, 1)[fin], ref_px.transpose(0, 2, 1)[fin], rtol=2e-5, atol=2e-3)
    edge = _near_border(ref_px, g[k + "P_pc"][:, 2], H, W, 1e-2)
    assert edge.sum() < 0.01 * edge.size
    assert np.array_equal(coarse[~edge], g[k + "coarse_labels"][~edge])
    assert (coarse == g[k + "coarse_labels"]).mean() > 0.9999
    inside = (g[k + "coarse_labels"] == 1) & ~_near_cell_edge(ref_px, scale) & ~edge
    assert inside.sum() > 50
    assert np.array_equal(fine[inside], g[k + "fine_labels"][inside])
    # loss assembly and accuracies on the REFERENCE's labels (the canned scores of the fixture)
    cs, fs = torch.from_numpy(g[k + "coarse_scores"]), torch.from_numpy(g[k + "fine_scores"])
    cl, fl = torch.from_numpy(g[k + "coarse_labels"]), torch.from_numpy(g[k + "fine_labels"])
    fl = torch.where(cl == 1, fl, torch.zeros_like(fl))          # labels of outside points are never read
    loss, lc, lf, ca, fa = losses_torch.classifier_loss(cs, fs, cl, fl, 50.0)
    np.testing.assert_allclose([float(loss), float(lc), float(lf)], g[k + "losses"], rtol=2e-6)
    np.testing.assert_allclose([float(ca), float(fa)], g[k + "accuracy"], rtol=1e-6)
    assert np.array_equal(cs.argmax(1).numpy(), g[k + "coarse_predictions"]) and np.array_equal(fs.argmax(1).numpy(), g[k + "fine_predictions"])


@pytest.mark.parametrize("ci", EV_CASES)
def test_oracle_eval_script_records_and_accuracies(golden, ci):
    g = golden("eval_script_golden.npz")
    k = "ev%d_" % ci
    H, W, scale, fine_model = [int(v) for v in g[k + "HW"]]
    pc, P, K = g[k + "pc"], g[k + "P"], g[k + "K"]
    rec = g[k + "pc_label"]                      # [B,7,N]: pc, coarse prediction, coarse label, fine prediction, fine label
    coarse, fine, pxpy = prep_np.project_labels(pc, P, K, H, W, scale)
    z = np.einsum("brk,bkn->brn", P[:, :3, :].astype(np.float32), np.concatenate((pc, np.ones_like(pc[:, :1])), axis=1))[:, 2]
    edge = _near_border(pxpy, z, H, W, 1e-2)
    assert np.array_equal(coarse[~edge], rec[:, 4][~edge].astype(np.int64))
    ok = (rec[:, 4] == 1) & ~edge & ~_near_cell_edge(pxpy, scale)
    assert np.array_equal(fine[ok], rec[:, 6][ok].astype(np.int64))
    # the record layout and dtype, built from the reference's own labels
    cpred = g[k + "coarse_pred"]
    fpred = g[k + "fine_pred"] if fine_model else cpred          # coarse-only models save the coarse prediction twice (:96-97)
    mine = prep_np.pack_pc_label(pc, cpred, rec[:, 4].astype(np.int64), fpred, rec[:, 6].astype(np.int64))
    assert mine.dtype == rec.dtype == np.float64 and np.array_equal(mine, rec)
    assert np.array_equal(g[k + "saved_K"], K) and np.array_equal(g[k + "saved_P"], P)
    acc = prep_np.accuracy(cpred, rec[:, 4].astype(np.int64), fpred, rec[:, 6].astype(np.int64))
    sums = g[k + "acc_sums"]
    assert int(sums[2]) == pc.shape[0]
    np.testing.assert_allclose(acc.astype(np.float64).sum(axis=0), sums[:2], rtol=1e-6)
    for b, line in enumerate(g[k + "acc_lines"]):
        assert str(line) == "%d coarse accuracy %.4f, fine accuracy %.4f" % (b, acc[b, 0], acc[b, 1])


@pytest.mark.parametrize("ci", PNP_CASES)
def test_oracle_pnp_front_end_vs_reference(golden, ci):
    from deepi2p_amd.registration_pnp import camera_matrix_scaling
    from scipy.spatial.transform import Rotation
    g = golden("pnp_frontend_golden.npz")
    k = "pnp%d_" % ci
    pc, coarse, fine, K = g[k + "pc"], g[k + "coarse"], g[k + "fine"], g[k + "K"]
    Himg, Wimg, s = g[k + "HWs"]
    W_fine = int(round(Wimg * s))
    X, uv = pnp_np.correspondences(pc, coarse, fine, W_fine)
    ok, n_inl, threw = bool(g[k + "ret"][0]), int(g[k + "ret"][1]), bool(g[k + "ret"][2])
    tvec = g[k + "ret"][3:6]
    if int(g[k + "called"]):
        assert X.shape[1] >= 4                                                   # :123
        assert np.array_equal(X.T, g[k + "arg_points"]) and np.array_equal(uv.T, g[k + "arg_pixels"])
        assert np.array_equal(camera_matrix_scaling(K, s), g[k + "arg_K"])
        assert list(g[k + "arg_scalars"]) == [500.0, 0.6, 1.0, 0.0]              # iterationsCount, reprojectionError, EPNP flag, no guess
    else:
        assert X.shape[1] < 4
    R = Rotation.from_rotvec([0.1, -0.2, 0.05]).as_matrix()
    Pm, ratio = pnp_np.accept(R, tvec, n_inl, max(X.shape[1], 1), success=ok and not threw and X.shape[1] >= 4)
    np.testing.assert_allclose(Pm, g[k + "P"], atol=1e-15)
    assert ratio == float(g[k + "ratio"])


@pytest.mark.parametrize("ci", RP_CASES)
def test_restart_driver_list_and_selection_vs_reference(golden, ci):
    from deepi2p_amd.registration import restart_list
    g = golden("lsq_restart_golden.npz")
    k = "rp%d_" % ci
    n, threads, seed = [int(v) for v in g[k + "args"]]
    sigma = 10 * math.pi / 180
    for fn in (lambda r: flm.reference_restart_list(r, 0.37, sigma, 10.0, n, threads), lambda r: restart_list(0.37, sigma, 10.0, n, threads, r)):
        ys, Ts = fn(random.Random(seed))
        assert len(ys) == int(g[k + "n_calls"])                                   # incl. the dropped last wave when threads | n
        assert np.array_equal(ys, g[k + "ry"]) and np.array_equal(Ts, g[k + "t"])
    assert (g[k + "max_iter"] == 500).all() and (g[k + "flags"] == [0, 1]).all()  # max_iter hard-coded (:176), is_debug False, is_2d passed on
    assert list(g[k + "bounds"]) == [-5, -0.1, -10, 5, 0.1, 10]
    best = flm.select_min_cost(g[k + "canned_cost"][:len(ys)])
    assert best == int(g[k + "best_call"]) == 3                                   # costs 3 and 7 tie: the first one is kept (strict <)
    assert float(g[k + "cost"]) == float(g[k + "canned_cost"][best])
    # without thread_num the build runs every restart
    assert len(restart_list(0.0, sigma, 10.0, n, None, np.random.default_rng(0))[0]) == n


# ------------------------------------------------------------------------------------------------ GPU: HIP vs reference fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("ci", FP_CASES)
def test_hip_labels_losses_accuracy_vs_forward_pass(dev, golden, ci):
    from deepi2p_amd import ops, prep, training
    g = golden("forward_pass_golden.npz")
    k = "fp%d_" % ci
    H, W, scale = [int(v) for v in g[k + "HW"]]
    pc, P, K = [torch.from_numpy(g[k + n]).to(dev) for n in ("pc", "P", "K")]
    coarse, fine, pxpy = prep.project_labels(pc, P, K, H, W, scale, want_pxpy=True)
    ref_px = g[k + "KP_pc_pxpy"]
    fin = np.isfinite(ref_px).all(axis=1) & (np.abs(ref_px) < 1e4).all(axis=1)          # far-away pixels (z ~ 0) are ill-conditioned and never inside
    np.testing.assert_allclose(pxpy.cpu().numpy().transpose(0, 2, 1)[fin], ref_px.transpose(0, 2, 1)[fin], rtol=2e-5, atol=2e-3)
    edge = _near_border(ref_px, g[k + "P_pc"][:, 2], H, W, 1e-2)
    c_np, f_np = coarse.cpu().numpy(), fine.cpu().numpy()
    assert np.array_equal(c_np[~edge], g[k + "coarse_labels"][~edge])
    inside = (g[k + "coarse_labels"] == 1) & ~_near_cell_edge(ref_px, scale) & ~edge
    assert np.array_equal(f_np[inside], g[k + "fine_labels"][inside])
    # losses + accuracies from the library on the reference's labels and canned scores
    cs, fs = torch.from_numpy(g[k + "coarse_scores"]).to(dev), torch.from_numpy(g[k + "fine_scores"]).to(dev)
    cl = torch.from_numpy(g[k + "coarse_labels"].astype(np.int32)).to(dev)
    fl = torch.from_numpy(np.where(g[k + "coarse_labels"] == 1, g[k + "fine_labels"], 0).astype(np.int32)).to(dev)
    out = training.classifier_loss(cs, cl, fs, fl, coarse_loss_alpha=50.0, want_grads=False)
    np.testing.assert_allclose([float(out["loss"]), float(out["coarse"]), float(out["fine"])], g[k + "losses"], rtol=1e-5)
    np.testing.assert_allclose([float(out["coarse_accuracy"]), float(out["fine_accuracy"])], g[k + "accuracy"], rtol=1e-6)
    cpred, fpred = ops.argmax_channels(cs), ops.argmax_channels(fs)
    assert np.array_equal(cpred.cpu().numpy(), g[k + "coarse_predictions"]) and np.array_equal(fpred.cpu().numpy(), g[k + "fine_predictions"])
    # foraward_pass's accuracies are over the whole batch (:194-199); the per-frame kernel values recombine to them
    acc = prep.label_accuracy(cpred.int(), cl, fpred.int(), fl).cpu().numpy().astype(np.float64)
    n_in = (g[k + "coarse_labels"] == 1).sum(axis=1)
    np.testing.assert_allclose(acc[:, 0].mean(), g[k + "accuracy"][0], rtol=1e-6)
    np.testing.assert_allclose((np.nan_to_num(acc[:, 1]) * n_in).sum() / n_in.sum(), g[k + "accuracy"][1], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", EV_CASES)
def test_hip_eval_script_records_and_accuracies(dev, golden, ci):
    from deepi2p_amd import prep
    g = golden("eval_script_golden.npz")
    k = "ev%d_" % ci
    H, W, scale, fine_model = [int(v) for v in g[k + "HW"]]
    rec = g[k + "pc_label"]
    pc, P, K = [torch.from_numpy(g[k + n]).to(dev) for n in ("pc", "P", "K")]
    coarse, fine, pxpy = prep.project_labels(pc, P, K, H, W, scale, want_pxpy=True)
    px = pxpy.cpu().numpy()
    z = np.einsum("brk,bkn->brn", g[k + "P"][:, :3, :], np.concatenate((g[k + "pc"], np.ones_like(g[k + "pc"][:, :1])), axis=1))[:, 2]
    edge = _near_border(px, z, H, W, 1e-2)
    assert np.array_equal(coarse.cpu().numpy()[~edge], rec[:, 4][~edge].astype(np.int32))
    ok = (rec[:, 4] == 1) & ~edge & ~_near_cell_edge(px, scale)
    assert np.array_equal(fine.cpu().numpy()[ok], rec[:, 6][ok].astype(np.int32))
    cpred = torch.from_numpy(g[k + "coarse_pred"].astype(np.int32)).to(dev)
    fpred = torch.from_numpy(g[k + "fine_pred"].astype(np.int32)).to(dev) if fine_model else cpred
    cgt, fgt = torch.from_numpy(rec[:, 4].astype(np.int32)).to(dev), torch.from_numpy(np.clip(rec[:, 6], -2 ** 31, 2 ** 31 - 1).astype(np.int32)).to(dev)
    mine = prep.pack_pc_label(pc, cpred, cgt, fpred, fgt).cpu().numpy()
    same = np.abs(rec[:, 6]) < 2 ** 31                                            # fine labels of far-outside points overflow the int32 hand-off; they are never read
    assert np.array_equal(mine[:, :6], rec[:, :6]) and np.array_equal(mine[:, 6][same], rec[:, 6][same])
    acc = prep.label_accuracy(cpred, cgt, fpred, fgt).cpu().numpy()
    np.testing.assert_allclose(acc.astype(np.float64).sum(axis=0), g[k + "acc_sums"][:2], rtol=1e-6)
    for b, line in enumerate(g[k + "acc_lines"]):
        assert str(line) == "%d coarse accuracy %.4f, fine accuracy %.4f" % (b, acc[b, 0], acc[b, 1])


@pytest.mark.gpu
@pytest.mark.parametrize("ci", PNP_CASES)
def test_hip_pnp_packing_and_rules_vs_reference(dev, golden, ci):
    from deepi2p_amd import registration_pnp as rp
    g = golden("pnp_frontend_golden.npz")
    k = "pnp%d_" % ci
    pc, coarse, fine, K = g[k + "pc"], g[k + "coarse"], g[k + "fine"], g[k + "K"]
    Himg, Wimg, s = g[k + "HWs"]
    W_fine = int(round(Wimg * s))
    corr, n = rp.pack_correspondences(torch.from_numpy(pc).to(dev).unsqueeze(0), torch.from_numpy(coarse.astype(np.int32)).to(dev).unsqueeze(0),
                                      torch.from_numpy(fine.astype(np.int32)).to(dev).unsqueeze(0), W_fine)
    n = int(n[0])
    assert n == int((coarse == 1).sum())
    if int(g[k + "called"]):
        c = corr[0, :n].cpu().numpy().astype(np.float64)
        assert np.array_equal(c[:, 0:3], g[k + "arg_points"]) and np.array_equal(c[:, 3:5], g[k + "arg_pixels"]) and not c[:, 5:].any()
    # the drop-in: fewer than four correspondences -> identity and outlier ratio 1 without a solve (:123,143-146)
    Pm, ratio = rp.solve_PnP(pc, coarse, fine, K, Himg, Wimg, s, 64, rng=np.random.default_rng(0))
    if n < 4:
        assert np.array_equal(Pm, np.identity(4)) and ratio == 1 and np.array_equal(Pm, g[k + "P"])
    else:
        assert Pm.shape == (4, 4) and 0.0 <= ratio <= 1.0
        if ratio == 1:
            assert np.array_equal(Pm, np.identity(4))                              # rejected poses come back as identity (:139-141)
        else:
            assert np.linalg.norm(Pm[:3, 3]) < 14.14


@pytest.mark.gpu
@pytest.mark.parametrize("ci", (0, 1))
def test_hip_restart_driver_runs_reference_list(dev, golden, ci):
    """solve_P_random_perturb on the HIP solver with the reference's restart list (seeded ``random``): as many hypotheses as the reference
    starts, max_iter 500, the minimum-cost restart returned -- equal to solving that list directly."""
    from deepi2p_amd import registration as reg
    from deepi2p_amd import synthetic
    g = golden("lsq_restart_golden.npz")
    k = "rp%d_" % ci
    n, threads, seed = [int(v) for v in g[k + "args"]]
    f = synthetic.make_frame(np.random.default_rng(50 + ci), N=2048, H=160, W=512, flip=0.05, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    sigma, lb, ub = 10 * math.pi / 180, [-5, -0.1, -10], [5, 0.1, 10]
    P, cost, res = reg.solve_P_random_perturb(pts, lab, f["K"], 160, 512, 10.0, 0.37, sigma, lb, ub, n, True, threads, rng=random.Random(seed))
    ys, Ts = g[k + "ry"], g[k + "t"]
    Ps, costs, best = reg.solvePGivenK_batched(pts, lab, f["K"], ys, Ts, 160, 512, lb, ub, 500, True)
    assert len(costs) == int(g[k + "n_calls"])
    assert best == flm.select_min_cost(costs) and cost == costs[best] and np.array_equal(P, Ps[best])
    assert res.shape[0] == 3 * int((lab == 1).sum()) + int((lab == 0).sum())
