"""Value-level parity at the BASELINE sizes (not only at the reduced golden sizes).

  configs[1]/[2]  N=20480, 160x512, B=1: HIP logits vs (1) the IMPORTED REFERENCE's full-size golden (sub-sampled logits,
                  statistics, all argmax labels) and (2) the oracle restatement run here on the same frame; coarse-only and
                  coarse+fine models; split-K on and off (the split-K / tile / XCD-order choices are shape dependent).
  configs[0]      Oxford single pair: N=8192, 160x512, B=1, R=1 -- network -> argmax -> solvePGivenK against the oracle.
  configs[2]      B=64 and configs[3] B=16: the stated batch sizes through the size-independent properties.
  configs[3]/[4]  one frame at the config's own shape (30000 pts / 896x1600 fine; 40960 pts / 384x640 coarse): every logit vs the oracle.
Tolerances: |dlogit| <= 1e-3 * max|logit|, label flips < 0.1 % (and none where the reference's own margin
exceeds the tolerance)."""
import math

import numpy as np
import pytest
import torch

from deepi2p_amd import synthetic
from oracle import frustum_lm as flm
from oracle import network_torch as nt
from tests import fullsize_golden as fg

pytestmark = pytest.mark.gpu
REL = 1e-3


def _det(dev, N, H, W, fine):
    from deepi2p_amd.networks import KeypointDetector
    opt = synthetic.OptLike(N, H, W, fine)
    det = KeypointDetector(opt)
    det.load_state_dict(synthetic.synthetic_state_dict(opt))
    return det.to(dev).eval(), opt


@pytest.mark.parametrize("fine", [False, True])
@pytest.mark.parametrize("nosplit", [0, 1])
def test_fullsize_logits_vs_reference_golden_and_oracle(dev, golden, fine, nosplit):
    from deepi2p_amd import _lib
    g = golden("network_fullsize_golden.npz")
    b, N, H, W, stride = fg.inputs(g)
    det, opt = _det(dev, N, H, W, fine)
    x = [torch.from_numpy(b[k]).to(dev) for k in fg.NAMES]
    with _lib.option("conv_nosplit", nosplit):
        out = det(*x)
    coarse = (out[0] if fine else out).cpu().numpy()
    fine_l = out[1].cpu().numpy() if fine else None
    res = fg.check_logits(g, "fine_model" if fine else "coarse_model", coarse, fine_l, REL, 1e-3)
    if nosplit == 0:      # the oracle on the same frame, every logit (not only the golden's sub-sample)
        torch.set_num_threads(16)
        sd = nt.synthetic_state_dict(nt.OptLike(N, H, W, fine))
        with torch.no_grad():
            ref = nt.keypoint_detector(sd, nt.OptLike(N, H, W, fine), *[torch.from_numpy(b[k]) for k in fg.NAMES])
        rc = (ref[0] if fine else ref).numpy()
        assert np.abs(coarse - rc).max() <= REL * np.abs(rc).max() + 1e-7
        if fine:
            rf = ref[1].numpy()
            assert np.abs(fine_l - rf).max() <= REL * np.abs(rf).max() + 1e-7
            assert (fine_l.argmax(1) != rf.argmax(1)).mean() < 1e-3
    assert res


def test_config0_oxford_single_pair(dev):
    """BASELINE configs[0]: one Oxford-shaped pair, 8192 points (normals are zeros for Oxford, oxford_pc_img_pose_loader.py:362),
    160x512 image, ONE Gauss-Newton start: network -> argmax -> initial guess -> solvePGivenK, each stage against the oracle."""
    from deepi2p_amd import ops, registration
    from deepi2p_amd.networks import MMClassiferCoarse
    N, H, W = 8192, 160, 512
    batch = synthetic.make_batch(77, 1, N=N, H=H, W=W)
    batch["sn"][:] = 0.0
    opt = synthetic.OptLike(N, H, W, False)
    opt.device = dev
    sd = synthetic.synthetic_state_dict(opt)
    mm = MMClassiferCoarse(opt)
    mm.detector.load_state_dict(sd)
    cpu = {k: torch.from_numpy(batch[k]) for k in fg.NAMES}
    mm.set_input(cpu["pc"], cpu["intensity"], cpu["sn"], cpu["node_a"], cpu["node_b"], torch.zeros(1, 3, 4), cpu["img"],
                 torch.from_numpy(batch["K"]).float())
    logits = mm.forward(mm.pc, mm.intensity, mm.sn, mm.node_a, mm.node_b, mm.img).cpu()
    pred = mm.inference_pass()
    assert pred.dtype == torch.int64 and pred.shape == (1, N)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = nt.keypoint_detector(sd, nt.OptLike(N, H, W, False), *[cpu[k] for k in fg.NAMES])
    assert float((logits - ref).abs().max()) <= REL * float(ref.abs().max()) + 1e-7
    assert (pred.cpu() != ref.argmax(1)).float().mean() < 1e-3
    # the pose stage on the pair's labels: the network has random weights, so the synthetic labels stand in
    pts, lab, K = batch["pc"][0].astype(np.float64), batch["labels"][0], batch["K"][0]
    P0, y0, pcf, labf = flm.get_initial_guess(pts, lab)
    P1, y1, pcf1, labf1 = registration.get_initial_guess(pts, lab)
    assert abs(y0 - y1) < 1e-12 and np.array_equal(labf, labf1)
    lb, ub = [-5, -0.1, -10], [5, 0.1, 10]
    Po, co, ro, info = flm.solvePGivenK(pcf, labf, K, y0, np.zeros(3), H, W, lb, ub, 500, False, True, return_info=True)
    Pg, cg, rg = registration.solvePGivenK(pcf1, labf1, K, y1, np.zeros(3), H, W, lb, ub, 500, False, True)
    assert rg.shape == ro.shape
    assert abs(cg - co) <= 1e-6 * co
    np.testing.assert_allclose(Pg, Po, atol=2e-3)
    np.testing.assert_allclose(rg, ro, atol=1e-6)
    # and the network's own prediction goes through the reference's skip rule when nothing is predicted inside
    lab_net = pred[0].cpu().numpy().astype(np.int32)
    if (lab_net == 1).sum() == 0:
        pipe = registration.RegistrationPipeline(H, W, R=1, seed=0)
        out = pipe(mm.pc, pred.int(), torch.from_numpy(batch["K"]).to(dev), pipe.draw(1, dev))
        assert float(out["cost"][0]) == 1e4 and int(out["best"][0]) == -1


def test_config2_stated_batch_64_fine_head_and_pnp(dev):
    """BASELINE configs[2] at its stated batch: 64 frames, coarse+fine heads, both argmax, PnP back end.  Properties: finite,
    deterministic, every frame equals the same frame run alone (batch independence); PnP runs on all 64 frames."""
    from deepi2p_amd import ops
    N, H, W, B = 20480, 160, 512, 64
    det, opt = _det(dev, N, H, W, True)
    b = synthetic.make_batch(301, B, N=N, H=H, W=W)
    x = [torch.from_numpy(b[k]).to(dev) for k in fg.NAMES]
    coarse, fine = det(*x)
    assert coarse.shape == (B, 2, N) and fine.shape == (B, 80, N)
    assert torch.isfinite(coarse).all() and torch.isfinite(fine).all()
    c2, f2 = det(*x)
    assert torch.equal(coarse, c2) and torch.equal(fine, f2)
    for i in (0, 37, 63):
        one = det(*[t[i:i + 1].contiguous() for t in x])
        assert torch.equal(one[0][0], coarse[i]) and torch.equal(one[1][0], fine[i])
    cl, fl = ops.argmax_channels(coarse), ops.argmax_channels(fine)
    assert torch.equal(cl.long(), coarse.argmax(1)) and torch.equal(fl.long(), fine.argmax(1))
    # PnP on GT-derived labels of the same 64 frames (the random-weight network predicts nothing useful)
    from deepi2p_amd import prep
    from deepi2p_amd.registration_pnp import camera_matrix_scaling, draw_samples, pnp_ransac
    P_gt = torch.from_numpy(b["P_gt"][:, :3, :]).float().to(dev)
    K32 = torch.from_numpy(b["K"]).float().to(dev)
    coarse_gt, fine_gt = prep.project_labels(x[0], P_gt, K32, H, W, 32)
    Kf = torch.from_numpy(np.stack([camera_matrix_scaling(k, 1 / 32) for k in b["K"]])).to(dev)
    samples = torch.from_numpy(draw_samples(np.random.default_rng(0), B, 500)).to(dev)
    # The reference's estimator (EPnP RANSAC, the default) under two observation models:
    #  (a) the reference's own front end (registration_pnp.py:107-109): the pixel of a correspondence is the top-left CORNER of its
    #      32 x 32 cell -- a systematic half-cell bias of ~2.6 deg per axis at fx = 359 px, hence the 8 deg bound;
    #  (b) cell-CENTRE observations, passed explicitly (the test controls them): unbiased, so the reference's 5 deg / 2 m success rule
    #      applies.  In both cases EPnP's unconditional least-squares re-fit on the inliers (as cv2.solvePnPRansac does it) flips to a
    #      mirrored solution on ~1 frame in 6 of this 5 m thick slab scene (oracle/epnp_np.py behaves identically: 20 of 24 frames pass
    #      either bound there); such frames have |t| > 14.14 and come back as identity / outlier ratio 1, as in the reference.
    py = torch.div(fine_gt, W // 32, rounding_mode="floor")
    centre = torch.stack(((fine_gt - py * (W // 32)).float() + 0.5, py.float() + 0.5), dim=1).contiguous()
    for name, pixels, r_max, in (("corner", None, 8.0), ("centre", centre, 5.0)):
        out = pnp_ransac(x[0], coarse_gt, fine_gt, Kf, W // 32, samples, pixels=pixels)
        out2 = pnp_ransac(x[0], coarse_gt, fine_gt, Kf, W // 32, samples, pixels=pixels)
        assert torch.equal(out["P"], out2["P"])                      # deterministic given the draws
        P = out["P"].cpu().numpy()
        errs = [flm.get_P_diff(P[i], b["P_gt"][i]) for i in range(B)]
        ok = sum(1 for t, r in errs if t < 2.0 and r < r_max)
        assert ok >= 0.7 * B, (name, ok)
        if name == "centre":
            assert np.median([r for t, r in errs]) < 2.5
    # the builder's DLT + local-optimisation variant stays available behind method="dlt_lo"
    out = pnp_ransac(x[0], coarse_gt, fine_gt, Kf, W // 32, samples, method="dlt_lo")
    P = out["P"].cpu().numpy()
    ok = sum(1 for i in range(B) if (lambda tr: tr[0] < 2.0 and tr[1] < 8.0)(flm.get_P_diff(P[i], b["P_gt"][i])))
    assert ok >= 0.8 * B, ok


def test_config3_shape_logits_vs_oracle(dev):
    """BASELINE configs[3] at its frame shape (nuScenes: 30000 points, 896 x 1600 image, L = 28 x 50 = 1400 fine classes), B = 1:
    every coarse and fine logit against the oracle restatement run here on the same frame (the 1402-output head and the 28 x 50
    attention maps are exercised by no golden fixture)."""
    N, H, W = 30000, 896, 1600
    det, opt = _det(dev, N, H, W, True)
    b = synthetic.make_batch(303, 1, N=N, H=H, W=W)
    x = [torch.from_numpy(b[k]).to(dev) for k in fg.NAMES]
    coarse, fine = det(*x)
    assert coarse.shape == (1, 2, N) and fine.shape == (1, 1400, N)
    torch.set_num_threads(32)
    sd = nt.synthetic_state_dict(nt.OptLike(N, H, W, True))
    with torch.no_grad():
        rc, rf = nt.keypoint_detector(sd, nt.OptLike(N, H, W, True), *[torch.from_numpy(b[k]) for k in fg.NAMES])
    coarse, fine = coarse.cpu(), fine.cpu()
    assert float((coarse - rc).abs().max()) <= REL * float(rc.abs().max()) + 1e-7
    assert float((fine - rf).abs().max()) <= REL * float(rf.abs().max()) + 1e-7
    assert (coarse.argmax(1) != rc.argmax(1)).float().mean() < 1e-3
    assert (fine.argmax(1) != rf.argmax(1)).float().mean() < 1e-3


def test_config4_shape_logits_vs_oracle(dev):
    """BASELINE configs[4] at its frame shape (Oxford: 40960 points, 384 x 640 image, coarse head), B = 1: every logit against the
    oracle restatement run here on the same frame.  At 384 x 640 the 3x3 layers take the RUN-TIME-row-length instances of
    `conv3x3_x3_kernel` (the template instances are the 160 x 512 shapes) and the fp32 stem + pool (`stem_x3` needs W % 128 == 0): a tile
    path no golden covers.  Also against the library's own fp32 kernels (conv_x3 / stem_x3 / pw_x3 / head_x3 off)."""
    from deepi2p_amd import _lib
    N, H, W = 40960, 384, 640
    det, opt = _det(dev, N, H, W, False)
    b = synthetic.make_batch(404, 1, N=N, H=H, W=W)
    x = [torch.from_numpy(b[k]).to(dev) for k in fg.NAMES]
    coarse = det(*x)
    assert coarse.shape == (1, 2, N)
    with _lib.option("conv_x3", 0), _lib.option("stem_x3", 0), _lib.option("pw_x3", 0), _lib.option("head_x3", 0):
        coarse_fp32 = det(*x)
    torch.set_num_threads(32)
    sd = nt.synthetic_state_dict(nt.OptLike(N, H, W, False))
    with torch.no_grad():
        rc = nt.keypoint_detector(sd, nt.OptLike(N, H, W, False), *[torch.from_numpy(b[k]) for k in fg.NAMES])
    for name, got in (("bf16x3 paths on", coarse.cpu()), ("fp32 kernels", coarse_fp32.cpu())):
        assert float((got - rc).abs().max()) <= REL * float(rc.abs().max()) + 1e-7, name
        assert (got.argmax(1) != rc.argmax(1)).float().mean() < 1e-3, name


def test_config3_stated_batch_16_per_gpu(dev):
    """BASELINE configs[3] at the per-GPU batch of the 8-GPU run: 16 frames of 30000 points / 896x1600 (L = 1400)."""
    N, H, W, B = 30000, 896, 1600, 16
    det, opt = _det(dev, N, H, W, True)
    b = synthetic.make_batch(302, B, N=N, H=H, W=W)
    x = [torch.from_numpy(b[k]).to(dev) for k in fg.NAMES]
    coarse, fine = det(*x)
    assert coarse.shape == (B, 2, N) and fine.shape == (B, 1400, N)
    assert torch.isfinite(coarse).all() and torch.isfinite(fine).all()
    c2, f2 = det(*x)
    assert torch.equal(coarse, c2) and torch.equal(fine, f2)
    del c2, f2
    for i in (0, 15):
        one = det(*[t[i:i + 1].contiguous() for t in x])
        assert torch.equal(one[0][0], coarse[i]) and torch.equal(one[1][0], fine[i])
