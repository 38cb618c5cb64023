"""GPU parity of the whole classification network (HIP kernels behind deepi2p_amd.networks) against
  (1) golden vectors produced by the IMPORTED REFERENCE (tests/golden/make_golden.py), and
  (2) the oracle restatement on fresh seeded inputs.
Stated tolerances: |dlogit| <= 1e-3 * max|logit| ; label flips < 0.1 % ; 3-NN indices equal except
at distance ties."""
import numpy as np
import pytest
import torch

from oracle import network_torch as nt

pytestmark = pytest.mark.gpu
REL = 1e-3


def _detector(dev, N, H, W, fine):
    from deepi2p_amd.networks import KeypointDetector
    opt = nt.OptLike(N, H, W, fine)
    det = KeypointDetector(opt)
    det.load_state_dict(nt.synthetic_state_dict(opt))
    return det.to(dev).eval(), opt


def _close(a, b, name, rel=REL):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    assert a.shape == b.shape, name
    err = float((a - b).abs().max())
    lim = rel * float(b.abs().max()) + 1e-7
    assert err <= lim, "%s: max err %.3g > %.3g" % (name, err, lim)


@pytest.mark.parametrize("fname", ["network_golden.npz", "network_coarse_golden.npz"])
def test_network_vs_reference_golden(dev, golden, fname):
    g = golden(fname)
    B, N, H, W, fine = [int(v) for v in g["meta"]]
    det, opt = _detector(dev, N, H, W, bool(fine))
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    enc = det.pc_encoder(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"])
    assert enc[2].dtype == torch.int64
    for i, name in enumerate(["pc_centers", "cluster_mean", None, "first_pn_out", "second_pn_out", "node_a_features",
                              "node_b_features", "global_feature"]):
        if name:
            _close(enc[i], g[name], name)
    mism = (enc[2].cpu().numpy() != g["min_k_idx"]).mean()
    assert mism < 2e-3, "3-NN index mismatch rate %.4f" % mism
    s16, s32, glob = det.img_encoder(t["img"])
    _close(s16, g["s16"], "s16"); _close(s32, g["s32"], "s32"); _close(glob, g["img_global"], "img_global")
    from deepi2p_amd import _lib
    with _lib.option("conv_nowinograd", 1):          # the direct implicit-GEMM kernel on the 3x3 stride-1 layers: same goldens
        d16, d32, dglob = det.img_encoder(t["img"])
    _close(d16, g["s16"], "s16 (direct)"); _close(d32, g["s32"], "s32 (direct)"); _close(dglob, g["img_global"], "img_global (direct)")
    assert not torch.equal(d32, s32)                  # (two different algorithms did run)
    out = det(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"])
    coarse = out[0] if fine else out
    _close(coarse, g["coarse"], "coarse logits")
    flips = (coarse.argmax(1).cpu().numpy() != g["coarse"].argmax(1)).mean()
    assert flips < 1e-3, "coarse label flip rate %.4f" % flips
    if fine:
        _close(out[1], g["fine"], "fine logits")
        assert (out[1].argmax(1).cpu().numpy() != g["fine"].argmax(1)).mean() < 1e-3


def test_network_vs_oracle_fresh_inputs(dev):
    """Different seed, KITTI-like geometry, B=1, N=2048, 96x160: logits and argmax labels vs the oracle."""
    from deepi2p_amd import synthetic
    from deepi2p_amd.networks import MMClassifer
    N, H, W = 2048, 96, 160
    batch = synthetic.make_batch(11, 1, N=N, H=H, W=W)
    opt = nt.OptLike(N, H, W, True)
    opt.device = dev
    sd = nt.synthetic_state_dict(opt, seed=3)
    mm = MMClassifer(opt)
    mm.detector.load_state_dict({("module." + k): v for k, v in sd.items()} if False else sd)
    cpu = {k: torch.from_numpy(batch[k]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    mm.set_input(cpu["pc"], cpu["intensity"], cpu["sn"], cpu["node_a"], cpu["node_b"], torch.zeros(1, 3, 4),
                 cpu["img"], torch.from_numpy(batch["K"]).float())
    cp, fp = mm.inference_pass()
    assert cp.dtype == torch.int64 and cp.shape == (1, N)
    with torch.no_grad():
        coarse, fine = nt.keypoint_detector(sd, opt, cpu["pc"], cpu["intensity"], cpu["sn"], cpu["node_a"], cpu["node_b"], cpu["img"])
    out = mm.forward(mm.pc, mm.intensity, mm.sn, mm.node_a, mm.node_b, mm.img)
    _close(out[0], coarse, "coarse"); _close(out[1], fine, "fine")
    assert (cp.cpu() != coarse.argmax(1)).float().mean() < 1e-3
    assert (fp.cpu() != fine.argmax(1)).float().mean() < 1e-3


def test_checkpoint_key_compat_module_prefix(dev, tmp_path):
    """load_model accepts DataParallel ('module.') checkpoints (util/pytorch_helper.py:24-33)."""
    from deepi2p_amd.networks import MMClassiferCoarse
    opt = nt.OptLike(512, 64, 64, False)
    opt.device = dev
    sd = nt.synthetic_state_dict(opt)
    path = tmp_path / "ckpt.pth"
    torch.save({"module." + k: v for k, v in sd.items()}, path)
    mm = MMClassiferCoarse(opt)
    mm.load_model(str(path))
    got = mm.detector.state_dict()
    assert list(got.keys()) == list(sd.keys())
    assert all(torch.equal(got[k].cpu(), sd[k]) for k in sd)


def test_pcencoder_rejects_wrong_N(dev):
    det, opt = _detector(dev, 512, 64, 64, False)
    x = torch.zeros(1, 3, 256, device=dev)
    with pytest.raises(RuntimeError):
        det.pc_encoder(x, x[:, :1], x, x[:, :, :128].contiguous(), x[:, :, :128].contiguous())


def test_full_size_forward_properties(dev):
    """BASELINE config-2 shape (N=20480, 160x512, coarse head), B=4: finite outputs, batch independence
    (frame b's logits do not depend on its batch neighbours) and determinism."""
    from deepi2p_amd import synthetic
    det, opt = _detector(dev, 20480, 160, 512, False)
    batch = synthetic.make_batch(5, 4)
    t = {k: torch.from_numpy(batch[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    out = det(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"])
    assert out.shape == (4, 2, 20480) and torch.isfinite(out).all()
    out2 = det(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"])
    assert torch.equal(out, out2)
    one = det(*[t[k][2:3].contiguous() for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")])
    assert torch.equal(one[0], out[2])


@pytest.mark.parametrize("N", [2048, 1000])
def test_fused_point_head_is_bit_identical(dev, N):
    """The coarse per_point_pn as one launch (hidden activations in LDS) must reproduce the three separate pointwise
    launches bit for bit: same K order in the MFMA chain, same epilogue arithmetic; N = 1000 leaves a ragged last tile."""
    from deepi2p_amd import synthetic
    H, W = 64, 128
    det, opt = _detector(dev, N, H, W, False)
    batch = synthetic.make_batch(5, 2, N=N, H=H, W=W)
    x = [torch.from_numpy(batch[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    from deepi2p_amd import _lib
    assert det.fuse_head
    with _lib.option("head_x3", 0):              # the fp32-MFMA fused head (round 5's default is the bf16x3 head: different arithmetic, below)
        fused = det(*x)
        det.fuse_head = False
        try:
            chain = det(*x)
        finally:
            del det.fuse_head
        assert fused.shape == chain.shape == (2, 2, N)
        assert torch.equal(fused, chain)
        with _lib.option("head_reg", 1):         # the wave-autonomous kernel (activations in registers; the LDS-tile kernel is the default)
            reg = det(*x)
        assert torch.equal(fused, reg)
    # the default head (di2p_point_head_x3: bf16 matrix instructions on exact three-way splits) against that chain: fp32 round-off of three
    # contractions (its accuracy against fp64 next to the fp32 head's is tests/test_gpu_head_x3.py's subject)
    x3 = det(*x)
    assert float((x3 - chain).abs().max()) <= 1e-4 * float(chain.abs().max()) + 1e-6


@pytest.mark.parametrize("N", [2048, 1000])
def test_fused_pointnet_chains_are_bit_identical(dev, N):
    """first_pointnet (7 -> 32 -> 32 -> 32) and second_pointnet ((32 + gathered 32) -> 64 -> 64) as ONE launch each
    (di2p_point_chain: hidden activations in LDS) against the separate pointwise launches (knob pw_nochain): every output of
    the point encoder bit for bit; N = 1000 leaves a ragged last tile."""
    from deepi2p_amd import synthetic, _lib
    H, W = 64, 128
    det, opt = _detector(dev, N, H, W, False)
    batch = synthetic.make_batch(6, 2, N=N, H=H, W=W)
    x = [torch.from_numpy(batch[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b")]
    _lib.WORK = {}
    try:
        with _lib.option("pw_nochain", 0):
            fused = det.pc_encoder(*x)
        assert "di2p_point_chain" in _lib.WORK
        _lib.WORK = {}
        with _lib.option("pw_nochain", 1):
            chain = det.pc_encoder(*x)
        assert "di2p_point_chain" not in _lib.WORK
    finally:
        _lib.WORK = None
    assert len(fused) == len(chain) == 8
    for a, b in zip(fused, chain):
        assert a.shape == b.shape and torch.equal(a, b)


def test_point_chain_argument_checks(dev):
    """di2p_point_chain directly: 3- and 2-layer chains of widths 32 and 64 at K0 = 8 / 7 / 32 / 40, N a multiple of 64, of 4 and of neither,
    bit-identical to the separate launches; unsupported shapes are refused (the host layer falls back to the separate layers)."""
    from deepi2p_amd import ops
    from deepi2p_amd._lib import DeepI2PHipError
    g = torch.Generator(device="cpu").manual_seed(1)
    mk = lambda k, m: (torch.randn(k, m, generator=g).to(dev), torch.rand(m, generator=g).to(dev) + 0.5, torch.randn(m, generator=g).to(dev), True)
    for M, K0, N in ((32, 8, 256), (32, 7, 250), (32, 32, 1000), (64, 32, 256), (64, 40, 330), (64, 7, 64), (32, 7, 1), (64, 33, 31)):
        x = torch.randn(3, K0, N, generator=g).to(dev)
        layers = [mk(K0, M), mk(M, M), (mk(M, M)[0], None, mk(M, M)[2], False)]
        for nl in (3, 2):
            assert ops.point_chain_ok([ops.Src(x)], layers[:nl], N)
            y = ops.point_chain([ops.Src(x)], layers[:nl], N)
            ref = x
            for l in layers[:nl]:
                ref = ops.pointwise_gemm([ops.Src(ref)], l[0], M, N, scale=l[1], shift=l[2], relu=l[3])
            assert torch.equal(y, ref), (M, K0, N, nl)
    x = torch.randn(2, 8, 256, generator=g).to(dev)
    layers = [mk(8, 32), mk(32, 32), mk(32, 32)]
    assert ops.point_chain([ops.Src(x[:0])], layers, 256).shape == (0, 32, 256)       # empty batch
    assert not ops.point_chain_ok([ops.Src(x)], [mk(8, 48), mk(48, 48)], 256)        # width not 32 / 64
    assert not ops.point_chain_ok([ops.Src(x)], [mk(8, 32), mk(32, 64)], 256)        # widths differ
    assert not ops.point_chain_ok([ops.Src(x), ops.Src(x)], layers, 256)             # two sources
    assert not ops.point_chain_ok([ops.Src(torch.randn(2, 40, 256, generator=g).to(dev))], [mk(40, 32), mk(32, 32)], 256)   # K0 > M
    with pytest.raises(DeepI2PHipError):
        ops.point_chain([ops.Src(x)], [mk(8, 48), mk(48, 48)], 256)


def test_forward_and_pose_are_run_to_run_identical(dev):
    """No result depends on the order in which atomics or workgroups retire: segment maxima merge through order-free
    u64 max, cluster sums are fixed-point integers, split-K partials and the solver's wave partials are combined in a fixed
    order.  Two runs of the network and of the pose pipeline must agree bit for bit."""
    from deepi2p_amd import synthetic
    from deepi2p_amd.registration import RegistrationPipeline
    N, H, W = 4096, 160, 512
    det, opt = _detector(dev, N, H, W, False)
    batch = synthetic.make_batch(9, 2, N=N, H=H, W=W)
    x = [torch.from_numpy(batch[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    a, b = det(*x), det(*x)
    assert torch.equal(a, b)
    pipe = RegistrationPipeline(H, W, R=12, seed=3)
    lab = torch.from_numpy(batch["labels"]).to(dev)
    K = torch.from_numpy(batch["K"]).to(dev)
    restarts = pipe.draw(2, dev)
    o1, o2 = pipe(x[0], lab, K, restarts), pipe(x[0], lab, K, restarts)
    for key in ("P", "costs", "iters", "best"):
        assert torch.equal(o1[key], o2[key]), key
