"""Training path on the GPU: every backward kernel against torch.autograd of the same op, the whole
train-mode step against the oracle (which tests/test_training_oracle.py pins to the imported reference's own gradients) and against
the reference-run fixture, and the optimiser loop."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import training_golden as tg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, tol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: max error / abs-max = %.3g (tolerance %.1g)" % (what, err, tol)


def _run_pair(fn_dev, fn_ref, tensors, tol=2e-5, grads=None):
    """fn_dev on CUDA fp32 leaves vs fn_ref on CPU fp64 leaves; compares outputs and the gradients of a random cotangent."""
    torch.manual_seed(0)
    dev = [t.to(DEV).requires_grad_(t.is_floating_point()) for t in tensors]
    ref = [t.double().requires_grad_(True) if t.is_floating_point() else t for t in tensors]
    yd, yr = fn_dev(*dev), fn_ref(*ref)
    _close(yd, yr, tol, "forward")
    ct = torch.randn(yr.shape, dtype=torch.float64)
    yd.backward(ct.float().to(DEV))
    yr.backward(ct)
    for i, (d, r) in enumerate(zip(dev, ref)):
        if not torch.is_tensor(r) or not r.is_floating_point() or (grads is not None and i not in grads):
            continue
        _close(d.grad, r.grad, tol, "gradient of input %d" % i)


@pytest.mark.parametrize("B,K,M,N", [(2, 7, 32, 1024), (3, 96, 128, 515), (1, 736, 256, 2048), (2, 128, 2, 640), (2, 67, 256, 128 * 16),
                                     (2, 200, 384, 2048), (3, 132, 130, 4100)])      # (the last two: 128 x 128 weight-gradient tiles with ragged edges)
def test_linear_backward(B, K, M, N):
    from deepi2p_amd import train_net as tn
    g = torch.Generator().manual_seed(B * 1000 + K)
    x, W, b = torch.randn(B, K, N, generator=g), torch.randn(M, K, 1, generator=g) / K ** 0.5, torch.randn(M, generator=g)
    _run_pair(lambda x, W, b: tn.linear(x, W, b), lambda x, W, b: F.conv1d(x, W, b), [x, W, b])


@pytest.mark.parametrize("shape,relu,res", [((4, 32, 1000), True, False), ((2, 64, 9, 13), True, True), ((3, 16, 128), False, False),
                                            ((2, 256, 2, 4), True, True), ((2, 8, 20480), True, False)])
def test_batchnorm_train(shape, relu, res):
    from deepi2p_amd import train_net as tn
    g = torch.Generator().manual_seed(7)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    r = torch.randn(shape, generator=g)
    rm_d, rv_d = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    rm_r, rv_r = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)

    def dev(x, gamma, beta, r):
        return tn._BatchNorm.apply(x, gamma, beta, rm_d, rv_d, 0.1, relu, r if res else None)

    def ref(x, gamma, beta, r):
        y = F.batch_norm(x, rm_r, rv_r, gamma, beta, True, 0.1, 1e-5)
        if res:
            y = y + r
        return F.relu(y) if relu else y

    _run_pair(dev, ref, [x, gamma, beta, r], tol=5e-5, grads=None if res else {0, 1, 2})
    _close(rm_d, rm_r, 1e-5, "running_mean")
    _close(rv_d, rv_r, 1e-5, "running_var")


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p", [(2, 3, 32, 64, 64, 7, 2, 3), (2, 64, 16, 32, 64, 3, 1, 1), (2, 64, 16, 32, 128, 3, 2, 1),
                                                 (2, 64, 16, 32, 128, 1, 2, 0), (1, 256, 4, 8, 512, 3, 2, 1), (3, 17, 9, 11, 5, 3, 1, 1)])
def test_conv2d_backward(B, Cin, H, W, Cout, k, s, p):
    from deepi2p_amd import train_net as tn
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    Wt = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    _run_pair(lambda x, W: tn._Conv2d.apply(x, W, s, p), lambda x, W: F.conv2d(x, W, None, stride=s, padding=p), [x, Wt], tol=3e-5)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,p", [(2, 64, 16, 32, 128, 3, 1), (2, 64, 16, 32, 128, 1, 0), (1, 256, 4, 8, 512, 3, 1), (2, 3, 32, 64, 64, 7, 3),
                                               (3, 17, 9, 11, 5, 3, 1), (2, 8, 7, 5, 4, 1, 0), (1, 5, 6, 6, 3, 2, 0)])
def test_conv2d_dgrad_stride2_parity_kernel_equals_dense_kernel(B, Cin, H, W, Cout, k, p):
    """Round 6: for stride 2 the input gradient is computed per parity class of input pixels (only the taps that reach the class: 1/4 of the
    dense kernel's matrix work, no divisibility tests).  Same non-zero products in the same order on an exact fma chain: the two kernels must
    agree bit for bit (zeros compare equal whatever their sign); odd sizes, 1 x 1 (three classes without taps), 7 x 7 and an even filter."""
    from deepi2p_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(H * W + k)
    OH, OW = (H + 2 * p - k) // 2 + 1, (W + 2 * p - k) // 2 + 1
    dy = torch.randn(B, Cout, OH, OW, generator=g).to(dev)
    Wt = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).to(dev)
    out = []
    for dense in (0, 1):
        _lib.set_option("conv_dgrad_dense", dense)
        try:
            dx = torch.full((B, Cin, H, W), float("nan"), device=dev)
            _lib.call("di2p_conv2d_dgrad", dy.data_ptr(), Wt.data_ptr(), dx.data_ptr(), B, Cin, H, W, Cout, k, k, 2, p, _lib.stream())
            out.append(dx)
        finally:
            _lib.set_option("conv_dgrad_dense", 0)
    assert torch.equal(out[0], out[1])
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), Wt.double().cpu(), dy.double().cpu(), stride=2, padding=p)
    assert float((out[0].double().cpu() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max()))


def test_winograd_dgrad_filter_transform_equals_flip_transpose_transform():
    """Round 6: the transformed filter of a stride-1 3x3 layer's input gradient comes straight from the forward filter (one launch); it must be
    the SAME tensor, bit for bit, as flipping, transposing and copying the filter and then transforming it (what the step did before) -- also
    for a layer with different channel counts (the channel roles swap)."""
    from deepi2p_amd import ops
    dev = torch.device("cuda", 0)
    for Cout, Cin in ((64, 64), (128, 64), (32, 96)):
        W = torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(Cout + Cin)).to(dev)
        a = ops.winograd_weights_dgrad(W)
        b = ops.winograd_weights(W.flip(2, 3).transpose(0, 1))
        assert a.shape == b.shape == (16, Cout, Cin) and torch.equal(a, b)


def test_bf16x3_pack_conv3x3_equals_host_side_permutation():
    """Round 6: the split operand of di2p_conv3x3_x3 straight from a filter bank, for the forward filter and for the input-gradient filter
    (flipped taps, channel roles swapped): the same bytes as permuting / flipping on the host side and packing the tap-major matrix."""
    from deepi2p_amd import ops
    dev = torch.device("cuda", 0)
    for Cout, Cin in ((256, 256), (512, 256), (48, 80)):
        W = torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(Cout + Cin)).to(dev)
        assert torch.equal(ops.bf16x3_pack_conv3x3(W), ops.bf16x3_pack(W.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous()))
        Wg = W.flip(2, 3).transpose(0, 1)                    # [Cin, Cout, 3, 3]: the filter of the input gradient
        assert torch.equal(ops.bf16x3_pack_conv3x3(W, dgrad=True), ops.bf16x3_pack(Wg.permute(2, 3, 1, 0).reshape(-1, Cin).contiguous()))
    with pytest.raises(RuntimeError):
        ops.bf16x3_pack_conv3x3(torch.randn(8, 8, 5, 5).to(dev))


def test_maxpool_avgpool_backward():
    from deepi2p_amd import train_net as tn
    g = torch.Generator().manual_seed(3)
    x = torch.relu(torch.randn(2, 16, 32, 64, generator=g))          # exact zeros: ties inside the windows
    _run_pair(lambda x: tn._MaxPool.apply(x), lambda x: F.max_pool2d(x, 3, 2, 1), [x])
    for shape in ((1, 3, 5, 7), (2, 4, 9, 256), (1, 2, 80, 256), (1, 1, 1, 1), (1, 2, 6, 2)):       # odd sizes, several row bands, degenerate planes
        x = torch.relu(torch.randn(*shape, generator=g))
        _run_pair(lambda x: tn._MaxPool.apply(x), lambda x: F.max_pool2d(x, 3, 2, 1), [x])
    x = torch.randn(2, 16, 5, 7, generator=g)
    _run_pair(lambda x: tn._AvgPool.apply(x), lambda x: F.adaptive_avg_pool2d(x, (1, 1)), [x])


def test_segment_max_gather_groupmax_backward():
    from deepi2p_amd import train_net as tn
    from oracle.network_torch import index_max_torch
    g = torch.Generator().manual_seed(5)
    B, C, N, Ma = 2, 32, 2048, 128
    data = torch.relu(torch.randn(B, C, N, generator=g))
    index = torch.randint(0, Ma - 5, (B, N), generator=g, dtype=torch.int32)        # the last clusters stay empty
    mask = torch.zeros(B, Ma)
    mask.scatter_(1, index.long(), 1.0)
    idx_d, mask_d = index.to(DEV), mask.to(DEV)

    def ref(d):
        gi = index_max_torch(d.float(), index, Ma)
        return d.gather(2, gi) * mask.double().unsqueeze(1)

    _run_pair(lambda d: tn._SegmentMax.apply(d, idx_d, Ma, mask_d), ref, [data])
    feats = torch.randn(B, C, Ma, generator=g)
    J = 3000
    jidx = torch.randint(0, Ma, (B, J), generator=g, dtype=torch.int32)
    jd = jidx.to(DEV)
    _run_pair(lambda f: tn._GatherCols.apply(f, jd), lambda f: f.gather(2, jidx.long().unsqueeze(1).expand(B, C, J)), [feats])
    y = torch.randn(B, C, Ma, 16, generator=g)
    _run_pair(lambda y: tn._GroupMax.apply(y), lambda y: y.max(dim=3)[0], [y])


def test_interpolate_attention_dropout_backward():
    from deepi2p_amd import train_net as tn
    g = torch.Generator().manual_seed(9)
    B, C, M, N = 2, 128, 128, 1500
    feats = torch.randn(B, C, M, generator=g)
    idx = torch.randint(0, M, (B, N, 3), generator=g, dtype=torch.int32)
    w = torch.rand(B, N, 3, generator=g)
    idx_d, w_d = idx.to(DEV), w.to(DEV)

    def ref(f):
        gathered = torch.gather(f.unsqueeze(3).expand(B, C, M, 3), 2, idx.long().unsqueeze(1).expand(B, C, N, 3))
        return (gathered * w.double().unsqueeze(1)).sum(3)

    _run_pair(lambda f: tn._Interpolate.apply(f, idx_d, w_d), ref, [feats])
    feat, score = torch.randn(B, 256, 80, generator=g), torch.randn(B, 80, M, generator=g)
    _run_pair(lambda f, s: tn._AttentionPool.apply(f, s), lambda f, s: torch.bmm(f, s) / f.shape[2], [feat, score])
    x = torch.randn(B, 64, 777, generator=g)
    mk = tn.dropout_mask((B, 64, 777), 0.5, 123, 0, DEV)
    keep = float(mk.float().mean())
    assert 0.48 < keep < 0.52
    assert torch.equal(mk, tn.dropout_mask((B, 64, 777), 0.5, 123, 0, DEV)) and not torch.equal(mk, tn.dropout_mask((B, 64, 777), 0.5, 123, 1, DEV))
    mc = mk.cpu().double()
    _run_pair(lambda x: tn._Dropout.apply(x, mk, 2.0), lambda x: x * mc * 2.0, [x])


def _hip_step(g, dropouts):
    """Train-mode forward + loss + backward of the product on the fixture's inputs -> (params dict, losses)."""
    from deepi2p_amd import train_net as tn
    from deepi2p_amd.synthetic import OptLike, random_state_dict
    from deepi2p_amd.training import classifier_loss
    opt = OptLike(g["N"], g["H"], g["W"], True)
    P = {k: v.to(DEV) for k, v in random_state_dict(opt, g["weight_seed"]).items()}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    inputs = [t.to(DEV) for t in g["inputs"]]
    scores = tn.keypoint_detector(P, opt, *inputs, dropouts=[m.to(DEV) for m in dropouts])
    L = classifier_loss(scores[:, 0:2].contiguous().detach(), g["coarse_labels"].to(DEV), scores[:, 2:].contiguous().detach(),
                        g["fine_labels"].to(DEV))
    scores.backward(torch.cat((L["d_coarse"], L["d_fine"]), dim=1))
    return P, scores, L


def test_full_train_step_vs_oracle_and_reference_fixture():
    """Train-mode BatchNorm makes the gradient of a freshly initialised 34-layer network ill-conditioned: two fp32 evaluations of
    the SAME graph (torch fp32 vs torch fp64) differ by ~0.4 % (median over the parameters) and up to a few per cent.  The bar is
    therefore relative to that yardstick: the HIP gradients must be as close to the fp64 oracle as torch's own fp32 evaluation is."""
    torch.set_num_threads(8)
    g = tg.load()
    z = g["z"]
    P, scores, L = _hip_step(g, g["masks"])
    sd64, (c64, f64), L64 = tg.oracle_step(g, torch.float64)
    sd32, _, _ = tg.oracle_step(g, torch.float32)
    assert abs(float(L["loss"]) - L64["loss"]) <= 1e-4 * abs(L64["loss"])
    assert abs(float(L["loss"]) - float(z["loss"])) <= 1e-4 * abs(float(z["loss"]))
    _close(scores[:, 0:2], c64, 2e-4, "train-mode coarse scores")
    _close(scores[:, 2:], f64, 2e-4, "train-mode fine scores")
    np.testing.assert_allclose(scores[:, 0:2].detach().cpu().numpy()[:, :, ::8], z["coarse_sub"], rtol=0, atol=3e-4 * np.abs(z["coarse_sub"]).max())
    names = g["names"]
    e_hip, e_t32, e_fix = [], [], []
    num = den_h = den_r = 0.0
    for i, k in enumerate(names):
        gh, g64, g32 = P[k].grad.detach().cpu().double(), sd64[k].grad, sd32[k].grad.double()
        if tg.zero_expected(z, i):
            assert float(gh.abs().max()) < tg.ZERO_GRAD_ABSMAX, k
            continue
        scale = float(g64.abs().max())
        e_hip.append(float((gh - g64).abs().max()) / scale)
        e_t32.append(float((g32 - g64).abs().max()) / scale)
        e_fix.append(abs(float(gh.norm()) - z["grad_digest"][i][0]) / z["grad_digest"][i][0])
        num += float((gh * g64).sum()); den_h += float((gh * gh).sum()); den_r += float((g64 * g64).sum())
    for k in g["unused"]:
        assert P[k].grad is None
    e_hip, e_t32, e_fix = np.array(e_hip), np.array(e_t32), np.array(e_fix)
    cos = num / (den_h * den_r) ** 0.5
    print("gradient error vs fp64 oracle / abs-max: HIP median %.3g p90 %.3g max %.3g | torch fp32 median %.3g p90 %.3g max %.3g | cosine %.6f"
          % (np.median(e_hip), np.percentile(e_hip, 90), e_hip.max(), np.median(e_t32), np.percentile(e_t32, 90), e_t32.max(), cos))
    print("l2 of each gradient vs the reference-run fixture: median %.3g max %.3g" % (np.median(e_fix), e_fix.max()))
    assert np.median(e_hip) <= 2.0 * np.median(e_t32) + 1e-4
    assert np.percentile(e_hip, 90) <= 2.0 * np.percentile(e_t32, 90) + 1e-4
    assert e_hip.max() <= 3.0 * e_t32.max() + 1e-3
    assert cos >= 0.9995
    assert np.median(e_fix) <= 2e-2 and e_fix.max() <= 0.15
    for j, k in enumerate(z["buffer_names"]):
        d, _ = tg.digest(P[str(k)])
        assert abs(d[0] - z["buffer_digest"][j][0]) <= 1e-3 * z["buffer_digest"][j][0] + 1e-6, k


def test_trainer_optimises_and_eval_path_follows():
    from deepi2p_amd import networks, synthetic
    from deepi2p_amd.training import ClassifierTrainer
    B, N, H, W = 4, 2048, 64, 128
    opt = synthetic.OptLike(N, H, W, True)
    opt.lr, opt.coarse_loss_alpha = 1e-3, 50.0
    det = networks.KeypointDetector(opt)
    det.load_state_dict(synthetic.random_state_dict(opt, 5))
    det = det.to(DEV)
    b = synthetic.make_batch(3, B, N=N, H=H, W=W)
    t = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(DEV) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    K = torch.from_numpy(b["K"]).float().to(DEV)
    Pgt = torch.from_numpy(b["P_gt"][:, :3, :]).float().contiguous().to(DEV)
    tr = ClassifierTrainer(det, opt)
    before = tr.flat.clone()
    losses = []
    for _ in range(8):
        L = tr.optimize(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"], K, Pgt)
        losses.append(float(L["loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert float((tr.flat - before).abs().max()) > 0
    # the BatchNorm step counters (round 6: bumped by one multi-tensor launch per forward instead of one launch per layer): 8 forwards = 8
    nbt = [v for k, v in det.state_dict().items() if k.endswith("num_batches_tracked")]
    assert len(nbt) > 40 and all(int(v) == 8 for v in nbt), sorted({int(v) for v in nbt})
    ev = tr.test_model(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"], K, Pgt)
    assert np.isfinite(float(ev["loss"])) and 0.0 <= float(ev["coarse_accuracy"]) <= 1.0
    print("losses", ["%.3f" % v for v in losses], "eval loss %.3f coarse acc %.3f" % (float(ev["loss"]), float(ev["coarse_accuracy"])))


def test_graphed_step_equals_eager_step():
    """ClassifierTrainer.optimize_graphed replays zero-gradients + forward + losses + backward from one captured hipGraph: the same kernels on the
    same values -- two trainers from the same weights, one eager and one graphed, hold the same parameters, BatchNorm buffers and losses after
    every step (the capture's warm-up passes must leave no trace in the BatchNorm statistics)."""
    from deepi2p_amd import networks, synthetic
    from deepi2p_amd.training import ClassifierTrainer
    B, N, H, W = 2, 2048, 64, 128
    opt = synthetic.OptLike(N, H, W, True)
    opt.lr, opt.coarse_loss_alpha = 1e-3, 50.0
    sd = synthetic.random_state_dict(opt, 11)
    trs = []
    for _ in range(2):
        det = networks.KeypointDetector(opt)
        det.load_state_dict(sd)
        trs.append(ClassifierTrainer(det.to(DEV), opt, seed=3))
    for step in range(4):
        b = synthetic.make_batch(40 + step, B, N=N, H=H, W=W)
        t = [torch.from_numpy(np.ascontiguousarray(b[k])).to(DEV) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
        K = torch.from_numpy(b["K"]).float().to(DEV)
        Pgt = torch.from_numpy(np.ascontiguousarray(b["P_gt"][:, :3, :])).float().to(DEV)
        Le = trs[0].optimize(*t, K, Pgt)
        Lg = trs[1].optimize_graphed(*t, K, Pgt)
        assert float(Le["loss"]) == float(Lg["loss"]), (step, float(Le["loss"]), float(Lg["loss"]))
        assert torch.equal(trs[0].flat, trs[1].flat), step
    for (ka, va), (kb, vb) in zip(trs[0].detector.state_dict().items(), trs[1].detector.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


def test_gradient_sinks_equal_autograd_accumulation():
    """ClassifierTrainer makes the backward kernels write each parameter's gradient straight into its slice of the flat buffer;
    the result must equal what autograd hands back for the same kernels (tolerance 1e-5, kept from when the max-pool backward added
    overlapping windows with float atomics; every backward kernel is deterministic now)."""
    from deepi2p_amd import networks, synthetic, train_net as tn
    from deepi2p_amd.training import ClassifierTrainer, classifier_loss
    B, N, H, W = 2, 1024, 64, 128
    opt = synthetic.OptLike(N, H, W, True)
    sd = synthetic.random_state_dict(opt, 9)
    b = synthetic.make_batch(5, B, N=N, H=H, W=W)
    t = [torch.from_numpy(np.ascontiguousarray(b[k])).to(DEV) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    K = torch.from_numpy(b["K"]).float().to(DEV)
    Pgt = torch.from_numpy(np.ascontiguousarray(b["P_gt"][:, :3, :])).float().to(DEV)
    masks = [tn.dropout_mask((B, 256, N), 0.5, 77, i, DEV) for i in range(2)]
    det = networks.KeypointDetector(opt)
    det.load_state_dict(sd)
    tr = ClassifierTrainer(det.to(DEV), opt)
    tr.flat_grad.zero_()
    scores, L = tr.forward_pass(*t, K, Pgt, True, masks)
    scores.backward(torch.cat((L["d_coarse"], L["d_fine"]), dim=1))
    P = {k: v.to(DEV) for k, v in sd.items()}
    for k, v in P.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    s2 = tn.keypoint_detector(P, opt, *t, dropouts=masks)
    assert torch.equal(s2, scores)
    s2.backward(torch.cat((L["d_coarse"], L["d_fine"]), dim=1))
    for name, p in det.named_parameters():
        if P[name].grad is None:
            assert float(p.grad.abs().max()) == 0.0, name
        else:
            _close(p.grad, P[name].grad, 1e-5, name)


def test_mmclassifer_optimize_api():
    """The reference's training entry points on the module mirror (multimodal_classifier.py:213-223,263-277): set_input -> optimize()
    records train_loss_dict / train_accuracy, test_model() the eval-mode ones, update_learning_rate() reaches the optimiser."""
    from deepi2p_amd import synthetic
    from deepi2p_amd.networks import MMClassifer
    B, N, H, W = 2, 1024, 64, 128
    opt = synthetic.OptLike(N, H, W, True)
    opt.device, opt.lr, opt.coarse_loss_alpha = torch.device(DEV), 1e-3, 50.0
    m = MMClassifer(opt)
    m.detector.load_state_dict(synthetic.random_state_dict(opt, 2))
    b = synthetic.make_batch(8, B, N=N, H=H, W=W)
    m.set_input(*[torch.from_numpy(np.ascontiguousarray(b[k])) for k in ("pc", "intensity", "sn", "node_a", "node_b")],
                torch.from_numpy(np.ascontiguousarray(b["P_gt"][:, :3, :])).float(), torch.from_numpy(b["img"]), torch.from_numpy(b["K"]).float())
    m.optimize()
    first = float(m.train_loss_dict["loss"])
    for _ in range(4):
        m.optimize()
    assert set(m.train_loss_dict) == {"loss", "coarse", "fine"} and set(m.train_accuracy) == {"coarse_accuracy", "fine_accuracy"}
    assert np.isfinite(first) and float(m.train_loss_dict["loss"]) < first
    m.update_learning_rate(0.5)
    assert m._trainer_obj.adam.lr == 5e-4
    m.test_model()
    assert np.isfinite(float(m.test_loss_dict["loss"])) and 0.0 <= float(m.test_accuracy["coarse_accuracy"]) <= 1.0
    labels = m.inference_pass()                      # the inference kernels pick up the updated weights
    assert labels[0].shape == (B, N)


def test_branch_streams_bit_identical():
    """The image branch on a second stream (train_net.keypoint_detector(branch_streams=True), what ClassifierTrainer runs) changes no
    kernel and no summation order: scores and every parameter gradient are bit-identical to the single-stream pass, run after run."""
    from deepi2p_amd import synthetic, train_net as tn
    B, N, H, W = 2, 1024, 64, 128
    opt = synthetic.OptLike(N, H, W, True)
    sd = synthetic.random_state_dict(opt, 4)
    b = synthetic.make_batch(6, B, N=N, H=H, W=W)
    t = [torch.from_numpy(np.ascontiguousarray(b[k])).to(DEV) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    masks = [tn.dropout_mask((B, 256, N), 0.5, 5, i, DEV) for i in range(2)]
    results = []
    for branch in (False, True, True):
        P = {k: v.to(DEV).clone() for k, v in sd.items()}
        for k, v in P.items():
            if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        s = tn.keypoint_detector(P, opt, *t, dropouts=masks, branch_streams=branch)
        g = torch.Generator(device="cpu").manual_seed(0)
        d = torch.randn(s.shape, generator=g).to(DEV)
        s.backward(d)
        torch.cuda.synchronize()
        results.append((s.detach().clone(), {k: v.grad.clone() for k, v in P.items() if v.requires_grad and v.grad is not None},
                        {k: v.clone() for k, v in P.items() if k.endswith(("running_mean", "running_var"))}))
    ref = results[0]
    for s, grads, stats in results[1:]:
        assert torch.equal(s, ref[0])
        assert set(grads) == set(ref[1])
        for k in grads:
            assert torch.equal(grads[k], ref[1][k]), k
        for k in stats:
            assert torch.equal(stats[k], ref[2][k]), k
