"""GPU parity of the fp32-MFMA contractions (pointwise GEMM with its virtual-concat loader and fused
epilogues, implicit-GEMM conv, pooling) vs plain PyTorch fp32 on CPU.  Tolerance: fp32 round-off of a
K-term dot product, |err| <= 2e-6 * K^0.5 * max|out| + 1e-6 (stated per test)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from deepi2p_amd import _lib

pytestmark = pytest.mark.gpu


def _tol(ref, K):
    return 3e-6 * (K ** 0.5) * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("B,M,K,N", [(2, 32, 7, 1000), (1, 64, 64, 4096), (2, 128, 96, 777), (3, 256, 67, 2048),
                                      (1, 1024, 300, 128), (2, 2, 128, 513), (1, 82, 256, 640), (1, 33, 17, 65)])
def test_pointwise_gemm_dense(dev, B, M, K, N):
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x, W = torch.randn(B, K, N, generator=g), torch.randn(M, K, generator=g) / K ** 0.5
    scale, shift = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)
    y = ops.pointwise_gemm([ops.Src(x.to(dev))], W.t().contiguous().to(dev), M, N, scale=scale.to(dev),
                           shift=shift.to(dev), relu=True).cpu()
    ref = torch.relu(torch.einsum("mk,bkn->bmn", W, x) * scale.view(1, M, 1) + shift.view(1, M, 1))
    assert (y - ref).abs().max() <= _tol(ref, K)


@pytest.mark.parametrize("B,chans,M,N", [(3, (24, 40, 5), 100, 1500), (2, (96,), 128, 20480), (1, (7,), 32, 260),
                                        (2, (64, 64), 512, 2048), (2, (33,), 4, 8), (2, (33,), 68, 12)])
def test_pointwise_gemm_dense_vector_path(dev, B, chans, M, N):
    """16-byte staged path (all sources dense, N % 4 == 0, M % 4 == 0): ragged K, M and N against torch and against the
    scalar stager."""
    from deepi2p_amd import _lib, ops
    g = torch.Generator().manual_seed(11 + N)
    xs = [torch.randn(B, c, N, generator=g) for c in chans]
    K = sum(chans)
    W = torch.randn(M, K, generator=g) / K ** 0.5
    scale, shift = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)
    bias = torch.randn(B, M, generator=g)
    Wt = W.t().contiguous().to(dev)
    args = dict(scale=scale.to(dev), shift=shift.to(dev), relu=True, batch_bias=bias.to(dev))
    srcs = [ops.Src(x.to(dev)) for x in xs]
    y = ops.pointwise_gemm(srcs, Wt, M, N, **args).cpu()
    ref = torch.relu((torch.einsum("mk,bkn->bmn", W, torch.cat(xs, 1)) + bias.unsqueeze(2)) * scale.view(1, M, 1) + shift.view(1, M, 1))
    assert (y - ref).abs().max() <= _tol(ref, K)
    with _lib.option("pw_novec", 1):
        y0 = ops.pointwise_gemm(srcs, Wt, M, N, **args).cpu()
    with _lib.option("conv_depth1", 1):
        assert torch.equal(ops.pointwise_gemm(srcs, Wt, M, N, **args).cpu(), y)       # prefetch depth does not change a bit
    assert (y - y0).abs().max() <= _tol(ref, K)
    yt = ops.pointwise_gemm(srcs, Wt, M, N, transpose_out=True, **args).cpu()
    assert torch.equal(yt.transpose(1, 2), y)


def test_pointwise_gemm_transpose_detecting(dev):
    """A = identity-like weights with an ASYMMETRIC input: catches row/col swaps in the MFMA C layout."""
    from deepi2p_amd import ops
    M = K = 64
    N = 128
    x = (torch.arange(K).view(1, K, 1) * 1000.0 + torch.arange(N).view(1, 1, N)).contiguous()
    y = ops.pointwise_gemm([ops.Src(x.to(dev))], torch.eye(K).to(dev), M, N).cpu()
    assert torch.equal(y, x)


def test_pointwise_gemm_concat_gather_group_bias(dev):
    from deepi2p_amd import _lib, ops
    g = torch.Generator().manual_seed(1)
    B, N, Mn, grp = 2, 2048, 128, 16
    a = torch.randn(B, 3, N, generator=g)                       # dense
    tab = torch.randn(B, 64, Mn, generator=g)                   # gathered by index
    gi = torch.randint(0, Mn, (B, N), generator=g, dtype=torch.int32)
    grpsrc = torch.randn(B, 20, N // grp, generator=g)          # group-broadcast
    bvec = torch.randn(B, 40, generator=g)                      # broadcast channels -> batch bias
    K, M = 3 + 64 + 20 + 40, 96
    W = torch.randn(M, K, generator=g) / K ** 0.5
    Wt = W.t().contiguous().to(dev)
    Wt_dense = torch.cat((Wt[:87],), 0).contiguous()
    bias = ops.batch_gemv(Wt, 87, bvec.to(dev))
    y = ops.pointwise_gemm([ops.Src(a.to(dev)), ops.Src(tab.to(dev), _lib.SRC_GATHER, gidx=gi.to(dev)),
                            ops.Src(grpsrc.to(dev), _lib.SRC_GROUP, group=grp)], Wt_dense, M, N, batch_bias=bias).cpu()
    full = torch.cat((a, torch.gather(tab, 2, gi.long().unsqueeze(1).expand(B, 64, N)),
                      grpsrc.repeat_interleave(grp, dim=2), bvec.unsqueeze(2).expand(B, 40, N)), dim=1)
    ref = torch.einsum("mk,bkn->bmn", W, full)
    assert (y - ref).abs().max() <= _tol(ref, K)
    with _lib.option("pw_novec", 1):            # scalar stager: same operands element by element
        y0 = ops.pointwise_gemm([ops.Src(a.to(dev)), ops.Src(tab.to(dev), _lib.SRC_GATHER, gidx=gi.to(dev)),
                                 ops.Src(grpsrc.to(dev), _lib.SRC_GROUP, group=grp)], Wt_dense, M, N, batch_bias=bias).cpu()
    assert (y - y0).abs().max() <= _tol(ref, K)
    # fused max over groups of 16 consecutive columns (torch.max(dim=3) of layers_pc.py:811,816)
    ym = ops.pointwise_gemm([ops.Src(full.to(dev))], Wt, M, N, relu=True, group_max=grp).cpu()
    refm = torch.relu(ref).view(B, M, N // grp, grp).max(dim=3)[0]
    assert (ym - refm).abs().max() <= _tol(ref, K)


def test_pointwise_gemm_gathered_add(dev):
    """per_point_pn layer 0: W @ cat(interp_a, interp_b, x) == gathered per-node products + W_x @ x."""
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(2)
    B, N, Ma, Mb, M = 2, 1500, 128, 128, 128
    fa, fb = torch.randn(B, 128, Ma, generator=g), torch.randn(B, 512, Mb, generator=g)
    x = torch.randn(B, 96, N, generator=g)
    ia = torch.randint(0, Ma, (B, N, 3), generator=g, dtype=torch.int32)
    ib = torch.randint(0, Mb, (B, N, 3), generator=g, dtype=torch.int32)
    wa, wb = torch.rand(B, N, 3, generator=g), torch.rand(B, N, 3, generator=g)
    W = torch.randn(M, 736, generator=g) / 736 ** 0.5
    scale, shift = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)
    Wt = W.t().contiguous().to(dev)
    Ga = ops.pointwise_gemm([ops.Src(fa.to(dev))], Wt[0:128], M, Ma, transpose_out=True)      # node-major [B,Ma,M]
    Gb = ops.pointwise_gemm([ops.Src(fb.to(dev))], Wt[128:640], M, Mb, transpose_out=True)
    Ga_plain = ops.pointwise_gemm([ops.Src(fa.to(dev))], Wt[0:128], M, Ma)
    assert torch.equal(Ga.transpose(1, 2), Ga_plain)                                            # same values, transposed store
    y = ops.pointwise_gemm([ops.Src(x.to(dev))], Wt[640:736], M, N, scale=scale.to(dev), shift=shift.to(dev), relu=True,
                           gathered=[(Ga, ia.to(dev), wa.to(dev)), (Gb, ib.to(dev), wb.to(dev))]).cpu()

    def interp(f, i, w):
        Bc, C, Mn = f.shape
        gth = torch.gather(f.unsqueeze(3).expand(Bc, C, Mn, 3), 2, i.long().unsqueeze(1).expand(Bc, C, N, 3))
        return (w.unsqueeze(1) * gth).sum(3)
    full = torch.cat((interp(fa, ia, wa), interp(fb, ib, wb), x), dim=1)
    ref = torch.relu(torch.einsum("mk,bkn->bmn", W, full) * scale.view(1, M, 1) + shift.view(1, M, 1))
    assert (y - ref).abs().max() <= _tol(ref, 736) * 2


def test_attention_pool(dev):
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(3)
    feat, score = torch.randn(2, 256, 320, generator=g), torch.randn(2, 320, 128, generator=g)
    out = ops.attention_pool(feat.to(dev), score.to(dev)).cpu()
    ref = torch.mean(feat.unsqueeze(3) * score.unsqueeze(1), dim=2)           # the reference's formulation
    assert (out - ref).abs().max() <= _tol(ref, 320)


@pytest.mark.parametrize("B,Cin,H,W,Cout,k,s,p", [(2, 3, 64, 128, 64, 7, 2, 3), (2, 64, 16, 32, 64, 3, 1, 1),
                                                  (1, 64, 16, 32, 128, 3, 2, 1), (2, 64, 16, 32, 128, 1, 2, 0),
                                                  (3, 256, 5, 16, 512, 3, 2, 1), (1, 512, 5, 16, 512, 3, 1, 1),
                                                  (1, 5, 9, 11, 7, 3, 1, 1), (2, 32, 3, 4, 32, 3, 1, 1),
                                                  (1, 64, 1, 8, 64, 3, 1, 1), (3, 32, 7, 12, 36, 1, 1, 0),
                                                  (2, 32, 6, 8, 64, 3, 2, 1), (2, 64, 40, 128, 128, 3, 2, 1), (2, 128, 20, 64, 256, 1, 2, 0)])
def test_conv2d(dev, B, Cin, H, W, Cout, k, s, p):
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv2d(x, w, None, stride=s, padding=p) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = torch.randn(ref0.shape, generator=g)
    Wt = w.reshape(Cout, -1).t().contiguous().to(dev)
    y = ops.conv2d(x.to(dev), Wt, scale.to(dev), shift.to(dev), k, k, s, p, True, residual=res.to(dev)).cpu()
    ref = torch.relu(ref0 + res)
    assert y.shape == ref.shape
    assert (y - ref).abs().max() <= _tol(ref0, Cin * k * k)
    y2 = ops.conv2d(x.to(dev), Wt, scale.to(dev), shift.to(dev), k, k, s, p, False).cpu()
    assert (y2 - ref0).abs().max() <= _tol(ref0, Cin * k * k)
    if Cin % 16 == 0:   # tap-major weight packing (kh,kw,ci): same result (split-K with its ordered reduce pass where the shape asks for it)
        Wtap = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous().to(dev)
        y3 = ops.conv2d(x.to(dev), Wtap, scale.to(dev), shift.to(dev), k, k, s, p, True, residual=res.to(dev), tap_major=True).cpu()
        assert (y3 - ref).abs().max() <= _tol(ref0, Cin * k * k)
        with _lib.option("conv_depth1", 1):          # depth-1 vs depth-2 register prefetch: same K order, same MFMA sequence
            y6 = ops.conv2d(x.to(dev), Wtap, scale.to(dev), shift.to(dev), k, k, s, p, True, residual=res.to(dev), tap_major=True).cpu()
        assert torch.equal(y6, y3)
        if s == 2:      # aligned 8-float window loads (default where the shape allows) vs four dword loads per staged row: the same values reach LDS
            with _lib.option("conv_s2scalar", 1):
                y7 = ops.conv2d(x.to(dev), Wtap, scale.to(dev), shift.to(dev), k, k, s, p, True, residual=res.to(dev), tap_major=True).cpu()
            assert torch.equal(y7, y3)
        with _lib.option("conv_nosplit", 1):
            y4 = ops.conv2d(x.to(dev), Wtap, scale.to(dev), shift.to(dev), k, k, s, p, True, residual=res.to(dev), tap_major=True).cpu()
        assert (y4 - y3).abs().max() <= _tol(ref0, Cin * k * k)
        y5 = ops.conv2d(x.to(dev), Wtap, scale.to(dev), shift.to(dev), k, k, s, p, True, residual=res.to(dev), tap_major=True).cpu()
        assert torch.equal(y5, y3)                                    # run-to-run identical (ordered split-K reduce, no atomics)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 64, 40, 128, 64), (2, 128, 20, 64, 128), (3, 256, 5, 16, 256), (1, 64, 7, 10, 96),
                                            (2, 32, 9, 12, 32), (1, 512, 5, 16, 512), (1, 64, 3, 4, 32)])
def test_conv3x3_winograd(dev, B, Cin, H, W, Cout):
    """Fused Winograd F(2x2,3x3) vs an fp64 convolution: same tolerance class as the direct kernel (its rounding error is a small
    multiple of the direct form's), odd heights / widths, residual + ReLU epilogue, both output-channel blockings."""
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(Cin * 3 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = (F.conv2d(x.double(), w.double(), None, stride=1, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    res = torch.randn(ref0.shape, generator=g)
    ref = torch.relu(ref0 + res.double())
    U = ops.winograd_weights(w.to(dev))
    direct = ops.conv2d(x.to(dev), w.reshape(Cout, -1).t().contiguous().to(dev), scale.to(dev), shift.to(dev), 3, 3, 1, 1, True, residual=res.to(dev)).cpu()
    e_direct = float((direct.double() - ref).abs().max())
    variants = [(32, 8, 1), (32, 4, 1), (0, 0, 2), (0, 0, 3)] + ([(64, 8, 1), (64, 4, 1)] if Cout % 64 == 0 else [])
    for cob, kc, reg in variants:       # reg 1: LDS-panel kernel (co-block, K-step), 2 / 3: register-resident kernel with 4 / 2 waves
        with _lib.option("wino_cob", cob), _lib.option("wino_kc", kc), _lib.option("wino_reg", reg):
            y = ops.conv3x3_winograd(x.to(dev), U, scale.to(dev), shift.to(dev), True, residual=res.to(dev)).cpu()
            y2 = ops.conv3x3_winograd(x.to(dev), U, scale.to(dev), shift.to(dev), False).cpu()
        assert y.shape == ref.shape
        err = float((y.double() - ref).abs().max())
        assert err <= 2 * _tol(ref0.float(), Cin * 9), (cob, err, e_direct)
        assert float((y2.double() - ref0).abs().max()) <= 2 * _tol(ref0.float(), Cin * 9)
    print("winograd max error %.3g vs direct %.3g (tolerance %.3g)" % (err, e_direct, 2 * float(_tol(ref0.float(), Cin * 9))))


@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (1, 160, 512), (3, 33, 70), (1, 7, 9)])
def test_conv_stem(dev, B, H, W):
    """Direct 7x7/2 stem kernel vs an fp64 convolution (and the generic implicit-GEMM path), odd sizes and partial column tiles."""
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(B, 3, H, W, generator=g) * 255
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = torch.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    y = ops.conv_stem(x.to(dev), ops.stem_weights(w.to(dev)), scale.to(dev), shift.to(dev), True).cpu()
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= _tol(ref.float(), 147)
    y0 = ops.conv2d(x.to(dev), w.reshape(64, -1).t().contiguous().to(dev), scale.to(dev), shift.to(dev), 7, 7, 2, 3, True).cpu()
    assert float((y - y0).abs().max()) <= _tol(ref.float(), 147)


def test_pools(dev):
    from deepi2p_amd import ops
    x = torch.randn(2, 64, 32, 64)
    assert torch.equal(ops.maxpool3x3s2(x.to(dev)).cpu(), F.max_pool2d(x, 3, 2, 1))
    x2 = torch.randn(2, 7, 33, 65)
    assert torch.equal(ops.maxpool3x3s2(x2.to(dev)).cpu(), F.max_pool2d(x2, 3, 2, 1))
    y = torch.randn(3, 512, 5, 16)
    torch.testing.assert_close(ops.global_avgpool(y.to(dev)).cpu(), F.adaptive_avg_pool2d(y, 1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("B,chans,M,N,mode", [(2, (256,), 256, 2048, "dense"), (2, (256,), 512, 2048, "gather"), (2, (512,), 256, 2048, "gmax"),
                                              (3, (256, 512), 1024, 128, "bias"), (2, (64, 512, 256), 512, 128, "dense"), (1, (1024,), 512, 128, "transpose"),
                                              (2, (130,), 128, 132, "dense"), (1, (128,), 384, 20, "dense")])
def test_pointwise_gemm_bf16x3_as_accurate_as_fp32_mfma(dev, B, chans, M, N, mode):
    """The GEMM-shaped layers run on bf16 matrix instructions with both fp32 operands split EXACTLY into three bf16 terms (six bf16
    products per fp32 product, fp32 accumulation).  Against an fp64 contraction of the same fp32 operands the result must be as accurate as
    the fp32-MFMA kernel's -- err_bf16x3 <= err_fp32mfma up to the run-to-run spread of two different summation orders (max error within
    1.25x, rms error within 1.1x) -- and inside the tolerance of every other contraction test.  Shapes: the kNN-fusion layers, the node-level
    PointNets with concatenated sources, ragged K / N, every epilogue the fp32 entry point has."""
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(5 + M + N)
    xs = [torch.randn(B, c, N, generator=g) * (1.0 + 3.0 * torch.rand(1, c, 1, generator=g)) for c in chans]
    K = sum(chans)
    W = torch.randn(M, K, generator=g) / K ** 0.5
    scale, shift = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)
    Wt = W.t().contiguous().to(dev)
    kw = dict(scale=scale.to(dev), shift=shift.to(dev), relu=False)
    srcs = [ops.Src(x.to(dev)) for x in xs]
    x64 = torch.cat(xs, 1).double()
    if mode == "gather":
        idx = torch.randint(0, N, (B, N), generator=g, dtype=torch.int32)
        srcs = [ops.Src(xs[0].to(dev), mode=_lib.SRC_GATHER, gidx=idx.to(dev))]
        x64 = torch.gather(xs[0].double(), 2, idx.long().unsqueeze(1).expand(B, K, N))
    ref = torch.einsum("mk,bkn->bmn", W.double(), x64)
    if mode == "bias":
        bias = torch.randn(B, M, generator=g)
        kw["batch_bias"] = bias.to(dev)
        ref = ref + bias.double().unsqueeze(2)
    ref = ref * scale.double().view(1, M, 1) + shift.double().view(1, M, 1)
    if mode == "gmax":
        kw["group_max"] = 16
        ref = ref.view(B, M, N // 16, 16).max(dim=3)[0]
    if mode == "transpose":
        kw["transpose_out"] = True
        ref = ref.transpose(1, 2)
    y3 = ops.pointwise_gemm(srcs, Wt, M, N, x3=True, **kw).double().cpu()
    y1 = ops.pointwise_gemm(srcs, Wt, M, N, x3=False, **kw).double().cpu()
    e3, e1 = (y3 - ref).abs(), (y1 - ref).abs()
    assert float(e3.max()) <= _tol(ref, K) and float(e1.max()) <= _tol(ref, K)
    assert float(e3.max()) <= 1.25 * float(e1.max()) + 1e-9, (float(e3.max()), float(e1.max()))
    assert float(e3.pow(2).mean().sqrt()) <= 1.1 * float(e1.pow(2).mean().sqrt()) + 1e-12, (float(e3.pow(2).mean().sqrt()), float(e1.pow(2).mean().sqrt()))
    # the automatic choice takes the bf16x3 kernel for K >= 128, M % 128 == 0 and at least 8 workgroups of 128 x 128 per frame (and the knob switches it off)
    auto = ops.pointwise_gemm(srcs, Wt, M, N, **kw).double().cpu()
    assert torch.equal(auto, y3 if (K >= 128 and M % 128 == 0 and ((N + 127) // 128) * (M // 128) >= 8) else y1)
    with _lib.option("pw_x3", 0):
        assert torch.equal(ops.pointwise_gemm(srcs, Wt, M, N, **kw).double().cpu(), y1)


@pytest.mark.parametrize("grp,x3", [(16, False), (16, True), (8, False), (32, False)])
def test_group_max_epilogue_torch_max_semantics(dev, grp, x3):
    """max over groups of consecutive columns in the epilogue == torch.max(dim): exact on the kernel's own full output (also_full), NaN
    propagates to its group only; groups of 16 take the DPP path, other sizes the shuffle path."""
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(grp)
    B, K, M, N = 2, 256, 256, 2048
    x = torch.randn(B, K, N, generator=g)
    x[0, 5, 37] = float("nan")                       # poisons column 37 of frame 0 in every row
    x[1, :, 100] = -1e30                             # a very negative column must not win or lose anything
    Wt = (torch.randn(K, M, generator=g) / K ** 0.5).to(dev)
    full, mx = ops.pointwise_gemm([ops.Src(x.to(dev))], Wt, M, N, group_max=grp, also_full=True, x3=x3)
    only = ops.pointwise_gemm([ops.Src(x.to(dev))], Wt, M, N, group_max=grp, x3=x3)
    ref = full.view(B, M, N // grp, grp).max(dim=3)[0]
    assert torch.equal(torch.isnan(mx), torch.isnan(ref)) and bool(torch.isnan(mx[0, :, 37 // grp]).all())
    assert int(torch.isnan(mx).sum()) == M
    assert torch.equal(torch.nan_to_num(mx, nan=0.0), torch.nan_to_num(ref, nan=0.0))
    assert torch.equal(torch.nan_to_num(only, nan=0.0), torch.nan_to_num(mx, nan=0.0))


def test_bf16x3_split_cache_follows_the_operand_not_the_address(dev):
    """The split weights are cached per fp32 operand.  A freed operand's address is handed to the next allocation of the same size: the
    entry must not survive its tensor (it remembers the tensor weakly), and _PackedModule._invalidate() drops everything."""
    import gc
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(3)
    B, K, M, N = 1, 256, 256, 2048
    x = torch.randn(B, K, N, generator=g).to(dev)
    seen = set()
    for trial in range(4):
        Wt = (torch.randn(K, M, generator=g) / K ** 0.5).to(dev)
        seen.add(Wt.data_ptr())
        y3 = ops.pointwise_gemm([ops.Src(x)], Wt, M, N, x3=True)
        y1 = ops.pointwise_gemm([ops.Src(x)], Wt, M, N, x3=False)
        assert float((y3 - y1).abs().max()) <= 2 * _tol(y1, K), trial          # a stale split would be off by O(1)
        xh = x[:, 0:128].contiguous()         # a row slice of the operand (a view: same base tensor) is its own cache entry
        assert torch.equal(ops.pointwise_gemm([ops.Src(xh)], Wt[0:128], M, N, x3=True), ops.pointwise_gemm([ops.Src(xh)], Wt[0:128].clone(), M, N, x3=True))
        del Wt, y3, y1
        gc.collect()
    assert len(seen) < 4              # the allocator did recycle an address: the case the weak reference exists for
    ops.x3_invalidate()
    assert not ops._X3_CACHE


@pytest.mark.gpu
def test_bf16x3_cache_entries_live_and_die_with_their_operand(dev):
    """Round-4 advisor finding: the split weights were dropped by ANY module's invalidation while captured graphs of another model still
    pointed to them.  An entry now lives exactly as long as the fp32 operand it was made from: a foreign module's .to() / load_state_dict
    leaves it alone, deleting the operand removes it."""
    import gc
    from deepi2p_amd import ops, networks
    ops.x3_invalidate()
    g = torch.Generator().manual_seed(5)
    B, K, M, N = 1, 256, 256, 2048
    x = torch.randn(B, K, N, generator=g).to(dev)
    Wt = (torch.randn(K, M, generator=g) / K ** 0.5).to(dev)
    ops.pointwise_gemm([ops.Src(x)], Wt, M, N, x3=True)
    assert len(ops._X3_CACHE) == 1
    Wp = ops.x3_live_operands()[0]
    other = networks._PackedModule()
    other._invalidate()                                   # another model changing must not clear this operand's split
    other.to(dev)
    assert ops.x3_live_operands() and ops.x3_live_operands()[0] is Wp
    y = ops.pointwise_gemm([ops.Src(x)], Wt, M, N, x3=True)
    assert ops.x3_live_operands()[0] is Wp                # found again, not re-packed
    del Wt, y
    gc.collect()
    assert not ops._X3_CACHE                              # expired with the operand


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(2, 256), (3, 2048)])
def test_bf16x3_split_planes_chain_is_bit_identical_to_fp32_handover(dev, B, N):
    """A chain of bf16x3 layers (GeneralKNNFusionModule, layers_pc.py:779-818: 256 -> 256 (+ max over the 16 neighbours) -> 512 (+ gathered table)
    -> 256 (max only)) may hand its activations on ALREADY SPLIT (di2p_epilogue_t.planes_out -> di2p_pointwise_gemm_x3p): the planes must hold
    exactly the fp32 values the fp32 hand-over stores, and every consumer must produce the same bits from them."""
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(77 + N)
    x = (torch.randn(B, 256, N, generator=g) * (1.0 + 3.0 * torch.rand(1, 256, 1, generator=g))).to(dev)

    def layer(K, M):
        W = (torch.randn(K, M, generator=g) / K ** 0.5).to(dev)
        return W, (torch.rand(M, generator=g) + 0.5).to(dev), torch.randn(M, generator=g).to(dev)

    (W1, s1, h1), (W2, s2, h2), (W3, s3, h3) = layer(256, 256), layer(256, 512), layer(512, 256)
    tab = torch.randn(B, N // 16, 512, generator=g).to(dev)
    gidx = (torch.arange(N, dtype=torch.int32) // 16).unsqueeze(0).expand(B, -1).contiguous().to(dev)
    gat = [(tab, gidx.reshape(B, N, 1), None)]

    def chain(planes):
        y1, m1 = ops.pointwise_gemm([ops.Src(x)], W1, 256, N, scale=s1, shift=h1, relu=True, group_max=16, also_full=True, x3=True, planes_out=planes)
        y2 = ops.pointwise_gemm([y1 if planes else ops.Src(y1)], W2, 512, N, scale=s2, shift=h2, relu=True, gathered=gat, x3=True, planes_out=planes)
        y3 = ops.pointwise_gemm([y2 if planes else ops.Src(y2)], W3, 256, N, scale=s3, shift=h3, relu=False, group_max=16, x3=True)
        return y1, m1, y2, y3

    a1, am, a2, a3 = chain(False)
    p1, pm, p2, p3 = chain(True)
    assert isinstance(p1, ops.X3Planes) and isinstance(p2, ops.X3Planes) and p1.shape == (B, 256, N) and p2.shape == (B, 512, N)
    assert torch.equal(p1.float(), a1) and torch.equal(pm, am)
    assert torch.equal(p2.float(), a2)
    assert torch.equal(p3, a3)
    # a plain (no epilogue extras) planes consumer with a full-size fp32 output, too
    yp = ops.pointwise_gemm([p2], W3, 256, N)
    yf = ops.pointwise_gemm([ops.Src(a2)], W3, 256, N, x3=True)
    assert torch.equal(yp, yf)
    # the planes-source entry point has two kernels: 256-row tiles with both operands in LDS (M % 256 == 0: everything above) and 128-row
    # tiles (any M % 4 == 0; knob pw_x3_planes = 2 forces it) -- the same bits from both
    W4 = (torch.randn(512, 384, generator=g) / 512 ** 0.5).to(dev)
    assert torch.equal(ops.pointwise_gemm([p2], W4, 384, N), ops.pointwise_gemm([ops.Src(a2)], W4, 384, N, x3=True))
    # the general group-maximum path (not the 16-lane DPP one) and a per-frame bias row, planes out
    bias = torch.randn(B, 256, generator=g).to(dev)
    for grp in (8, 32):
        fa, fm = ops.pointwise_gemm([ops.Src(x)], W1, 256, N, scale=s1, shift=h1, batch_bias=bias, group_max=grp, also_full=True, x3=True)
        fp, pm2 = ops.pointwise_gemm([ops.Src(x)], W1, 256, N, scale=s1, shift=h1, batch_bias=bias, group_max=grp, also_full=True, x3=True, planes_out=True)
        assert torch.equal(fp.float(), fa) and torch.equal(pm2, fm)
    _lib.set_option("pw_x3_planes", 2)
    try:
        q1, qm, q2, q3 = chain(True)
        assert torch.equal(q1.t, p1.t) and torch.equal(qm, pm) and torch.equal(q2.t, p2.t) and torch.equal(q3, p3)
    finally:
        _lib.set_option("pw_x3_planes", 1)


@pytest.mark.gpu
def test_bf16x3_split_planes_argument_checks(dev):
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 256, 256, generator=g).to(dev)
    W = torch.randn(256, 256, generator=g).to(dev)
    with pytest.raises(RuntimeError):                                # N % 128 != 0
        ops.pointwise_gemm([ops.Src(x[:, :, :192].contiguous())], W, 256, 192, x3=True, planes_out=True)
    with pytest.raises(RuntimeError):                                # the fp32 kernel does not write planes
        ops.pointwise_gemm([ops.Src(x)], W, 256, 256, x3=False, planes_out=True)
    with pytest.raises(RuntimeError):                                # no transposed planes
        ops.pointwise_gemm([ops.Src(x)], W, 256, 256, x3=True, planes_out=True, transpose_out=True)
    pl = ops.pointwise_gemm([ops.Src(x)], W, 256, 256, x3=True, planes_out=True)
    with pytest.raises(RuntimeError):                                # planes are the ONLY source, with matching K and N
        ops.pointwise_gemm([pl, ops.Src(x)], torch.randn(512, 256).to(dev), 256, 256)
    with pytest.raises(RuntimeError):
        ops.pointwise_gemm([ops.Src(x), pl], torch.randn(512, 256).to(dev), 256, 256)
    with pytest.raises(RuntimeError):
        ops.pointwise_gemm([pl], torch.randn(128, 256).to(dev), 256, 256)
    # the C entry points refuse what the host layer would never send
    e = _lib.EpilogueT()
    e.group_max = 1
    e.planes_out = pl.t.data_ptr()
    y = torch.empty(1, 256, 256, device=dev)
    arr = ops._fill_srcs([ops.Src(x)])
    with pytest.raises(_lib.DeepI2PHipError):                        # fp32 entry point + planes_out
        _lib.call("di2p_pointwise_gemm", arr, 1, W.data_ptr(), y.data_ptr(), 1, 256, 256, 256, ctypes.byref(e), _lib.stream())
    Wp = ops.bf16x3_pack(W)
    with pytest.raises(_lib.DeepI2PHipError):                        # K % 32 != 0
        _lib.call("di2p_pointwise_gemm_x3p", pl.t.data_ptr(), Wp.data_ptr(), y.data_ptr(), 1, 256, 200, 256, None, _lib.stream())
    with pytest.raises(_lib.DeepI2PHipError):                        # Y == NULL without planes_out
        _lib.call("di2p_pointwise_gemm_x3p", pl.t.data_ptr(), Wp.data_ptr(), None, 1, 256, 256, 256, None, _lib.stream())
    assert _lib.load().di2p_bf16x3_planes_bytes(2, 256, 2048) == 2 * 3 * 256 * 2048 * 2
