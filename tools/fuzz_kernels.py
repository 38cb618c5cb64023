"""Randomised cross-check of the convolution and pointwise kernels against torch (GPU box).  Not a unit test: a wider
net than the parametrised cases (shapes around every dispatch boundary: vector / scalar stager, stride 1 / 2, split-K,
stem, ragged M / N / K, gathered and group sources)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from deepi2p_amd import _lib, ops

dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
g = torch.Generator().manual_seed(int(os.environ.get("SEED", 0)))
tol = lambda ref, K: 3e-6 * (K ** 0.5) * float(ref.abs().max()) + 1e-6
bad = 0
n_conv = int(os.environ.get("N_CONV", 150))
for it in range(n_conv):
    k = int(rng.choice([1, 3, 3, 3, 7]))
    s = int(rng.choice([1, 1, 2]))
    Cin = int(rng.choice([3, 5, 16, 32, 64, 96, 128, 256])) if k != 7 else int(rng.choice([3, 4]))
    Cout = int(rng.choice([4, 7, 32, 36, 64, 68, 128, 132, 256, 512]))
    H, W = int(rng.integers(1, 14)), int(rng.choice([4, 5, 8, 12, 16, 20, 32, 64]))
    B = int(rng.integers(1, 5))
    p = k // 2
    if (H + 2 * p - k) // s + 1 < 1 or (W + 2 * p - k) // s + 1 < 1:
        continue
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv2d(x, w, None, stride=s, padding=p) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = torch.randn(ref0.shape, generator=g)
    ref = torch.relu(ref0 + res)
    tap = Cin % 16 == 0
    Wt = (w.permute(2, 3, 1, 0).reshape(-1, Cout) if tap else w.reshape(Cout, -1).t()).contiguous().to(dev)
    y = ops.conv2d(x.to(dev), Wt, sc.to(dev), sh.to(dev), k, k, s, p, True, residual=res.to(dev), tap_major=tap).cpu()
    err = float((y - ref).abs().max())
    if not err <= tol(ref0, Cin * k * k):
        bad += 1
        print("CONV MISMATCH", dict(B=B, Cin=Cin, H=H, W=W, Cout=Cout, k=k, s=s), err)
# Winograd (both kernels, all blockings), the direct stem, and the training-side convolution gradients on random shapes
n_w = int(os.environ.get("N_WINO", 60))
for it in range(n_w):
    Cin, Cout = int(rng.choice([16, 32, 64, 96, 128, 256])), int(rng.choice([32, 64, 96, 128, 256]))
    H, W, B = int(rng.integers(1, 24)), int(rng.choice([4, 6, 8, 10, 16, 32, 34, 64])), int(rng.integers(1, 5))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref0 = F.conv2d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    res = torch.randn(ref0.shape, generator=g)
    ref = torch.relu(ref0 + res.double())
    U = ops.winograd_weights(w.to(dev))
    for cob, kc, reg in ((32, 8, 1), (32, 4, 1), (64, 8, 1), (0, 0, 2), (0, 0, 3)):
        if cob == 64 and Cout % 64:
            continue
        with _lib.option("wino_cob", cob), _lib.option("wino_kc", kc), _lib.option("wino_reg", reg):
            y = ops.conv3x3_winograd(x.to(dev), U, sc.to(dev), sh.to(dev), True, residual=res.to(dev)).cpu()
        err = float((y.double() - ref).abs().max())
        if not err <= 2 * tol(ref0.float(), Cin * 9):
            bad += 1
            print("WINOGRAD MISMATCH", dict(B=B, Cin=Cin, H=H, W=W, Cout=Cout, cob=cob, kc=kc, reg=reg), err)
for it in range(int(os.environ.get("N_STEM", 20))):
    H, W, B = int(rng.integers(1, 70)), int(rng.integers(1, 300)), int(rng.integers(1, 4))
    x = torch.rand(B, 3, H, W, generator=g) * 255
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = torch.relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=3) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    y = ops.conv_stem(x.to(dev), ops.stem_weights(w.to(dev)), sc.to(dev), sh.to(dev), True).cpu()
    if not float((y.double() - ref).abs().max()) <= tol(ref.float(), 147):
        bad += 1
        print("STEM MISMATCH", dict(B=B, H=H, W=W), float((y.double() - ref).abs().max()))
from deepi2p_amd import train_net as tn
for it in range(int(os.environ.get("N_GRAD", 30))):
    k, s = (3, int(rng.choice([1, 2]))) if rng.random() < 0.7 else (1, 2)
    Cin, Cout = int(rng.choice([3, 16, 17, 64, 128])), int(rng.choice([5, 32, 64, 128]))
    H, W, B = int(rng.integers(2, 20)), int(rng.choice([4, 7, 8, 16, 32])), int(rng.integers(1, 4))
    p = k // 2
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride=s, padding=p)
    ct = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(ct)
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    yd = tn._Conv2d.apply(xd, wd, s, p)
    yd.backward(ct.float().to(dev))
    for name, a, b_ in (("y", yd, yr), ("dx", xd.grad, xr.grad), ("dW", wd.grad, wr.grad)):
        e = float((a.detach().cpu().double() - b_.detach()).abs().max()) / max(float(b_.detach().abs().max()), 1e-30)
        if not e <= 5e-5:
            bad += 1
            print("CONV GRAD MISMATCH", name, dict(B=B, Cin=Cin, H=H, W=W, Cout=Cout, k=k, s=s), e)
n_pw = int(os.environ.get("N_PW", 150))
for it in range(n_pw):
    B = int(rng.integers(1, 4))
    N = int(rng.choice([4, 8, 12, 100, 128, 132, 260, 1000, 2048]))
    M = int(rng.choice([2, 4, 32, 36, 64, 68, 100, 128, 132, 256]))
    nsrc = int(rng.integers(1, 4))
    srcs, cols = [], []
    grp = int(rng.choice([2, 4])) if N % 4 == 0 else 1
    for i in range(nsrc):
        C = int(rng.integers(1, 70))
        mode = int(rng.choice([0, 0, 1, 2])) if N % grp == 0 else int(rng.choice([0, 1]))
        if mode == 0:
            t = torch.randn(B, C, N, generator=g); srcs.append(ops.Src(t.to(dev))); cols.append(t)
        elif mode == 1:
            Mn = int(rng.integers(1, 50)); t = torch.randn(B, C, Mn, generator=g)
            gi = torch.randint(0, Mn, (B, N), generator=g, dtype=torch.int32)
            srcs.append(ops.Src(t.to(dev), _lib.SRC_GATHER, gidx=gi.to(dev))); cols.append(torch.gather(t, 2, gi.long().unsqueeze(1).expand(B, C, N)))
        else:
            t = torch.randn(B, C, N // grp, generator=g); srcs.append(ops.Src(t.to(dev), _lib.SRC_GROUP, group=grp)); cols.append(t.repeat_interleave(grp, dim=2))
    full = torch.cat(cols, 1)
    K = full.shape[1]
    Wm = torch.randn(M, K, generator=g) / K ** 0.5
    sc, sh, bias = torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g), torch.randn(B, M, generator=g)
    y = ops.pointwise_gemm(srcs, Wm.t().contiguous().to(dev), M, N, scale=sc.to(dev), shift=sh.to(dev), relu=True, batch_bias=bias.to(dev)).cpu()
    ref = torch.relu((torch.einsum("mk,bkn->bmn", Wm, full) + bias.unsqueeze(2)) * sc.view(1, M, 1) + sh.view(1, M, 1))
    err = float((y - ref).abs().max())
    if not err <= tol(ref, K):
        bad += 1
        print("PW MISMATCH", dict(B=B, N=N, M=M, K=K, modes=[s_.mode for s_ in srcs]), err)
print("fuzz: %d conv + %d winograd + stem + conv-gradient + %d pointwise cases, %d mismatches" % (n_conv, n_w, n_pw, bad))
sys.exit(1 if bad else 0)
