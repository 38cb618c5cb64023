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
print("fuzz: %d conv + %d pointwise cases, %d mismatches" % (n_conv, n_pw, bad))
sys.exit(1 if bad else 0)
