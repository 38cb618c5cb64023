import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from deepi2p_amd import _lib, ops
dev = torch.device("cuda", 0)
B, Cin, H, W, Cout, s, cfg = 1, 32, 12, 64, 16, 2, 3
x = torch.zeros(B, Cin, H, W)
for r in range(H):
    for c in range(W):
        x[0, 0, r, c] = r * 100 + c + 1
wd = torch.zeros(Cout, Cin, 1, 1)
one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
Wpd = ops.bf16x3_pack(wd.reshape(Cout, Cin).t().contiguous().to(dev))
pat = x[0, 0].clone()
for ci, tap in [(c, t) for c in (0, 1, 7, 8, 9, 15, 16, 17, 24, 31) for t in (0, 2, 4)]:
    x = torch.zeros(B, Cin, H, W)
    x[0, ci] = pat
    w = torch.zeros(Cout, Cin, 3, 3)
    w[:, ci, tap // 3, tap % 3] = 1.0
    ref = F.conv2d(x, w, stride=s, padding=1)
    Wp = ops.bf16x3_pack(w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous().to(dev))
    with _lib.option("conv_x3_cfg", cfg):
        y, yd = ops.conv3x3_x3(x.to(dev), Wp, Cout, one, zero, 2, False, downsample=(Wpd, one, zero))
    y = y.cpu()
    bad = (y[0, 0] != ref[0, 0]).nonzero()
    print("ci", ci, "tap", tap, "mismatches", len(bad))
    for (r, c) in bad[:6].tolist():
        print("   out(%d,%d): got %g want %g" % (r, c, float(y[0, 0, r, c]), float(ref[0, 0, r, c])))
