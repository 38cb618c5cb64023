import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from deepi2p_amd import _lib, ops
dev = torch.device("cuda", 0)
B, Cin, H, W, Cout, s = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (1, 128, 20, 64, 256, 2))]
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, W, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
wd = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
ref = F.conv2d(x.double(), w.double(), stride=s, padding=1)
refd = F.conv2d(x.double(), wd.double(), stride=s)
Wt = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous().to(dev)
Wp = ops.bf16x3_pack(Wt)
Wpd = ops.bf16x3_pack(wd.reshape(Cout, Cin).t().contiguous().to(dev))
one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
for cfg in range(4):
    with _lib.option("conv_x3_cfg", cfg):
        if not ops.conv3x3_x3_supported((B, Cin, H, W), Cout, s):
            continue
        if s == 2:
            y, yd = ops.conv3x3_x3(x.to(dev), Wp, Cout, one, zero, 2, False, downsample=(Wpd, one, zero))
        else:
            y, yd = ops.conv3x3_x3(x.to(dev), Wp, Cout, one, zero, 1, False), None
    e = (y.cpu().double() - ref).abs()
    print("cfg", cfg, "max err", float(e.max()), "ds", None if yd is None else float((yd.cpu().double() - refd).abs().max()))
    if float(e.max()) > 1e-3:
        em = e.amax(dim=(0, 1))          # [OH, OW]
        OW = em.shape[1]
        for r in range(em.shape[0]):
            print("  row %2d:" % r, " ".join("%.0e" % float(em[r, c0:c0 + 16].max()) for c0 in range(0, OW, 16)))
        ec = e.amax(dim=(0, 2, 3))
        print("  by channel block of 16:", " ".join("%.0e" % float(ec[c:c + 16].max()) for c in range(0, Cout, 16)))
