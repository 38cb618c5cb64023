"""Error of di2p_conv3x3_x3 against an fp64 convolution per tile configuration, next to the fp32-MFMA kernels' (direct implicit GEMM,
Winograd), on the ResNet-34 layer shapes.  B=3.  (Every shipped instance keeps the small partial products in a second accumulator set; the
round-5 knob that switched it off is gone.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from deepi2p_amd import _lib, ops
dev = torch.device("cuda", 0)
B = 3
for Cin, H, W, Cout in ((64, 40, 128, 64), (128, 20, 64, 128), (256, 10, 32, 256), (512, 5, 16, 512)):
    g = torch.Generator().manual_seed(Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    Wt = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous().to(dev)
    Wp = ops.bf16x3_pack(Wt)
    one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    xd = x.to(dev)
    def err(y):
        d = (y.cpu().double() - ref).abs()
        return "max %.2e rms %.2e" % (float(d.max()), float((d ** 2).mean().sqrt()))
    print("K=%d (%d,%d,%d,%d): direct fp32 %s | winograd %s" % (9 * Cin, Cin, H, W, Cout, err(ops.conv2d(xd, Wt, one, zero, 3, 3, 1, 1, False, tap_major=True)),
                                                               err(ops.conv3x3_winograd(xd, ops.winograd_weights(w.to(dev)), one, zero, False))))
    for cfg in range(4):
        with _lib.option("conv_x3_cfg", cfg):
            if ops.conv3x3_x3_supported(xd.shape, Cout, 1):
                print("    bf16x3 cfg%d: %s" % (cfg, err(ops.conv3x3_x3(xd, Wp, Cout, one, zero, 1, False))))
