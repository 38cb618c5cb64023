"""The head of the image branch at the benchmark shape (32 frames of 160 x 512): di2p_stem_x3 (one launch, bf16 matrix instructions, exact
splits) against di2p_conv7x7s2_stem + di2p_maxpool3x3s2 (fp32 MFMA).  REPS=20 python tools/bench_stem_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, H, W, REPS = int(os.environ.get("B", 32)), 160, 512, int(os.environ.get("REPS", 20))
g = torch.Generator().manual_seed(0)
x = (torch.rand(B, 3, H, W, generator=g) * 255).to(dev)
w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).to(dev)
sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
Wp1, Wp3 = ops.stem_weights(w), ops.stem_x3_weights(w)


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


flop = 2.0 * B * 64 * 147 * (H // 2) * (W // 2)
t1 = timed(lambda: ops.conv_stem(x, Wp1, sc, sh, True))
t2 = timed(lambda: ops.maxpool3x3s2(ops.conv_stem(x, Wp1, sc, sh, True)))
t3 = timed(lambda: ops.stem_x3(x, Wp3, sc, sh))
print("fp32-MFMA stem %.1f us (%.1f TFLOP/s), + max-pool %.1f us" % (t1, flop / t1 / 1e6, t2))
print("bf16x3 stem + pool, one launch %.1f us (%.1f TFLOP/s fp32-equivalent)" % (t3, flop / t3 / 1e6))
a, b = ops.stem_x3(x, Wp3, sc, sh), ops.maxpool3x3s2(ops.conv_stem(x, Wp1, sc, sh, True))
print("max |difference| %.3g of max %.3g" % (float((a - b).abs().max()), float(b.abs().max())))
