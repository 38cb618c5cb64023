"""Per-layer timings of di2p_conv3x3_x3 (bf16x3 direct convolution) against the kernels it replaces, on the seven 3x3 layer shapes of
ResNet-34 at 160 x 512, B frames (GPU box).  Prints microseconds per call and TFLOP/s (fp32-equivalent algorithmic 2*MAC; executed bf16
flops are 6 x that) for every tile configuration that runs the shape, then the whole image encoder under the `conv_x3` masks.
    B=32 REPS=20 python tools/bench_conv_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import _lib, ops, synthetic as nt
from deepi2p_amd.networks import ImageEncoder

B = int(os.environ.get("B", 32))
REPS = int(os.environ.get("REPS", 20))
dev = torch.device("cuda", 0)
SHAPES = [(64, 40, 128, 64, 1, 6), (128, 20, 64, 128, 1, 7), (256, 10, 32, 256, 1, 11), (512, 5, 16, 512, 1, 5),
          (64, 40, 128, 128, 2, 1), (128, 20, 64, 256, 2, 1), (256, 10, 32, 512, 2, 1)]


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


tot_old = tot_new = 0.0
print("%-24s %6s | %-38s | %s" % ("Cin,H,W,Cout,stride", "calls", "fp32-MFMA kernels us (TF)", "bf16x3 us (TF fp32-eq / bf16 executed) per configuration"))
for Cin, H, W, Cout, s, calls in SHAPES:
    g = torch.Generator().manual_seed(Cin)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
    Wt = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous()
    Wp = ops.bf16x3_pack(Wt)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Cout, OH, OW, device=dev)
    flop = 2.0 * B * OH * OW * Cout * Cin * 9
    if s == 1:
        U = ops.winograd_weights(w)
        t_old = timed(lambda: ops.conv3x3_winograd(x, U, sc, sh, True, residual=res))
        old = "winograd %7.1f (%5.1f)" % (t_old, flop / t_old / 1e6)
        run = lambda: ops.conv3x3_x3(x, Wp, Cout, sc, sh, 1, True, residual=res)
    else:
        Wtd = (torch.randn(Cin, Cout, generator=g) / Cin ** 0.5).to(dev)
        Wpd = ops.bf16x3_pack(Wtd)
        flop += 2.0 * B * OH * OW * Cout * Cin
        t_a = timed(lambda: ops.conv2d(x, Wt, sc, sh, 3, 3, 2, 1, True, tap_major=True))
        t_b = timed(lambda: ops.conv2d(x, Wtd, sc, sh, 1, 1, 2, 0, False, tap_major=True))
        t_old = t_a + t_b
        old = "direct %6.1f + ds %5.1f (%5.1f)" % (t_a, t_b, flop / t_old / 1e6)
        run = lambda: ops.conv3x3_x3(x, Wp, Cout, sc, sh, 2, True, downsample=(Wpd, sc, sh))
    cells, best = [], None
    for cfg in (-1, 0, 1, 2, 3):
        with _lib.option("conv_x3_cfg", cfg):
            if not ops.conv3x3_x3_supported(x.shape, Cout, s):
                continue
            t = timed(run)
        cells.append("%s %6.1f (%5.1f / %4.0f)" % ("auto" if cfg < 0 else "cfg%d" % cfg, t, flop / t / 1e6, 6 * flop / t / 1e6))
        if cfg < 0:
            best = t
    tot_old += calls * t_old
    tot_new += calls * (best if best is not None else t_old)
    print("%-24s %6d | %-38s | %s" % ("%d,%d,%d,%d,%d" % (Cin, H, W, Cout, s), calls, old, "  ".join(cells)))
print("3x3 layers of one 32-frame encoder pass: fp32-MFMA kernels %.3f ms, bf16x3 (auto) %.3f ms" % (tot_old / 1e3, tot_new / 1e3))

opt = nt.OptLike(20480, 160, 512, False)
sd = {k[len("img_encoder."):]: v for k, v in nt.synthetic_state_dict(opt).items() if k.startswith("img_encoder.")}
enc = ImageEncoder(opt)
enc.load_state_dict(sd)
enc = enc.to(dev)
img = torch.rand(B, 3, 160, 512, device=dev) * 255
for mask in (0, 1, 2, 4, 8, 16, 12, 28, 31):
    with _lib.option("conv_x3", mask):
        t = timed(lambda: enc(img))
    print("image encoder B=%d conv_x3=%2d: %.3f ms -> %.1f TFLOP/s algorithmic (%.2f of the fp32-MFMA peak)" % (B, mask, t / 1e3, 2 * 5.981e9 * B / t / 1e6, 2 * 5.981e9 * B / t / 1e6 / 157.3))
