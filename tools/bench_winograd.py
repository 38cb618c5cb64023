"""Winograd F(2x2,3x3) vs the direct implicit-GEMM kernel on the four 3x3 stride-1 layer shapes of ResNet-34 at B=32 (GPU box).
TFLOP/s are ALGORITHMIC (2 * direct-convolution MACs / time) for both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import _lib, ops

B = int(os.environ.get("B", 32))
dev = torch.device("cuda", 0)


def timeit(f, reps=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("%-22s %10s %8s | Winograd us" % ("Cin,H,W,Cout", "direct us", "TF"))
tot = [0.0, 0.0]
for (C, H, W, calls) in ((64, 40, 128, 6), (128, 20, 64, 7), (256, 10, 32, 11), (512, 5, 16, 5)):
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = torch.randn(B, C, H, W, device=dev)
    Wtap = w.permute(2, 3, 1, 0).reshape(-1, C).contiguous()
    U = ops.winograd_weights(w)
    fl = 2.0 * B * H * W * C * C * 9
    td = timeit(lambda: ops.conv2d(x, Wtap, sc, sh, 3, 3, 1, 1, True, residual=res, tap_major=True))
    out = []
    for kc, cob, reg in ((8, 32, 1), (4, 32, 1), (0, 0, 2), (0, 0, 3)):
        with _lib.option("wino_cob", cob), _lib.option("wino_kc", kc), _lib.option("wino_reg", reg):
            out.append(timeit(lambda: ops.conv3x3_winograd(x, U, sc, sh, True, residual=res)))
    pipe, same = float("nan"), None
    if _lib.get_option("wino_pipe") >= 0:
        with _lib.option("wino_reg", 2):
            y0 = ops.conv3x3_winograd(x, U, sc, sh, True, residual=res)
            with _lib.option("wino_pipe", 1):
                pipe = timeit(lambda: ops.conv3x3_winograd(x, U, sc, sh, True, residual=res))
                same = bool(torch.equal(y0, ops.conv3x3_winograd(x, U, sc, sh, True, residual=res)))
    best = min(v for v in out if v == v)
    tot[0] += td * calls; tot[1] += best * calls
    print("%-22s %10.1f %8.1f | LDS panels K-step 8 %6.1f K-step 4 %6.1f | register-resident 4 waves %6.1f 2 waves %6.1f%s | best %.1f TF" % (
        "%d,%d,%d,%d" % (C, H, W, C), td, fl / td / 1e6, out[0], out[1], out[2], out[3],
        (" pipelined %6.1f (bit-identical %s)" % (pipe, same)) if same is not None else "", fl / best / 1e6))
print("26 layers per 32-frame step: direct %.3f ms, Winograd (better blocking per shape) %.3f ms" % (tot[0] / 1e3, tot[1] / 1e3))

x = torch.rand(B, 3, 160, 512, device=dev) * 255
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
Wt, Wp = w.reshape(64, -1).t().contiguous(), ops.stem_weights(w)
fl = 2.0 * B * 80 * 256 * 64 * 147
t0 = timeit(lambda: ops.conv2d(x, Wt, sc, sh, 7, 7, 2, 3, True))
t1 = timeit(lambda: ops.conv_stem(x, Wp, sc, sh, True))
print("stem 3->64 7x7/2 at 160x512: implicit GEMM %.1f us (%.1f TF), direct kernel %.1f us (%.1f TF algorithmic)" % (t0, fl / t0 / 1e6, t1, fl / t1 / 1e6))
