"""Image-encoder driver for profiling the conv kernels (GPU box): whole-encoder time and a per-layer table
(HIP events around every di2p_conv2d call, serial)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd.networks import ImageEncoder
from deepi2p_amd import synthetic as nt, _lib, ops

B = int(os.environ.get("B", 32))
dev = torch.device("cuda", 0)
opt = nt.OptLike(20480, 160, 512, False)
sd = {k[len("img_encoder."):]: v for k, v in nt.synthetic_state_dict(opt).items() if k.startswith("img_encoder.")}
enc = ImageEncoder(opt)
enc.load_state_dict(sd)
enc = enc.to(dev)
img = torch.rand(B, 3, 160, 512, device=dev) * 255
for name, val in [kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv]:
    _lib.set_option(name, int(val))
for _ in range(3):
    enc(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    enc(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("image encoder B=%d [%s]: %.3f ms  -> %.1f TFLOP/s algorithmic" % (B, os.environ.get("OPTS", ""), dt * 1e3, 2 * 5.981e9 * B / dt / 1e12))
if os.environ.get("LAYERS"):
    shapes = []
    orig = ops.conv2d
    def rec(x, Wt, scale, shift, KH, KW, stride, pad, relu, residual=None, tap_major=False):
        shapes.append((x.shape[1], x.shape[2], x.shape[3], Wt.shape[1], KH, stride))
        return orig(x, Wt, scale, shift, KH, KW, stride, pad, relu, residual=residual, tap_major=tap_major)
    ops.conv2d = rec
    import deepi2p_amd.networks as nw
    _lib.TIMED = {"di2p_conv2d": [], "di2p_conv2d_ws": []}
    reps = 5
    for _ in range(reps):
        shapes.clear()
        enc(img)
    torch.cuda.synchronize()
    ev = sorted(_lib.TIMED["di2p_conv2d"] + _lib.TIMED["di2p_conv2d_ws"], key=lambda e: e[0].elapsed_time(e[1]) * 0)  # keep order per list
    # events were appended per entry point; rebuild call order from the two lists by replaying the shapes' split decision
    _lib.TIMED = None
    # simpler: time each distinct shape separately
    seen = {}
    for sh in shapes:
        seen[sh] = seen.get(sh, 0) + 1
    tot = 0.0
    print("%-34s %5s %9s %8s" % ("Cin,H,W,Cout,k,stride", "calls", "us/call", "TFLOP/s"))
    for (Cin, H, W, Cout, k, s), n in seen.items():
        x = torch.randn(B, Cin, H, W, device=dev)
        Wt = torch.randn(Cin * k * k, Cout, device=dev) * 0.05
        sc, sh_ = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        pad = 3 if k == 7 else (1 if k == 3 else 0)
        res = None
        f = lambda: orig(x, Wt, sc, sh_, k, k, s, pad, True, residual=res, tap_major=(k != 7))
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        fl = 2.0 * B * OH * OW * Cout * Cin * k * k
        tot += us * n
        print("%-34s %5d %9.1f %8.1f" % ("%d,%d,%d,%d,%d,%d" % (Cin, H, W, Cout, k, s), n, us, fl / us / 1e6))
    print("sum over layers %.3f ms" % (tot / 1e3))
