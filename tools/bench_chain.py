"""Device time of the two fused PointNet chains at the config-2 size (B = 32, N = 20480), against the separate launches.
   python tools/bench_chain.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops, _lib
from deepi2p_amd.ops import Src

dev = torch.device("cuda", 0)
B, N, Ma = 32, 20480, 128
g = torch.Generator().manual_seed(0)
mk = lambda k, m: (torch.randn(k, m, generator=g).to(dev) * 0.1, torch.rand(m, generator=g).to(dev) + 0.5, torch.randn(m, generator=g).to(dev) * 0.1, True)
aug = torch.randn(B, 7, N, generator=g).to(dev)
first_layers = [mk(7, 32), mk(32, 32), mk(32, 32)]
first = torch.randn(B, 32, N, generator=g).to(dev)
G = torch.randn(B, Ma, 64, generator=g).to(dev)
idx = torch.randint(0, Ma, (B, N, 1), generator=g, dtype=torch.int32).to(dev)
second_layers = [mk(32, 64), mk(64, 64)]


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def separate(x, layers, gathered=None):
    for i, l in enumerate(layers):
        x = ops.pointwise_gemm([Src(x)], l[0], l[0].shape[1], N, scale=l[1], shift=l[2], relu=l[3], gathered=gathered if i == 0 else None)
    return x


a = t(lambda: ops.point_chain([Src(aug)], first_layers, N))
b = t(lambda: ops.point_chain([Src(first)], second_layers, N, gathered=[(G, idx, None)]))
print("%s  first %.1f us  second %.1f us" % (os.environ.get("DI2P_LIB", "default").split("/")[-2:][0], a, b))
if not os.environ.get("DI2P_LIB"):
    print("separate  first %.1f us  second %.1f us" % (t(lambda: separate(aug, first_layers)), t(lambda: separate(first, second_layers, [(G, idx, None)]))))
    assert torch.equal(ops.point_chain([Src(first)], second_layers, N, gathered=[(G, idx, None)]), separate(first, second_layers, [(G, idx, None)]))
