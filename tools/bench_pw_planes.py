"""The kNN-fusion chain (256 -> 256 + max16, -> 512 + gathered, -> 256 max16 at 32 frames x 2048 columns) with fp32 hand-over against split
planes hand-over: microseconds per layer and for the chain.   REPS=20 python tools/bench_pw_planes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N, REPS = int(os.environ.get("B", 32)), 2048, int(os.environ.get("REPS", 20))


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


def layer(K, M):
    return torch.randn(K, M, device=dev) / K ** 0.5, torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev)


x = torch.randn(B, 256, N, device=dev)
(W1, s1, h1), (W2, s2, h2), (W3, s3, h3) = layer(256, 256), layer(256, 512), layer(512, 256)
tab = torch.randn(B, N // 16, 512, device=dev)
gidx = (torch.arange(N, dtype=torch.int32, device=dev) // 16).unsqueeze(0).expand(B, -1).contiguous()
gat = [(tab, gidx.reshape(B, N, 1), None)]
for planes in (False, True, False, True):
    l1 = lambda: ops.pointwise_gemm([ops.Src(x)], W1, 256, N, scale=s1, shift=h1, relu=True, group_max=16, also_full=True, x3=True, planes_out=planes)
    y1, _ = l1()
    l2 = lambda: ops.pointwise_gemm([y1 if planes else ops.Src(y1)], W2, 512, N, scale=s2, shift=h2, relu=True, gathered=gat, x3=True, planes_out=planes)
    y2 = l2()
    l3 = lambda: ops.pointwise_gemm([y2 if planes else ops.Src(y2)], W3, 256, N, scale=s3, shift=h3, relu=False, group_max=16, x3=True)
    t1, t2, t3 = timed(l1), timed(l2), timed(l3)
    print("%-6s before.1 %6.1f us  after.0 %6.1f us  after.1 %6.1f us  chain %6.1f us" % ("planes" if planes else "fp32", t1, t2, t3, t1 + t2 + t3), flush=True)
