"""Where the time of the bf16x3 pointwise kernel goes: K sweep (slope = K loop, intercept = prologue + epilogue + launch rounds) per epilogue
variant, at the column counts of the kNN-fusion layers (32 frames x 2048 columns).  REPS=20 [PLANES=1] python tools/bench_pw_x3.py
(PLANES=1: the source handed over as split bf16 planes, di2p_pointwise_gemm_x3p)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N, REPS = 32, 2048, int(os.environ.get("REPS", 20))


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


g = torch.Generator(device="cpu").manual_seed(0)
for M in (256, 512):
    sc, sh = torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev)
    nodes = 128
    table = torch.randn(B, nodes, M, device=dev)
    idx = torch.randint(0, nodes, (B, N, 3), dtype=torch.int32, device=dev)
    w = torch.rand(B, N, 3, device=dev)
    for name, kw in (("plain", {}), ("bn+relu", dict(scale=sc, shift=sh, relu=True)), ("bn+relu+gmax16", dict(scale=sc, shift=sh, relu=True, group_max=16)),
                     ("bn+relu+gathered", dict(scale=sc, shift=sh, relu=True, gathered=[(table, idx, w)]))):
        row = []
        for K in (32, 128, 256, 512):
            x = torch.randn(B, K, N, device=dev)
            Wt = torch.randn(K, M, device=dev) / K ** 0.5
            src = ops.Src(x)
            if os.environ.get("PLANES"):     # the activation handed over as split planes (made by a layer of K output rows)
                src = ops.pointwise_gemm([ops.Src(x)], torch.randn(K, K, device=dev), K, N, x3=True, planes_out=True)
            t = timed(lambda: ops.pointwise_gemm([src], Wt, M, N, x3=True, **kw))
            row.append("K=%3d %6.1f us (%5.1f TF)" % (K, t, 2.0 * B * M * K * N / t / 1e6))
        print("M=%d %-18s %s" % (M, name, "  ".join(row)), flush=True)
