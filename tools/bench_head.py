import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N, Ma = 32, 20480, 128
g = torch.Generator().manual_seed(0)
first, second = torch.randn(B, 32, N, device=dev), torch.randn(B, 64, N, device=dev)
W0 = torch.randn(96, 128, device=dev); W1 = torch.randn(128, 128, device=dev); W2 = torch.randn(128, 2, device=dev)
sc, sh = torch.rand(128, device=dev), torch.rand(128, device=dev)
Ga, Gb = torch.randn(B, Ma, 128, device=dev), torch.randn(B, Ma, 128, device=dev)
ia = torch.randint(0, Ma, (B, N, 3), dtype=torch.int32, device=dev); ib = torch.randint(0, Ma, (B, N, 3), dtype=torch.int32, device=dev)
wa, wb = torch.rand(B, N, 3, device=dev), torch.rand(B, N, 3, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
l0, l1, l2 = (W0, sc, sh, True), (W1, sc, sh, True), (W2, None, sh[:2].contiguous(), False)
S = [ops.Src(first), ops.Src(second)]
print("fused head with gathered add : %.0f us" % t(lambda: ops.point_head(S, l0, l1, l2, N, gathered=[(Ga, ia, wa), (Gb, ib, wb)])))
print("fused head without gathers   : %.0f us" % t(lambda: ops.point_head(S, l0, l1, l2, N)))
print("layer 0 alone with gathers   : %.0f us" % t(lambda: ops.pointwise_gemm(S, W0, 128, N, scale=sc, shift=sh, relu=True, gathered=[(Ga, ia, wa), (Gb, ib, wb)])))
print("layer 0 alone without gathers: %.0f us" % t(lambda: ops.pointwise_gemm(S, W0, 128, N, scale=sc, shift=sh, relu=True)))
from deepi2p_amd import _lib
new = ops.point_head(S, l0, l1, l2, N, gathered=[(Ga, ia, wa), (Gb, ib, wb)])
new_ng = ops.point_head(S, l0, l1, l2, N)
with _lib.option("head_reg", 1):
    print("wave-autonomous head with gathered add : %.0f us" % t(lambda: ops.point_head(S, l0, l1, l2, N, gathered=[(Ga, ia, wa), (Gb, ib, wb)])))
    print("wave-autonomous head without gathers   : %.0f us" % t(lambda: ops.point_head(S, l0, l1, l2, N)))
    old = ops.point_head(S, l0, l1, l2, N, gathered=[(Ga, ia, wa), (Gb, ib, wb)])
    old_ng = ops.point_head(S, l0, l1, l2, N)
print("bit-identical:", torch.equal(new, old), torch.equal(new_ng, old_ng), float((new - old).abs().max()))
