"""What would the fused head's gathered add cost if the points came sorted by their nearest node?  Same kernel, same sizes; the neighbour
indices of consecutive points are (a) random (today: the cloud is in arbitrary order), (b) runs of one nearest node with the other two
neighbours drawn from 4 nodes around it (what a counting sort by nearest node would give)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N, Ma = 32, 20480, 128
g = torch.Generator().manual_seed(0)
first, second = torch.randn(B, 32, N, device=dev), torch.randn(B, 64, N, device=dev)
W0 = torch.randn(96, 128, device=dev); W1 = torch.randn(128, 128, device=dev); W2 = torch.randn(128, 2, device=dev)
sc, sh = torch.rand(128, device=dev), torch.rand(128, device=dev)
Ga, Gb = torch.randn(B, Ma, 128, device=dev), torch.randn(B, Ma, 128, device=dev)
wa, wb = torch.rand(B, N, 3, device=dev), torch.rand(B, N, 3, device=dev)
l0, l1, l2 = (W0, sc, sh, True), (W1, sc, sh, True), (W2, None, sh[:2].contiguous(), False)
S = [ops.Src(first), ops.Src(second)]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def idx(sorted_):
    if not sorted_:
        return torch.randint(0, Ma, (B, N, 3), generator=g, dtype=torch.int32).to(dev)
    near = (torch.arange(N) * Ma // N).view(1, N, 1).expand(B, N, 1)
    other = (near + torch.randint(-2, 3, (B, N, 2), generator=g)).clamp(0, Ma - 1)
    return torch.cat((near, other), dim=2).to(torch.int32).contiguous().to(dev)
for name, s_ in (("random", False), ("sorted", True)):
    ia, ib = idx(s_), idx(s_)
    print("%s neighbours: head %.0f us, layer 0 alone %.0f us, second-chain-like gathered layer %.0f us" % (
        name, t(lambda: ops.point_head(S, l0, l1, l2, N, gathered=[(Ga, ia, wa), (Gb, ib, wb)])),
        t(lambda: ops.pointwise_gemm(S, W0, 128, N, scale=sc, shift=sh, relu=True, gathered=[(Ga, ia, wa), (Gb, ib, wb)])),
        t(lambda: ops.pointwise_gemm([ops.Src(first)], W0[:32, :64].contiguous(), 64, N, gathered=[(Ga[:, :, :64].contiguous(), ia[:, :, :1].contiguous(), None)]))))
    # index_max on sorted / random segment ids
    seg = ia[:, :, 0].contiguous()
    mask = torch.ones(B, Ma, dtype=torch.bool, device=dev)
    try:
        print("   index_max(second): %.0f us" % t(lambda: ops.index_max(second, seg, Ma, return_values=True, mask=None)))
    except Exception as e:
        print("   index_max probe failed:", e)
