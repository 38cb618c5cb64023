"""Eight launches of the fused per-point head at the config-2 size (for the counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N, Ma = 32, 20480, 128
first, second = torch.randn(B, 32, N, device=dev), torch.randn(B, 64, N, device=dev)
W0 = torch.randn(96, 128, device=dev); W1 = torch.randn(128, 128, device=dev); W2 = torch.randn(128, 2, device=dev)
sc, sh = torch.rand(128, device=dev), torch.rand(128, device=dev)
Ga, Gb = torch.randn(B, Ma, 128, device=dev), torch.randn(B, Ma, 128, device=dev)
ia = torch.randint(0, Ma, (B, N, 3), dtype=torch.int32, device=dev); ib = torch.randint(0, Ma, (B, N, 3), dtype=torch.int32, device=dev)
wa, wb = torch.rand(B, N, 3, device=dev), torch.rand(B, N, 3, device=dev)
l0, l1, l2 = (W0, sc, sh, True), (W1, sc, sh, True), (W2, None, sh[:2].contiguous(), False)
S = [ops.Src(first), ops.Src(second)]
g = [(Ga, ia, wa), (Gb, ib, wb)] if not os.environ.get("NOGATHER") else None
for _ in range(8):
    ops.point_head(S, l0, l1, l2, N, gathered=g)
torch.cuda.synchronize()
