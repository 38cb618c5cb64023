"""Eight launches of each fused PointNet chain at the config-2 size (for the counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
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
for _ in range(8):
    ops.point_chain([Src(aug)], first_layers, N)
    ops.point_chain([Src(first)], second_layers, N, gathered=[(G, idx, None)])
torch.cuda.synchronize()
