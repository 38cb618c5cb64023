"""di2p_stem_x3 at the benchmark shape (32 frames of 160 x 512), six launches (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = (torch.rand(32, 3, 160, 512, generator=g) * 255).to(dev)
w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).to(dev)
sc, sh = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
Wp = ops.stem_x3_weights(w)
for _ in range(6):
    ops.stem_x3(x, Wp, sc, sh)
torch.cuda.synchronize()
