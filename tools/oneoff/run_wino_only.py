"""The product's Winograd kernel choice on the four 3x3 stride-1 layer shapes of ResNet-34 at B = 32, eight launches each (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B = 32
for (C, H, W) in ((64, 40, 128), (128, 20, 64), (256, 10, 32), (512, 5, 16)):
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = torch.randn(B, C, H, W, device=dev)
    U = ops.winograd_weights(w)
    for _ in range(8):
        ops.conv3x3_winograd(x, U, sc, sh, True, residual=res)
torch.cuda.synchronize()
