"""di2p_conv3x3_x3 on the seven 3x3 layer shapes of ResNet-34 at B = 32, six launches each (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B = 32
for (Cin, H, W, Cout, s) in ((64, 40, 128, 64, 1), (128, 20, 64, 128, 1), (256, 10, 32, 256, 1), (512, 5, 16, 512, 1), (64, 40, 128, 128, 2), (128, 20, 64, 256, 2), (256, 10, 32, 512, 2)):
    x = torch.randn(B, Cin, H, W, device=dev)
    Wp = ops.bf16x3_pack(torch.randn(9 * Cin, Cout, device=dev) * 0.05)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Cout, OH, OW, device=dev)
    ds = (ops.bf16x3_pack(torch.randn(Cin, Cout, device=dev) * 0.05), sc, sh) if s == 2 else None
    for _ in range(6):
        ops.conv3x3_x3(x, Wp, Cout, sc, sh, s, True, residual=res if s == 1 else None, downsample=ds)
torch.cuda.synchronize()
