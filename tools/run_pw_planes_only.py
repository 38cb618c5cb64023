"""Only the planes-source bf16x3 layer (layers_after.0's shape: K = 256 -> M = 512 with a gathered table, 32 frames x 2048 columns, planes in and out),
for tools/prof_kernel_counters.sh.   [PLANES=0 for the fp32-source kernel]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N = 32, 2048
planes = os.environ.get("PLANES", "1") != "0"
x = torch.randn(B, 256, N, device=dev)
W1, W2 = torch.randn(256, 256, device=dev) / 16, torch.randn(256, 512, device=dev) / 16
s2, h2 = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev)
tab = torch.randn(B, N // 16, 512, device=dev)
gidx = (torch.arange(N, dtype=torch.int32, device=dev) // 16).unsqueeze(0).expand(B, -1).contiguous()
y1 = ops.pointwise_gemm([ops.Src(x)], W1, 256, N, x3=True, planes_out=planes)
for _ in range(12):
    ops.pointwise_gemm([y1 if planes else ops.Src(y1)], W2, 512, N, scale=s2, shift=h2, relu=True, gathered=[(tab, gidx.reshape(B, N, 1), None)], x3=True,
                       planes_out=planes)
torch.cuda.synchronize()
