"""Tile-engine calibration: pointwise GEMM on ideal shapes (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
for (B, M, K, N) in [(1, 4096, 4096, 4096), (1, 64, 4096, 65536), (32, 64, 576, 5120), (32, 128, 128, 20480), (32, 512, 512, 2048)]:
    x = torch.randn(B, K, N, device=dev)
    Wt = torch.randn(K, M, device=dev)
    for _ in range(2):
        y = ops.pointwise_gemm([ops.Src(x)], Wt, M, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        y = ops.pointwise_gemm([ops.Src(x)], Wt, M, N)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B=%d M=%d K=%d N=%d: %.3f ms  %.1f TFLOP/s" % (B, M, K, N, dt * 1e3, 2.0 * B * M * K * N / dt / 1e12))
