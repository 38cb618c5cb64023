"""Tile-engine calibration: pointwise GEMM shapes of the network (GPU box).  DI2P_PW_NOVEC=1 selects the scalar stager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
shapes = [(1, 4096, 4096, 4096), (32, 32, 7, 20480), (32, 32, 32, 20480), (32, 64, 32, 20480), (32, 64, 64, 20480),
          (32, 128, 96, 20480), (32, 128, 128, 20480), (32, 512, 256, 2048), (32, 256, 512, 2048)]
for (B, M, K, N) in shapes:
    x = torch.randn(B, K, N, device=dev)
    Wt = torch.randn(K, M, device=dev)
    sc, sh = torch.rand(M, device=dev), torch.rand(M, device=dev)
    for _ in range(2):
        y = ops.pointwise_gemm([ops.Src(x)], Wt, M, N, scale=sc, shift=sh, relu=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        y = ops.pointwise_gemm([ops.Src(x)], Wt, M, N, scale=sc, shift=sh, relu=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    gb = 4.0 * B * N * (K + M) / 1e9
    print("B=%d M=%d K=%d N=%d: %.3f ms  %.1f TFLOP/s  %.0f GB/s" % (B, M, K, N, dt * 1e3, 2.0 * B * M * K * N / dt / 1e12, gb / dt))
