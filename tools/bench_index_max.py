"""index_max driver for PMC traffic collection (GPU box): BASELINE config-2 shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
B, N, K = 32, 20480, 128
g = torch.Generator().manual_seed(0)
index = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32).to(dev)
for C in (32, 64):
    data = torch.relu(torch.randn(B, C, N, generator=g)).to(dev)
    for _ in range(3):
        ops.index_max(data, index, K, return_values=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        ops.index_max(data, index, K, return_values=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    alg = B * (4 * C * N + 4 * N + 2 * 4 * C * K)
    print("index_max C=%d: %.1f us  algorithmic %.1f MB -> %.0f GB/s" % (C, dt * 1e6, alg / 1e6, alg / dt / 1e9))
