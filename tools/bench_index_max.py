"""index_max driver (GPU box): BASELINE config-2 shapes.  The inputs rotate through > 1 GB of distinct buffers so that every
call streams from HBM (the 256 MB Infinity Cache would otherwise serve a repeated 84 / 168 MB input: the in-pipeline figure
is the cold one)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops, _lib
dev = torch.device("cuda", 0)
B, N, K = 32, 20480, 128
g = torch.Generator().manual_seed(0)
index = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32).to(dev)
for name, val in [kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv]:
    _lib.set_option(name, int(val))
for C in [int(c) for c in os.environ.get("CS", "32,64").split(",")]:
    nbuf = max(2, int(1.2e9 // (B * C * N * 4)))
    datas = [torch.relu(torch.randn(B, C, N, generator=g)).to(dev) for _ in range(nbuf)]
    for i in range(3):
        ops.index_max(datas[i % nbuf], index, K, return_values=True)
    torch.cuda.synchronize()
    reps = 4 * nbuf
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        ops.index_max(datas[i % nbuf], index, K, return_values=True)
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) * 1e-3 / reps
    alg = B * (4 * C * N + 4 * N + 2 * 4 * C * K)
    print("index_max C=%d [%s]: %.1f us (cold inputs, %d rotating buffers)  algorithmic %.1f MB -> %.0f GB/s = %.2f of 8 TB/s" % (
        C, os.environ.get("OPTS", ""), dt * 1e6, nbuf, alg / 1e6, alg / dt / 1e9, alg / dt / 8e12))
