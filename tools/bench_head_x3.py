"""The coarse per-point head at the benchmark shape (32 frames x 20480 points, 128 + 128 nodes): di2p_point_head_x3 (tables in LDS / from
memory) against di2p_point_head (fp32 MFMA, LDS tile).  REPS=20 python tools/bench_head_x3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from deepi2p_amd import _lib
import test_gpu_head_x3 as T
dev = torch.device("cuda", 0)
B, N, REPS = int(os.environ.get("B", 32)), 20480, int(os.environ.get("REPS", 20))
d = T._case(dev, B, N, (128, 128), 2, 1)
# IDX=same: every point gathers node 0 (LDS broadcasts, no bank conflicts); IDX=seq: lane l gathers nodes l, l+1, l+2 (all banks distinct)
if os.environ.get("IDX") == "same":
    d["ia"].zero_(); d["ib"].zero_()
elif os.environ.get("IDX") == "seq":
    ar = torch.arange(N, device=dev, dtype=torch.int32).view(1, N, 1) + torch.arange(3, device=dev, dtype=torch.int32).view(1, 1, 3)
    d["ia"].copy_((ar % 128).expand(B, N, 3)); d["ib"].copy_(((ar + 64) % 128).expand(B, N, 3))


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


flop = 2.0 * B * N * (96 * 128 + 128 * 128 + 128 * 2)
t = timed(lambda: T._run_fp32(d, N))
print("fp32-MFMA fused head (LDS tile)        %7.1f us  %6.1f TFLOP/s" % (t, flop / t / 1e6))
for tab in (1, 2, 0) * int(os.environ.get("ROUNDS", 2)):      # (several rounds: the first timings of a process run at lower clocks)
    with _lib.option("head_x3_tab", tab):
        t = timed(lambda: T._run_x3(d, N))
    print("bf16x3 head, tables %-24s %7.1f us  %6.1f TFLOP/s fp32-equivalent (%.0f of executed bf16 products)" % ({1: "in LDS, 8 waves", 2: "in LDS, 4 waves", 0: "from memory"}[tab], t, flop / t / 1e6, 6 * flop / t / 1e6))
