"""di2p_point_head_x3 at the benchmark shape, six launches (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_head_x3 as T
dev = torch.device("cuda", 0)
d = T._case(dev, 32, 20480, (128, 128), 2, 1)
for _ in range(6):
    T._run_x3(d, 20480)
torch.cuda.synchronize()
