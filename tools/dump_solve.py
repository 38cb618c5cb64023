"""Solve one seeded config-2 shaped batch and dump the raw outputs (A/B of two library builds through DI2P_LIB: bit-identity checks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepi2p_amd import ops, synthetic
from deepi2p_amd.registration import RegistrationPipeline
dev = torch.device("cuda", 0)
F, N, R, H, W = 4, 20480, 24, 160, 512
rng = np.random.default_rng(7)
frames = [synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.05, with_image=False) for _ in range(F)]
pc = torch.from_numpy(np.stack([f["pc"] for f in frames])).to(dev)
K = torch.from_numpy(np.stack([f["K"] for f in frames])).to(dev)
lab = torch.from_numpy(np.stack([f["labels"] for f in frames])).to(dev)
pipe = RegistrationPipeline(H, W, R=R, seed=1)
restarts = pipe.draw(F, dev)
yaw0, lab_front, has = ops.initial_guess(pc.double(), lab)
sweeps = torch.zeros((F, R), dtype=torch.int32, device=dev)
p, c, it = ops.solve_batched(pc, lab_front, K, restarts[0], restarts[1], H, W, pipe.lb, pipe.ub, 500, True, yaw0=yaw0, sweeps=sweeps)
np.savez(sys.argv[1], p=p.cpu().numpy(), c=c.cpu().numpy(), it=it.cpu().numpy(), sw=sweeps.cpu().numpy())
print("dumped", sys.argv[1], float(c.min()), int(sweeps.sum()))
