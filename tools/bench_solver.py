"""Solver micro-benchmark (GPU box): time / iterations / sweeps of di2p_solve_batched_f32 on config-2 shape."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepi2p_amd import ops, synthetic
from deepi2p_amd.registration import RegistrationPipeline

F, N, R, H, W = int(os.environ.get("F", 32)), 20480, 60, 160, 512
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
frames = [synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.05, with_image=False) for _ in range(F)]
pc = torch.from_numpy(np.stack([f["pc"] for f in frames])).to(dev)
K = torch.from_numpy(np.stack([f["K"] for f in frames])).to(dev)
for name, lab_np in (("noisy-gt", np.stack([f["labels"] for f in frames])),):
    lab = torch.from_numpy(lab_np).to(dev)
    pipe = RegistrationPipeline(H, W, R=R, seed=1)
    restarts = pipe.draw(F, dev)
    pts64 = pc.double()
    yaw0, lab_front, has = ops.initial_guess(pts64, lab)
    sweeps = torch.zeros((F, R), dtype=torch.int32, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        params, cost, iters = ops.solve_batched(pc, lab_front, K, restarts[0], restarts[1], H, W, pipe.lb, pipe.ub, 500, True, yaw0=yaw0, sweeps=sweeps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    kept = (lab_front >= 0).sum(1).float().mean().item()
    print("%s CFG=%s: %.2f ms  iters mean %.1f max %d  sweeps mean %.1f max %d  kept pts %.0f  -> %.1f us/sweep/hyp-wave" % (
        name, os.environ.get("DI2P_SOLVER_CFG", "443"), dt * 1e3, iters.float().mean().item(), iters.max().item(),
        sweeps.float().mean().item(), sweeps.max().item(), kept, dt * 1e6 / max(sweeps.float().mean().item(), 1)))
