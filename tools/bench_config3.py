"""BASELINE config 3 on one GPU (not the headline): B=64 KITTI-shaped frames, coarse + fine heads (L=80), argmax of both,
EPnP-RANSAC (the reference's estimator, 500 iterations; METHOD=dlt_lo for the builder's DLT variant) instead of the Gauss-Newton solver.  The fine labels fed to PnP are the synthetic GT cells
(random-init weights carry no information), the network forward + both argmax run in full."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepi2p_amd import ops, prep, synthetic
from deepi2p_amd.networks import MMClassifer
from deepi2p_amd.registration_pnp import camera_matrix_scaling, draw_samples, pnp_ransac

B, N, H, W = int(os.environ.get("B", 64)), 20480, 160, 512
METHOD = os.environ.get("METHOD", "epnp")
dev = torch.device("cuda", 0)
opt = synthetic.OptLike(N, H, W, True)
opt.device = dev
mm = MMClassifer(opt)
mm.detector.load_state_dict(synthetic.synthetic_state_dict(opt))
batch = synthetic.make_batch(5, B, N=N, H=H, W=W)
t = {k: torch.from_numpy(batch[k]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
mm.set_input(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], torch.zeros(B, 3, 4), t["img"], torch.from_numpy(batch["K"]).float())
P_gt = torch.from_numpy(batch["P_gt"][:, :3, :]).float().to(dev)
K32 = torch.from_numpy(batch["K"]).float().to(dev)
coarse_gt, fine_gt = prep.project_labels(mm.pc, P_gt, K32, H, W, 32)
Kf = torch.from_numpy(np.stack([camera_matrix_scaling(k, 1 / 32) for k in batch["K"]])).to(dev)
samples = torch.from_numpy(draw_samples(np.random.default_rng(0), B, 500)).to(dev)

def step():
    coarse_pred, fine_pred = mm.inference_pass()
    out = pnp_ransac(mm.pc, coarse_gt, fine_gt, Kf, W // 32, samples, method=METHOD)
    return coarse_pred, fine_pred, out

for _ in range(2):
    step()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t0 = time.perf_counter()
n = 5
for _ in range(n):
    e[0].record(); mm.inference_pass(); e[1].record(); pnp_ransac(mm.pc, coarse_gt, fine_gt, Kf, W // 32, samples, method=METHOD); e[2].record()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
o = step()[2]
P = o["P"].cpu().numpy()
ok = 0
from deepi2p_amd.registration import get_P_diff
for i in range(B):
    tt, rr = get_P_diff(P[i], batch["P_gt"][i])
    ok += (tt < 2.0 and rr < 8.0)
print("config 3 (%s), B=%d: %.2f ms per batch = %.0f frames/s (network + argmax %.2f ms, PnP-RANSAC %.2f ms); %d/%d frames within 2 m / 8 deg (cell-corner bias)"
      % (METHOD, B, dt * 1e3, B / dt, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), ok, B))
