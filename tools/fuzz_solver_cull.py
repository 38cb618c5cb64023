"""Randomised check that the cluster shortcut of the pose solver never changes a result: for many random frames, point
counts, intrinsics and hypotheses the outputs with and without DI2P_SOLVER_NOCULL (and with the classification cache off, DI2P_SOLVER_NOCACHE) must be bit-identical (GPU box)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepi2p_amd import _lib, ops, synthetic

dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))
bad = 0
cases = int(os.environ.get("CASES", 40))
for it in range(cases):
    N = int(rng.choice([37, 64, 500, 2048, 5000, 20480, 30000]))
    H, W = [(160, 512), (384, 640), (64, 128)][int(rng.integers(0, 3))]
    is_2d = bool(rng.integers(0, 2))
    f32 = bool(rng.integers(0, 2))
    f = synthetic.make_frame(rng, N=N, H=H, W=W, flip=float(rng.uniform(0, 0.2)), with_image=False)
    pts = f["pc"].astype(np.float32 if f32 else np.float64)
    if rng.random() < 0.3:
        pts = (pts * np.array([[1.0], [float(rng.uniform(0.1, 3))], [1.0]])).astype(pts.dtype)      # squash / stretch heights
    lab = f["labels"].astype(np.int32)
    R = 8
    ys = rng.normal(f["yaw_gt"], 0.5, R)
    Ts = rng.uniform(-6, 6, (R, 3)); Ts[:, 1] = rng.uniform(-0.1, 0.1, R)
    args = (torch.from_numpy(pts).to(dev).unsqueeze(0), torch.from_numpy(lab).to(dev).unsqueeze(0),
            torch.from_numpy(f["K"]).to(dev).view(1, 3, 3), torch.from_numpy(ys).to(dev).view(1, R),
            torch.from_numpy(Ts).to(dev).view(1, R, 3), H, W, [-5, -0.1, -10], [5, 0.1, 10], 80, is_2d)

    def run():
        sw = torch.zeros((1, R), dtype=torch.int32, device=dev)
        p, c, i = ops.solve_batched(*args, sweeps=sw)
        return [t.cpu().numpy().tobytes() for t in (p, c, i, sw)]
    a = run()
    with _lib.option("solver_nocull", 1), _lib.option("solver_noprefilter", 1):
        b = run()
    with _lib.option("solver_nocache", 1):
        c = run()
    if a != b or a != c:
        bad += 1
        print("MISMATCH", dict(N=N, H=H, W=W, is_2d=is_2d, f32=f32, vs_nocull=a != b, vs_nocache=a != c))
print("solver cull fuzz: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
