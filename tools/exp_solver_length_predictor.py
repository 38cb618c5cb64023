import os, sys, numpy as np, torch, math
sys.path.insert(0, "/root/repo")
from deepi2p_amd import ops, synthetic, _lib
from deepi2p_amd.registration import RegistrationPipeline
F, N, R, H, W = 32, 20480, 60, 160, 512
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
frames = [synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.05, with_image=False) for _ in range(F)]
pc = torch.from_numpy(np.stack([f["pc"] for f in frames])).to(dev)
K = torch.from_numpy(np.stack([f["K"] for f in frames])).to(dev)
lab = torch.from_numpy(np.stack([f["labels"] for f in frames])).to(dev)
pipe = RegistrationPipeline(H, W, R=R, seed=1)
restarts = pipe.draw(F, dev)
yaw0, lab_front, has = ops.initial_guess(pc.double(), lab)
sweeps = torch.zeros((F, R), dtype=torch.int32, device=dev)
params, cost, iters = ops.solve_batched(pc, lab_front, K, restarts[0], restarts[1], H, W, pipe.lb, pipe.ub, 500, True, yaw0=yaw0, sweeps=sweeps)
# cost after 0 iterations (max_iter=0): initial cost
p0, c0, i0 = ops.solve_batched(pc, lab_front, K, restarts[0], restarts[1], H, W, pipe.lb, pipe.ub, 0, True, yaw0=yaw0)
sw = sweeps.cpu().numpy().astype(float); c0 = c0.cpu().numpy(); cf = cost.cpu().numpy()
noise = restarts[0].cpu().numpy(); tz = restarts[1][..., 2].cpu().numpy()
yaw_gt = np.array([f["yaw_gt"] for f in frames]); tz_gt = np.array([f["t_gt"][2] for f in frames])
y0 = yaw0.cpu().numpy()
def wrap(a): return (a + np.pi) % (2*np.pi) - np.pi
dyaw = np.abs(wrap(noise + y0[:, None] - yaw_gt[:, None])); dtz = np.abs(tz - tz_gt[:, None])
for name, v in (("init cost", c0), ("|noise|", np.abs(noise)), ("|dyaw to gt|", dyaw), ("|dtz to gt|", dtz), ("final cost", cf), ("c0 - cf", c0 - cf), ("log c0/cf", np.log(c0/cf))):
    print("corr(sweeps, %s) = %.3f ; spearman-ish %.3f" % (name, np.corrcoef(sw.ravel(), v.ravel())[0, 1], np.corrcoef(np.argsort(np.argsort(sw.ravel())), np.argsort(np.argsort(v.ravel())))[0, 1]))
# per frame rank correlation of c0 with sweeps
print("sweeps mean per frame min/max", sw.mean(1).min(), sw.mean(1).max())
# how good is sorting by c0: fraction of top-25% longest captured in top-25% by c0
k = sw.size // 4
top_s = set(np.argsort(-sw.ravel())[:k]); 
for name, v in (("init cost", c0), ("log c0/cf", np.log(c0/cf)), ("dyaw", dyaw), ("dtz", dtz)):
    top_v = set(np.argsort(-v.ravel())[:k]); print(name, "captures", len(top_s & top_v) / k)

# list-scheduling model: 768 workgroup slots, a hypothesis occupies a slot for sweeps x t_s; per-sweep time shrinks when the
# chip empties (3 WG/CU -> 1 WG/CU: ~59k -> ~35k cycles), modelled by t_s(n_running) = 35k + 24k * min(1, n_running / 768)
import heapq
def simulate(order):
    t = 0.0; running = []   # (remaining sweeps, id)
    pending = list(order)[::-1]
    # event-driven with uniform progress: all running tasks progress at the same per-sweep time
    active = {}
    clock = 0.0
    while pending or active:
        while pending and len(active) < 768:
            i = pending.pop(); active[i] = float(sw.ravel()[i])
        n = len(active)
        ts = (35e3 + 24e3 * min(1.0, n / 768.0)) / 2.4e9
        m = min(active.values())
        clock += m * ts
        for i in list(active):
            active[i] -= m
            if active[i] <= 1e-9: del active[i]
    return clock * 1e3
idx = np.arange(sw.size)
# kernel block order: block b -> f = b % F, r = b // F
blocks = np.array([(b % F) * R + (b // F) for b in range(F * R)])
print("model: index order %.2f ms" % simulate(blocks))
an = np.abs(noise)
order_noise = np.array([(b % F) * R + np.argsort(-an[b % F], kind="stable")[b // F] for b in range(F * R)])
print("model: |noise|-descending order %.2f ms" % simulate(order_noise))
order_lpt = np.array([(b % F) * R + np.argsort(-sw[b % F], kind="stable")[b // F] for b in range(F * R)])
print("model: true longest-first order %.2f ms" % simulate(order_lpt))
print("model: packed lower bound %.2f ms, longest chain alone %.2f ms" % (sw.sum() * 59e3 / 2.4e9 / 768 * 1e3, sw.max() * 35e3 / 2.4e9 * 1e3))
