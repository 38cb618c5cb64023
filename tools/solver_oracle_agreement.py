import sys, math, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from deepi2p_amd import registration, synthetic
from oracle import frustum_lm as flm
LB, UB = [-5, -0.1, -10], [5, 0.1, 10]
def agreement(po, pg, is_2d):
    toff = 1 if is_2d else 3
    dt = np.linalg.norm(po[:, toff:] - pg[:, toff:], axis=1); dr = np.linalg.norm(po[:, :toff] - pg[:, :toff], axis=1)
    return (dt <= 1e-3) & (dr <= 1e-3), dt, dr
for seed, is_2d, N, HW, flip, R in [(31, True, 8192, (384, 640), 0.1, 12), (32, False, 3000, (64, 128), 0.02, 12), (33, True, 20480, (160, 512), 0.05, 12),
                                   (41, True, 20480, (160, 512), 0.05, 60), (42, True, 20480, (160, 512), 0.05, 60), (43, False, 8192, (160, 512), 0.05, 24)]:
    h, w = HW
    rng = np.random.default_rng(seed)
    f = synthetic.make_frame(rng, N=N, H=h, W=w, flip=flip, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    _, y0, pcf, labf = flm.get_initial_guess(pts, lab)
    ys, Ts = flm.draw_restarts(rng, R, y0, 10 * math.pi / 180, 10)
    Po, co, it_o, term_o, par_o = flm.solve_restarts(pcf, labf, f["K"], ys, Ts, h, w, LB, UB, 500, is_2d, nthreads=32)
    Pg, cg, best, par_g, it_g = registration.solvePGivenK_batched(pcf, labf, f["K"], ys, Ts, h, w, LB, UB, 500, is_2d, return_all=True)
    ok, dt, dr = agreement(par_o, par_g, is_2d)
    rel = np.abs(cg - co) / co
    print("seed %d 2d=%s N=%d: agree %d/%d ; iters equal %d/%d ; max rel cost diff (agreeing) %.2e ; best cost rel %.2e ; disagreeing cost ratios %s" % (
        seed, is_2d, N, ok.sum(), R, (it_g == it_o).sum(), R, rel[ok].max(), abs(cg.min() - co.min()) / co.min(), np.round(cg[~ok] / co[~ok], 4).tolist()))
