"""Solver micro-benchmark (GPU box): time / iterations / sweeps of di2p_solve_batched_f32 on config-2 shape."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepi2p_amd import ops, synthetic
from deepi2p_amd.registration import RegistrationPipeline

PFC = int(os.environ.get("PFC", 4))     # clusters per straight-line batch of the classification walk (for the fill figure)
F, N, R, H, W = int(os.environ.get("F", 32)), int(os.environ.get("N", 20480)), 60, 160, 512
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
    def timed(reps=6):
        best = 1e9
        for it in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = ops.solve_batched(pc, lab_front, K, restarts[0], restarts[1], H, W, pipe.lb, pipe.ub, 500, True, yaw0=yaw0, sweeps=sweeps)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best, out
    from deepi2p_amd import _lib
    with _lib.option("solver_nocull", 1):
        dt_nocull, (p_nc, c_nc, i_nc) = timed(2)
    dt_nocache = float("nan")
    if _lib.load().di2p_get_option(b"solver_nocache") >= 0:
        with _lib.option("solver_nocache", 1):
            dt_nocache, (p_n, c_n, i_n) = timed()
    dt, (params, cost, iters) = timed()
    print("per-point classification of every cluster (DI2P_SOLVER_NOCULL=1): %.2f ms ; cluster test without the classification cache: %.2f ms ; with it: %.2f ms" % (
        dt_nocull * 1e3, dt_nocache * 1e3, dt * 1e3))
    print("  bit-identical to the unculled solve: params %s cost %s iterations %s" % (torch.equal(p_nc, params), torch.equal(c_nc, cost), torch.equal(i_nc, iters)))
    kept = (lab_front >= 0).sum(1).float().mean().item()
    if os.environ.get("PROF"):
        from deepi2p_amd import _lib
        ver = _lib.load().di2p_version()
        PW = (36 if os.environ.get("LMPROF") else 28) if ver >= 6 else (20 if ver >= 4 else (16 if ver >= 3 else 8))       # int64 words per hypothesis (version 3: finer phases, 4: classification-cache hits, 6: walk batches / rounds)
        prof = torch.zeros((F, R, PW), dtype=torch.int64, device=dev)
        _lib.load().di2p_solver_set_profile_buffer(prof.data_ptr())
        ops.solve_batched(pc, lab_front, K, restarts[0], restarts[1], H, W, pipe.lb, pipe.ub, 500, True, yaw0=yaw0, sweeps=sweeps)
        torch.cuda.synchronize()
        _lib.load().di2p_solver_set_profile_buffer(None)
        p = prof.double().cpu().numpy().reshape(-1, PW); sw = sweeps.cpu().numpy().reshape(-1)
        tot = p[:, :3].sum(1)
        print("  per-sweep cycles (mean over hyps): sweep %.0f  barrier-wait %.0f  LM %.0f ; active evals/sweep (wave0) %.1f of %d records/wave"
              % ((p[:, 0] / sw).mean(), (p[:, 1] / sw).mean(), (p[:, 2] / sw).mean(), (p[:, 3] / sw).mean(), kept / 4))
        print("  clusters per sweep (wave 0): per-point %.1f  all-active %.1f  guard-only %s ; partial-combine cycles per sweep %.0f (part of LM)" % (
            (p[:, 4] / sw).mean(), (p[:, 5] / sw).mean(), ("%.1f" % (p[:, 11] / sw).mean()) if PW >= 16 else "n/a", (p[:, 6] / sw).mean()))
        if PW >= 20:
            print("  classification cache (wave 0, clusters per sweep): re-used masks %.1f (misses = per-point above)  guard walks skipped %.1f (misses = guard-only above)" % (
                (p[:, 16] / sw).mean(), (p[:, 17] / sw).mean()))
        if PW >= 28:
            print("  walk (wave 0, per sweep): cluster-test rounds %.2f  guard batches %.2f (fill %.2f)  classification batches %.2f (fill %.2f)  phase-II appends %.1f  phase-B rounds label-1 %.2f label-0 %.2f  guard-only clusters on T/B/Z only %.1f" % (
                (p[:, 23] / sw).mean(), (p[:, 18] / sw).mean(), p[:, 11].sum() / max(4 * p[:, 18].sum(), 1), (p[:, 19] / sw).mean(), p[:, 4].sum() / max(PFC * p[:, 19].sum(), 1),
                (p[:, 20] / sw).mean(), (p[:, 21] / sw).mean(), (p[:, 22] / sw).mean(), (p[:, 24] / sw).mean()))
        if PW >= 16:
            print("  LM stages per sweep: combine %.0f  decide %.0f  wave minimiser %.0f  finish+begin %.0f" % tuple((p[:, i] / sw).mean() for i in (6, 8, 9, 10)))
            print("  inside the sweep (wave 0, cycles per sweep): set-up %.0f  cluster-test rounds %.0f  phase B (drains) %.0f  log+reduction %.0f  -> cluster walk / phase A %.0f" % (
                (p[:, 14] / sw).mean(), (p[:, 12] / sw).mean(), (p[:, 13] / sw).mean(), (p[:, 15] / sw).mean(), ((p[:, 0] - p[:, 12] - p[:, 13] - p[:, 14] - p[:, 15]) / sw).mean()))
        print("  sweeps per hypothesis: percentiles 50/75/90/95/99/100 = %s" % np.percentile(sw, [50, 75, 90, 95, 99, 100]).round(0).tolist())
        q = prof.cpu().numpy().reshape(-1, PW)[:, 7]
        print("  line search: extra trials %.1f / hyp, of which accepted %.1f, re-sweeps %.1f (iterations %.1f, sweeps %.1f)" % (
            (q & 0xfffff).mean(), ((q >> 20) & 0xfffff).mean(), (q >> 40).mean(), iters.float().mean().item(), sw.mean()))
        if os.environ.get("LMPROF"):       # variant build -DDI2P_SOLVER_LMPROF: cycles of the Cholesky solves (packed where the re-sweep count was) and of the sincos
            print("  LM detail (cycles per sweep): Cholesky solves %.0f  sincos of the next iterate %.0f" % (((q >> 40) / sw).mean(), (p[:, 25] / sw).mean()))
            it_ = np.maximum(iters.cpu().numpy().reshape(-1).astype(float), 1.0); ex_ = np.maximum((q & 0xfffff).astype(float), 1.0)
            print("  LM stages, cycles per ITERATION: finish %.0f  begin: fetch + scaled matrix %.0f  Cholesky solve %.0f  model change %.0f  step + projection + stores %.0f ; "
                  "per SWEEP: decide without the fit %.0f ; per FAILED TRIAL: interpolating fit %.0f  trial-next %.0f" % (
                      (p[:, 28] / it_).mean(), (p[:, 29] / it_).mean(), (p[:, 30] / it_).mean(), (p[:, 31] / it_).mean(), (p[:, 32] / it_).mean(),
                      ((p[:, 33] - p[:, 34]) / sw).mean(), (p[:, 34] / ex_).mean(), (p[:, 35] / ex_).mean()))
        X = np.stack([iters.cpu().numpy().reshape(-1).astype(float), (q & 0xfffff).astype(float), np.ones(sw.size)], axis=1)
        coef = np.linalg.lstsq(X, p[:, 2], rcond=None)[0]
        print("  LM cycles per hypothesis ~ %.0f x iterations + %.0f x extra line-search trials + %.0f" % tuple(coef))
        i = int(np.argmax(sw))
        print("  slowest hyp: sweeps %d cycles total %.3g (sweep %.3g wait %.3g lm %.3g) active/sweep %.1f" % (sw[i], tot[i], p[i, 0], p[i, 1], p[i, 2], p[i, 3] / sw[i]))
        print("  sum over hyps of block cycles %.3g ; max %.3g" % (tot.sum(), tot.max()))
    for tier in [int(t) for t in os.environ.get("TIERS", "").split(",") if t]:
        with _lib.option("solver_tier_sweeps", tier):
            dtt, (p2, c2, i2) = timed()
        same = torch.equal(c2, cost)
        print("  two-tier launch, park after %d sweeps: %.2f ms (costs bit-identical to the single launch: %s; max rel diff %.2e)" % (
            tier, dtt * 1e3, same, float(((c2 - cost).abs() / cost.abs().clamp(min=1e-30)).max())))
    print("%s CFG=%s: %.2f ms  iters mean %.1f max %d  sweeps mean %.1f max %d  kept pts %.0f  -> %.1f us/sweep/hyp-wave" % (
        name, os.environ.get("DI2P_SOLVER_CFG", "44"), dt * 1e3, iters.float().mean().item(), iters.max().item(),
        sweeps.float().mean().item(), sweeps.max().item(), kept, dt * 1e6 / max(sweeps.float().mean().item(), 1)))
