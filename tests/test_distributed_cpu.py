"""CPU, world_size 2, gloo: the N>1 logic of the registration path -- frame sharding (no collective) and the
hypothesis fan-out with ONE all_gather + identical argmin on every rank.  The per-rank solve is the ORACLE here
(test infrastructure standing in for the HIP solver, which needs a GPU); the sharding/gather/argmin code under test
is the product's (deepi2p_amd/distributed.py)."""
import math
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepi2p_amd import synthetic
from deepi2p_amd.distributed import shard_frames, solve_hypotheses_sharded
from oracle import frustum_lm as flm

H, W = 160, 512
LB, UB = [-5, -0.1, -10], [5, 0.1, 10]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame():
    rng = np.random.default_rng(4)
    f = synthetic.make_frame(rng, N=1200, H=H, W=W, with_image=False)
    pts, lab = f["pc"].astype(np.float64), f["labels"]
    _, y0, pcf, labf = flm.get_initial_guess(pts, lab)
    ys, Ts = flm.draw_restarts(rng, 9, y0, 10 * math.pi / 180, 10)      # 9 restarts over 2 ranks: ragged 5 + 4
    return f, pcf, labf, ys, Ts


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f, pcf, labf, ys, Ts = _frame()

    def solve_fn(iy, iT):
        P, cost, it, term, params = flm.solve_restarts(pcf, labf, f["K"], iy[0].numpy(), iT[0].numpy(), H, W, LB, UB, 200, True)
        return torch.from_numpy(params).unsqueeze(0), torch.from_numpy(cost).unsqueeze(0)

    best, bp, bc, allc = solve_hypotheses_sharded(solve_fn, torch.from_numpy(ys).unsqueeze(0), torch.from_numpy(Ts).unsqueeze(0))
    # frame sharding: pure slicing, every frame owned by exactly one rank
    shard, (lo, hi) = shard_frames({"x": torch.arange(7).view(7, 1)}, rank, world)
    q.put((rank, int(best[0]), bp[0].numpy(), float(bc[0]), allc[0].numpy(), (lo, hi), shard["x"].view(-1).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_hypothesis_fanout_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    f, pcf, labf, ys, Ts = _frame()
    P, cost, it, term, params = flm.solve_restarts(pcf, labf, f["K"], ys, Ts, H, W, LB, UB, 200, True)
    for rank, best, bp, bc, allc, rng_, shard in res:
        np.testing.assert_array_equal(allc, cost)                 # gathered in hypothesis order, ragged shards
        assert best == int(np.argmin(cost)) and bc == cost.min()
        np.testing.assert_array_equal(bp, params[best])
    assert res[0][5] == (0, 4) and res[1][5] == (4, 7)
    assert res[0][6] + res[1][6] == list(range(7))


def test_single_process_path_needs_no_process_group():
    ys, Ts = torch.zeros(2, 5, dtype=torch.float64), torch.zeros(2, 5, 3, dtype=torch.float64)

    def solve_fn(iy, iT):
        c = torch.tensor([[3.0, 1.0, 1.0, float("nan"), 2.0]] * 2, dtype=torch.float64)
        return torch.arange(2 * 5 * 4, dtype=torch.float64).view(2, 5, 4), c
    best, bp, bc, _ = solve_hypotheses_sharded(solve_fn, ys, Ts)
    assert best.tolist() == [1, 1] and bc.tolist() == [1.0, 1.0]      # tie -> lowest id, NaN never wins
    assert bp[1].tolist() == [24.0, 25.0, 26.0, 27.0]
