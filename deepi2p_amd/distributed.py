"""Multi-GPU sharding of the registration path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on the GPU box; "gloo" in the CPU tests).

The path shards two ways and needs no data-path all-reduce:
  * frames are independent            -> contiguous frame shards per rank, no collective at all;
  * restarts of a frame are independent -> each rank solves hypotheses [lo,hi), then ONE small all_gather of
    (cost, params) per frame and a local, identical argmin on every rank (ties -> lowest hypothesis id).
    Replaces the racy Manager().dict() min-reduce of evaluation/registration_lsq.py:136-139,147-184.
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) of n items for `rank` of `world` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_frames(tensors, rank, world):
    """Slice every [B, ...] tensor to this rank's frame shard."""
    B = next(iter(tensors.values())).shape[0]
    lo, hi = shard_range(B, rank, world)
    return {k: v[lo:hi] for k, v in tensors.items()}, (lo, hi)


def solve_hypotheses_sharded(solve_fn, init_y, init_T, group=None, gather_events=None):
    """Config-5 style fan-out.  init_y [F,R], init_T [F,R,3] are identical on every rank; rank r solves its slice
    of R with `solve_fn(init_y_slice, init_T_slice) -> (params [F,r,np], cost [F,r])`, results are all-gathered
    (R*(1+np) doubles per frame: latency-bound, not link-bandwidth-bound) and every rank returns the same
    (best [F], best_params [F,np], best_cost [F], all_cost [F,R]).  gather_events: optional (start, end) CUDA events
    recorded around the all_gather on the current stream (bench.py's collective latency)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    F, R = init_y.shape
    lo, hi = shard_range(R, rank, world)
    params, cost = solve_fn(init_y[:, lo:hi].contiguous(), init_T[:, lo:hi].contiguous())
    npar = params.shape[2]
    if world == 1:
        all_params, all_cost = params, cost
    else:
        width = max(shard_range(R, r, world)[1] - shard_range(R, r, world)[0] for r in range(world))
        buf = torch.full((F, width, npar + 1), float("inf"), dtype=torch.float64, device=cost.device)
        buf[:, : hi - lo, 0] = cost
        buf[:, : hi - lo, 1:] = params
        gathered = [torch.empty_like(buf) for _ in range(world)]
        if gather_events is not None:
            gather_events[0].record()
        dist.all_gather(gathered, buf, group=group)
        if gather_events is not None:
            gather_events[1].record()
        pieces_c, pieces_p = [], []
        for r in range(world):
            l2, h2 = shard_range(R, r, world)
            pieces_c.append(gathered[r][:, : h2 - l2, 0])
            pieces_p.append(gathered[r][:, : h2 - l2, 1:])
        all_cost, all_params = torch.cat(pieces_c, dim=1), torch.cat(pieces_p, dim=1)
    # argmin with ties -> lowest hypothesis id, NaN never wins
    c = torch.where(torch.isnan(all_cost), torch.full_like(all_cost, float("inf")), all_cost)
    # explicit tie rule (not left to argmin's implementation): among the hypotheses that attain the minimum, the lowest id
    cmin = c.min(dim=1, keepdim=True).values
    ids = torch.arange(R, device=c.device).view(1, R).expand(F, R)
    best = torch.where(c == cmin, ids, torch.full_like(ids, R)).min(dim=1).values
    best = best.clamp(max=R - 1)            # all-NaN row: cmin = inf == inf holds everywhere -> id 0; clamp is a guard only
    idx = best.view(F, 1, 1).expand(F, 1, npar)
    return best, torch.gather(all_params, 1, idx).squeeze(1), torch.gather(all_cost, 1, best.view(F, 1)).squeeze(1), all_cost
