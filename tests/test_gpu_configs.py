"""BASELINE.json configs[2..4] as parity/property test cases (configs[1] is the bench line):
  configs[2]  KITTI shape, coarse+fine head (L=80)                       -> logits vs oracle at reduced N, argmax API
  configs[3]  nuScenes-like 30000 pts / 896x1600 (L=1400), frame-sharded -> full-size forward properties + shard equality
  configs[4]  Oxford-like 40960 pts / 384x640, 256 hypotheses per frame  -> hypothesis-sharded driver == single launch
900x1600 (BASELINE text) is inconsistent with the reference's own shape arithmetic: 896x1600 is used."""
import math

import numpy as np
import pytest
import torch

from deepi2p_amd import synthetic

pytestmark = pytest.mark.gpu


def _det(dev, N, H, W, fine):
    from deepi2p_amd.networks import KeypointDetector
    opt = synthetic.OptLike(N, H, W, fine)
    det = KeypointDetector(opt)
    det.load_state_dict(synthetic.synthetic_state_dict(opt))
    return det.to(dev).eval(), opt


def _inputs(dev, seed, B, N, H, W):
    b = synthetic.make_batch(seed, B, N=N, H=H, W=W)
    return [torch.from_numpy(b[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")], b


def test_config2_fine_head_kitti_shape(dev):
    """KITTI 20480 pts / 160x512 with the fine head: shapes, finiteness, determinism, batch independence, and the
    coarse logits are unaffected by the presence of the fine head's extra output rows (same first 2 rows of layer 2)."""
    det, opt = _det(dev, 20480, 160, 512, True)
    x, _ = _inputs(dev, 21, 3, 20480, 160, 512)
    coarse, fine = det(*x)
    assert coarse.shape == (3, 2, 20480) and fine.shape == (3, 80, 20480)
    assert torch.isfinite(coarse).all() and torch.isfinite(fine).all()
    c2, f2 = det(*x)
    assert torch.equal(coarse, c2) and torch.equal(fine, f2)
    one = det(*[t[1:2].contiguous() for t in x])
    assert torch.equal(one[0][0], coarse[1]) and torch.equal(one[1][0], fine[1])
    from deepi2p_amd import ops
    lab = ops.argmax_channels(fine)
    assert torch.equal(lab.long(), fine.argmax(1)) and int(lab.max()) < 80


def test_config3_nuscenes_shape_frame_sharding(dev):
    """30000 points, 896x1600 image (L = 28*50 = 1400 fine classes): a batch of 2 equals its two frame shards run
    separately (what the 8-GPU data-parallel run does; no collective on the data path)."""
    from deepi2p_amd.distributed import shard_frames
    N, H, W = 30000, 896, 1600
    det, opt = _det(dev, N, H, W, True)
    x, _ = _inputs(dev, 22, 2, N, H, W)
    coarse, fine = det(*x)
    assert coarse.shape == (2, 2, N) and fine.shape == (2, 1400, N) and torch.isfinite(fine).all()
    names = ("pc", "intensity", "sn", "node_a", "node_b", "img")
    for rank in range(2):
        shard, (lo, hi) = shard_frames(dict(zip(names, x)), rank, 2)
        c, f = det(*[shard[k].contiguous() for k in names])
        assert torch.equal(c[0], coarse[lo]) and torch.equal(f[0], fine[lo])


def test_config4_oxford_shape_hypothesis_fanout(dev):
    """40960 points, 384x640, 256 hypotheses per frame: the rank-sliced solve + gather + argmin of
    deepi2p_amd.distributed (here world = 1 and an emulated 8-way split) equals one 256-hypothesis launch."""
    from deepi2p_amd import ops
    from deepi2p_amd.distributed import shard_range, solve_hypotheses_sharded
    from deepi2p_amd.registration import RegistrationPipeline
    N, H, W, R, F = 40960, 384, 640, 256, 2
    rng = np.random.default_rng(23)
    frames = [synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.03, with_image=False) for _ in range(F)]
    pc = torch.from_numpy(np.stack([f["pc"] for f in frames])).to(dev)
    lab = torch.from_numpy(np.stack([f["labels"] for f in frames])).to(dev)
    K = torch.from_numpy(np.stack([f["K"] for f in frames])).to(dev)
    pipe = RegistrationPipeline(H, W, R=R, seed=2)
    noise, Ts = pipe.draw(F, dev)
    full = pipe(pc, lab, K, (noise, Ts))
    pts64 = pc.double()
    yaw0, lab_front, has = ops.initial_guess(pts64, lab)

    def solve_fn(iy, iT):
        p, c, _ = ops.solve_batched(pc, lab_front, K, iy, iT, H, W, pipe.lb, pipe.ub, 500, True, yaw0=yaw0)
        return p, c
    best, bp, bc, allc = solve_hypotheses_sharded(solve_fn, noise, Ts)          # world = 1 path
    assert torch.equal(best.int(), full["best"]) and torch.equal(bc, full["cost"]) and torch.equal(allc, full["costs"])
    # emulate 8 ranks: each solves its slice; concatenated costs/params must be bit-identical to the single launch
    parts = []
    for r in range(8):
        lo, hi = shard_range(R, r, 8)
        parts.append(solve_fn(noise[:, lo:hi].contiguous(), Ts[:, lo:hi].contiguous()))
    assert torch.equal(torch.cat([c for _, c in parts], dim=1), full["costs"])
    assert torch.equal(torch.cat([p for p, _ in parts], dim=1), full["params"])
    for i, f in enumerate(frames):
        from deepi2p_amd.registration import get_P_diff
        t, rr = get_P_diff(full["P"][i].cpu().numpy(), f["P_gt"])
        assert t < 2.0 and rr < 5.0
