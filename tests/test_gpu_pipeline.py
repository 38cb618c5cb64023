"""The product's stream / graph executor (deepi2p_amd/pipeline.py): a graph-replayed step is BIT-identical to eager launches and to the
plain operator calls (MMClassifer.inference_pass-style argmax + RegistrationPipeline), with and without the H2D copies in the step,
on every slot; new host data reaches the device whichever way the copies are issued (second stream per slot, the slot's stream, graph nodes)."""
import numpy as np
import pytest
import torch

from deepi2p_amd import ops, synthetic

pytestmark = pytest.mark.gpu
NAMES = ("pc", "intensity", "sn", "node_a", "node_b", "img")
KEYS = ("pred", "P", "cost", "best", "costs", "iters", "params", "yaw0")


def _setup(dev, B=3, N=2048, H=64, W=128, R=6):
    from deepi2p_amd.networks import MMClassiferCoarse
    from deepi2p_amd.registration import RegistrationPipeline
    opt = synthetic.OptLike(N, H, W, False)
    opt.device = dev
    mm = MMClassiferCoarse(opt)
    mm.detector.load_state_dict(synthetic.synthetic_state_dict(opt))
    batches = [synthetic.make_batch(500 + i, B, N=N, H=H, W=W) for i in range(3)]
    host = [{k: torch.from_numpy(b[k]) for k in NAMES} for b in batches]
    K = torch.from_numpy(batches[0]["K"]).to(dev)
    pipe = RegistrationPipeline(H, W, R=R, seed=3)
    restarts = pipe.draw(B, dev)
    return mm, pipe, K, restarts, batches, host


def _reference(mm, pipe, K, restarts, hb, labels, dev):
    d = {k: hb[k].to(dev) for k in NAMES}
    pred = ops.argmax_channels(mm.detector(*[d[k] for k in NAMES]))
    out = pipe(d["pc"], labels if labels is not None else pred, K, restarts)
    out["pred"] = pred
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("override,h2d_mode", [(False, "copy_stream"), (True, "copy_stream"), (True, "eager"), (True, "graph")])
def test_graph_replay_equals_eager_and_operator_calls(dev, override, h2d_mode):
    from deepi2p_amd.pipeline import RegistrationExecutor
    mm, pipe, K, restarts, batches, host = _setup(dev)
    labels = torch.from_numpy(batches[0]["labels"]).to(dev) if override else None     # the benchmark's synthetic labels / the network's own
    outs = {}
    for graph in (True, False):
        ex = RegistrationExecutor(mm, pipe, K, host[0], n_streams=2, use_graph=graph, restarts=restarts, labels_override=labels, h2d_mode=h2d_mode)
        ex.warm_up(with_h2d=True)
        assert ex.use_graph == graph, ex.graph_error
        got = []
        for i in (0, 1, 2, 1):                       # four steps over two slots: every slot is reused with NEW host data
            t = ex.submit(host[i])
            got.append({k: ex.result(t)[k].clone() for k in KEYS})
        # resident replay (no H2D in the step) of what slot 0 holds now (batch 2's data... slot order: 0,1,0,1 -> slot 0 holds batch 2)
        t = ex.submit(None, with_h2d=False)
        got.append({k: ex.result(t)[k].clone() for k in KEYS})
        assert ex.latency_ms(t) > 0.0
        outs[graph] = got
    for a, b in zip(outs[True], outs[False]):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k
    for i, g in zip((0, 1, 2, 1, 2), outs[True]):
        ref = _reference(mm, pipe, K, restarts, host[i], labels, dev)
        for k in KEYS:
            assert torch.equal(g[k], ref[k]), (i, k)
    if not override:                                 # random-init weights: nothing predicted inside -> the reference's skip rule
        assert int((outs[True][0]["best"] >= 0).sum()) == int(((outs[True][0]["pred"] == 1).sum(dim=1) > 0).sum())


def test_run_iterator_and_backpressure(dev):
    from deepi2p_amd.pipeline import RegistrationExecutor
    mm, pipe, K, restarts, batches, host = _setup(dev)
    labels = torch.from_numpy(batches[0]["labels"]).to(dev)
    ex = RegistrationExecutor(mm, pipe, K, host[0], n_streams=2, restarts=restarts, labels_override=labels)
    order = [0, 1, 2, 0, 2, 1, 1]
    res = list(ex.run([host[i] for i in order]))
    assert [j for j, _ in res] == list(range(len(order)))
    refs = {i: _reference(mm, pipe, K, restarts, host[i], labels, dev) for i in set(order)}
    for (j, o), i in zip(res, order):
        assert torch.equal(o["P"], refs[i]["P"]) and torch.equal(o["pred"], refs[i]["pred"])
