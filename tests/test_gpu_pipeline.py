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


def test_double_buffered_inputs_equal_single_set(dev):
    """double_buffer=True: two device input sets per slot, one graph per set, copies of the next batch issued while the slot's step runs --
    the same bits as the single-set executor, including the resident replay of the newest data and tickets kept across a re-submit."""
    from deepi2p_amd.pipeline import RegistrationExecutor
    mm, pipe, K, restarts, batches, host = _setup(dev)
    outs = {}
    for dbuf in (False, True):
        ex = RegistrationExecutor(mm, pipe, K, host[0], n_streams=2, restarts=restarts, double_buffer=dbuf)
        ex.warm_up(with_h2d=True)
        assert ex.use_graph and ex.double_buffer == dbuf, ex.graph_error
        got = []
        for i in (0, 1, 2, 1, 0):                   # slot 0: batches 0, 2, 0; slot 1: batches 1, 1 -- submitted without waiting
            got.append(ex.submit(host[i]))
        res = []
        for i, t in enumerate(got[-2:]):             # the last step of each slot
            res.append({k: ex.result(t)[k].clone() for k in KEYS})
        t = ex.submit(None, with_h2d=False)          # resident replay of what the next slot (slot 1) holds: batch 1
        res.append({k: ex.result(t)[k].clone() for k in KEYS})
        outs[dbuf] = res
    for a, b in zip(outs[True], outs[False]):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k


def test_split_solver_graphs_equal_single_graph(dev):
    """split_solver=True (experiment): classifier and pose solve as two graphs on two streams joined by events -- the same bits as the
    one-graph step, with slots reused with new host data while their previous solve may still be running."""
    from deepi2p_amd.pipeline import RegistrationExecutor
    mm, pipe, K, restarts, batches, host = _setup(dev)
    outs = {}
    for split in (False, True):
        ex = RegistrationExecutor(mm, pipe, K, host[0], n_streams=2, restarts=restarts, split_solver=split)
        ex.warm_up(with_h2d=True)
        assert ex.use_graph and ex.split_solver == split, ex.graph_error
        tickets = [ex.submit(host[i]) for i in (0, 1)]
        got = [{k: ex.result(t)[k].clone() for k in KEYS} for t in tickets]
        for i in (2, 1, 0):
            t = ex.submit(host[i])
            got.append({k: ex.result(t)[k].clone() for k in KEYS})
            assert ex.latency_ms(t) > 0.0
        outs[split] = got
    for a, b in zip(outs[True], outs[False]):
        for k in KEYS:
            assert torch.equal(a[k], b[k]), k


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


def test_per_batch_camera_matrix_shape_check_and_weight_updates(dev):
    """(1) K travels with the batch: a host batch that brings its own camera matrices is solved with THEM (the reference's loader hands K
    per batch: visualize_and_save_data.py:81-90); (2) a host batch of another shape is refused instead of being solved as the example's;
    (3) after the classifier's weights change the executor re-packs and re-captures: the next step equals the plain operator calls with
    the NEW weights."""
    from deepi2p_amd.pipeline import RegistrationExecutor
    mm, pipe, K, restarts, batches, host = _setup(dev)
    labels = torch.from_numpy(batches[0]["labels"]).to(dev)
    ex = RegistrationExecutor(mm, pipe, K, host[0], n_streams=2, restarts=restarts, labels_override=labels)
    base = {k: ex.result(ex.submit(host[0]))[k].clone() for k in KEYS}
    K2 = K.clone()
    K2[:, 0, 0] *= 1.1
    K2[:, 1, 1] *= 0.9
    K2[:, 0, 2] += 3.0
    hb = dict(host[0])
    hb["K"] = K2.cpu().float()                         # f32 like the reference's loader; staged as f64
    for _ in range(3):                                 # both slots see the new K, and then the old one again
        got = {k: ex.result(ex.submit(hb))[k].clone() for k in KEYS}
        ref = _reference(mm, pipe, K2.cpu().float().double().to(dev), restarts, host[0], labels, dev)
        for k in KEYS:
            assert torch.equal(got[k], ref[k]), k
    assert not torch.equal(got["P"], base["P"])
    hb["K"] = K.cpu()
    for _ in range(2):
        again = {k: ex.result(ex.submit(hb))[k].clone() for k in KEYS}
    for k in KEYS:
        assert torch.equal(again[k], base[k]), k
    # (2) shapes are part of the captured graphs
    short = {k: v[:2] for k, v in host[1].items()}
    with pytest.raises(ValueError):
        ex.submit(short)
    with pytest.raises(ValueError):
        ex.submit(host[1], with_h2d=False)
    # (3) new weights: the prediction must follow them
    ex2 = RegistrationExecutor(mm, pipe, K, host[0], n_streams=2, restarts=restarts)          # the network's own labels feed the solver
    before = ex2.result(ex2.submit(host[1]))["pred"].clone()
    sd = {k: v.clone() for k, v in mm.detector.state_dict().items()}
    for k in sd:
        if k.endswith("per_point_pn.layers.2.conv.bias") or k.endswith("per_point_pn.layers.2.conv.weight"):
            sd[k] = -sd[k]                              # flips the two logits of every point
    v0 = mm.detector.weights_version
    mm.detector.load_state_dict(sd)
    assert mm.detector.weights_version != v0
    after = {k: ex2.result(ex2.submit(host[1]))[k].clone() for k in KEYS}
    assert getattr(ex2, "weights_refreshes", 0) == 1
    ref = _reference(mm, pipe, K, restarts, host[1], None, dev)
    for k in KEYS:
        assert torch.equal(after[k], ref[k]), k
    assert not torch.equal(after["pred"], before)
    # (4) round-5 advisor finding: an executor pins the bf16x3 splits of ITS model only -- a split made for some other operand that happens
    # to be cached in the process when the graphs are captured is not kept alive by it
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(5)
    other_w = (torch.randn(256, 256, generator=g) / 16).to(dev)
    ops.pointwise_gemm([ops.Src(torch.randn(1, 256, 2048, generator=g).to(dev))], other_w, 256, 2048, x3=True)
    foreign = [t for t in ops.x3_live_operands() if all(t is not u for u in ops.x3_live_operands(mm.detector.packed_operands()))]
    assert len(foreign) >= 1
    ex3 = RegistrationExecutor(mm, pipe, K, host[0], n_streams=1, restarts=restarts)
    ex3.result(ex3.submit(host[1]))                    # the graphs are captured at the first step of a slot
    assert len(ex3._x3_refs) >= 1
    assert all(all(t is not f for f in foreign) for t in ex3._x3_refs)
