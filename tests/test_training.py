"""Training-side slice: losses + gradients w.r.t. the scores, Adam, gradient all-reduce.

CPU: the torch restatement (oracle/losses_torch.py) against golden values AND autograd gradients of the reference's own
models/focal_loss.py + torch's CrossEntropyLoss, assembled as models/multimodal_classifier.py:169-191 (tests/golden/loss_golden.npz);
the gradient all-reduce over gloo (world_size 2).
GPU: di2p_classifier_loss against the same golden (loss 1e-5 relative, gradients 1e-5 * max|grad|), di2p_adam_step against
torch.optim.Adam."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import losses_torch as lt

CASES = ("small", "kitti_L80", "coarse_only")


def _case(g, name):
    coarse = torch.from_numpy(g[name + "_coarse"])
    clab = torch.from_numpy(g[name + "_clab"])
    fine = torch.from_numpy(g[name + "_fine"]) if (name + "_fine") in g else None
    flab = torch.from_numpy(g[name + "_flab"]) if fine is not None else None
    return coarse, clab, fine, flab


@pytest.mark.parametrize("name", CASES)
def test_loss_oracle_vs_reference_golden(golden, name):
    g = golden("loss_golden.npz")
    coarse, clab, fine, flab = _case(g, name)
    coarse = coarse.clone().requires_grad_(True)
    fine_r = fine.clone().requires_grad_(True) if fine is not None else None
    loss, cl, fl, ca, fa = lt.classifier_loss(coarse, fine_r, clab.long(), flab.long() if flab is not None else None)
    loss.backward()
    assert abs(float(loss) - float(g[name + "_loss"])) <= 1e-6 * float(g[name + "_loss"])
    assert abs(float(cl) - float(g[name + "_coarse_loss"])) <= 1e-6 * float(g[name + "_coarse_loss"])
    assert abs(float(ca) - float(g[name + "_coarse_acc"])) < 1e-7
    ref = g[name + "_d_coarse"]
    assert np.abs(coarse.grad.numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    if fine is not None:
        ref = g[name + "_d_fine"]
        assert np.abs(fine_r.grad.numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
        assert abs(float(fa) - float(g[name + "_fine_acc"])) < 1e-7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ar_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepi2p_amd.training import allreduce_gradients
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_gradients(g)
    q.put((rank, g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.arange(1000, dtype=np.float32) * 1.5
    for _, got in res:
        np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_loss_and_gradients_vs_reference_golden(dev, golden, name):
    from deepi2p_amd import training
    g = golden("loss_golden.npz")
    coarse, clab, fine, flab = _case(g, name)
    out = training.classifier_loss(coarse.to(dev), clab.to(dev), fine.to(dev) if fine is not None else None,
                                   flab.to(dev) if flab is not None else None)
    for key, gk in (("loss", "_loss"), ("coarse", "_coarse_loss")):
        assert abs(float(out[key]) - float(g[name + gk])) <= 1e-5 * float(g[name + gk]), key
    assert abs(float(out["coarse_accuracy"]) - float(g[name + "_coarse_acc"])) < 1e-6
    ref = g[name + "_d_coarse"]
    assert np.abs(out["d_coarse"].cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    if fine is not None:
        assert abs(float(out["fine"]) - float(g[name + "_fine_loss"])) <= 1e-5 * float(g[name + "_fine_loss"])
        assert abs(float(out["fine_accuracy"]) - float(g[name + "_fine_acc"])) < 1e-6
        ref = g[name + "_d_fine"]
        assert np.abs(out["d_fine"].cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
        assert float(out["inside"]) == float((clab == 1).sum())
    out2 = training.classifier_loss(coarse.to(dev), clab.to(dev), fine.to(dev) if fine is not None else None,
                                    flab.to(dev) if flab is not None else None)
    assert torch.equal(out["d_coarse"], out2["d_coarse"]) and float(out["loss"]) == float(out2["loss"])     # deterministic reduction


@pytest.mark.gpu
def test_hip_adam_matches_torch(dev):
    from deepi2p_amd.training import FlatAdam
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(10007, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.999), weight_decay=0)
    mine = p0.clone().to(dev)
    adam = FlatAdam(mine)
    for step in range(5):
        grad = torch.randn(10007, generator=g) * (0.1 + step)
        ref.grad = grad.clone()
        opt.step()
        adam.step(grad.to(dev))
    assert (mine.cpu() - ref.detach()).abs().max() <= 2e-6 * ref.detach().abs().max()
