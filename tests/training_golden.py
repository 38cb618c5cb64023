"""Shared loader of tests/golden/training_golden.npz (the imported reference's train-mode step; see make_golden.py
make_training) and the oracle's restatement of that step."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES = 16


def load():
    z = np.load(os.path.join(HERE, "golden", "training_golden.npz"))
    n = np.load(os.path.join(HERE, "golden", "network_golden.npz"))      # same seeded inputs (seed 7, B=2, N=1024, 64x128)
    inputs = [torch.from_numpy(n[k]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    masks = []
    for i in range(2):
        shp = tuple(int(v) for v in z["mask%d_shape" % i])
        bits = np.unpackbits(z["mask%d" % i])[: int(np.prod(shp))]
        masks.append(torch.from_numpy(bits.reshape(shp).astype(np.uint8)))
    B, N, H, W, fine, wseed = (int(v) for v in z["meta"])
    return dict(z=z, inputs=inputs, masks=masks, B=B, N=N, H=H, W=W, weight_seed=wseed,
                coarse_labels=torch.from_numpy(z["coarse_labels"]), fine_labels=torch.from_numpy(z["fine_labels"]),
                names=[str(s) for s in z["param_names"]], unused=[str(s) for s in z["unused_params"]])


def digest(t):
    a = t.detach().double().reshape(-1).cpu()
    pos = torch.linspace(0, a.numel() - 1, SAMPLES).long()
    return np.array([float(a.norm()), float(a.sum()), float(a.abs().max())]), a[pos].numpy()


def oracle_step(g, dtype=torch.float32):
    """Train-mode forward + backward of oracle/network_torch.py on the fixture's inputs -> (sd with .grad, scores, losses).
    dtype float64: the same graph in double precision (the yardstick for how far two fp32 evaluations may differ)."""
    from oracle import losses_torch as lt
    from oracle import network_torch as nt
    opt = nt.OptLike(g["N"], g["H"], g["W"], True)
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in nt.random_state_dict(opt, g["weight_seed"]).items()}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    with nt.train_mode(momentum=0.1, dropouts=g["masks"]):
        coarse, fine = nt.keypoint_detector(sd, opt, *[t.to(dtype) for t in g["inputs"]])
    loss, cl, fl, ca, fa = lt.classifier_loss(coarse, fine, g["coarse_labels"].long(), g["fine_labels"].long())
    loss.backward()
    return sd, (coarse, fine), dict(loss=loss.item(), coarse=cl.item(), fine=fl.item())


ZERO_GRAD_ABSMAX = 1e-4


def zero_expected(z, i):
    """Parameters whose exact gradient is 0 -- a bias (or shift) that the next train-mode BatchNorm removes again: conv biases in
    front of a norm layer, the last-layer biases of node_a_pn / node_b_pn and the last shift of final_pointnet (they move every
    column of the next BatchNorm's input by the same amount).  The reference's own fp32 gradient there is round-off (< 1e-5)."""
    return z["grad_digest"][i][2] < 1e-5
