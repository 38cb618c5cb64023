"""The oracle's train-mode step (oracle/network_torch.py train_mode + oracle/losses_torch.py) against the fixture generated from the
IMPORTED reference (tests/golden/training_golden.npz): losses, train-mode scores and the gradient of every parameter."""
import numpy as np
import torch

from tests import training_golden as tg


def test_oracle_train_step_matches_reference_fixture():
    torch.set_num_threads(8)
    g = tg.load()
    z = g["z"]
    sd, (coarse, fine), L = tg.oracle_step(g)
    assert abs(L["loss"] - float(z["loss"])) <= 2e-5 * abs(float(z["loss"]))
    assert abs(L["coarse"] - float(z["coarse_loss"])) <= 2e-5 * abs(float(z["coarse_loss"]))
    assert abs(L["fine"] - float(z["fine_loss"])) <= 2e-5 * abs(float(z["fine_loss"]))
    np.testing.assert_allclose(coarse.detach().numpy()[:, :, ::8], z["coarse_sub"], rtol=0, atol=2e-4 * np.abs(z["coarse_sub"]).max())
    np.testing.assert_allclose(fine.detach().numpy()[:, :, ::8], z["fine_sub"], rtol=0, atol=2e-4 * np.abs(z["fine_sub"]).max())
    names = g["names"]
    assert set(names) | set(g["unused"]) == {k for k, v in sd.items() if v.requires_grad}
    for k in g["unused"]:
        assert sd[k].grad is None
    worst = 0.0
    for i, k in enumerate(names):
        d, s = tg.digest(sd[k].grad)
        ref_d, ref_s = z["grad_digest"][i], z["grad_samples"][i]
        if tg.zero_expected(z, i):
            assert d[2] < tg.ZERO_GRAD_ABSMAX, k
            continue
        scale = ref_d[2]
        assert abs(d[0] - ref_d[0]) <= 2e-3 * ref_d[0], (k, d, ref_d)
        err = np.abs(s - ref_s).max() / scale
        worst = max(worst, err)
        assert err <= 2e-3, (k, err)
    for j, k in enumerate(z["buffer_names"]):
        d, _ = tg.digest(sd[str(k)])
        assert abs(d[0] - z["buffer_digest"][j][0]) <= 1e-4 * z["buffer_digest"][j][0] + 1e-7, k
    print("worst sampled gradient error / abs-max:", worst)
