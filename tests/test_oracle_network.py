"""CPU: the oracle network restatement against golden outputs of the IMPORTED REFERENCE (and, where the
reference is present, against the live reference).  Tolerance 1e-4 * max|x| (CPU fp32, same op graph)."""
import numpy as np
import pytest
import torch

from oracle import network_torch as nt


def _run(g):
    B, N, H, W, fine = [int(v) for v in g["meta"]]
    opt = nt.OptLike(N, H, W, bool(fine))
    sd = nt.synthetic_state_dict(opt)
    t = {k: torch.from_numpy(g[k]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    with torch.no_grad():
        c, f, inter = nt.keypoint_detector(sd, opt, t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"],
                                           return_intermediates=True)
    return c, f, inter, opt, sd, t


@pytest.mark.parametrize("fname", ["network_golden.npz", "network_coarse_golden.npz"])
def test_oracle_network_vs_reference_golden(golden, fname):
    g = golden(fname)
    torch.set_num_threads(8)
    c, f, inter, opt, sd, t = _run(g)

    def close(a, name):
        b = g[name]
        assert float(np.abs(a.numpy() - b).max()) <= 1e-4 * float(np.abs(b).max()) + 1e-7, name
    close(c, "coarse")
    if f is not None and "fine" in g:
        close(f, "fine")
    for mine, name in (("pc_center", "pc_centers"), ("cluster_mean", "cluster_mean"), ("first_pn_out", "first_pn_out"),
                       ("second_pn_out", "second_pn_out"), ("node_a_features", "node_a_features"),
                       ("node_b_features", "node_b_features"), ("global_feature", "global_feature"), ("s16", "s16"),
                       ("s32", "s32"), ("img_global", "img_global")):
        close(inter[mine], name)
    assert np.array_equal(inter["a_min_k_idx"].numpy(), g["min_k_idx"])
    assert np.array_equal(c.argmax(1).numpy(), g["coarse"].argmax(1))


def test_state_dict_spec_is_the_reference_layout():
    opt = nt.OptLike(1024, 64, 128, True)
    spec = nt.state_dict_spec(opt)
    assert len(spec) == 361
    keys = [k for k, _ in spec]
    assert keys[0] == "pc_encoder.first_pointnet.layers.0.conv.weight" and spec[0][1] == (32, 7, 1)
    assert "img_encoder.backbone.fc.weight" in keys and "img_encoder.backbone.layer1.0.conv1.weight" in keys
    assert nt.strip_module_prefix({"module." + k: 0 for k in keys}).keys() == set(keys)


def test_index_max_torch_equals_numpy_oracle():
    from oracle import ops_np
    rng = np.random.default_rng(1)
    data = np.maximum(rng.standard_normal((2, 6, 700)), 0).astype(np.float32)
    index = rng.integers(0, 30, (2, 700)).astype(np.int32)
    a = nt.index_max_torch(torch.from_numpy(data), torch.from_numpy(index), 32).numpy()
    np.testing.assert_array_equal(a, ops_np.index_max_forward(data, index, 32))


def test_oracle_network_vs_live_reference():
    from oracle import ref_network as rn
    if not rn.available():
        pytest.skip("reference not present")
    torch.manual_seed(0)
    N, H, W, B = 512, 64, 64, 1
    opt_ref = rn.make_opt(N, H, W, True, B=B)
    det = rn.load_reference_detector(opt_ref)
    opt = nt.OptLike(N, H, W, True)
    sd = nt.synthetic_state_dict(opt, seed=2)
    det.load_state_dict(sd)
    pc = torch.randn(B, 3, N) * 10
    args = (pc, torch.rand(B, 1, N), torch.randn(B, 3, N), pc[:, :, :128].contiguous(), pc[:, :, 128:256].contiguous(),
            torch.rand(B, 3, H, W) * 255)
    with torch.no_grad():
        r = det(*args)
        o = nt.keypoint_detector(sd, opt, *args)
    assert float((r[0] - o[0]).abs().max()) <= 1e-4 * float(r[0].abs().max())
    assert float((r[1] - o[1]).abs().max()) <= 1e-4 * float(r[1].abs().max())


@pytest.mark.parametrize("fine", [False, True])
def test_oracle_network_vs_reference_fullsize_golden(golden, fine):
    """Oracle pin at the BASELINE config-2/3 size (N=20480, 160x512): sub-sampled logits, statistics and all argmax labels
    of the imported reference (tests/golden/network_fullsize_golden.npz)."""
    from tests import fullsize_golden as fg
    g = golden("network_fullsize_golden.npz")
    b, N, H, W, stride = fg.inputs(g)
    torch.set_num_threads(8)
    opt = nt.OptLike(N, H, W, fine)
    sd = nt.synthetic_state_dict(opt)
    t = [torch.from_numpy(b[k]) for k in fg.NAMES]
    with torch.no_grad():
        c, f, inter = nt.keypoint_detector(sd, opt, *t, return_intermediates=True)
    fg.check_logits(g, "fine_model" if fine else "coarse_model", c.numpy(), f.numpy() if fine else None, 1e-4, 1e-4)
    if not fine:
        for name in ("first_pn_out", "second_pn_out", "node_a_features", "node_b_features", "global_feature", "s16", "s32", "img_global"):
            ref = g["stage_%s_stats" % name]
            assert np.abs(fg.stats(inter[name].numpy()) - ref).max() <= 1e-4 * ref[-1] + 1e-7, name
        assert np.abs(inter["global_feature"].numpy() - g["stage_global_feature"]).max() <= 1e-4 * np.abs(g["stage_global_feature"]).max()
