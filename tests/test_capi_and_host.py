"""CPU: the C-ABI library loads and exports every symbol include/deepi2p_hip.h declares (no compute calls without
a GPU); host-side logic (state-dict layout, weight packing, error behaviour) without touching the device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "deepi2p_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(di2p_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from deepi2p_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    # and the python binding covers exactly the declared ABI
    assert sorted(_lib.EXPORTS) == names
    l = _lib.load()
    assert l.di2p_version() >= 1 and l.di2p_last_error() == b"ok"
    assert l.di2p_solve_workspace_bytes(32, 60, 20480) >= 32 * 20480 * 32


def test_argument_errors_are_reported_not_crashed():
    from deepi2p_amd import _lib
    l = _lib.load()
    with pytest.raises(_lib.DeepI2PHipError, match="k must be"):
        _lib.call("di2p_knn_nodes", None, None, None, None, 1, 10, 8, 99, None)
    with pytest.raises(_lib.DeepI2PHipError, match="null"):
        _lib.call("di2p_conv2d", None, None, None, None, None, None, 1, 3, 8, 8, 4, 3, 3, 1, 1, 0, 0, None)
    # empty problems are valid no-ops (edge cases: B == 0)
    _lib.call("di2p_ball_query_forward", None, None, 0.5, 4, 0, 3, 10, None)
    _lib.call("di2p_index_max_forward", None, None, None, 0, 4, 10, 8, None, None)


def test_host_side_shape_queries_of_the_bf16x3_entry_points():
    """Pure host logic behind the C ABI (no device needed): which shapes the round-5 kernels take, and the sizes of their packed operands."""
    from deepi2p_amd import _lib
    lib = _lib.load()
    # stem + pool in one launch: H % 4 == 0, W % 128 == 0, W <= 512 (include/deepi2p_hip.h)
    assert lib.di2p_stem_x3_supported(160, 512) == 1 and lib.di2p_stem_x3_supported(4, 128) == 1
    for H, W in ((30, 512), (160, 640), (160, 192), (0, 128), (2, 128)):
        assert lib.di2p_stem_x3_supported(H, W) == 0, (H, W)
    assert lib.di2p_stem_x3_packed_bytes() == 11 * 2 * 3 * 64 * 16           # [K-step][channel tile][plane][lane] x 16 B
    # the head's fragment-ordered weights: [K / 16][4 row tiles][3 planes][64 lanes] x 16 B
    assert lib.di2p_head_x3_packed_bytes(96) == 6 * 4 * 3 * 1024 and lib.di2p_head_x3_packed_bytes(128) == 8 * 4 * 3 * 1024
    assert lib.di2p_head_x3_packed_bytes(100) == 0                           # K % 16 != 0: nothing to pack
    # the seven 3x3 layer shapes of ResNet-34 at 160 x 512 have a bf16x3 instance; odd sizes under stride 2 and Cin < 16 do not
    for Cin, H, W, Cout, s in ((64, 40, 128, 64, 1), (128, 20, 64, 128, 1), (256, 10, 32, 256, 1), (512, 5, 16, 512, 1),
                               (64, 40, 128, 128, 2), (128, 20, 64, 256, 2), (256, 10, 32, 512, 2)):
        assert lib.di2p_conv3x3_x3_supported(32, Cin, H, W, Cout, s) == 1, (Cin, H, W, Cout, s)
    assert lib.di2p_conv3x3_x3_supported(32, 64, 41, 128, 128, 2) == 0 and lib.di2p_conv3x3_x3_supported(32, 8, 40, 128, 64, 1) == 0
    assert lib.di2p_conv3x3_x3_supported(1, 64, 40, 128, 64, 1) == lib.di2p_conv3x3_x3_supported(64, 64, 40, 128, 64, 1) == 1     # batch independent


def test_no_cpu_fallback_paths():
    from deepi2p_amd import ball_query, index_max, ops
    x = torch.zeros(1, 2, 8)
    idx = torch.zeros(1, 8, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="CUDA"):
        index_max.forward_cuda_shared_mem(x, idx, 4)
    with pytest.raises(NotImplementedError):
        index_max.forward_cpu(x, idx, 4)
    with pytest.raises(RuntimeError, match="CUDA"):
        ball_query.forward_cuda_shared_mem(torch.zeros(1, 2, 8), 0.1, 2)
    with pytest.raises(RuntimeError):
        ops.knn_nodes(torch.zeros(1, 3, 8), torch.zeros(1, 3, 4), 2)
    if not torch.cuda.is_available():
        from deepi2p_amd import FrustumRegistration
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            FrustumRegistration.solvePGivenK(np.zeros((3, 4)), np.zeros(4, np.int32), np.eye(3), 0.0, np.zeros(3), 160, 512,
                                             [-1, -1, -1], [1, 1, 1], 5, False, True)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "deepi2p_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("oracle's", "").replace("oracle/", "") \
                    or not f.endswith(".py"), f
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_state_dict_layout_and_packing():
    from deepi2p_amd import networks
    from oracle import network_torch as nt
    opt = nt.OptLike(256, 64, 64, True)
    det = networks.KeypointDetector(opt)
    spec = nt.state_dict_spec(opt)
    sd = det.state_dict()
    assert list(sd.keys()) == [k for k, _ in spec] and len(sd) == 361
    assert all(tuple(sd[k].shape) == tuple(s) for k, s in spec)
    syn = nt.synthetic_state_dict(opt)
    det.load_state_dict(syn)
    # BN folding used by every kernel epilogue: y = scale * conv + shift == BN(conv + bias)
    Wt, scale, shift, act = networks._fold(syn, "pc_encoder.first_pointnet.layers.0.conv", "pc_encoder.first_pointnet.layers.0.norm")
    x = torch.randn(5, 7)
    ref = torch.nn.functional.batch_norm(x @ syn["pc_encoder.first_pointnet.layers.0.conv.weight"][:, :, 0].t()
                                         + syn["pc_encoder.first_pointnet.layers.0.conv.bias"],
                                         syn["pc_encoder.first_pointnet.layers.0.norm.running_mean"],
                                         syn["pc_encoder.first_pointnet.layers.0.norm.running_var"],
                                         syn["pc_encoder.first_pointnet.layers.0.norm.weight"],
                                         syn["pc_encoder.first_pointnet.layers.0.norm.bias"], False, 0.0, 1e-5)
    torch.testing.assert_close((x @ Wt) * scale + shift, ref, rtol=1e-5, atol=1e-6)
    assert act and Wt.shape == (7, 32)
    # last layers have no norm: identity scale, shift = bias, no activation
    Wt2, sc2, sh2, act2 = networks._fold(syn, "per_point_pn.layers.2.conv", "per_point_pn.layers.2.norm")
    assert sc2 is None and not act2 and torch.equal(sh2, syn["per_point_pn.layers.2.conv.bias"])
    # repacking is invalidated by load_state_dict
    det._packed = {"stale": True}
    det.load_state_dict(syn)
    assert det._packed is None
    # 'module.'-prefixed checkpoints (util/pytorch_helper.py:24-33)
    conv = networks.model_state_dict_convert_auto({"module." + k: v for k, v in syn.items()})
    assert list(conv.keys()) == list(syn.keys())


def test_shard_ranges():
    from deepi2p_amd.distributed import shard_range
    for n in (0, 1, 7, 60, 128, 256):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_classifier_checkpoint_and_lr_helpers(tmp_path):
    """save_network / load_model / update_learning_rate of the MMClassifer mirror (multimodal_classifier.py:75-80,263-277): host logic
    only -- the 361-key state_dict round-trips through a file, the learning-rate rule clips at 1e-5."""
    import torch
    from deepi2p_amd import synthetic
    from deepi2p_amd.networks import MMClassifer
    opt = synthetic.OptLike(256, 32, 64, True)
    opt.device, opt.checkpoints_dir, opt.lr = torch.device("cpu"), str(tmp_path), 1e-3
    m = MMClassifer(opt)
    sd = synthetic.random_state_dict(opt, 1)
    m.detector.load_state_dict(sd)
    m.save_network(m.detector, "gpu0_0_net_detector.pth")
    m2 = MMClassifer(opt)
    m2.load_model(str(tmp_path / "gpu0_0_net_detector.pth"))
    got = m2.detector.state_dict()
    assert set(got) == set(sd) and len(got) == 361
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    m.update_learning_rate(0.5)
    assert abs(m.old_lr_detector - 5e-4) < 1e-12
    m.update_learning_rate(1e-6)
    assert m.old_lr_detector == 0.00001
