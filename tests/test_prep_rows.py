"""'Next' rows: farthest point sampling, index gather, GT label projection, accuracies and the
7 x N pc_label hand-off record.  CPU: oracle vs goldens from the reference's own FarthestSampler.  GPU: HIP vs oracle."""
import numpy as np
import pytest
import torch

from oracle import prep_np


@pytest.mark.parametrize("i", [0, 1, 2])
def test_fps_oracle_vs_reference_golden(golden, i):
    g = golden("prep_golden.npz")
    far, idx = prep_np.farthest_point_sampling(g["fps%d_pts" % i], int(g["fps%d_k" % i]), int(g["fps%d_init" % i]))
    np.testing.assert_array_equal(idx, g["fps%d_idx" % i])
    np.testing.assert_array_equal(far, g["fps%d_nodes" % i])


@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1, 2])
def test_fps_hip_vs_reference_golden(dev, golden, i):
    """Bit-exact indices (fp64 distances, first-occurrence argmax; case 1 holds duplicated points = exact ties)."""
    from deepi2p_amd import prep
    g = golden("prep_golden.npz")
    pts = torch.from_numpy(g["fps%d_pts" % i]).to(dev).unsqueeze(0)
    init = torch.tensor([int(g["fps%d_init" % i])], dtype=torch.int32, device=dev)
    nodes, idx = prep.farthest_point_sampling(pts, int(g["fps%d_k" % i]), init)
    np.testing.assert_array_equal(idx[0].cpu().numpy(), g["fps%d_idx" % i])
    np.testing.assert_array_equal(nodes[0].double().cpu().numpy(), g["fps%d_nodes" % i])
    # drop-in class (numpy in / numpy out, same return types as the reference)
    far, fidx = prep.FarthestSampler().sample(g["fps%d_pts" % i], int(g["fps%d_k" % i]), init_idx=int(g["fps%d_init" % i]))
    assert far.dtype == np.float64 and fidx.dtype == np.int64
    np.testing.assert_array_equal(fidx, g["fps%d_idx" % i])


@pytest.mark.gpu
def test_fps_batch_kitti_shape_and_properties(dev):
    """KITTI shape: 32 frames, 1024 candidates -> 128 nodes (x2 node sets): every frame equals the oracle, indices unique."""
    from deepi2p_amd import prep
    rng = np.random.default_rng(0)
    B, N = 8, 20480
    pc = torch.from_numpy((rng.standard_normal((B, 3, N)) * 25).astype(np.float32)).to(dev)
    cand = torch.from_numpy(np.stack([rng.choice(N, 1024, replace=False) for _ in range(B)]).astype(np.int32)).to(dev)
    init = torch.from_numpy(rng.integers(0, 3, B).astype(np.int32)).to(dev)
    sub = prep.gather_points(pc, cand)
    assert torch.equal(sub, torch.gather(pc, 2, cand.long().unsqueeze(1).expand(B, 3, 1024)))
    nodes, idx = prep.farthest_point_sampling(sub, 128, init)
    for b in range(B):
        far, oi = prep_np.farthest_point_sampling(sub[b].cpu().numpy(), 128, int(init[b]))
        np.testing.assert_array_equal(idx[b].cpu().numpy(), oi)
        assert len(set(oi.tolist())) == 128
    assert torch.equal(prep.sample_nodes(pc, 128, cand, init), nodes)


@pytest.mark.gpu
def test_label_projection_accuracy_and_handoff(dev):
    from deepi2p_amd import prep, synthetic
    H, W, scale = 160, 512, 32
    batch = synthetic.make_batch(3, 4, N=4096, H=H, W=W, with_image=False)
    pc = torch.from_numpy(batch["pc"]).to(dev)
    P = torch.from_numpy(batch["P_gt"][:, :3, :].astype(np.float32)).to(dev).contiguous()
    K = torch.from_numpy(batch["K"].astype(np.float32)).to(dev)
    coarse, fine, pxpy = prep.project_labels(pc, P, K, H, W, scale, want_pxpy=True)
    oc, of, opx = prep_np.project_labels(batch["pc"], batch["P_gt"], batch["K"], H, W, scale)
    # fp32 dot-product association may differ by an ulp: labels must agree except within 1e-3 px of a frustum edge
    torch.testing.assert_close(pxpy.cpu(), torch.from_numpy(opx), rtol=1e-5, atol=1e-3)
    edge = (np.abs(opx[:, 0]) < 1e-2) | (np.abs(opx[:, 0] - (W - 1)) < 1e-2) | (np.abs(opx[:, 1]) < 1e-2) | (np.abs(opx[:, 1] - (H - 1)) < 1e-2)
    assert np.array_equal(coarse.cpu().numpy()[~edge], oc[~edge])
    cell_edge = (np.abs(opx / scale - np.round(opx / scale)) < 1e-4).any(axis=1)
    assert np.array_equal(fine.cpu().numpy()[~cell_edge], of[~cell_edge])
    assert (coarse.cpu().numpy() == batch["labels_gt"]).mean() > 0.9995       # fp32 vs the fp64 generator labels
    # 4x4 P is accepted as well (registration_lsq.py:297-298 pads 3x4 to 4x4)
    P4 = torch.from_numpy(batch["P_gt"].astype(np.float32)).to(dev)
    c4, f4 = prep.project_labels(pc, P4, K, H, W, scale)
    assert torch.equal(c4, coarse) and torch.equal(f4, fine)
    # accuracies and the hand-off record
    g = torch.Generator().manual_seed(0)
    cpred = torch.where(torch.rand(coarse.shape, generator=g).to(dev) < 0.1, 1 - coarse, coarse)
    fpred = torch.where(torch.rand(fine.shape, generator=g).to(dev) < 0.3, fine + 1, fine)
    acc = prep.label_accuracy(cpred, coarse, fpred, fine).cpu().numpy()
    ref = prep_np.accuracy(cpred.cpu().numpy(), coarse.cpu().numpy(), fpred.cpu().numpy(), fine.cpu().numpy())
    np.testing.assert_allclose(acc, ref, rtol=1e-6)
    rec = prep.pack_pc_label(pc, cpred, coarse, fpred, fine)
    assert rec.dtype == torch.float64 and rec.shape == (4, 7, 4096)
    np.testing.assert_array_equal(rec.cpu().numpy(), prep_np.pack_pc_label(batch["pc"], cpred.cpu().numpy(), coarse.cpu().numpy(),
                                                                            fpred.cpu().numpy(), fine.cpu().numpy()))
    # the record is exactly what registration_lsq.py:291-296 unpacks
    r0 = rec[0].cpu().numpy()
    assert np.array_equal(r0[0:3].astype(np.float32), batch["pc"][0]) and np.array_equal(r0[3].astype(np.int64), cpred[0].cpu().numpy())
    # no gt-inside point -> fine accuracy is NaN like np.mean([])
    z = torch.zeros_like(coarse)
    assert np.isnan(prep.label_accuracy(z, z, fpred, fine).cpu().numpy()[:, 1]).all()
