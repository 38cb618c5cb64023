"""GPU parity of the point<->node kernels vs plain torch restatements of the reference idioms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(B, N, M, seed):
    g = torch.Generator().manual_seed(seed)
    pc = torch.randn(B, 3, N, generator=g) * 20
    sel = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)])
    nodes = torch.gather(pc, 2, sel.unsqueeze(1).expand(B, 3, M)).contiguous()
    return pc, nodes


def _check_knn(idx, pc, nodes, k):
    """idx must equal torch.topk up to permutations among (near-)exact distance ties."""
    d = torch.norm(pc.unsqueeze(3) - nodes.unsqueeze(2), dim=1)          # B x N x M
    dk, ik = torch.topk(d, k=k, dim=2, largest=False, sorted=True)
    mine = torch.gather(d, 2, idx.long())
    assert torch.allclose(mine, dk, rtol=1e-6, atol=1e-6)                  # same distances in the same order
    mism = (idx.long() != ik)
    if mism.any():                                                          # only allowed at ties
        assert torch.all((mine - dk).abs()[mism] <= 1e-6 * dk[mism].abs() + 1e-7)
    return float(mism.float().mean())


@pytest.mark.parametrize("B,N,M,k", [(2, 1024, 128, 3), (1, 20480, 128, 3), (2, 128, 128, 16), (3, 77, 5, 1), (2, 500, 64, 4)])
def test_knn_nodes(dev, B, N, M, k):
    from deepi2p_amd import ops
    pc, nodes = _scene(B, N, M, 10 + k)
    idx, w = ops.knn_nodes(pc.to(dev), nodes.to(dev), k, want_weights=True)
    frac = _check_knn(idx.cpu(), pc, nodes, k)
    assert frac < 1e-3
    d = torch.norm(pc.unsqueeze(3) - torch.gather(nodes.unsqueeze(2).expand(B, 3, N, M), 3,
                                                   idx.cpu().long().unsqueeze(1).expand(B, 3, N, k)), dim=1)
    w_ref = 1 - d / d.sum(dim=2, keepdim=True)
    if k > 1:
        torch.testing.assert_close(w.cpu(), w_ref, rtol=1e-5, atol=1e-6)


def test_knn_duplicate_nodes_tie_rule(dev):
    """Duplicated nodes (padding by repetition, data/kitti_pc_img_pose_loader.py:158-171) give exact ties:
    the lower node id must come first."""
    from deepi2p_amd import ops
    pc, nodes = _scene(1, 256, 16, 3)
    nodes = torch.cat((nodes, nodes), dim=2).contiguous()                  # node j == node j+16
    idx = ops.knn_nodes(pc.to(dev), nodes.to(dev), 4).cpu()
    assert torch.all(idx[:, :, 1] == idx[:, :, 0] + 16)
    assert torch.all(idx[:, :, 3] == idx[:, :, 2] + 16)


def test_cluster_stats_and_point_input(dev):
    from deepi2p_amd import ops
    B, N, M = 2, 4096, 128
    pc, nodes = _scene(B, N, M, 4)
    nodes[:, :, -3:] = 1e4                                                  # three nodes nobody is assigned to
    inten, sn = torch.rand(B, 1, N), torch.randn(B, 3, N)
    idx = ops.knn_nodes(pc.to(dev), nodes.to(dev), 3)
    mean, mask, min_idx = ops.cluster_stats(pc.to(dev), idx, M)
    mi = idx[:, :, 0].long().cpu()
    assert torch.equal(min_idx.cpu().long(), mi)
    onehot = torch.nn.functional.one_hot(mi, M).float()                     # B x N x M
    cnt = onehot.sum(1)
    ref_mean = torch.einsum("bcn,bnm->bcm", pc.double(), onehot.double()).float() / (cnt.unsqueeze(1) + 1e-5)
    torch.testing.assert_close(mean.cpu(), ref_mean, rtol=2e-6, atol=2e-5)
    assert torch.equal(mask.cpu(), (cnt > 0).float())
    assert float(mask[:, -3:].sum()) == 0 and torch.all(mean[:, :, -3:] == 0)
    centers, aug = ops.build_point_input(pc.to(dev), inten.to(dev), sn.to(dev), mean, min_idx)
    ref_c = torch.gather(mean.cpu(), 2, mi.unsqueeze(1).expand(B, 3, N))
    assert torch.equal(centers.cpu(), ref_c)
    assert torch.equal(aug.cpu(), torch.cat((pc - ref_c, inten, sn), dim=1))
    # determinism: fixed-point accumulation -> bit-identical across runs
    mean2, _, _ = ops.cluster_stats(pc.to(dev), idx, M)
    assert torch.equal(mean, mean2)


def test_interpolate_gather_argmax_channelmax(dev):
    from deepi2p_amd import ops
    g = torch.Generator().manual_seed(0)
    B, C, M, Nq, k = 2, 70, 128, 1000, 3
    feats = torch.randn(B, C, M, generator=g)
    idx = torch.randint(0, M, (B, Nq, k), generator=g, dtype=torch.int32)
    w = torch.rand(B, Nq, k, generator=g)
    out = ops.interpolate(feats.to(dev), idx.to(dev), w.to(dev)).cpu()
    gath = torch.gather(feats.unsqueeze(3).expand(B, C, M, k), 2, idx.long().unsqueeze(1).expand(B, C, Nq, k))
    torch.testing.assert_close(out, (w.unsqueeze(1) * gath).sum(3), rtol=1e-5, atol=1e-6)
    # many columns (training: 20480 points) run the kernel that keeps the 32-channel table slice in LDS: the same sums in the same order, i.e.
    # the same bits as the gather-from-memory kernel on the same columns (which the first 1000 columns alone still select)
    for kk, Nl in ((3, 2500), (1, 1024), (4, 1500)):
        idx_l = torch.randint(0, M, (B, Nl, kk), generator=g, dtype=torch.int32)
        w_l = torch.rand(B, Nl, kk, generator=g)
        big = ops.interpolate(feats.to(dev), idx_l.to(dev), w_l.to(dev)).cpu()
        small = ops.interpolate(feats.to(dev), idx_l[:, :1000].contiguous().to(dev), w_l[:, :1000].contiguous().to(dev)).cpu()
        assert torch.equal(big[:, :, :1000], small)
        g_l = torch.gather(feats.unsqueeze(3).expand(B, C, M, kk), 2, idx_l.long().unsqueeze(1).expand(B, C, Nl, kk))
        torch.testing.assert_close(big, (w_l.unsqueeze(1) * g_l).sum(3), rtol=1e-5, atol=1e-6)
    db, q = torch.randn(B, 3, M, generator=g), torch.randn(B, 3, 50, generator=g)
    kn = torch.randint(0, M, (B, 50, 16), generator=g, dtype=torch.int32)
    gn = ops.gather_neighbors(db.to(dev), q.to(dev), kn.to(dev)).cpu().view(B, 3, 50, 16)
    ref = torch.gather(db, 2, kn.long().view(B, 1, 800).expand(B, 3, 800)).view(B, 3, 50, 16) - q.unsqueeze(3)
    assert torch.equal(gn, ref)
    s = torch.randn(B, 82, 777, generator=g)
    s[0, 5, 10] = s[0, 3, 10] = 99.0                                         # tie -> first (lowest channel) wins
    am = ops.argmax_channels(s.to(dev)).cpu()
    assert torch.equal(am.long(), torch.max(s, dim=1)[1]) and am[0, 10] == 3
    sl = s.to(dev)[:, 2:, :]                                                 # channel slice with foreign batch stride
    assert torch.equal(ops.argmax_channels(sl).cpu().long(), torch.max(s[:, 2:, :], dim=1)[1])
    assert torch.equal(ops.channel_max(s.to(dev)).cpu(), s.max(dim=2)[0])
    for n in (1, 3, 16, 33, 64, 65):                                         # short rows share a wavefront
        t = torch.randn(3, 37, n, generator=g)
        assert torch.equal(ops.channel_max(t.to(dev)).cpu(), t.max(dim=2)[0])
