"""GPU parity: index_max / ball_query (HIP, through the C ABI) vs the oracle and the reference-generated
golden vectors.  Bit-exact (integer index outputs)."""
import numpy as np
import pytest
import torch

from oracle import ops_np

pytestmark = pytest.mark.gpu


def _im(dev, data, index, K, **kw):
    from deepi2p_amd import ops
    return ops.index_max(torch.from_numpy(data).to(dev), torch.from_numpy(index).to(dev), K, **kw)


@pytest.mark.parametrize("case", ["rand", "relu_ties", "tiny", "ragged"])
def test_index_max_golden(dev, golden, case):
    g = golden("index_max_golden.npz")
    out = _im(dev, g[case + "_data"], g[case + "_index"], int(g[case + "_K"]))
    assert out.dtype == torch.int32
    np.testing.assert_array_equal(out.cpu().numpy(), g[case + "_out"])


@pytest.mark.parametrize("B,C,N,K", [(2, 32, 20480, 128), (1, 64, 4096, 128), (3, 7, 1000, 5), (1, 1, 1, 1),
                                      (2, 3, 257, 300), (4, 16, 8192, 1024)])
def test_index_max_vs_oracle(dev, B, C, N, K):
    rng = np.random.default_rng(B * 1000 + C)
    data = np.maximum(rng.standard_normal((B, C, N)), 0).astype(np.float32)     # post-ReLU: many ties
    index = rng.integers(0, max(1, K - 2), (B, N)).astype(np.int32)
    out = _im(dev, data, index, K).cpu().numpy()
    np.testing.assert_array_equal(out, ops_np.index_max_forward(data, index, K))


def test_index_max_specials(dev):
    """NaN never wins, values <= -1000 never win, -0.0 == +0.0 (first occurrence), empty cluster -> 0."""
    data = np.array([[[np.nan, -0.0, 0.0, -1000.0, -1500.0, 5.0, 5.0, np.nan]]], dtype=np.float32)
    index = np.array([[0, 1, 1, 2, 2, 3, 3, 3]], dtype=np.int32)
    out = _im(dev, data, index, 6).cpu().numpy()
    np.testing.assert_array_equal(out, ops_np.index_max_forward_loops(data, index, 6))
    np.testing.assert_array_equal(out[0, 0], [0, 1, 0, 5, 0, 0])


def test_index_max_values_fused(dev):
    rng = np.random.default_rng(5)
    B, C, N, K = 2, 32, 4096, 128
    data = np.maximum(rng.standard_normal((B, C, N)), 0).astype(np.float32)
    index = rng.integers(0, K - 5, (B, N)).astype(np.int32)
    mask = np.zeros((B, K), np.float32)
    for b in range(B):
        mask[b, np.unique(index[b])] = 1
    idx, val = _im(dev, data, index, K, return_values=True, mask=torch.from_numpy(mask).cuda())
    ref_idx = ops_np.index_max_forward(data, index, K)
    ref_val = np.take_along_axis(data, ref_idx.astype(np.int64), axis=2) * mask[:, None, :]
    np.testing.assert_array_equal(idx.cpu().numpy(), ref_idx)
    np.testing.assert_array_equal(val.cpu().numpy(), ref_val)


def test_index_max_full_size_properties(dev):
    """BASELINE config-2 size (B=32, C=64, N=20480, K=128): size-independent properties."""
    from deepi2p_amd import ops
    g = torch.Generator(device="cpu").manual_seed(0)
    B, C, N, K = 32, 64, 20480, 128
    data = torch.relu(torch.randn(B, C, N, generator=g)).to(dev)
    index = torch.randint(0, K, (B, N), generator=g, dtype=torch.int32).to(dev)
    idx, val = ops.index_max(data, index, K, return_values=True)
    li = idx.long()
    # (1) the winner belongs to its cluster, (2) its value is the segment max, (3) permuting points and
    # un-permuting indices gives the same maxima (order independence of the max itself)
    owner = torch.gather(index.long().unsqueeze(1).expand(B, C, N), 2, li)
    nonempty = val > 0
    assert torch.all((owner == torch.arange(K, device=dev).view(1, 1, K)) | ~nonempty)
    seg = torch.full((B, C, K), -1000.0, device=dev).scatter_reduce(2, index.long().unsqueeze(1).expand(B, C, N), data, "amax")
    assert torch.equal(torch.where(seg > -1000, seg, torch.zeros_like(seg)), val)
    perm = torch.randperm(N, generator=g).to(dev)
    _, val_p = ops.index_max(data[:, :, perm].contiguous(), index[:, perm].contiguous(), K, return_values=True)
    assert torch.equal(val_p, val)


def test_index_max_errors(dev):
    from deepi2p_amd import index_max
    with pytest.raises(RuntimeError):
        index_max.forward_cuda_shared_mem(torch.zeros(1, 2, 3), torch.zeros(1, 3, dtype=torch.int32), 4)   # CPU tensor
    x = torch.zeros(1, 2, 6, device=dev)[:, :, ::2]
    with pytest.raises(RuntimeError):
        index_max.forward_cuda_shared_mem(x, torch.zeros(1, 3, dtype=torch.int32, device=dev), 4)            # non-contiguous
    out = index_max.forward_cuda(torch.zeros(1, 2, 3, device=dev), torch.zeros(1, 3, dtype=torch.int32, device=dev), 4)
    assert out.shape == (1, 2, 4) and out.dtype == torch.int32


@pytest.mark.parametrize("B,M,N,K,radius", [(2, 16, 1000, 8, 0.3), (1, 128, 20480, 64, 0.02), (3, 5, 63, 4, 0.5),
                                            (2, 7, 300, 500, 0.9), (1, 3, 10, 4, -1.0)])
def test_ball_query_vs_oracle(dev, B, M, N, K, radius):
    from deepi2p_amd import ops
    rng = np.random.default_rng(K)
    d = rng.random((B, M, N)).astype(np.float32)
    d[0, 0, :] = 2.0                 # a row with zero hits
    if N > 5:
        d[0, 1 % M, 3] = radius      # exactly == radius counts as a hit
    out = ops.ball_query(torch.from_numpy(d).to(dev), radius, K).cpu().numpy()
    np.testing.assert_array_equal(out, ops_np.ball_query_forward(d, radius, K))


def test_ball_query_module_api(dev):
    from deepi2p_amd import ball_query
    d = torch.rand(2, 4, 100, device=dev)
    out = ball_query.forward_cuda_shared_mem(d, 0.2, 6)
    assert out.shape == (2, 4, 6) and out.dtype == torch.int32
    with pytest.raises(NotImplementedError):
        ball_query.forward_cuda(d, 0.2, 6)
