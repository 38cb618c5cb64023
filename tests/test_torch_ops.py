"""The torch.library registration of the reference's extension ops (deepi2p_amd/torch_ops.py): schemas, fake (meta) kernels and the
"no CPU kernel" behaviour without a GPU; on the GPU: opcheck, equality with the ctypes front-ends, autograd of the fused segment max."""
import numpy as np
import pytest
import torch

from deepi2p_amd import torch_ops


def test_ops_are_registered_with_schemas():
    for name in torch_ops.OPS:
        op = getattr(torch.ops.deepi2p_amd, name)
        assert str(op.default._schema).startswith("deepi2p_amd::" + name + "(")
    s = str(torch.ops.deepi2p_amd.index_max.default._schema)
    assert s == "deepi2p_amd::index_max(Tensor data, Tensor index, int K) -> Tensor"       # forward_cuda_shared_mem(data, index, K), index_max.cpp:154-159


def test_fake_kernels_give_the_output_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        d = torch.empty(2, 32, 1000, device="cuda")
        i = torch.empty(2, 1000, dtype=torch.int32, device="cuda")
        out = torch.ops.deepi2p_amd.index_max(d, i, 64)
        assert out.shape == (2, 32, 64) and out.dtype == torch.int32 and out.device.type == "cuda"
        val, idx = torch.ops.deepi2p_amd.index_max_values(d, i, None, 64)
        assert val.shape == (2, 32, 64) and val.dtype == torch.float32 and idx.dtype == torch.int32
        bq = torch.ops.deepi2p_amd.ball_query(torch.empty(2, 16, 1000, device="cuda"), 0.3, 8)
        assert bq.shape == (2, 16, 8) and bq.dtype == torch.int32
        ki, kw = torch.ops.deepi2p_amd.knn_nodes(torch.empty(2, 3, 500, device="cuda"), torch.empty(2, 3, 64, device="cuda"), 3)
        assert ki.shape == (2, 500, 3) and kw.dtype == torch.float32
        p, c, it = torch.ops.deepi2p_amd.solve_pose_batched(torch.empty(2, 3, 500, device="cuda"), torch.empty(2, 500, dtype=torch.int32, device="cuda"),
                                                            torch.empty(2, 3, 3, dtype=torch.float64, device="cuda"), torch.empty(2, 5, dtype=torch.float64, device="cuda"),
                                                            torch.empty(2, 5, 3, dtype=torch.float64, device="cuda"), None, 160.0, 512.0, [-5, -0.1, -10], [5, 0.1, 10], 500, True)
        assert p.shape == (2, 5, 4) and c.shape == (2, 5) and it.dtype == torch.int32


def test_no_cpu_kernel_is_registered():
    with pytest.raises(NotImplementedError):
        torch.ops.deepi2p_amd.index_max(torch.zeros(1, 2, 8), torch.zeros(1, 8, dtype=torch.int32), 4)


@pytest.mark.gpu
def test_ops_match_the_front_ends_and_pass_opcheck(dev):
    from deepi2p_amd import ops
    rng = np.random.default_rng(0)
    data = torch.from_numpy(np.maximum(rng.standard_normal((2, 32, 4096)), 0).astype(np.float32)).to(dev)
    index = torch.from_numpy(rng.integers(0, 60, (2, 4096)).astype(np.int32)).to(dev)
    assert torch.equal(torch.ops.deepi2p_amd.index_max(data, index, 64), ops.index_max(data, index, 64))
    torch.library.opcheck(torch.ops.deepi2p_amd.index_max.default, (data, index, 64), test_utils=("test_schema", "test_faketensor"))
    dist = torch.from_numpy(rng.random((2, 8, 512)).astype(np.float32)).to(dev)
    assert torch.equal(torch.ops.deepi2p_amd.ball_query(dist, 0.25, 6), ops.ball_query(dist, 0.25, 6))
    torch.library.opcheck(torch.ops.deepi2p_amd.ball_query.default, (dist, 0.25, 6), test_utils=("test_schema", "test_faketensor"))
    q = torch.from_numpy(rng.standard_normal((2, 3, 777)).astype(np.float32)).to(dev)
    nodes = torch.from_numpy(rng.standard_normal((2, 3, 64)).astype(np.float32)).to(dev)
    ki, kw = torch.ops.deepi2p_amd.knn_nodes(q, nodes, 3)
    ri, rw = ops.knn_nodes(q, nodes, 3, want_weights=True)
    assert torch.equal(ki, ri) and torch.equal(kw, rw)


@pytest.mark.gpu
def test_fused_segment_max_is_differentiable(dev):
    """d(masked maxima)/d(data) routes to the arg-max positions: against torch autograd of gather(data, max_idx) * mask."""
    import deepi2p_amd.index_max as index_max
    rng = np.random.default_rng(1)
    B, C, N, K = 2, 8, 1024, 16
    data = torch.from_numpy(rng.standard_normal((B, C, N)).astype(np.float32)).abs().to(dev).requires_grad_(True)
    index = torch.from_numpy(rng.integers(0, K - 2, (B, N)).astype(np.int32)).to(dev)      # the last two clusters stay empty
    mask = torch.zeros((B, K), device=dev)
    mask[:, : K - 2] = 1.0
    val, idx = index_max.forward(data, index, K, mask)
    g = torch.from_numpy(rng.standard_normal((B, C, K)).astype(np.float32)).to(dev)
    val.backward(g)
    ref = data.detach().clone().requires_grad_(True)
    rv = torch.gather(ref, 2, idx.long()) * mask.unsqueeze(1)
    assert torch.equal(rv.detach(), val.detach())
    rv.backward(g)
    assert torch.allclose(data.grad, ref.grad, rtol=0, atol=0)
