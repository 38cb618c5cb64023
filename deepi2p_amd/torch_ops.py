"""PyTorch custom operators (`torch.library`) over the C-ABI kernels -- the dispatcher-level form of the reference's three torch
extensions (north_star: "called from Python through PyTorch-ROCm custom ops that keep the reference's module/operator API"):

    torch.ops.deepi2p_amd.index_max(data, index, K)                    models/index_max_ext/index_max.cpp:154-159  (forward_cuda_shared_mem)
    torch.ops.deepi2p_amd.index_max_values(data, index, mask, K)       the fused variant of this build: (masked maxima, arg-max)
    torch.ops.deepi2p_amd.ball_query(node_to_point_dist, radius, K)    models/ball_query_ext/ball_query.cpp:45-48
    torch.ops.deepi2p_amd.solve_pose_batched(...)                      evaluation/frustum_reg/src/registration.cpp:190-213 (solvePGivenK), F x R hypotheses
    torch.ops.deepi2p_amd.knn_nodes(query, nodes, k)                   the dense norm + topk of models/networks_pc.py:61-65 as one op

Each op has a schema, a CUDA (= HIP) implementation that calls the library on the current stream, and a fake (meta) implementation,
so the ops are visible to dispatcher-level tooling (`torch.library.opcheck`, FakeTensor / export tracing, the profiler's op names).
`deepi2p_amd.index_max`, `.ball_query` and `.FrustumRegistration` -- the reference's module names -- call these ops.  There is no
CPU kernel registered: on a CPU tensor the dispatcher raises, as the rest of the package does (no fallback).

`index_max_values` also registers autograd (the gradient of the masked maxima goes to the arg-max positions, the segment-max backward
kernel of the training path); the integer-valued ops have no gradient.
"""
import torch

from . import ops

_lib = torch.library.Library("deepi2p_amd", "DEF")
_lib.define("index_max(Tensor data, Tensor index, int K) -> Tensor")
_lib.define("index_max_values(Tensor data, Tensor index, Tensor? mask, int K) -> (Tensor, Tensor)")
_lib.define("ball_query(Tensor node_to_point_dist, float radius, int K) -> Tensor")
_lib.define("knn_nodes(Tensor query, Tensor nodes, int k) -> (Tensor, Tensor)")
_lib.define("solve_pose_batched(Tensor points, Tensor labels, Tensor K, Tensor init_y, Tensor init_T, Tensor? yaw0, float H, float W, "
            "float[] t_lowerbound, float[] t_upperbound, int max_iter, bool is_2d) -> (Tensor, Tensor, Tensor)")


# ------------------------------------------------------------------------------------------------------------ HIP implementations
def _index_max_cuda(data, index, K):
    return ops.index_max(data, index, K)


def _index_max_values_cuda(data, index, mask, K):
    idx, val = ops.index_max(data, index, K, return_values=True, mask=mask)
    return val, idx


def _ball_query_cuda(node_to_point_dist, radius, K):
    return ops.ball_query(node_to_point_dist, radius, K)


def _knn_nodes_cuda(query, nodes, k):
    return ops.knn_nodes(query, nodes, k, want_weights=True)


def _solve_cuda(points, labels, K, init_y, init_T, yaw0, H, W, lb, ub, max_iter, is_2d):
    return ops.solve_batched(points, labels, K, init_y, init_T, H, W, list(lb), list(ub), max_iter, is_2d, yaw0=yaw0)


for _name, _fn in (("index_max", _index_max_cuda), ("index_max_values", _index_max_values_cuda), ("ball_query", _ball_query_cuda),
                   ("knn_nodes", _knn_nodes_cuda), ("solve_pose_batched", _solve_cuda)):
    _lib.impl(_name, _fn, "CUDA")


# ------------------------------------------------------------------------------------------------------------ fake (meta) implementations
@torch.library.register_fake("deepi2p_amd::index_max")
def _(data, index, K):
    B, C, _N = data.shape
    return data.new_empty((B, C, K), dtype=torch.int32)


@torch.library.register_fake("deepi2p_amd::index_max_values")
def _(data, index, mask, K):
    B, C, _N = data.shape
    return data.new_empty((B, C, K)), data.new_empty((B, C, K), dtype=torch.int32)


@torch.library.register_fake("deepi2p_amd::ball_query")
def _(node_to_point_dist, radius, K):
    B, M, _N = node_to_point_dist.shape
    return node_to_point_dist.new_empty((B, M, K), dtype=torch.int32)


@torch.library.register_fake("deepi2p_amd::knn_nodes")
def _(query, nodes, k):
    B, _three, Nq = query.shape
    return query.new_empty((B, Nq, k), dtype=torch.int32), query.new_empty((B, Nq, k))


@torch.library.register_fake("deepi2p_amd::solve_pose_batched")
def _(points, labels, K, init_y, init_T, yaw0, H, W, lb, ub, max_iter, is_2d):
    F, R = init_y.shape
    return (init_y.new_empty((F, R, 4 if is_2d else 6)), init_y.new_empty((F, R)), init_y.new_empty((F, R), dtype=torch.int32))


# ------------------------------------------------------------------------------------------------------------ autograd of the fused segment max
def _imv_setup(ctx, inputs, output):
    data, _index, mask, _K = inputs
    ctx.save_for_backward(output[1], mask if mask is not None else data.new_empty(0))
    ctx.has_mask = mask is not None
    ctx.shape = tuple(data.shape)
    ctx.set_materialize_grads(True)


def _imv_backward(ctx, d_val, _d_idx):
    from ._lib import call, ptr, stream
    idx, mask = ctx.saved_tensors
    d_val = d_val.contiguous()
    B, C, M = d_val.shape
    dx = torch.empty(ctx.shape, dtype=torch.float32, device=d_val.device)
    call("di2p_segment_max_backward", ptr(d_val), ptr(idx), ptr(mask) if ctx.has_mask else None, ptr(dx), B, C, ctx.shape[2], M, stream())
    return dx, None, None, None


torch.library.register_autograd("deepi2p_amd::index_max_values", _imv_backward, setup_context=_imv_setup)

OPS = ("index_max", "index_max_values", "ball_query", "knn_nodes", "solve_pose_batched")
