"""Drop-in for the reference's ``ball_query`` extension module (models/ball_query_ext/ball_query.cpp:45-48)."""
import torch

from . import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.deepi2p_amd.*)


def forward_cuda_shared_mem(node_to_point_dist, radius, K):
    ops.require_cuda(node_to_point_dist)
    return torch.ops.deepi2p_amd.ball_query(node_to_point_dist, float(radius), int(K))


def forward_cuda(node_to_point_dist, radius, K):
    # the reference's own forward_cuda is an unimplemented stub (ball_query.cpp:23-31)
    raise NotImplementedError("ball_query.forward_cuda is not implemented in the reference either; use forward_cuda_shared_mem")
