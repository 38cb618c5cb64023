"""Drop-in for the reference's ``ball_query`` extension module (models/ball_query_ext/ball_query.cpp:45-48)."""
from . import ops


def forward_cuda_shared_mem(node_to_point_dist, radius, K):
    return ops.ball_query(node_to_point_dist, radius, K)


def forward_cuda(node_to_point_dist, radius, K):
    # the reference's own forward_cuda is an unimplemented stub (ball_query.cpp:23-31)
    raise NotImplementedError("ball_query.forward_cuda is not implemented in the reference either; use forward_cuda_shared_mem")
