"""Drop-in for the reference's ``index_max`` extension module (models/index_max_ext/index_max.cpp:154-159).

    import deepi2p_amd.index_max as index_max
    idx = index_max.forward_cuda_shared_mem(data, index, K)      # i32[B,C,K]

Same names, argument order and error behaviour (RuntimeError for non-CUDA / non-contiguous input,
index_max.cpp:119-121).  Both CUDA entry points run the same HIP kernel.  There is no CPU path here.
"""
import torch

from . import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.deepi2p_amd.*)


def _check(data, index):
    ops.require_cuda(data, index)          # the reference's CHECK_INPUT (index_max.cpp:119-121): RuntimeError, before the dispatcher


def forward_cuda_shared_mem(data, index, K):
    _check(data, index)
    return torch.ops.deepi2p_amd.index_max(data, index, int(K))


def forward_cuda(data, index, K):
    _check(data, index)
    return torch.ops.deepi2p_amd.index_max(data, index, int(K))


def forward(data, index, K, mask=None):
    """Fused variant (differentiable in `data`): -> (max values with empty clusters zeroed f32[B,C,K], max_idx i32[B,C,K])."""
    _check(data, index)
    return torch.ops.deepi2p_amd.index_max_values(data, index, mask, int(K))


def forward_cpu(data, index, K):
    raise NotImplementedError("deepi2p_amd.index_max is GPU-only: there is deliberately no CPU fallback")


def forward_multi_thread_cpu(data, index, K, thread_num):
    raise NotImplementedError("deepi2p_amd.index_max is GPU-only: there is deliberately no CPU fallback")
