"""Training-side pieces on the device (first slice of SURVEY.md 8f rank 4).

classifier_loss        losses + d loss / d scores of models/multimodal_classifier.py:189-191 (FocalLoss of models/focal_loss.py:55-112
                       for the coarse head, cross-entropy over the inside points for the fine head) and the accuracies (:195-201)
adam_step              torch.optim.Adam as the reference builds it (multimodal_classifier.py:44-47) on a flat parameter buffer
allreduce_gradients    the data-parallel gradient exchange (replaces nn.DataParallel's gather/reduce, :37-38): ONE all-reduce of a
                       flat fp32 buffer over RCCL (or gloo in the CPU tests), averaged

The backward of the network itself (dgrad/wgrad of the contractions, train-mode BatchNorm, the fused gather / segment-max
epilogues) is NOT built: this module is the part of a training step that sits after the logits and after the gradients.
"""
import torch

from . import _lib
from ._lib import call, ptr, require_cuda, stream


def classifier_loss(coarse_scores, coarse_labels, fine_scores=None, fine_labels=None, alpha=0.5, gamma=2.0, coarse_loss_alpha=50.0,
                    want_grads=True):
    """coarse_scores f32[B,2,N], coarse_labels i32[B,N], fine_scores f32[B,L,N] | None, fine_labels i32[B,N]
    -> dict(loss, coarse, fine, coarse_accuracy, fine_accuracy, inside (f64 device scalars), d_coarse, d_fine)."""
    require_cuda(coarse_scores, coarse_labels, fine_scores, fine_labels)
    B, C, N = coarse_scores.shape
    if C != 2 or coarse_scores.dtype != torch.float32 or coarse_labels.dtype != torch.int32:
        raise RuntimeError("coarse_scores must be f32[B,2,N] and coarse_labels i32[B,N]")
    L = fine_scores.shape[1] if fine_scores is not None else 0
    dev = coarse_scores.device
    out = torch.empty((8,), dtype=torch.float64, device=dev)
    d_coarse = torch.empty_like(coarse_scores) if want_grads else None
    d_fine = torch.empty_like(fine_scores) if (want_grads and fine_scores is not None) else None
    ws = torch.empty((_lib.load().di2p_classifier_loss_workspace_bytes(B, N),), dtype=torch.uint8, device=dev)
    call("di2p_classifier_loss", ptr(coarse_scores), ptr(fine_scores), ptr(coarse_labels), ptr(fine_labels), B, N, L, float(alpha),
         float(gamma), float(coarse_loss_alpha), ptr(out), ptr(d_coarse), ptr(d_fine), ptr(ws), stream())
    return dict(loss=out[0], coarse=out[1], fine=out[2], coarse_accuracy=out[3], fine_accuracy=out[4], inside=out[5],
                d_coarse=d_coarse, d_fine=d_fine)


class FlatAdam:
    """Adam over ONE flat fp32 parameter buffer (lr 0.001, betas (0.9, 0.999), eps 1e-8, weight decay 0: kitti/options.py:53)."""

    def __init__(self, flat_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        require_cuda(flat_params)
        self.p, self.lr, self.betas, self.eps, self.t = flat_params, lr, betas, eps, 0
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)

    def step(self, flat_grads):
        self.t += 1
        call("di2p_adam_step", ptr(self.p), ptr(flat_grads), ptr(self.m), ptr(self.v), self.p.numel(), self.t, float(self.lr),
             float(self.betas[0]), float(self.betas[1]), float(self.eps), stream())


def allreduce_gradients(flat_grads, group=None):
    """Average one flat gradient buffer over the ranks (a single all-reduce; no-op without a process group)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat_grads
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    flat_grads.div_(dist.get_world_size(group))
    return flat_grads
