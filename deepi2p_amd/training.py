"""Training on the device.

classifier_loss        losses + d loss / d scores of models/multimodal_classifier.py:189-191 (FocalLoss of models/focal_loss.py:55-112
                       for the coarse head, cross-entropy over the inside points for the fine head) and the accuracies (:195-201)
adam_step              torch.optim.Adam as the reference builds it (multimodal_classifier.py:44-47) on a flat parameter buffer
allreduce_gradients    the data-parallel gradient exchange (replaces nn.DataParallel's gather/reduce, :37-38): ONE all-reduce of a
                       flat fp32 buffer over RCCL (or gloo in the CPU tests), averaged

ClassifierTrainer      MMClassifer.optimize / test_model (models/multimodal_classifier.py:119-225): labels by projection, train-mode
                       forward and backward on the HIP kernels (deepi2p_amd/train_net.py), gradient all-reduce, Adam
"""
import os

import torch

from . import _lib, ops, prep, train_net
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


def _allreduce_sum(t, group=None):
    """In-place sum over the ranks of a (device) tensor; through host memory when the backend is gloo (CPU tests)."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_gradients(flat_grads, group=None):
    """Average one flat gradient buffer over the ranks (a single all-reduce; no-op without a process group)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat_grads
    if flat_grads.is_cuda and dist.get_backend(group) == "gloo":      # test mode (no RCCL): through host memory
        h = flat_grads.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        flat_grads.copy_(h)
    else:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    flat_grads.div_(dist.get_world_size(group))
    return flat_grads


class ClassifierTrainer:
    """One optimisation step of the reference's MMClassifer (multimodal_classifier.py:213-218) for a
    deepi2p_amd.networks.KeypointDetector on the device.

    All parameters are re-seated as views of ONE flat fp32 buffer (and their gradients as views of one flat gradient buffer): the
    data-parallel exchange is a single all-reduce and the optimiser a single launch.  `optimize` returns device scalars; nothing
    synchronises with the host."""

    def __init__(self, detector, opt, lr=None, betas=(0.9, 0.999), group=None, seed=0):
        import torch.distributed as dist
        # every replica draws its OWN dropout masks (nn.DataParallel replicas do): the rank is folded into the Philox seed
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.detector, self.opt, self.group, self.seed, self.steps = detector, opt, group, int(seed) ^ (rank << 32), 0
        params = [p for _, p in detector.named_parameters()]
        if not params or not params[0].is_cuda:
            raise RuntimeError("the detector must live on a GPU")
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty((n,), dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros((n,), dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.requires_grad_(True)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
                p._di2p_grad = p.grad          # the backward kernels write here directly (train_net._sink): no accumulation launches
                off += k
        self.adam = FlatAdam(self.flat, lr=getattr(opt, "lr", 1e-3) if lr is None else lr, betas=betas)

    def tensors(self):
        P = dict(self.detector.named_parameters())
        P.update(dict(self.detector.named_buffers()))
        return P

    def labels(self, pc, P_gt, K, H, W):
        return prep.project_labels(pc, P_gt, K, H, W, getattr(self.opt, "img_fine_resolution_scale", 32))

    def forward_pass(self, pc, intensity, sn, node_a, node_b, img, K, P_gt, train=True, dropouts="draw", want_grads=True):
        """foraward_pass (sic) of the reference (:119-211): scores, labels, losses, accuracies.  train=False is test_model's
        eval-mode pass (:220-223) on the inference kernels."""
        B, N = pc.shape[0], pc.shape[2]
        H, W = img.shape[2], img.shape[3]
        coarse_labels, fine_labels = self.labels(pc, P_gt, K, H, W)
        fine_on = bool(getattr(self.opt, "is_fine_resolution", True))
        if train:
            P = self.tensors()
            if dropouts == "draw":
                c0, c1 = train_net.head_widths(P)
                dropouts = [train_net.dropout_mask((B, c, N), 0.5, self.seed, 2 * self.steps + i, pc.device) for i, c in enumerate((c0, c1))]
            scores = train_net.keypoint_detector(P, self.opt, pc, intensity, sn, node_a, node_b, img, dropouts,
                                                 branch_streams=bool(getattr(self.opt, "branch_streams", os.environ.get("DI2P_TRAIN_BRANCH_STREAMS", "1") != "0")))
        else:
            with torch.no_grad():
                out = self.detector(pc, intensity, sn, node_a, node_b, img)
            scores = torch.cat(out, dim=1) if fine_on else out
        coarse = scores[:, 0:2].contiguous()
        fine = scores[:, 2:].contiguous() if fine_on else None
        L = classifier_loss(coarse.detach(), coarse_labels, fine.detach() if fine is not None else None, fine_labels if fine_on else None,
                            coarse_loss_alpha=getattr(self.opt, "coarse_loss_alpha", 50.0), want_grads=want_grads and train)
        L["coarse_labels"], L["fine_labels"] = coarse_labels, fine_labels
        return scores, L

    def optimize(self, pc, intensity, sn, node_a, node_b, img, K, P_gt, dropouts="draw"):
        import torch.distributed as dist
        self.flat_grad.zero_()
        scores, L = self.forward_pass(pc, intensity, sn, node_a, node_b, img, K, P_gt, True, dropouts)
        if L["d_fine"] is not None and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # The reference gathers the replicas' scores and takes ONE mean over the inside points of the whole batch
            # (multimodal_classifier.py:189-191).  Here every rank normalised its fine-loss gradient by its OWN inside count and the
            # gradients are averaged over the ranks afterwards: rescale by local_count * world / global_count so that the average is
            # that single mean (device-side, no host synchronisation).  The coarse loss is a mean over B * N points on every rank,
            # equal counts, so its average of means already is the global mean.
            world = dist.get_world_size(self.group)
            local = L["inside"].detach().to(torch.float64).reshape(1).clone()
            total = _allreduce_sum(local.clone(), self.group)
            L["d_fine"].mul_((local * world / total.clamp(min=1.0)).to(torch.float32))
            L["inside_global"] = total[0]
        d = L["d_coarse"] if L["d_fine"] is None else torch.cat((L["d_coarse"], L["d_fine"]), dim=1)
        scores.backward(d)
        allreduce_gradients(self.flat_grad, self.group)
        self.adam.step(self.flat_grad)
        self.steps += 1
        self.detector._invalidate()          # the packed inference operands (folded BN, transposed weights) are stale now
        return L

    # ------------------------------------------------------------------ the same step, forward + losses + backward replayed from ONE hipGraph
    def _capture(self, inputs):
        """Static copies of the step's inputs and dropout masks, two eager passes on a side stream (first-use allocations and library
        set-up happen outside the capture; the BatchNorm buffers they touch are put back), then the capture of zero-gradients + forward +
        losses + backward."""
        dev = inputs[0].device
        self._g_in = [x.clone() for x in inputs]
        P = self.tensors()
        c0, c1 = train_net.head_widths(P)
        B, N = inputs[0].shape[0], inputs[0].shape[2]
        self._g_masks = [torch.empty((B, c, N), dtype=torch.uint8, device=dev) for c in (c0, c1)]
        for i, m in enumerate(self._g_masks):
            m.copy_(train_net.dropout_mask(tuple(m.shape), 0.5, self.seed, 2 * self.steps + i, dev))
        buffers = [(b, b.clone()) for _, b in self.detector.named_buffers()]
        # one stream inside the capture: with the eager step's branch streams captured as parallel branches the replay measured 22.9 ms per step
        # (8 ms of it host time inside the graph launch) against 14.8 ms without them (eager: 13.9 with, 15.0 without; profiles/r06_c42_train_graph.txt)
        had, old = hasattr(self.opt, "branch_streams"), getattr(self.opt, "branch_streams", None)
        self.opt.branch_streams = False

        def fwd_bwd():
            self.flat_grad.zero_()
            scores, L = self.forward_pass(*self._g_in, True, self._g_masks)
            d = L["d_coarse"] if L["d_fine"] is None else torch.cat((L["d_coarse"], L["d_fine"]), dim=1)
            scores.backward(d)
            return L

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd_bwd()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        self._g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self._g):
                self._g_L = fwd_bwd()
        finally:
            if had:
                self.opt.branch_streams = old
            else:
                del self.opt.branch_streams
        self._g_key = tuple(tuple(x.shape) for x in inputs)

    def optimize_graphed(self, pc, intensity, sn, node_a, node_b, img, K, P_gt):
        """optimize() with zero-gradients + forward + losses + backward replayed from one captured hipGraph (the ~1000 launches of a step become
        one); inputs are copied into the capture's static buffers, the dropout masks are drawn into them with the step's seeds, the
        gradient exchange and Adam run eagerly behind the replay.  Same kernels on the same values as optimize(): the same parameters after
        every step (tests/test_gpu_training.py).  NOT the default and not faster: the step is bound by its kernels, not by their launches
        (measured at batch 8: 14.8 ms against 13.9 ms for the eager step, which overlaps the image and point branches on two streams;
        0.3 ms of host time per step instead of 11-13 -- the reason to use it is a busy host).  Single process only (the multi-process fine-loss rescale needs a collective inside the
        step); the capture is redone when an input shape changes.  The returned dict's tensors are the capture's: read them before the
        next call."""
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            return self.optimize(pc, intensity, sn, node_a, node_b, img, K, P_gt)
        inputs = (pc, intensity, sn, node_a, node_b, img, K, P_gt)
        if getattr(self, "_g", None) is None or self._g_key != tuple(tuple(x.shape) for x in inputs):
            self._capture(inputs)
        for dst, src in zip(self._g_in, inputs):
            dst.copy_(src)
        for i, m in enumerate(self._g_masks):
            m.copy_(train_net.dropout_mask(tuple(m.shape), 0.5, self.seed, 2 * self.steps + i, m.device))
        self._g.replay()
        self.adam.step(self.flat_grad)
        self.steps += 1
        self.detector._invalidate()
        return self._g_L

    def test_model(self, pc, intensity, sn, node_a, node_b, img, K, P_gt):
        self.detector.eval()
        return self.forward_pass(pc, intensity, sn, node_a, node_b, img, K, P_gt, train=False, want_grads=False)[1]
