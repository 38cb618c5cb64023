"""Oracle: plain-torch restatement of the training losses.  TEST INFRASTRUCTURE ONLY.

coarse   models/focal_loss.py:55-112 (focal_loss, via FocalLoss(alpha=0.5, gamma=2, reduction='mean'),
         models/multimodal_classifier.py:33) times opt.coarse_loss_alpha (:189)
fine     nn.CrossEntropyLoss over the points inside the image only (:169-190)
PINNED by tests/golden/loss_golden.npz: values and autograd gradients of the IMPORTED reference focal_loss module + torch's
cross-entropy, assembled exactly as foraward_pass does (tests/golden/make_golden.py make_losses)."""
import torch
import torch.nn.functional as F


def focal_loss(scores, labels, alpha=0.5, gamma=2.0):
    p = F.softmax(scores, dim=1) + 1e-6          # FocalLoss passes its self.eps = 1e-6 (focal_loss.py:159,165), not the function default 1e-8
    h = torch.zeros_like(scores).scatter_(1, labels.unsqueeze(1), 1.0) + 1e-6
    focal = -alpha * torch.pow(-p + 1.0, gamma) * torch.log(p)
    return torch.mean(torch.sum(h * focal, dim=1))


def classifier_loss(coarse, fine, coarse_labels, fine_labels, coarse_loss_alpha=50.0):
    """coarse [B,2,N], fine [B,L,N] | None, labels int64 [B,N] -> (loss, coarse loss, fine loss, coarse acc, fine acc)."""
    cl = focal_loss(coarse, coarse_labels) * coarse_loss_alpha
    ca = (coarse.argmax(1) == coarse_labels).float().mean()
    if fine is None:
        return cl, cl, torch.zeros(()), ca, torch.zeros(())
    B, L, N = fine.shape
    inside = coarse_labels.reshape(B * N) == 1
    fs = fine.permute(0, 2, 1).reshape(B * N, L)[inside]
    fl_lab = fine_labels.reshape(B * N)[inside]
    fl = F.cross_entropy(fs, fl_lab)
    fa = (fs.argmax(1) == fl_lab).float().mean()
    return cl + fl, cl, fl, ca, fa
