"""Oracle restatements of the "next" rows.  TEST INFRASTRUCTURE ONLY.

farthest_point_sampling   data/kitti_helper.py:231-243 (FarthestSampler.sample).  PINNED by
                          tests/golden/prep_golden.npz generated from the reference class itself.
project_labels / accuracy / pc_label record
                          models/multimodal_classifier.py:135-156,194-199 (foraward_pass) and
                          evaluation/visualize_and_save_data.py:100-115,138-147,174-181 (the evaluation script's loop body).
                          PINNED by tests/golden/forward_pass_golden.npz and eval_script_golden.npz: the reference's own method /
                          loop body, extracted with ``ast`` and run against stub collaborators (tests/golden/make_golden.py
                          make_forward_pass / make_eval_script) -- coarse labels, fine labels of the inside points, pixel
                          coordinates, accuracies (sums and the printed lines) and the saved 7 x N records.
"""
import numpy as np


def farthest_point_sampling(pts, k, init_idx):
    pts = np.asarray(pts)
    far = np.zeros((3, k))
    idx = np.zeros(k, dtype=np.int64)
    far[:, 0] = pts[:, init_idx]
    idx[0] = init_idx
    dist = ((far[:, 0:1] - pts) ** 2).sum(axis=0)
    for i in range(1, k):
        j = int(np.argmax(dist))
        far[:, i] = pts[:, j]
        idx[i] = j
        dist = np.minimum(dist, ((far[:, i:i + 1] - pts) ** 2).sum(axis=0))
    return far, idx


def project_labels(pc, P, K, H, W, scale):
    """float32 arithmetic like the reference's torch tensors.  pc [B,3,N], P [B,3|4,4], K [B,3,3]."""
    pc, P, K = np.asarray(pc, np.float32), np.asarray(P, np.float32), np.asarray(K, np.float32)
    B, _, N = pc.shape
    homo = np.concatenate((pc, np.ones((B, 1, N), np.float32)), axis=1)
    cam = np.einsum("brk,bkn->brn", P[:, :3, :], homo).astype(np.float32)
    kp = np.einsum("brk,bkn->brn", K, cam).astype(np.float32)
    pxpy = (kp[:, 0:2] / kp[:, 2:3]).astype(np.float32)
    inside = (pxpy[:, 0] >= 0) & (pxpy[:, 0] <= W - 1) & (pxpy[:, 1] >= 0) & (pxpy[:, 1] <= H - 1) & (cam[:, 2] > 0.1)
    W_fine = int(round(W / scale))
    cell = np.floor(pxpy / np.float32(scale)).astype(np.int64)
    fine = cell[:, 0] + cell[:, 1] * W_fine
    return inside.astype(np.int32), fine.astype(np.int32), pxpy


def accuracy(coarse_pred, coarse_gt, fine_pred, fine_gt):
    out = np.zeros((coarse_pred.shape[0], 2), np.float32)
    for b in range(coarse_pred.shape[0]):
        out[b, 0] = np.mean((coarse_pred[b] == coarse_gt[b]).astype(np.float64))
        m = coarse_gt[b] == 1
        out[b, 1] = np.mean((fine_pred[b][m] == fine_gt[b][m]).astype(np.float64)) if m.any() else np.nan
    return out


def pack_pc_label(pc, coarse_pred, coarse_gt, fine_pred, fine_gt):
    return np.concatenate((pc.astype(np.float64), coarse_pred[:, None].astype(np.float64), coarse_gt[:, None].astype(np.float64),
                           fine_pred[:, None].astype(np.float64), fine_gt[:, None].astype(np.float64)), axis=1)
