"""Synthetic KITTI-shaped frames: no datasets are available, so bench and
tests use this seeded generator.  numpy only; nothing here touches the reference or the oracle.

Input contract mirrored: data/kitti_pc_img_pose_loader.py:431-446 (pc 3xN f32 camera frame,
intensity 1xN, sn 3xN unit normals, node_a/node_b = FPS of 1024 random points
(:416-423, data/kitti_helper.py:224-243), img 3xHxW f32 0..255, K 3x3).
"""
import math

import numpy as np


def farthest_point_sampling(pts, k, start=0):
    """pts 3xM -> k column indices; greedy max-min distance (data/kitti_helper.py:224-243)."""
    M = pts.shape[1]
    sel = np.zeros(k, dtype=np.int64)
    sel[0] = start
    d = np.full(M, np.inf)
    for i in range(1, k):
        d = np.minimum(d, np.sum((pts - pts[:, sel[i - 1]:sel[i - 1] + 1]) ** 2, axis=0))
        sel[i] = int(np.argmax(d))
    return sel


def ry_matrix(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def make_K(H, W, fx_scale=0.7):
    fx = fx_scale * W
    return np.array([[fx, 0, W / 2.0], [0, fx, H / 2.0], [0, 0, 1.0]])


def inside_mask(pc, P, K, H, W):
    """Frustum labels as evaluation/registration_lsq.py:67-84 (<= W-1, z > 0.1)."""
    cam = P[:3, :3] @ pc + P[:3, 3:4]
    px = K[0, 0] * cam[0] / cam[2] + K[0, 2]
    py = K[1, 1] * cam[1] / cam[2] + K[1, 2]
    return (px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1) & (cam[2] > 0.1)


def make_scene(rng, N, r_min=2.0, r_max=80.0):
    ang = rng.uniform(-math.pi, math.pi, N)
    r = np.sqrt(rng.uniform(r_min ** 2, r_max ** 2, N))          # uniform over the annulus area
    y = rng.uniform(-2.0, 3.0, N)
    return np.stack([r * np.cos(ang), y, r * np.sin(ang)], axis=0)  # x right, y down, z forward


def make_frame(rng, N=20480, H=160, W=512, Ma=128, Mb=128, flip=0.05, with_image=True):
    """One frame: network inputs (f32) + GT pose + solver labels (exact frustum labels with `flip`
    random flips emulating classifier error)."""
    pc = make_scene(rng, N)
    yaw = rng.uniform(-math.pi, math.pi)
    t = np.array([rng.uniform(-5, 5), rng.uniform(-0.1, 0.1), rng.uniform(-10, 10)])
    P = np.eye(4)
    P[:3, :3] = ry_matrix(yaw)
    P[:3, 3] = t
    K = make_K(H, W)
    labels = inside_mask(pc, P, K, H, W).astype(np.int32)
    flips = rng.random(N) < flip
    labels_noisy = np.where(flips, 1 - labels, labels).astype(np.int32)
    sub = rng.choice(N, size=min(1024, N), replace=False)
    na = sub[farthest_point_sampling(pc[:, sub], Ma)]
    sub2 = rng.choice(N, size=min(1024, N), replace=False)
    nb = sub2[farthest_point_sampling(pc[:, sub2], Mb)]
    sn = rng.standard_normal((3, N))
    sn /= np.linalg.norm(sn, axis=0, keepdims=True)
    out = dict(pc=pc.astype(np.float32), intensity=rng.random((1, N)).astype(np.float32),
               sn=sn.astype(np.float32), node_a=np.ascontiguousarray(pc[:, na], dtype=np.float32), node_b=np.ascontiguousarray(pc[:, nb], dtype=np.float32),
               K=K, P_gt=P, yaw_gt=yaw, t_gt=t, labels_gt=labels, labels=labels_noisy)
    if with_image:
        out["img"] = rng.uniform(0, 255, (3, H, W)).astype(np.float32)
    return out


def make_batch(seed, B, **kw):
    rng = np.random.default_rng(seed)
    frames = [make_frame(rng, **kw) for _ in range(B)]
    return {k: np.ascontiguousarray(np.stack([np.asarray(f[k]) for f in frames], axis=0)) for k in frames[0]}
