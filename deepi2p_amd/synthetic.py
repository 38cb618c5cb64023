"""Synthetic KITTI-shaped frames: no datasets are available, so bench and
tests use this seeded generator.  numpy only; nothing here touches the reference or the oracle.

Input contract mirrored: data/kitti_pc_img_pose_loader.py:431-446 (pc 3xN f32 camera frame,
intensity 1xN, sn 3xN unit normals, node_a/node_b = FPS of 1024 random points
(:416-423, data/kitti_helper.py:224-243), img 3xHxW f32 0..255, K 3x3).
"""
import math

import numpy as np
import torch


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

# ----------------------------------------------------------------------------------------
# Option bag + closed-form ("synthetic") weights: no checkpoints can be downloaded, so bench and tests use
# deterministic formula weights (SURVEY.md 8c fixture policy).  Pure data generation, no network compute.
# ----------------------------------------------------------------------------------------
class OptLike:
    """Attribute bag with the field names of kitti/options.py:6-60 used on the path."""

    def __init__(self, input_pt_num=20480, img_H=160, img_W=512, is_fine_resolution=False,
                 node_a_num=128, node_b_num=128, k_ab=16, k_interp_ab=3, k_interp_point_a=3,
                 k_interp_point_b=3, img_fine_resolution_scale=32, batch_size=8):
        self.input_pt_num = input_pt_num
        self.img_H, self.img_W = img_H, img_W
        self.is_fine_resolution = is_fine_resolution
        self.node_a_num, self.node_b_num = node_a_num, node_b_num
        self.k_ab, self.k_interp_ab = k_ab, k_interp_ab
        self.k_interp_point_a, self.k_interp_point_b = k_interp_point_a, k_interp_point_b
        self.img_fine_resolution_scale = img_fine_resolution_scale
        self.batch_size = batch_size
        self.normalization, self.activation, self.norm_momentum = "batch", "relu", 0.1
        self.gpu_ids = [0]


def state_dict_spec(opt):
    """(key, shape) list of the reference KeypointDetector state_dict, in its own order
    (networks_united.py:19-74, networks_pc.py:19-42, resnet.py:125-152)."""
    spec = []

    def pn(prefix, cin, couts, norm_last):
        c = cin
        for i, co in enumerate(couts):
            q = "%s.layers.%d" % (prefix, i)
            spec.append((q + ".conv.weight", (co, c, 1)))
            spec.append((q + ".conv.bias", (co,)))
            if i < len(couts) - 1 or norm_last:
                for s in ("weight", "bias", "running_mean", "running_var"):
                    spec.append((q + ".norm." + s, (co,)))
                spec.append((q + ".norm.num_batches_tracked", ()))
            c = co

    def c2d(prefix, cin, co):
        spec.append((prefix + ".conv.weight", (co, cin, 1, 1)))
        spec.append((prefix + ".conv.bias", (co,)))
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec.append((prefix + ".norm." + s, (co,)))
        spec.append((prefix + ".norm.num_batches_tracked", ()))

    def bn(prefix, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec.append((prefix + "." + s, (c,)))
        spec.append((prefix + ".num_batches_tracked", ()))

    Ca, Cb, Cg = 64, 256, 512
    pn("pc_encoder.first_pointnet", 7, [Ca // 2] * 3, True)
    pn("pc_encoder.second_pointnet", Ca, [Ca, Ca], True)
    c2d("pc_encoder.knnlayer.layers_before.0", 3 + Ca, Cb)
    c2d("pc_encoder.knnlayer.layers_before.1", Cb, Cb)
    c2d("pc_encoder.knnlayer.layers_after.0", 2 * Cb, 2 * Cb)
    c2d("pc_encoder.knnlayer.layers_after.1", 2 * Cb, Cb)
    pn("pc_encoder.final_pointnet", 3 + Cb, [Cg // 2, Cg], True)
    r = "img_encoder.backbone"
    spec.append((r + ".conv1.weight", (64, 3, 7, 7)))
    bn(r + ".bn1", 64)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for bi in range(nb):
            q = "%s.layer%d.%d" % (r, li, bi)
            spec.append((q + ".conv1.weight", (planes, inpl, 3, 3)))
            bn(q + ".bn1", planes)
            spec.append((q + ".conv2.weight", (planes, planes, 3, 3)))
            bn(q + ".bn2", planes)
            if bi == 0 and li > 1:
                spec.append((q + ".downsample.0.weight", (planes, inpl, 1, 1)))
                bn(q + ".downsample.1", planes)
            inpl = planes
    spec.append((r + ".fc.weight", (1000, 512)))
    spec.append((r + ".fc.bias", (1000,)))
    L = int(round(opt.img_H / opt.img_fine_resolution_scale)) * int(round(opt.img_W / opt.img_fine_resolution_scale))
    pn("node_b_attention_pn", 256 + 512, [256, L], False)
    pn("node_b_pn", 256 + 512 + 512 + 512, [1024, 512, 512], False)
    pn("node_a_attention_pn", 64 + 512, [256, L * 4], False)
    pn("node_a_pn", 64 + 256 + 512, [512, 128, 128], False)
    if opt.is_fine_resolution:
        pn("per_point_pn", 736, [256, 256, 2 + L], False)
    else:
        pn("per_point_pn", 736, [128, 128, 2], False)
    return spec


def synthetic_state_dict(opt, seed=0):
    """Closed-form deterministic weights (no RNG state, reproducible anywhere):
    w[i] = amp * sin(a*i + b) with per-tensor (a, b) from the tensor's ordinal.
    conv weights get He-like amplitude so activations stay O(1) through 34 layers;
    BN running_var in [0.5, 1.5], gamma in [0.8, 1.2]; small biases / means."""
    sd = {}
    for t, (key, shape) in enumerate(state_dict_spec(opt)):
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(1, dtype=torch.long)
            continue
        n = 1
        for s in shape:
            n *= s
        i = torch.arange(n, dtype=torch.float64)
        a = 0.731 + 0.0137 * ((t * 7 + seed) % 53)
        b = 0.37 * t + 0.11 * seed
        base = torch.sin(a * i + b)
        if key.endswith("conv.weight") or key.endswith(".conv1.weight") or key.endswith(".conv2.weight") \
                or key.endswith("downsample.0.weight") or key.endswith("fc.weight"):
            fan_in = n // shape[0]
            v = base * math.sqrt(3.0 / fan_in) * 1.3
        elif key.endswith("running_var"):
            v = 1.0 + 0.5 * base
        elif key.endswith("running_mean"):
            v = 0.1 * base
        elif key.endswith("norm.weight") or key.endswith("bn1.weight") or key.endswith("bn2.weight") \
                or key.endswith("downsample.1.weight"):
            v = 1.0 + 0.2 * base
        else:  # biases
            v = 0.05 * base
        sd[key] = v.to(torch.float32).reshape(shape)
    return sd


def random_state_dict(opt, seed=0):
    """He-normal weights, BN gamma in [0.9, 1.1], small biases, fresh running buffers: a freshly initialised network as the
    reference's constructors leave it (layers_pc.py:308-323, resnet.py:148-160), drawn from a seeded CPU generator (the
    training-parity fixtures use it: train-mode gradients of the closed-form weights above are ill-conditioned in fp32)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in state_dict_spec(opt):
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.long)
        elif key.endswith("running_var"):
            sd[key] = torch.ones(shape)
        elif key.endswith("running_mean"):
            sd[key] = torch.zeros(shape)
        elif len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[key] = torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_in)
        elif key.endswith("weight"):
            sd[key] = 1.0 + 0.2 * (torch.rand(shape, generator=gen) - 0.5)
        else:
            sd[key] = 0.1 * torch.randn(shape, generator=gen)
    return sd
