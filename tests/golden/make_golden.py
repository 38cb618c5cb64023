"""Generates the committed golden fixtures from the REFERENCE itself (run in the build container only,
where /root/reference exists):   python tests/golden/make_golden.py

  index_max_golden.npz   reference models/index_max_ext/index_max.cpp forward_cpu (compiled unmodified into
                         oracle/_ref by oracle/Makefile) on seeded inputs incl. ties, empty clusters, floor cases
  network_golden.npz     reference models/networks_united.py KeypointDetector (imported, CPU, fp32) with the
                         closed-form weights of oracle/network_torch.synthetic_state_dict on seeded inputs:
                         PCEncoder 8-tuple, ImageEncoder maps, coarse+fine logits (B=2, N=1024, 64x128)
  network_coarse_golden.npz   same, coarse-only head
  prep_golden.npz        data/kitti_helper.py FarthestSampler.sample (class extracted with ast), incl. duplicate points
  lsq_driver_golden.npz  evaluation/registration_lsq.py get_initial_guess / wrap_in_pi / get_P_diff /
                         get_inside_img_mask and data/augmentation.py angles2rotation_matrix: the function
                         DEFINITIONS are extracted from the reference files with ``ast`` and exec'd here
                         (the modules themselves need open3d/cv2/TkAgg); only inputs/outputs are stored.
  network_fullsize_golden.npz   the imported reference at the BASELINE config-2/3 size (B=1, N=20480, 160x512), coarse-only
                         and coarse+fine models, on the seeded synthetic frame deepi2p_amd.synthetic.make_batch(41, 1)
                         (regenerated on the GPU box; its SHA-256 is stored to detect generator drift): every 40th point's
                         logits, ALL argmax labels (coarse bit-packed, fine uint8), per-output percentiles / sums, and
                         percentiles of the encoder stages (SURVEY.md 8c fixture policy: "one full-size KITTI-shape run
                         reduced to checksums/percentiles")
  loss_golden.npz        the reference's models/focal_loss.py (imported) + torch CrossEntropyLoss assembled as
                         models/multimodal_classifier.py:169-191: loss values, accuracies and AUTOGRAD gradients w.r.t. the scores
  training_golden.npz    the imported reference KeypointDetector in TRAIN mode (batch-statistics BatchNorm, Dropout(0.5) in per_point_pn
                         with recorded keep-masks: torch.nn.functional.dropout is replaced by a mask multiply for the run) + the
                         reference's FocalLoss / CrossEntropyLoss assembled as foraward_pass does, then loss.backward():
                         train-mode scores (sub-sampled), losses, and for EVERY parameter the gradient's (l2, sum, abs-max) and 16
                         sampled entries, plus the BatchNorm running buffers after the step (B=2, N=1024, 64x128, coarse+fine)
Fixtures hold data only (inputs + expected outputs), never reference source.
"""
import ast
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import network_torch as nt  # noqa: E402
from oracle import ref_network as rn  # noqa: E402
from oracle.ref_loader import load_ref_index_max  # noqa: E402

REF = rn.REF


def network_inputs(seed, B, N, H, W, Ma=128, Mb=128):
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand(B, N, generator=g) * 2 * math.pi
    r = torch.sqrt(torch.rand(B, N, generator=g) * (80.0 ** 2 - 4.0) + 4.0)
    pc = torch.stack((r * torch.cos(ang), torch.rand(B, N, generator=g) * 5 - 2, r * torch.sin(ang)), dim=1)
    inten = torch.rand(B, 1, N, generator=g)
    sn = torch.nn.functional.normalize(torch.randn(B, 3, N, generator=g), dim=1)
    pa = torch.stack([torch.randperm(N, generator=g)[:Ma] for _ in range(B)])
    pb = torch.stack([torch.randperm(N, generator=g)[:Mb] for _ in range(B)])
    node_a = torch.gather(pc, 2, pa.unsqueeze(1).expand(B, 3, Ma)).contiguous()
    node_b = torch.gather(pc, 2, pb.unsqueeze(1).expand(B, 3, Mb)).contiguous()
    img = torch.rand(B, 3, H, W, generator=g) * 255
    return pc.contiguous(), inten, sn.contiguous(), node_a, node_b, img


def make_index_max():
    m = load_ref_index_max()
    rng = np.random.default_rng(123)
    out = {}
    cases = {"rand": (2, 8, 3000, 16), "relu_ties": (3, 5, 2049, 128), "tiny": (1, 1, 7, 4), "ragged": (2, 3, 1001, 37)}
    for name, (B, C, N, K) in cases.items():
        data = rng.standard_normal((B, C, N)).astype(np.float32)
        if name == "relu_ties":
            data = np.maximum(data, 0)              # many exact ties at 0 (and -0.0 below)
            data[0, 0, :7] = -0.0
            data[1, 1, :50] = -1000.0               # exactly the floor: can never win
            data[2, 2, :50] = -2000.0
        index = rng.integers(0, max(1, K - 3), (B, N)).astype(np.int32)   # last clusters stay empty
        if name == "ragged":
            index[0, :] = 5                          # one cluster owns everything
        ref = m.forward_cpu(torch.from_numpy(data), torch.from_numpy(index), K).numpy()
        out[name + "_data"], out[name + "_index"], out[name + "_K"], out[name + "_out"] = data, index, np.int32(K), ref
    np.savez_compressed(os.path.join(HERE, "index_max_golden.npz"), **out)
    print("index_max_golden.npz", {k: v.shape for k, v in out.items() if k.endswith("_out")})


def make_network(fine, fname):
    B, N, H, W = 2, 1024, 64, 128
    opt = rn.make_opt(N, H, W, fine, B=B)
    det = rn.load_reference_detector(opt)
    ol = nt.OptLike(N, H, W, fine)
    det.load_state_dict(nt.synthetic_state_dict(ol))
    pc, inten, sn, na, nb, img = network_inputs(7, B, N, H, W)
    with torch.no_grad():
        enc = det.pc_encoder(pc, inten, sn, na, nb)
        s16, s32, glob = det.img_encoder(img)
        res = det(pc, inten, sn, na, nb, img)
    out = dict(pc=pc, intensity=inten, sn=sn, node_a=na, node_b=nb, img=img,
               pc_centers=enc[0], cluster_mean=enc[1], min_k_idx=enc[2].to(torch.int32), first_pn_out=enc[3],
               second_pn_out=enc[4], node_a_features=enc[5], node_b_features=enc[6], global_feature=enc[7],
               s16=s16, s32=s32, img_global=glob)
    if fine:
        out["coarse"], out["fine"] = res
    else:
        out["coarse"] = res
    out = {k: v.numpy() for k, v in out.items()}
    out["meta"] = np.array([B, N, H, W, int(fine)], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "coarse absmax", float(np.abs(out["coarse"]).max()))


FULLSIZE_SEED, FULLSIZE_STRIDE = 41, 40
PCTS = np.array([0, 1, 5, 25, 50, 75, 95, 99, 100], dtype=np.float64)


def stats(a):
    """[percentiles..., mean, abs-mean, abs-max] of a tensor/array (float64)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.concatenate((np.percentile(a, PCTS), [a.mean(), np.abs(a).mean(), np.abs(a).max()]))


def fullsize_inputs():
    import hashlib
    from deepi2p_amd import synthetic
    b = synthetic.make_batch(FULLSIZE_SEED, 1)
    names = ("pc", "intensity", "sn", "node_a", "node_b", "img")
    h = hashlib.sha256()
    for k in names:
        h.update(np.ascontiguousarray(b[k]).tobytes())
    return b, names, h.hexdigest()


def make_fullsize():
    b, names, digest = fullsize_inputs()
    N, H, W = b["pc"].shape[2], b["img"].shape[2], b["img"].shape[3]
    t = [torch.from_numpy(b[k]) for k in names]
    out = {"input_sha256": np.array(digest), "meta": np.array([1, N, H, W, FULLSIZE_SEED, FULLSIZE_STRIDE], dtype=np.int32)}
    for fine in (False, True):
        tag = "fine_model" if fine else "coarse_model"
        opt = rn.make_opt(N, H, W, fine, B=1)
        det = rn.load_reference_detector(opt)
        det.load_state_dict(nt.synthetic_state_dict(nt.OptLike(N, H, W, fine)))
        with torch.no_grad():
            enc = det.pc_encoder(*t[:5])
            s16, s32, glob = det.img_encoder(t[5])
            res = det(*t)
        coarse = (res[0] if fine else res).numpy()
        out[tag + "_coarse_sub"] = coarse[:, :, ::FULLSIZE_STRIDE].copy()
        out[tag + "_coarse_labels"] = np.packbits(coarse.argmax(1).astype(np.uint8), axis=1)
        out[tag + "_coarse_stats"] = stats(coarse)
        out[tag + "_coarse_absmax"] = np.float64(np.abs(coarse).max())
        # margin between the two coarse logits: label flips are only meaningful where it exceeds the tolerance
        out[tag + "_coarse_margin"] = np.abs(coarse[:, 0] - coarse[:, 1]).astype(np.float32)
        if fine:
            f = res[1].numpy()
            out[tag + "_fine_sub"] = f[:, :, ::FULLSIZE_STRIDE].copy()
            out[tag + "_fine_labels"] = f.argmax(1).astype(np.uint8)
            out[tag + "_fine_stats"] = stats(f)
            out[tag + "_fine_absmax"] = np.float64(np.abs(f).max())
            srt = np.sort(f, axis=1)
            out[tag + "_fine_margin"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
        if not fine:   # encoder stages are identical for both models
            for name, v in (("first_pn_out", enc[3]), ("second_pn_out", enc[4]), ("node_a_features", enc[5]),
                            ("node_b_features", enc[6]), ("global_feature", enc[7]), ("s16", s16), ("s32", s32), ("img_global", glob)):
                out["stage_" + name + "_stats"] = stats(v.numpy())
            out["stage_global_feature"] = enc[7].numpy()
            out["stage_img_global"] = glob.numpy()
            out["stage_node_b_features_sub"] = enc[6].numpy()[:, ::8, ::4].copy()
        print("fullsize", tag, "coarse absmax", float(np.abs(coarse).max()))
    np.savez_compressed(os.path.join(HERE, "network_fullsize_golden.npz"), **out)
    print("network_fullsize_golden.npz written")


def make_losses():
    """Loss values and autograd gradients from the reference's OWN modules: models/focal_loss.py (imported) + torch's
    CrossEntropyLoss, assembled as models/multimodal_classifier.py:169-191 does (sort-based gather of the inside points)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_focal_loss", os.path.join(REF, "models", "focal_loss.py"))
    fl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fl)
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, (B, L, N, frac) in {"small": (2, 8, 1024, 0.3), "kitti_L80": (1, 80, 768, 0.15), "coarse_only": (2, 0, 777, 0.5)}.items():
        coarse = (torch.randn(B, 2, N, generator=g) * 2).requires_grad_(True)
        clab = (torch.rand(B, N, generator=g) < frac).long()
        crit_c = fl.FocalLoss(alpha=0.5, gamma=2, reduction="mean")
        coarse_loss = crit_c(coarse, clab) * 50
        out[name + "_coarse"], out[name + "_clab"] = coarse.detach().numpy(), clab.numpy().astype(np.int32)
        if L:
            fine = (torch.randn(B, L, N, generator=g) * 3).requires_grad_(True)
            flab = torch.randint(0, L, (B, N), generator=g)
            inside_Bn = clab.reshape(B * N).to(torch.int32)
            insider_num = int(inside_Bn.sum())
            _, idx = torch.sort(inside_Bn, descending=True)
            insider_idx = idx[:insider_num]
            flab_in = torch.gather(flab.view(B * N), 0, insider_idx)
            fs = fine.permute(0, 2, 1).reshape(B * N, L).contiguous()
            fs_in = torch.gather(fs, 0, insider_idx.unsqueeze(1).expand(insider_num, L))
            fine_loss = torch.nn.CrossEntropyLoss()(fs_in, flab_in)
            loss = coarse_loss + fine_loss
            loss.backward()
            out[name + "_fine"], out[name + "_flab"] = fine.detach().numpy(), flab.numpy().astype(np.int32)
            out[name + "_d_fine"] = fine.grad.numpy()
            out[name + "_fine_acc"] = np.float64((fs_in.argmax(1) == flab_in).float().mean())
            out[name + "_fine_loss"] = np.float64(fine_loss.item())
        else:
            loss = coarse_loss
            loss.backward()
        out[name + "_d_coarse"] = coarse.grad.numpy()
        out[name + "_loss"], out[name + "_coarse_loss"] = np.float64(loss.item()), np.float64(coarse_loss.item())
        out[name + "_coarse_acc"] = np.float64((coarse.argmax(1) == clab).float().mean())
    np.savez_compressed(os.path.join(HERE, "loss_golden.npz"), **out)
    print("loss_golden.npz written", {k: float(v) for k, v in out.items() if k.endswith("_loss")})


TRAIN_SAMPLES = 16
TRAIN_WEIGHT_SEED = 3          # deepi2p_amd.synthetic.random_state_dict(opt, 3)


def grad_digest(t):
    """(l2, sum, abs-max) in float64 and TRAIN_SAMPLES entries at evenly spaced flat positions."""
    a = t.detach().double().reshape(-1)
    pos = torch.linspace(0, a.numel() - 1, TRAIN_SAMPLES).long()
    return np.array([float(a.norm()), float(a.sum()), float(a.abs().max())]), a[pos].numpy()


def training_labels(pc, H, W, scale=32):
    """Coarse / fine labels of models/multimodal_classifier.py:135-156 for P = identity, K = [[0.7W,0,W/2],[0,0.7W,H/2],[0,0,1]]
    (inputs of the loss; stored in the fixture)."""
    fx = 0.7 * W
    z = pc[:, 2]
    px = fx * pc[:, 0] / z + W / 2
    py = fx * pc[:, 1] / z + H / 2
    inside = (px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1) & (z > 0.1)
    Wf = int(round(W / scale))
    fine = torch.floor(px / scale).long() + torch.floor(py / scale).long() * Wf
    return inside.long(), torch.where(inside, fine, torch.zeros_like(fine))


def run_training_reference(det, inputs, clab, flab, masks, focal_module, coarse_loss_alpha=50.0):
    """One train-mode forward + backward of the imported reference detector with the given dropout keep-masks."""
    import torch.nn.functional as F
    queue = list(masks)
    real = F.dropout

    def fixed_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return x
        m = queue.pop(0)
        return x * m.to(x.dtype) / (1.0 - p)

    F.dropout = fixed_dropout
    try:
        det.train()
        det.zero_grad()
        coarse, fine = det(*inputs)
        B, L, N = fine.shape
        closs = focal_module.FocalLoss(alpha=0.5, gamma=2, reduction="mean")(coarse, clab) * coarse_loss_alpha
        inside_Bn = clab.reshape(B * N).to(torch.int32)
        insider_num = int(inside_Bn.sum())
        _, idx = torch.sort(inside_Bn, descending=True)
        insider_idx = idx[:insider_num]
        flab_in = torch.gather(flab.view(B * N), 0, insider_idx)
        fs = fine.permute(0, 2, 1).reshape(B * N, L).contiguous()
        fs_in = torch.gather(fs, 0, insider_idx.unsqueeze(1).expand(insider_num, L))
        floss = torch.nn.CrossEntropyLoss()(fs_in, flab_in)
        loss = closs + floss
        loss.backward()
    finally:
        F.dropout = real
    assert not queue, "dropout was called fewer times than masks were provided"
    return coarse, fine, loss, closs, floss


def make_training():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_focal_loss", os.path.join(REF, "models", "focal_loss.py"))
    fl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fl)
    B, N, H, W = 2, 1024, 64, 128
    opt = rn.make_opt(N, H, W, True, B=B)
    det = rn.load_reference_detector(opt)
    det.load_state_dict(nt.random_state_dict(nt.OptLike(N, H, W, True), TRAIN_WEIGHT_SEED))
    inputs = network_inputs(7, B, N, H, W)
    clab, flab = training_labels(inputs[0], H, W)
    g = torch.Generator().manual_seed(11)
    widths = [det.per_point_pn.layers[0].conv.out_channels, det.per_point_pn.layers[1].conv.out_channels]
    masks = [(torch.rand(B, c, N, generator=g) >= 0.5) for c in widths]
    coarse, fine, loss, closs, floss = run_training_reference(det, inputs, clab, flab, masks, fl)
    out = {"meta": np.array([B, N, H, W, 1, TRAIN_WEIGHT_SEED], dtype=np.int32), "coarse_labels": clab.numpy().astype(np.int32),
           "fine_labels": flab.numpy().astype(np.int32), "loss": np.float64(loss.item()), "coarse_loss": np.float64(closs.item()),
           "fine_loss": np.float64(floss.item()), "coarse_sub": coarse.detach().numpy()[:, :, ::8].copy(),
           "fine_sub": fine.detach().numpy()[:, :, ::8].copy()}
    for i, m in enumerate(masks):
        out["mask%d" % i] = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
        out["mask%d_shape" % i] = np.array(m.shape, dtype=np.int32)
    names, dig, samp = [], [], []
    unused = []
    for k, p in det.named_parameters():
        if p.grad is None:          # parameters the forward never touches get no gradient from the reference either
            unused.append(k)
            continue
        d, sm = grad_digest(p.grad)
        names.append(k); dig.append(d); samp.append(sm)
    out["unused_params"] = np.array(unused)
    out["param_names"] = np.array(names)
    out["grad_digest"] = np.stack(dig)
    out["grad_samples"] = np.stack(samp)
    bn, bd = [], []
    for k, b in det.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            bn.append(k); bd.append(grad_digest(b)[0])
    out["buffer_names"] = np.array(bn)
    out["buffer_digest"] = np.stack(bd)
    np.savez_compressed(os.path.join(HERE, "training_golden.npz"), **out)
    print("training_golden.npz written: loss", loss.item(), "inside", int(clab.sum()), "params", len(names))


def _extract_functions(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"np": np, "math": math}
    from scipy.spatial.transform import Rotation
    ns["Rotation"] = Rotation
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    return ns


def make_lsq_driver():
    ns = _extract_functions(os.path.join(REF, "data", "augmentation.py"), {"angles2rotation_matrix"})
    ns2 = _extract_functions(os.path.join(REF, "evaluation", "registration_lsq.py"),
                             {"wrap_in_pi", "get_initial_guess", "get_P_diff", "get_inside_img_mask"})
    ns2["angles2rotation_matrix"] = ns["angles2rotation_matrix"]
    rng = np.random.default_rng(99)
    out = {}
    xs = np.concatenate((rng.uniform(-20, 20, 30), [0.0, math.pi, -math.pi, 3 * math.pi, -3 * math.pi]))
    out["wrap_in"] = xs
    out["wrap_out"] = np.array([ns2["wrap_in_pi"](float(x)) for x in xs])
    angs = rng.uniform(-math.pi, math.pi, (10, 3))
    out["a2r_in"] = angs
    out["a2r_out"] = np.stack([ns["angles2rotation_matrix"](a) for a in angs])
    for i in range(3):
        N = 500
        pc = rng.uniform(-40, 40, (3, N))
        lab = (rng.random(N) < 0.3).astype(np.int64)
        P_init, y, pcf, labf = ns2["get_initial_guess"](pc, lab)
        out["ig%d_pc" % i], out["ig%d_lab" % i] = pc, lab
        out["ig%d_P" % i], out["ig%d_y" % i], out["ig%d_pcf" % i], out["ig%d_labf" % i] = P_init, np.float64(y), pcf, labf
        A = np.eye(4)
        A[:3, :3] = ns["angles2rotation_matrix"](rng.uniform(-0.3, 0.3, 3))
        A[:3, 3] = rng.uniform(-3, 3, 3)
        Bm = np.eye(4)
        Bm[:3, :3] = ns["angles2rotation_matrix"](rng.uniform(-0.3, 0.3, 3))
        Bm[:3, 3] = rng.uniform(-3, 3, 3)
        t, r = ns2["get_P_diff"](A, Bm)
        out["pd%d_A" % i], out["pd%d_B" % i], out["pd%d_t" % i], out["pd%d_r" % i] = A, Bm, np.float64(t), np.float64(r)
        K = np.array([[350.0, 0, 256], [0, 350.0, 80], [0, 0, 1]])
        out["im%d_mask" % i] = ns2["get_inside_img_mask"](pc, A, K, 160, 512)
        out["im%d_K" % i] = K
    np.savez_compressed(os.path.join(HERE, "lsq_driver_golden.npz"), **out)
    print("lsq_driver_golden.npz written")


def make_prep():
    """FarthestSampler.sample of data/kitti_helper.py, class extracted with ast (the module imports open3d)."""
    src = open(os.path.join(REF, "data", "kitti_helper.py")).read()
    tree = ast.parse(src)
    ns = {"np": np}
    np.int = int   # the reference predates numpy 1.24 (np.int removed); alias for exec only
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == "FarthestSampler":
            exec(compile(ast.Module(body=[node], type_ignores=[]), "kitti_helper.py", "exec"), ns)
    fs = ns["FarthestSampler"]()
    rng = np.random.default_rng(5)
    out = {}
    for i, (M, k) in enumerate([(1024, 128), (300, 17), (64, 64)]):
        pts = (rng.standard_normal((3, M)) * 20).astype(np.float32)
        if i == 1:
            pts[:, 100:110] = pts[:, 50:60]          # duplicated points -> exact distance ties
        np.random.seed(100 + i)
        init = np.random.randint(len(pts))           # what the reference would draw
        np.random.seed(100 + i)
        far, idx = fs.sample(pts, k)
        out["fps%d_pts" % i], out["fps%d_k" % i], out["fps%d_init" % i] = pts, np.int32(k), np.int32(init)
        out["fps%d_nodes" % i], out["fps%d_idx" % i] = far, idx.astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "prep_golden.npz"), **out)
    print("prep_golden.npz written")


# ----------------------------------------------------------------------------------------------------------------------
# Front ends of the path that are METHODS / functions / script bodies of the reference: extracted with ``ast`` and run here against
# stub collaborators (canned network scores, a recording cv2, a recording FrustumRegistration), exactly like make_lsq_driver does for
# the LSQ helpers.  Only inputs and outputs are stored.
# ----------------------------------------------------------------------------------------------------------------------
class _NpProxy:
    """numpy with the aliases the reference's era still had (np.int / np.float) and a recording np.save"""

    def __init__(self, saved=None):
        self._saved = saved
        self.int = int
        self.float = float

    def save(self, path, arr):
        self._saved.append((os.path.basename(path), np.array(arr)))

    def __getattr__(self, name):
        return getattr(np, name)


def _method_as_function(path, cls, name, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    exec(compile(ast.Module(body=[sub], type_ignores=[]), path, "exec"), ns)
                    return ns[name]
    raise KeyError(name)


def _label_inputs(seed, B, N, H, W, P_rows):
    """pc / P / K of a seeded synthetic batch, with a few points planted close to (not on) the image borders and the z = 0.1 plane"""
    from deepi2p_amd import synthetic
    b = synthetic.make_batch(seed, B, N=N, H=H, W=W, with_image=False)
    pc = b["pc"].astype(np.float32).copy()
    P = b["P_gt"].astype(np.float32)[:, :P_rows, :].copy()
    K = b["K"].astype(np.float32)
    rng = np.random.default_rng(seed + 1)
    for bi in range(B):
        Pi = b["P_gt"][bi]
        Rm, t = Pi[:3, :3], Pi[:3, 3]
        for j in range(24):
            z = rng.uniform(1.0, 40.0)
            u = [0.0, W - 1.0, rng.uniform(0, W - 1)][j % 3] + rng.choice([-1, 1]) * rng.uniform(2e-2, 0.4)
            v = [rng.uniform(0, H - 1), 0.0, H - 1.0][(j // 3) % 3] + rng.choice([-1, 1]) * rng.uniform(2e-2, 0.4)
            if j >= 18:
                z = 0.1 + rng.choice([-1, 1]) * rng.uniform(1e-3, 5e-2)
            cam = np.array([(u - K[bi, 0, 2]) * z / K[bi, 0, 0], (v - K[bi, 1, 2]) * z / K[bi, 1, 1], z])
            pc[bi, :, j] = (Rm.T @ (cam - t)).astype(np.float32)
    return pc, P, K


def make_forward_pass():
    """MMClassifer.foraward_pass (models/multimodal_classifier.py:119-212) run on a stub ``self``: label projection, fine labels,
    loss assembly (reference FocalLoss + CrossEntropyLoss), accuracies."""
    from types import SimpleNamespace
    import torch.nn as nn
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_focal_loss", os.path.join(REF, "models", "focal_loss.py"))
    focal = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(focal)
    ns = {"torch": torch}
    fwd = _method_as_function(os.path.join(REF, "models", "multimodal_classifier.py"), "MMClassifer", "foraward_pass", ns)
    out = {}
    for ci, (seed, B, N, H, W) in enumerate([(71, 2, 2048, 64, 128), (72, 1, 4096, 160, 512)]):
        scale = 32
        L = int(round(H / scale)) * int(round(W / scale))
        pc, P, K = _label_inputs(seed, B, N, H, W, 3)
        g = torch.Generator().manual_seed(seed)
        coarse_scores = torch.randn((B, 2, N), generator=g)
        fine_scores = torch.randn((B, L, N), generator=g)
        self_ = SimpleNamespace(
            pc=torch.from_numpy(pc), intensity=None, sn=None, node_a=None, node_b=None, img=torch.zeros((B, 3, H, W)),
            P=torch.from_numpy(P), K=torch.from_numpy(K),
            opt=SimpleNamespace(node_a_num=128, node_b_num=128, img_fine_resolution_scale=scale, coarse_loss_alpha=50, is_debug=False),
            forward=lambda *a: (coarse_scores, fine_scores),
            coarse_ce_criteria=focal.FocalLoss(alpha=0.5, gamma=2, reduction="mean"), fine_ce_criteria=nn.CrossEntropyLoss())
        loss_dict, vis, acc = fwd(self_)
        k = "fp%d_" % ci
        out[k + "pc"], out[k + "P"], out[k + "K"] = pc, P, K
        out[k + "HW"] = np.array([H, W, scale], dtype=np.int64)
        out[k + "coarse_scores"], out[k + "fine_scores"] = coarse_scores.numpy(), fine_scores.numpy()
        out[k + "coarse_labels"] = vis["coarse_labels"].numpy()
        out[k + "fine_labels"] = vis["fine_labels"].numpy()
        out[k + "KP_pc_pxpy"] = vis["KP_pc_pxpy"].numpy()
        out[k + "P_pc"] = vis["pc"].numpy()
        out[k + "coarse_predictions"] = vis["coarse_predictions"].numpy()
        out[k + "fine_predictions"] = vis["fine_predictions"].numpy()
        out[k + "losses"] = np.array([float(loss_dict[n]) for n in ("loss", "coarse", "fine")])
        out[k + "accuracy"] = np.array([float(acc["coarse_accuracy"]), float(acc["fine_accuracy"])])
    np.savez_compressed(os.path.join(HERE, "forward_pass_golden.npz"), **out)
    print("forward_pass_golden.npz written")


def make_eval_script():
    """The per-batch body of evaluation/visualize_and_save_data.py (the ``for i, data in enumerate(testloader)`` loop, :81-187): GT labels,
    per-frame accuracies and the saved pc_label / K / P records, with a stub model (canned predictions), stub visualisation and a
    recording np.save."""
    from types import SimpleNamespace
    path = os.path.join(REF, "evaluation", "visualize_and_save_data.py")
    tree = ast.parse(open(path).read())
    loop = None
    for node in ast.walk(tree):
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Call) and getattr(node.iter.func, "id", "") == "enumerate" \
                and getattr(node.iter.args[0], "id", "") == "testloader":
            loop = node
    assert loop is not None
    code = compile(ast.Module(body=[loop], type_ignores=[]), path, "exec")
    out = {}
    for ci, (seed, B, N, H, W, fine, P_rows) in enumerate([(81, 2, 2048, 64, 128, True, 3), (82, 2, 1024, 160, 512, False, 4)]):
        scale = 32
        L = int(round(H / scale)) * int(round(W / scale))
        pc, P, K = _label_inputs(seed, B, N, H, W, P_rows)
        rng = np.random.default_rng(seed)
        cpred = torch.from_numpy(rng.integers(0, 2, (B, N)).astype(np.int64))
        fpred = torch.from_numpy(rng.integers(0, L, (B, N)).astype(np.int64))
        img = torch.from_numpy(rng.uniform(0, 255, (B, 3, H, W)).astype(np.float32))
        saved, sums = [], []

        class Model:
            def set_input(self, *a):
                pass

            def inference_pass(self):
                return (cpred, fpred) if fine else cpred

        data = (torch.from_numpy(pc), torch.zeros((B, 1, N)), torch.zeros((B, 3, N)), torch.zeros((B, 3, 128)), torch.zeros((B, 3, 128)),
                torch.from_numpy(P), img, torch.from_numpy(K), torch.zeros((B, 3)))
        vis = SimpleNamespace(get_classification_visualization=lambda *a, **k: np.zeros((4, 4, 3), np.uint8),
                              get_classification_visualization_coarse=lambda *a, **k: np.zeros((4, 4, 3), np.uint8))
        lines = []
        ns = {"np": _NpProxy(saved), "torch": torch, "os": os, "testloader": [data], "model": Model(), "vis_tools": vis, "cv2": None,
              "opt": SimpleNamespace(img_fine_resolution_scale=scale, is_fine_resolution=fine), "counter": 0, "coarse_accuracy_sum": 0,
              "fine_accuracy_sum": 0, "is_save_visualization": False, "is_save_data": True, "is_plot": False, "iter_max": 1e9,
              "data_output_path": "data", "visualization_output_path": "vis", "print": lambda *a: lines.append(a[0])}
        with np.errstate(all="ignore"):
            exec(code, ns)
        k = "ev%d_" % ci
        out[k + "pc"], out[k + "P"], out[k + "K"] = pc, P, K
        out[k + "HW"] = np.array([H, W, scale, int(fine)], dtype=np.int64)
        out[k + "coarse_pred"], out[k + "fine_pred"] = cpred.numpy(), fpred.numpy()
        recs = [a for n, a in saved if n.endswith("pc_label.npy")]
        assert len(recs) == B and recs[0].shape == (7, N)
        out[k + "pc_label"] = np.stack(recs)
        out[k + "saved_K"] = np.stack([a for n, a in saved if n.endswith("_K.npy")])
        out[k + "saved_P"] = np.stack([a for n, a in saved if n.endswith("_P.npy")])
        out[k + "acc_sums"] = np.array([ns["coarse_accuracy_sum"], ns["fine_accuracy_sum"], ns["counter"]], dtype=np.float64)
        out[k + "acc_lines"] = np.array(lines)
    np.savez_compressed(os.path.join(HERE, "eval_script_golden.npz"), **out)
    print("eval_script_golden.npz written")


def make_pnp_frontend():
    """solve_PnP + camera_matrix_scaling (evaluation/registration_pnp.py:58-61,95-148) with a cv2 that records what reaches
    solvePnPRansac and returns canned answers: correspondence packing, the >= 4 rule, the |t| < 14.14 acceptance, the outlier ratio."""
    from scipy.spatial.transform import Rotation
    path = os.path.join(REF, "evaluation", "registration_pnp.py")
    out = {}
    cases = [dict(n_in=300, ok=True, t=[1.0, -0.5, 3.0], n_inl=120), dict(n_in=300, ok=True, t=[10.0, 1.0, 10.0], n_inl=80),
             dict(n_in=300, ok=False, t=[0.1, 0.1, 0.1], n_inl=10), dict(n_in=3, ok=True, t=[0, 0, 1.0], n_inl=3),
             dict(n_in=4, ok=True, t=[0.2, 0, 1.0], n_inl=4), dict(n_in=50, ok=True, t=[0, 0, 1.0], n_inl=20, throw=True)]
    for ci, c in enumerate(cases):
        rng = np.random.default_rng(200 + ci)
        N, Hf, Wf, scale = 1000, 5, 16, 1.0 / 32          # the caller passes the FINE grid size and the 1/32 factor (registration_pnp.py:207)
        pc = rng.uniform(-30, 30, (3, N)).astype(np.float32)
        coarse = np.zeros(N, np.int64)
        coarse[rng.choice(N, c["n_in"], replace=False)] = 1
        fine = rng.integers(0, Hf * Wf, N).astype(np.int64)
        K = np.array([[358.4, 0, 256.0], [0, 358.4, 80.0], [0, 0, 1.0]])
        rec = {}

        class CV2:
            SOLVEPNP_EPNP = 1

            @staticmethod
            def solvePnPRansac(points, pixels, Kf, useExtrinsicGuess, iterationsCount, reprojectionError, flags, distCoeffs):
                rec.update(points=np.array(points), pixels=np.array(pixels), K=np.array(Kf), iters=iterationsCount, err=reprojectionError,
                           flags=flags, guess=useExtrinsicGuess)
                if c.get("throw"):
                    raise RuntimeError("cv2 error")
                return c["ok"], np.array([[0.1], [-0.2], [0.05]]), np.array(c["t"], dtype=np.float64).reshape(3, 1), np.arange(c["n_inl"]).reshape(-1, 1)

            @staticmethod
            def Rodrigues(rvec):
                return Rotation.from_rotvec(np.asarray(rvec).reshape(3)).as_matrix(), None

        ns = {"np": _NpProxy(), "cv2": CV2}
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in ("camera_matrix_scaling", "solve_PnP"):
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        Pm, ratio = ns["solve_PnP"](pc, coarse, fine, K.copy(), Hf * 32, Wf * 32, scale, 500, CV2.SOLVEPNP_EPNP)
        k = "pnp%d_" % ci
        out[k + "pc"], out[k + "coarse"], out[k + "fine"], out[k + "K"] = pc, coarse, fine, K
        out[k + "HWs"] = np.array([Hf * 32, Wf * 32, scale])
        out[k + "called"] = np.int64(1 if rec else 0)
        if rec:
            out[k + "arg_points"], out[k + "arg_pixels"], out[k + "arg_K"] = rec["points"], rec["pixels"], rec["K"]
            out[k + "arg_scalars"] = np.array([rec["iters"], rec["err"], rec["flags"], float(rec["guess"])])
        out[k + "ret"] = np.array([float(c["ok"]), c["n_inl"], float(bool(c.get("throw")))] + list(c["t"]))
        out[k + "P"], out[k + "ratio"] = Pm, np.float64(ratio)
    np.savez_compressed(os.path.join(HERE, "pnp_frontend_golden.npz"), **out)
    print("pnp_frontend_golden.npz written")


def make_lsq_restart_driver():
    """solve_P_random_perturb + solver_wrapper (evaluation/registration_lsq.py:127-186) with a recording FrustumRegistration, an in-process
    ``multiprocessing`` and a seeded ``random``: the restart list (ry = y0 + gauss, t = (0, 0, U)), the hard-coded max_iter = 500, the
    strict-< minimum, and the wave arithmetic (iteration_num % thread_num == 0 drops the last wave)."""
    import random as pyrandom
    path = os.path.join(REF, "evaluation", "registration_lsq.py")
    out = {}
    for ci, (iteration_num, thread_num) in enumerate([(60, 8), (64, 8), (5, 8), (16, 4)]):
        calls = []
        costs_rng = np.random.default_rng(300 + ci)
        canned = costs_rng.uniform(5.0, 50.0, 200)
        canned[7] = canned[3] = canned.min() - 1.0          # an exact tie: the FIRST one must win (strict <)

        class FR:
            @staticmethod
            def solvePGivenK(pc, lab, K, R_init, t_init, H, W, lb, ub, max_iter, is_debug, is_2d):
                i = len(calls)
                calls.append((float(R_init), np.array(t_init, dtype=np.float64), H, W, list(lb), list(ub), max_iter, is_debug, is_2d))
                P = np.eye(4)
                P[0, 3] = i
                return P, float(canned[i]), np.full(3, float(i))

        class Proc:
            def __init__(self, target, args):
                self.t, self.a = target, args

            def start(self):
                self.t(*self.a)

            def join(self):
                pass

        class MP:
            Process = Proc

            @staticmethod
            def Manager():
                class M:
                    @staticmethod
                    def dict():
                        return {}
                return M()

        ns = {"np": np, "math": math, "random": pyrandom.Random(1234 + ci), "multiprocessing": MP, "FrustumRegistration": FR}
        tree = ast.parse(open(path).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in ("solver_wrapper", "solve_P_random_perturb"):
                exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        pc, lab, K = np.zeros((3, 8)), np.ones(8, np.int32), np.eye(3)
        P, cost, res = ns["solve_P_random_perturb"](pc, lab, K, 160, 512, 10.0, 0.37, 10 * math.pi / 180, [-5, -0.1, -10], [5, 0.1, 10],
                                                    iteration_num, True, thread_num)
        k = "rp%d_" % ci
        out[k + "args"] = np.array([iteration_num, thread_num, 1234 + ci], dtype=np.int64)
        out[k + "n_calls"] = np.int64(len(calls))
        out[k + "ry"] = np.array([c[0] for c in calls])
        out[k + "t"] = np.array([c[1] for c in calls]).reshape(len(calls), 3)
        out[k + "max_iter"] = np.array([c[6] for c in calls], dtype=np.int64)
        out[k + "flags"] = np.array([[int(c[7]), int(c[8])] for c in calls], dtype=np.int64).reshape(len(calls), 2)
        out[k + "bounds"] = np.array(calls[0][4] + calls[0][5]) if calls else np.zeros(6)
        out[k + "canned_cost"] = canned[:max(len(calls), 1)]
        out[k + "best_call"] = np.int64(-1 if P is None else int(P[0, 3]))
        out[k + "cost"] = np.float64(cost)
    np.savez_compressed(os.path.join(HERE, "lsq_restart_golden.npz"), **out)
    print("lsq_restart_golden.npz written")


def make_front_ends():
    make_forward_pass()
    make_eval_script()
    make_pnp_frontend()
    make_lsq_restart_driver()


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else None
    assert rn.available(), "needs /root/reference and oracle/_ref (make -C oracle ref)"
    torch.set_num_threads(8)
    if only == "fullsize":
        make_fullsize()
        sys.exit(0)
    if only == "losses":
        make_losses()
        sys.exit(0)
    if only == "training":
        make_training()
        sys.exit(0)
    if only == "frontends":
        make_front_ends()
        sys.exit(0)
    make_index_max()
    make_network(True, "network_golden.npz")
    make_network(False, "network_coarse_golden.npz")
    make_lsq_driver()
    make_prep()
    make_fullsize()
    make_losses()
    make_training()
    make_front_ends()
