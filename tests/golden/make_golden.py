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
    make_index_max()
    make_network(True, "network_golden.npz")
    make_network(False, "network_coarse_golden.npz")
    make_lsq_driver()
    make_prep()
    make_fullsize()
    make_losses()
    make_training()
