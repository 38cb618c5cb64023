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
    assert rn.available(), "needs /root/reference and oracle/_ref (make -C oracle ref)"
    torch.set_num_threads(8)
    make_index_max()
    make_network(True, "network_golden.npz")
    make_network(False, "network_coarse_golden.npz")
    make_lsq_driver()
    make_prep()
