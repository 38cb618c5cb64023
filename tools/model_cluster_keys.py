#!/usr/bin/env python3
"""CPU model of the solver's cluster test (csrc/solver.hip: prepare_kernel + cluster_status) for different sort keys: how many
64-point clusters of a config-2 frame need per-point work (status 1 = classify, 3 = zero-guard only) at the ground-truth pose and at
perturbed poses.  Exact interval arithmetic (the fp32 margins of the kernel are negligible at this scale).
    python tools/model_cluster_keys.py"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepi2p_amd import synthetic  # noqa: E402

CL = 64


def spread(v):
    v = v & 0x3ff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v


def morton(qx, qz):
    return spread(qx) | (spread(qz) << 1)


def hilbert(qx, qz, bits=10):
    x, y = qx.copy(), qz.copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        # rotate
        flip = (ry == 0)
        swap_flip = flip & (rx == 1)
        x = np.where(swap_flip, s - 1 - x, x)
        y = np.where(swap_flip, s - 1 - y, y)
        x, y = np.where(flip, y, x), np.where(flip, x, y)
        s >>= 1
    return d


def kd_order(x, z, leaf=CL):
    """balanced k-d split order (alternating median splits down to `leaf` points)"""
    idx = np.arange(x.size)
    out = []

    def rec(ids, axis):
        if ids.size <= leaf:
            out.append(ids)
            return
        v = (x if axis == 0 else z)[ids]
        o = ids[np.argsort(v, kind="stable")]
        # split at a multiple of leaf
        h = (o.size // 2 + leaf - 1) // leaf * leaf
        rec(o[:h], 1 - axis)
        rec(o[h:], 1 - axis)
    ext = (x.max() - x.min(), z.max() - z.min())
    rec(idx, 0 if ext[0] >= ext[1] else 1)
    return np.concatenate(out)


def statuses(pts, lab, order, P, K, H, W):
    """-> counts of cluster statuses {0 skip, 1 classify, 2 all active, 3 guard only} for the records in `order`"""
    R, t = P[:3, :3], P[:3, 3]
    fx, fy, cx, cy, W1, H1 = K[0, 0], K[1, 1], K[0, 2], K[1, 2], W - 1.0, H - 1.0
    normals = np.array([[fx, 0, cx], [-fx, 0, W1 - cx], [0, fy, cy], [0, -fy, H1 - cy], [0, 0, 1.0]])   # f_i(p) = n_i . p
    res = {}
    for L in (1, 0):
        ids = order[lab[order] == L]
        cnt = np.zeros(4, int)
        for s in range(0, ids.size, CL):
            p = pts[:, ids[s:s + CL]]
            lo, hi = p.min(1), p.max(1)
            c, h = 0.5 * (lo + hi), 0.5 * (hi - lo)
            pc = R @ c + t
            f = normals @ pc
            sup = np.abs(normals @ R) @ h
            decided = np.abs(f) > sup
            inside = np.all(f - sup > 0)
            if decided.all():
                st = (0 if inside else 2) if L == 1 else (2 if inside else 0)
            elif np.any(f + sup < 0):
                st = 2 if L == 1 else 3
            else:
                st = 1
            cnt[st] += 1
        res[L] = cnt
    return res


def main():
    rng = np.random.default_rng(0)
    N, H, W = 20480, 160, 512
    tot = {}
    for fi in range(6):
        f = synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.05, with_image=False)
        pts = f["pc"].astype(np.float64)
        lab = f["labels"]
        x, z = pts[0], pts[2]
        ext = max(x.max() - x.min(), z.max() - z.min())
        qx = np.clip(((x - x.min()) * 1023.0 / ext).astype(np.int64), 0, 1023)
        qz = np.clip(((z - z.min()) * 1023.0 / ext).astype(np.int64), 0, 1023)
        orders = {"morton": np.argsort(morton(qx, qz), kind="stable"), "hilbert": np.argsort(hilbert(qx, qz), kind="stable"),
                  "kd-tree": kd_order(x, z)}
        # layered: ground-plane groups of G points (Hilbert), sorted by height inside the group
        for G in (128, 256):
            o = orders["hilbert"]
            parts = []
            for L in (1, 0):
                ids = o[lab[o] == L]
                for s in range(0, ids.size, G):
                    g = ids[s:s + G]
                    parts.append(g[np.argsort(pts[1][g], kind="stable")])
            orders["hilbert+height/%d" % G] = np.concatenate(parts)
        poses = [("gt", f["P_gt"])]
        for k in range(3):
            P = f["P_gt"].copy()
            P[:3, :3] = synthetic.ry_matrix(f["yaw_gt"] + rng.normal(0, 0.1))
            P[:3, 3] += rng.normal(0, 1.0, 3) * np.array([1, 0.05, 1])
            poses.append(("perturbed", P))
        for name, o in orders.items():
            for pn, P in poses:
                r = statuses(pts, lab, o, P, f["K"], H, W)
                a = tot.setdefault(name, np.zeros((2, 4)))
                a[0] += r[1]; a[1] += r[0]
    print("clusters per frame and pose (mean): label 1 {skip, classify, all-active} | label 0 {skip, classify, all-active, guard-only} | per-point work")
    for name, a in tot.items():
        a = a / (6 * 4)
        print("%-20s  L1 %5.1f %5.1f %5.1f | L0 %6.1f %5.1f %5.1f %5.1f | %6.1f of %.0f" % (name, a[0, 0], a[0, 1], a[0, 2], a[1, 0], a[1, 1], a[1, 2], a[1, 3],
                                                                                  a[0, 1] + a[1, 1] + a[1, 3], a.sum()))


main()
