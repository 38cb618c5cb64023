#!/usr/bin/env python3
"""CPU model of TEMPORAL COHERENCE in the pose solver's cluster walk (csrc/solver.hip).

The oracle's trace of trial points (oracle_solve_trace: the points every cost+gradient evaluation of one solve is made at, i.e. the
kernel's sweeps) is replayed over the kernel's cluster layout (Hilbert-sorted 64-point clusters per label).  Per sweep and cluster:
the box status (0 skip / 1 classify per point / 2 all active / 3 zero-guard only) and, for the clusters that need per-point work
(1, 3), the smallest normalised distance of any of its points to any frustum plane ("slack").  A cache entry {slack, recorded iterate}
stays valid while  |d theta| * rho_cluster + |d t|_inf < slack : every point keeps its sign pattern, so the recorded active mask
(or "nothing to guard") can be re-used without touching the points.  Prints the hit rates such a cache would have.
    python tools/model_temporal.py [frames] [restarts]"""
import ctypes
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepi2p_amd import synthetic  # noqa: E402
from oracle import frustum_lm as flm  # noqa: E402

CL = 64


def hilbert(qx, qz, bits=10):
    x, y = qx.copy(), qz.copy()
    d = np.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0)
        swap_flip = flip & (rx == 1)
        x = np.where(swap_flip, s - 1 - x, x)
        y = np.where(swap_flip, s - 1 - y, y)
        x, y = np.where(flip, y, x), np.where(flip, x, y)
        s >>= 1
    return d


def trace_solve(pts, lab, K, y, T, H, W, lb, ub):
    lib = flm._lib()
    lib.oracle_solve_trace.restype = ctypes.c_int
    cap = 2048
    tr = np.zeros((cap, 4))
    cost = ctypes.c_double()
    n = lib.oracle_solve_trace(flm._dp(pts), flm._ip(lab), pts.shape[1], flm._dp(K), ctypes.c_double(y), flm._dp(np.ascontiguousarray(T)),
                               ctypes.c_double(H), ctypes.c_double(W), flm._dp(lb), flm._dp(ub), 500, 1, ctypes.byref(cost), flm._dp(tr), cap)
    return tr[:min(n, cap)].copy()


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ring = [8, 16, 32, 10 ** 9]
    rng = np.random.default_rng(0)
    N, H, W = 20480, 160, 512
    lb = np.array([-5.0, -0.1, -10.0]); ub = np.array([5.0, 0.1, 10.0])
    tot = {}
    nsweeps = []
    for fi in range(nframes):
        f = synthetic.make_frame(rng, N=N, H=H, W=W, flip=0.05, with_image=False)
        _, yaw0, pts, lab = flm.get_initial_guess(f["pc"].astype(np.float64), f["labels"])
        pts = np.ascontiguousarray(pts); lab = np.ascontiguousarray(lab.astype(np.int32))
        K = f["K"]
        fx, fy, cx, cy, W1, H1 = K[0, 0], K[1, 1], K[0, 2], K[1, 2], W - 1.0, H - 1.0
        normals = np.array([[fx, 0, cx], [-fx, 0, W1 - cx], [0, fy, cy], [0, -fy, H1 - cy], [0, 0, 1.0]])
        n1 = np.abs(normals).sum(1)
        x, z = pts[0], pts[2]
        ext = max(x.max() - x.min(), z.max() - z.min())
        qx = np.clip(((x - x.min()) * 1023.0 / ext).astype(np.int64), 0, 1023)
        qz = np.clip(((z - z.min()) * 1023.0 / ext).astype(np.int64), 0, 1023)
        order = np.argsort(hilbert(qx, qz), kind="stable")
        clusters = []           # (label, ids)
        for L in (1, 0):
            ids = order[lab[order] == L]
            for s in range(0, ids.size, CL):
                clusters.append((L, ids[s:s + CL]))
        nc = len(clusters)
        lo = np.array([pts[:, ids].min(1) for _, ids in clusters]); hi = np.array([pts[:, ids].max(1) for _, ids in clusters])
        cc, hh = 0.5 * (lo + hi), 0.5 * (hi - lo)
        rho = np.array([np.sqrt(pts[0, ids] ** 2 + pts[2, ids] ** 2).max() for _, ids in clusters])
        labs = np.array([L for L, _ in clusters])
        ys, Ts = flm.draw_restarts(rng, R, yaw0, 10.0 * math.pi / 180.0, 10.0)
        for r in range(R):
            tr = trace_solve(pts, lab, K, ys[r], Ts[r], H, W, lb, ub)
            nsweeps.append(len(tr))
            # cache per ring size: recorded sweep index and slack per cluster
            rec_it = {g: np.full(nc, -1) for g in ring}
            rec_sl = {g: np.zeros(nc) for g in ring}
            rec2_it = np.full(nc, -1); rec2_sl = np.zeros((nc, 2))         # two-group variant (ring 8): slack towards {L, R, Z} and {T, B}
            for s, xi in enumerate(tr):
                Rm = synthetic.ry_matrix(xi[0]); t = xi[1:4]
                pc = (Rm @ cc.T).T + t
                fcl = pc @ normals.T
                sup = hh @ np.abs(normals @ Rm).T
                decided = np.abs(fcl) > sup
                inside = np.all(fcl - sup > 0, axis=1)
                alld = decided.all(1)
                anyneg = np.any(fcl + sup < 0, axis=1)
                st = np.where(alld, np.where(labs == 1, np.where(inside, 0, 2), np.where(inside, 2, 0)),
                              np.where(anyneg, np.where(labs == 1, 2, 3), 1))
                work = np.nonzero((st == 1) | (st == 3))[0]
                for kind in (1, 3):
                    tot.setdefault(("n", kind), 0)
                    tot[("n", kind)] += int((st == kind).sum())
                tot["sweeps"] = tot.get("sweeps", 0) + 1
                tot["nc"] = tot.get("nc", 0) + nc
                # slack of the clusters that need per-point work
                sl = np.zeros(nc)
                sl2 = np.zeros((nc, 2))
                for c in work:
                    p = (Rm @ pts[:, clusters[c][1]]).T + t
                    fp = np.abs(p @ normals.T) / n1
                    if labs[c] == 1:
                        sg = (p @ normals.T) / n1
                        neg = sg < 0
                        out = neg.any(1)
                        # outside: stays active while its most negative plane stays negative; inside: all five must stay positive
                        slp = np.where(out, np.where(neg, -sg, 0).max(1), fp.min(1))
                    else:
                        slp = fp.min(1)
                    sl[c] = slp.min()
                    # two groups: planes L, R, Z do not see t_y, planes T, B see the rotation only through their (small) p2 coefficient
                    gA, gB = [0, 1, 4], [2, 3]
                    if labs[c] == 1:
                        mnA, mnB = sg[:, gA].min(1), sg[:, gB].min(1)
                        useA = mnA <= mnB
                        sA = np.where(out, np.where(useA, -mnA, np.inf), mnA)
                        sB = np.where(out, np.where(useA, np.inf, -mnB), mnB)
                    else:
                        sA, sB = fp[:, gA].min(1), fp[:, gB].min(1)
                    sl2[c] = (sA.min(), sB.min())
                cT = max(normals[2, 1] / n1[2], normals[3, 1] / n1[3] * -1 if normals[3, 1] < 0 else normals[3, 1] / n1[3])
                bT = max(abs(normals[2, 2]) / n1[2], abs(normals[3, 2]) / n1[3])
                aT = max(abs(normals[2, 1]) / n1[2], abs(normals[3, 1]) / n1[3])
                for c in work:
                    hit = False
                    if rec2_it[c] >= 0 and s - rec2_it[c] < 8:
                        xr = tr[rec2_it[c]]
                        rot = abs(xi[0] - xr[0]) * rho[c]
                        muA = rot + max(abs(xi[1] - xr[1]), abs(xi[3] - xr[3]))
                        muB = aT * abs(xi[2] - xr[2]) + bT * (rot + abs(xi[3] - xr[3]))
                        hit = muA * 1.0001 < rec2_sl[c, 0] and muB * 1.0001 < rec2_sl[c, 1]
                    key = ("hit" if hit else "miss", int(st[c]), "2grp")
                    tot[key] = tot.get(key, 0) + 1
                    if not hit:
                        rec2_it[c] = s; rec2_sl[c] = sl2[c]
                for g in ring:
                    ri, rs = rec_it[g], rec_sl[g]
                    for c in work:
                        hit = False
                        if ri[c] >= 0 and s - ri[c] < g:
                            xr = tr[ri[c]]
                            mu = abs(xi[0] - xr[0]) * rho[c] + np.abs(xi[1:4] - xr[1:4]).max()
                            hit = mu * 1.0001 < rs[c]
                        key = ("hit" if hit else "miss", int(st[c]), g)
                        tot[key] = tot.get(key, 0) + 1
                        if not hit:
                            ri[c] = s; rs[c] = sl[c]
    sw = tot["sweeps"]
    print("sweeps per hypothesis: mean %.1f median %.0f max %d ; clusters per frame %.0f" % (np.mean(nsweeps), np.median(nsweeps), max(nsweeps), tot["nc"] / sw))
    print("per sweep: classify %.1f guard-only %.1f clusters" % (tot[("n", 1)] / sw, tot[("n", 3)] / sw))
    for g in ring + ["2grp"]:
        for kind, nm in ((1, "classify"), (3, "guard")):
            h, m = tot.get(("hit", kind, g), 0), tot.get(("miss", kind, g), 0)
            print("ring %-10s %-9s hit rate %.3f  (%d / %d)" % (g if (g == "2grp" or g < 10 ** 9) else "unbounded", nm, h / max(h + m, 1), h, h + m))


main()
