"""Oracle: numpy restatement of the PnP-RANSAC algorithm of deepi2p_amd/csrc/pnp.hip.  TEST INFRASTRUCTURE ONLY.

Replaces cv2.solvePnPRansac as called by evaluation/registration_pnp.py:95-148.  PARITY UNPINNED: OpenCV is neither in
the reference tree nor in this image and its RANSAC sampling is internal; the reference holds no vector for this path.
The FRONT END around the OpenCV call IS pinned (tests/golden/pnp_frontend_golden.npz: the reference's solve_PnP extracted with ``ast`` and
run against a recording cv2): `correspondences` (what reaches solvePnPRansac), camera_matrix_scaling, the >= 4 rule, `accept` (|t| < 14.14,
outlier ratio).  The estimator itself is pinned by pose recovery on synthetic exact correspondences (tests/test_pnp.py) and uses independent
numerics for the two linear-algebra steps (SVD null vector, SVD polar factor) so it cross-checks the kernel's Gaussian
elimination and Newton polar iteration.
"""
import numpy as np


def correspondences(pc, coarse, fine, W_fine, pixels=None):
    m = np.asarray(coarse) == 1
    X = np.asarray(pc, np.float32)[:, m].astype(np.float64)
    if pixels is not None:
        uv = np.asarray(pixels, np.float32)[:, m].astype(np.float64)
    else:
        f = np.asarray(fine)[m]
        py = np.floor(f.astype(np.float32) / np.float32(W_fine)).astype(np.int64)
        uv = np.stack((f - py * W_fine, py)).astype(np.float64)
    return X, uv


def accept(R, t, n_inliers, n_corr, success=True):
    """evaluation/registration_pnp.py:133-141: the pose is kept iff the estimator succeeded and |t| < 14.14; "cost" = outlier ratio.
    PINNED by tests/golden/pnp_frontend_golden.npz (the reference's solve_PnP run against a recording cv2)."""
    P = np.identity(4)
    if success and np.linalg.norm(t) < 14.14:
        P[:3, :3], P[:3, 3] = R, np.asarray(t).reshape(3)
        return P, 1.0 - n_inliers / n_corr
    return P, 1.0


def dlt6(X, uv, K):
    """X 3x6, uv 2x6 -> (R, t) or None."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    xh, yh = (uv[0] - cx) / fx, (uv[1] - cy) / fy
    cen = X.mean(axis=1, keepdims=True)
    Xc = X - cen
    sc = np.mean(np.linalg.norm(Xc, axis=0))
    if not sc > 1e-9:
        return None
    Xn = Xc / sc
    rows = []
    for j in range(6):
        x, y, z = Xn[:, j]
        rows.append([x, y, z, 1, 0, 0, 0, 0, -xh[j] * x, -xh[j] * y, -xh[j] * z, -xh[j]])
        rows.append([0, 0, 0, 0, x, y, z, 1, -yh[j] * x, -yh[j] * y, -yh[j] * z, -yh[j]])
    A = np.array(rows[:11])
    _, s, Vt = np.linalg.svd(A)
    if s[10] < 1e-10 * max(s[0], 1e-300):
        return None
    p = Vt[-1].reshape(3, 4)
    Pm = np.empty((3, 4))
    Pm[:, :3] = p[:, :3] / sc
    Pm[:, 3] = p[:, 3] - (p[:, :3] @ cen[:, 0]) / sc
    n3 = np.linalg.norm(Pm[2, :3])
    if not n3 > 1e-300:
        return None
    s0 = 1.0 / n3
    if (Pm[2, :3] @ X[:, 0] + Pm[2, 3]) * s0 < 0:
        s0 = -s0
    R, t = Pm[:, :3] * s0, Pm[:, 3] * s0
    U, _, Vt2 = np.linalg.svd(R)
    Rn = U @ Vt2
    if np.linalg.det(Rn) < 0.5 or np.linalg.det(R) <= 0:
        return None
    return Rn, t


def inlier_mask(X, uv, K, R, t, thr):
    p = R @ X + t[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        du = K[0, 0] * p[0] / p[2] + K[0, 2] - uv[0]
        dv = K[1, 1] * p[1] / p[2] + K[1, 2] - uv[1]
    return (p[2] > 1e-9) & (du * du + dv * dv < thr * thr)


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th * th <= np.finfo(float).eps:
        return np.array([[1, -w[2], w[1]], [w[2], 1, -w[0]], [-w[1], w[0], 1]])
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * np.outer(k, k)


def refine(X, uv, K, R, t, mask, iters):
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    Xi, ui = X[:, mask], uv[:, mask]
    for _ in range(iters):
        q = R @ Xi
        p = q + t[:, None]
        iz = 1.0 / p[2]
        ru, rv = fx * p[0] * iz + cx - ui[0], fy * p[1] * iz + cy - ui[1]
        n = Xi.shape[1]
        dp = np.zeros((3, 6, n))
        dp[0, 1], dp[0, 2], dp[0, 3] = q[2], -q[1], 1
        dp[1, 0], dp[1, 2], dp[1, 4] = -q[2], q[0], 1
        dp[2, 0], dp[2, 1], dp[2, 5] = q[1], -q[0], 1
        Ju = fx * iz * dp[0] - fx * p[0] * iz * iz * dp[2]
        Jv = fy * iz * dp[1] - fy * p[1] * iz * iz * dp[2]
        A = Ju @ Ju.T + Jv @ Jv.T
        g = Ju @ ru + Jv @ rv
        A = A + np.diag(1e-9 * (1.0 + np.diag(A)))
        try:
            d = np.linalg.solve(A, -g)
        except np.linalg.LinAlgError:
            break
        R = _rodrigues(d[:3]) @ R
        t = t + d[3:]
    return R, t


def pnp_ransac(pc, coarse, fine, K_scaled, W_fine, samples, reproj_err=0.6, refine_rounds=20, refine_iters=5, pixels=None):
    """One frame.  -> (P 4x4, outlier_ratio, n_inliers, n_corr, best, per-hypothesis inlier counts)."""
    X, uv = correspondences(pc, coarse, fine, W_fine, pixels)
    cnt = X.shape[1]
    iters = samples.shape[0]
    counts = np.full(iters, -1, dtype=np.int64)
    models = [None] * iters
    if cnt >= 6:
        for it in range(iters):
            idx = samples[it].astype(np.int64) % cnt
            m = dlt6(X[:, idx], uv[:, idx], K_scaled)
            if m is None:
                continue
            models[it] = m
            counts[it] = int(inlier_mask(X, uv, K_scaled, m[0], m[1], reproj_err).sum())
    best = int(np.argmax(counts)) if cnt >= 6 else -1
    if cnt < 6 or counts[best] < 6:
        return np.eye(4), 1.0, 0, cnt, -1, counts
    R, t = models[best]
    nin = int(counts[best])
    for _ in range(refine_rounds):          # locally optimised RANSAC: inliers -> Gauss-Newton, kept if no inlier is lost
        mask = inlier_mask(X, uv, K_scaled, R, t, reproj_err)
        R1, t1 = refine(X, uv, K_scaled, R, t, mask, refine_iters)
        c1 = int(inlier_mask(X, uv, K_scaled, R1, t1, reproj_err).sum())
        if c1 < nin:
            break
        R, t, nin = R1, t1, c1
    P, ratio = accept(R, t, nin, cnt)
    return P, ratio, nin, cnt, best, counts
