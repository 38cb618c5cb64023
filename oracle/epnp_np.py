"""Oracle: numpy restatement of EPnP (Lepetit, Moreno-Noguer, Fua, "EPnP: An Accurate O(n) Solution to the PnP Problem",
IJCV 2009) in the arrangement of OpenCV's calib3d/src/epnp.cpp, and of the RANSAC wrapper the reference calls
(cv2.solvePnPRansac(flags=SOLVEPNP_EPNP), evaluation/registration_pnp.py:123-132).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: OpenCV (un-vendored `opencv-python`, unpinned) is absent from the reference tree and from this image, and its
RANSAC sampling is internal.  What is restated from the published algorithm / OpenCV's documented behaviour:
  * four control points = centroid + principal directions scaled by the standard deviations; barycentric coordinates;
    the 2n x 12 matrix M, the four eigenvectors of M^T M with the smallest eigenvalues;
  * the 6 x 10 system L beta = rho of control-point distances, the three linearisations (N = 1, 2, 3) each refined by five
    Gauss-Newton steps, pose by Horn/Arun absolute orientation, the N with the smallest mean reprojection error wins;
  * RANSAC: minimal sample 5 (OpenCV's model_points for EPNP; 4 points when only 4 correspondences exist -- OpenCV switches to
    P3P there), inlier iff squared reprojection error <= threshold^2, the model with the most inliers, then ONE EPnP re-fit on
    all its inliers (OpenCV >= 3.x), success iff there are inliers; the explicit sample table replaces OpenCV's RNG.
Pinned by exact-correspondence pose recovery for n = 4, 5, 6, 50, 2000 and noisy / outlier cases (tests/test_pnp.py)."""
import numpy as np

PAIRS = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]


def _control_points(X):
    n = X.shape[1]
    c0 = X.mean(axis=1)
    D = (X - c0[:, None])
    w, U = np.linalg.eigh(D @ D.T)                       # ascending
    cws = [c0]
    for i in (2, 1, 0):                                  # OpenCV: singular values in descending order
        u = U[:, i]
        u = -u if u[np.argmax(np.abs(u))] < 0 else u     # sign convention: largest component positive (an eigen-solver's sign is
        cws.append(c0 + np.sqrt(max(w[i], 0.0) / n) * u)  # arbitrary, and with noisy data EPnP is not invariant to the mirroring)
    return np.array(cws)                                 # 4 x 3


def _alphas(X, cws):
    CC = (cws[1:] - cws[0]).T                            # columns = c_j - c_0
    a = np.linalg.solve(CC, X - cws[0][:, None])         # 3 x n
    return np.vstack((1.0 - a.sum(axis=0), a))           # 4 x n


def _fill_M(alphas, uv, fu, fv, uc, vc):
    n = uv.shape[1]
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[j] * fu
        M[0::2, 3 * j + 2] = alphas[j] * (uc - uv[0])
        M[1::2, 3 * j + 1] = alphas[j] * fv
        M[1::2, 3 * j + 2] = alphas[j] * (vc - uv[1])
    return M


def _L_6x10(v):
    """v: 4 x 12 (v[0] = eigenvector of the smallest eigenvalue)."""
    dv = np.zeros((4, 6, 3))
    for i in range(4):
        for p, (a, b) in enumerate(PAIRS):
            dv[i, p] = v[i, 3 * a:3 * a + 3] - v[i, 3 * b:3 * b + 3]
    L = np.zeros((6, 10))
    for p in range(6):
        d = dv[:, p]
        L[p] = [d[0] @ d[0], 2 * d[0] @ d[1], d[1] @ d[1], 2 * d[0] @ d[2], 2 * d[1] @ d[2], d[2] @ d[2],
                2 * d[0] @ d[3], 2 * d[1] @ d[3], 2 * d[2] @ d[3], d[3] @ d[3]]
    return L


def _rho(cws):
    return np.array([np.sum((cws[a] - cws[b]) ** 2) for a, b in PAIRS])


def _lstsq(A, b):
    return np.linalg.lstsq(A, b, rcond=None)[0]


def _betas_approx(L, rho, N):
    betas = np.zeros(4)
    if N == 1:                                           # betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44] -> [B11 B12 B13 B14]
        b = _lstsq(L[:, [0, 1, 3, 6]], rho)
        s = np.sqrt(abs(b[0]))
        sign = -1.0 if b[0] < 0 else 1.0
        betas[:] = [s, sign * b[1] / s, sign * b[2] / s, sign * b[3] / s]
    elif N == 2:                                         # [B11 B12 B22]
        b = _lstsq(L[:, [0, 1, 2]], rho)
        if b[0] < 0:
            betas[0] = np.sqrt(-b[0]); betas[1] = np.sqrt(-b[2]) if b[2] < 0 else 0.0
        else:
            betas[0] = np.sqrt(b[0]); betas[1] = np.sqrt(b[2]) if b[2] > 0 else 0.0
        if b[1] < 0:
            betas[0] = -betas[0]
    else:                                                # [B11 B12 B22 B13 B23]
        b = _lstsq(L[:, 0:5], rho)
        if b[0] < 0:
            betas[0] = np.sqrt(-b[0]); betas[1] = np.sqrt(-b[2]) if b[2] < 0 else 0.0
        else:
            betas[0] = np.sqrt(b[0]); betas[1] = np.sqrt(b[2]) if b[2] > 0 else 0.0
        if b[1] < 0:
            betas[0] = -betas[0]
        betas[2] = b[3] / betas[0]
    return betas


def _gauss_newton(L, rho, betas, iters=5):
    b = betas.copy()
    for _ in range(iters):
        A = np.stack((2 * L[:, 0] * b[0] + L[:, 1] * b[1] + L[:, 3] * b[2] + L[:, 6] * b[3],
                      L[:, 1] * b[0] + 2 * L[:, 2] * b[1] + L[:, 4] * b[2] + L[:, 7] * b[3],
                      L[:, 3] * b[0] + L[:, 4] * b[1] + 2 * L[:, 5] * b[2] + L[:, 8] * b[3],
                      L[:, 6] * b[0] + L[:, 7] * b[1] + L[:, 8] * b[2] + 2 * L[:, 9] * b[3]), axis=1)
        r = rho - (L[:, 0] * b[0] * b[0] + L[:, 1] * b[0] * b[1] + L[:, 2] * b[1] * b[1] + L[:, 3] * b[0] * b[2] +
                   L[:, 4] * b[1] * b[2] + L[:, 5] * b[2] * b[2] + L[:, 6] * b[0] * b[3] + L[:, 7] * b[1] * b[3] +
                   L[:, 8] * b[2] * b[3] + L[:, 9] * b[3] * b[3])
        b = b + _lstsq(A, r)
    return b


def _pose_from_betas(v, betas, alphas, X):
    ccs = sum(betas[k] * v[k].reshape(4, 3) for k in range(4))          # control points in the camera frame
    pcs = alphas.T @ ccs                                                 # n x 3
    if pcs[0, 2] < 0:
        ccs, pcs = -ccs, -pcs
    pc0, pw0 = pcs.mean(axis=0), X.mean(axis=1)
    ABt = (pcs - pc0).T @ (X - pw0[:, None]).T
    U, _, Vt = np.linalg.svd(ABt)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R[2] = -R[2]
    return R, pc0 - R @ pw0


def reprojection_error(X, uv, R, t, fu, fv, uc, vc):
    p = R @ X + t[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        ue, ve = uc + fu * p[0] / p[2], vc + fv * p[1] / p[2]
        return float(np.mean(np.sqrt((uv[0] - ue) ** 2 + (uv[1] - ve) ** 2)))


def epnp(X, uv, K):
    """X 3 x n (n >= 4), uv 2 x n, K 3x3 -> (R, t, mean reprojection error) or None when the configuration is degenerate."""
    fu, fv, uc, vc = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    X = np.asarray(X, np.float64)
    uv = np.asarray(uv, np.float64)
    cws = _control_points(X)
    try:
        alphas = _alphas(X, cws)
    except np.linalg.LinAlgError:
        return None
    if not np.all(np.isfinite(alphas)):
        return None
    M = _fill_M(alphas, uv, fu, fv, uc, vc)
    w, V = np.linalg.eigh(M.T @ M)                                        # ascending: columns 0..3 = null-space basis
    v = V[:, :4].T.copy()
    L, rho = _L_6x10(v), _rho(cws)
    best = None
    for N in (1, 2, 3):
        with np.errstate(all="ignore"):
            betas = _gauss_newton(L, rho, _betas_approx(L, rho, N))
            if not np.all(np.isfinite(betas)):
                continue
            R, t = _pose_from_betas(v, betas, alphas, X)
            err = reprojection_error(X, uv, R, t, fu, fv, uc, vc)
        if np.isfinite(err) and (best is None or err < best[2]):
            best = (R, t, err)
    return best


def inlier_mask(X, uv, K, R, t, thr):
    p = R @ X + t[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        du = K[0, 0] * p[0] / p[2] + K[0, 2] - uv[0]
        dv = K[1, 1] * p[1] / p[2] + K[1, 2] - uv[1]
        return (du * du + dv * dv) <= thr * thr


def epnp_ransac(X, uv, K, samples, reproj_err=0.6):
    """X 3 x cnt, uv 2 x cnt, samples int[iters, >= 5] (reduced modulo cnt; a sample with repeated indices is skipped).
    -> (R, t, inlier mask, best hypothesis, per-hypothesis inlier counts) ; R is None when nothing was found."""
    cnt = X.shape[1]
    iters = samples.shape[0]
    counts = np.full(iters, -1, dtype=np.int64)
    models = [None] * iters
    m = 5 if cnt >= 5 else 4
    if cnt < 4:
        return None, None, np.zeros(cnt, bool), -1, counts
    for it in range(iters):
        idx = samples[it, :m].astype(np.int64) % cnt
        if len(set(idx.tolist())) < m:
            continue
        sol = epnp(X[:, idx], uv[:, idx], K)
        if sol is None:
            continue
        models[it] = sol
        counts[it] = int(inlier_mask(X, uv, K, sol[0], sol[1], reproj_err).sum())
    best = int(np.argmax(counts))
    if counts[best] < m:
        return None, None, np.zeros(cnt, bool), -1, counts
    mask = inlier_mask(X, uv, K, models[best][0], models[best][1], reproj_err)
    sol = epnp(X[:, mask], uv[:, mask], K) if mask.sum() >= 4 else None    # the re-fit on all inliers of the winning model
    if sol is None:
        sol = models[best]
    return sol[0], sol[1], mask, best, counts
