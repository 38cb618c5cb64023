"""Numpy restatement of the device random draws (deepi2p_amd/csrc/rng.hip): Philox4x32-10 (Salmon et al., "Parallel random
numbers: as easy as 1, 2, 3", SC'11; constants of the Random123 reference implementation), 53-bit uniforms in (0, 1],
Box-Muller normals, and the key-sort random choice.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The REFERENCE draws from unseeded host Mersenne Twisters
(evaluation/registration_lsq.py:163-164 `random.gauss / random.uniform`; data/kitti_pc_img_pose_loader.py:158-171,416-423
`np.random.choice(..., replace=False)`): no stream of its own exists to compare with, only the distributions.  Pinned by the
Random123 known-answer vectors (tests/test_rng.py)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint64 arrays holding 32-bit values); scalar key.  -> 4 uint64 arrays (32-bit)."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & np.uint64(MASK) for c in (c0, c1, c2, c3)]
    k0, k1 = int(k0) & MASK, int(k1) & MASK
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & np.uint64(MASK)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & np.uint64(MASK)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def u53(hi, lo):
    m = ((hi >> np.uint64(5)) << np.uint64(26)) | (lo >> np.uint64(6))
    return (m.astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)


def draw_restarts(seed, F, R, ry_sigma, t_amplitude):
    """-> (ry_noise f64[F,R], init_T f64[F,R,3]) exactly as di2p_draw_restarts (normals up to libm rounding)."""
    n = F * R
    i = np.arange(n, dtype=np.uint64)
    lo, hi = i & np.uint64(MASK), i >> np.uint64(32)
    z = np.zeros(n, dtype=np.uint64)
    a = philox4x32_10(lo, hi, z, z, seed & MASK, (seed >> 32) & MASK)
    b = philox4x32_10(lo, hi, z, z + np.uint64(1), seed & MASK, (seed >> 32) & MASK)
    u1, u2, u3 = u53(a[0], a[1]), u53(a[2], a[3]), u53(b[0], b[1])
    ry = ry_sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)
    T = np.zeros((n, 3))
    T[:, 2] = t_amplitude * (2.0 * u3 - 1.0)
    return ry.reshape(F, R), T.reshape(F, R, 3), (u1.reshape(F, R), u2.reshape(F, R), u3.reshape(F, R))


def random_choice(seed, stream_id, B, n_src, n_out):
    """-> i32[B, n_out]: indices of the n_out smallest (philox key, index) pairs of every frame, in key order."""
    out = np.zeros((B, n_out), dtype=np.int32)
    n = np.arange(n_src, dtype=np.uint64)
    for b in range(B):
        r = philox4x32_10(n, np.full(n_src, b, np.uint64), np.full(n_src, stream_id, np.uint64), np.full(n_src, 2, np.uint64),
                          seed & MASK, (seed >> 32) & MASK)
        keys = (r[0] << np.uint64(32)) | n
        out[b] = (np.sort(keys)[:n_out] & np.uint64(MASK)).astype(np.int32)
    return out
