"""Shared checks against tests/golden/network_fullsize_golden.npz (the imported reference at the BASELINE config-2/3 size;
generator: tests/golden/make_golden.py make_fullsize).  Used by the CPU oracle pin and by the GPU parity test."""
import hashlib

import numpy as np

PCTS = np.array([0, 1, 5, 25, 50, 75, 95, 99, 100], dtype=np.float64)
NAMES = ("pc", "intensity", "sn", "node_a", "node_b", "img")


def stats(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.concatenate((np.percentile(a, PCTS), [a.mean(), np.abs(a).mean(), np.abs(a).max()]))


def inputs(g):
    """Regenerate the seeded frame and verify it is the one the reference saw."""
    from deepi2p_amd import synthetic
    _, N, H, W, seed, stride = [int(v) for v in g["meta"]]
    b = synthetic.make_batch(seed, 1, N=N, H=H, W=W)
    h = hashlib.sha256()
    for k in NAMES:
        h.update(np.ascontiguousarray(b[k]).tobytes())
    assert h.hexdigest() == str(g["input_sha256"]), "synthetic generator drifted: the golden inputs cannot be regenerated"
    return b, N, H, W, stride


def check_logits(g, tag, coarse, fine, rel, max_flip):
    """coarse [1,2,N] (and fine [1,L,N]) numpy: sub-sampled logits within rel * max|logit|, percentile statistics, and ALL
    argmax labels: flips are only allowed where the reference's own decision margin is below the tolerance."""
    stride = int(g["meta"][5])
    out = {}
    for head, val in (("coarse", coarse), ("fine", fine)):
        if val is None:
            continue
        key = "%s_%s" % (tag, head)
        amax = float(g[key + "_absmax"])
        tol = rel * amax + 1e-7
        err = float(np.abs(val[:, :, ::stride] - g[key + "_sub"]).max())
        assert err <= tol, "%s logits: max err %.3g > %.3g" % (key, err, tol)
        st = stats(val)
        assert np.abs(st - g[key + "_stats"]).max() <= 2 * tol, key + " statistics"
        lab = val.argmax(1)
        ref = np.unpackbits(g[key + "_labels"], axis=1)[:, :lab.shape[1]] if head == "coarse" else g[key + "_labels"]
        diff = lab != ref
        margin = g[key + "_margin"]
        assert not np.any(diff & (margin > 2 * tol)), key + ": label differs where the reference's margin exceeds the tolerance"
        assert diff.mean() <= max_flip, "%s: label flip rate %.4f" % (key, diff.mean())
        out[key] = (err, float(diff.mean()))
    return out
