"""Oracle restatements of the two native torch extensions (numpy, CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

index_max  -- restates models/index_max_ext/index_max.cpp:73-112 (forward_cpu; the
              CUDA kernels index_max_cuda.cu:10-26 / :30-62 have identical
              semantics).  PINNED: checked bit-exact against the reference's own
              ``forward_cpu`` compiled from /root/reference (oracle/_ref, see
              oracle/Makefile) and against tests/golden/index_max_*.npz generated
              from it (tests/golden/make_golden.py).
ball_query -- restates models/ball_query_ext/ball_query_cuda.cu:11-50.  The
              reference has NO CPU variant and no test for it: PARITY UNPINNED
              beyond the kernel text; known-answer cases are hand-derived from
              that text in tests/test_ball_query_oracle.py.
"""
import numpy as np


def index_max_forward(data: np.ndarray, index: np.ndarray, K: int) -> np.ndarray:
    """Segment arg-max.  data f32[B,C,N], index i32[B,N] in [0,K) -> i32[B,C,K].

    Reference loop (index_max.cpp:98-109): running max initialised to -1000
    (:81), index to 0 (:80); strict ``>`` so the FIRST n attaining the max wins,
    values <= -1000 and NaN never win, empty clusters keep index 0.
    Vectorised: for each (b, k) take the points of the cluster in increasing n
    and use argmax (first occurrence) -- identical to the sequential scan.
    """
    data = np.asarray(data, dtype=np.float32)
    index = np.asarray(index)
    B, C, N = data.shape
    out = np.zeros((B, C, K), dtype=np.int32)
    for b in range(B):
        order = np.argsort(index[b], kind="stable")          # cluster-sorted, n ascending inside
        sorted_idx = index[b][order]
        starts = np.searchsorted(sorted_idx, np.arange(K), side="left")
        ends = np.searchsorted(sorted_idx, np.arange(K), side="right")
        for k in range(K):
            if ends[k] == starts[k]:
                continue
            members = order[starts[k]:ends[k]]                # ascending n
            vals = data[b][:, members]                        # C x m
            # NaN never wins a strict '>' comparison: treat as -inf
            vals = np.where(np.isnan(vals), -np.inf, vals)
            am = np.argmax(vals, axis=1)                      # first occurrence
            best = vals[np.arange(C), am]
            win = best > np.float32(-1000.0)
            out[b, :, k] = np.where(win, members[am], 0).astype(np.int32)
    return out


def index_max_forward_loops(data, index, K):
    """Literal triple loop of index_max.cpp:98-109 (small inputs only)."""
    B, C, N = data.shape
    max_idx = np.zeros((B, C, K), dtype=np.int32)
    max_val = np.full((B, C, K), -1000.0, dtype=np.float32)
    for b in range(B):
        for c in range(C):
            for n in range(N):
                k = int(index[b, n])
                v = data[b, c, n]
                if v > max_val[b, c, k]:
                    max_val[b, c, k] = v
                    max_idx[b, c, k] = n
    return max_idx


def ball_query_forward(node_to_point_dist: np.ndarray, radius: float, K: int) -> np.ndarray:
    """Radius query.  dist f32[B,M,N] -> i32[B,M,K].

    ball_query_cuda.cu:23-33: scan n ascending, append n while dist <= radius
    until K found.  :37-41: zero hits -> all zeros.  :42-47: 0<u<K hits -> slot
    u+i = slot (i % u) (cyclic repetition of the first u hits).  The comparison is
    done in float (``const float radius``).
    """
    d = np.asarray(node_to_point_dist, dtype=np.float32)
    B, M, N = d.shape
    r = np.float32(radius)
    out = np.zeros((B, M, K), dtype=np.int32)
    for b in range(B):
        for m in range(M):
            hits = np.nonzero(d[b, m] <= r)[0][:K]
            u = hits.shape[0]
            if u == 0:
                continue
            out[b, m, :u] = hits
            if u < K:
                i = np.arange(K - u)
                out[b, m, u:] = hits[i % u]
    return out
