"""CPU: the oracle's index_max / ball_query restatements against the reference-generated golden vectors and
hand-derived known answers."""
import numpy as np
import pytest

from oracle import ops_np


@pytest.mark.parametrize("case", ["rand", "relu_ties", "tiny", "ragged"])
def test_index_max_oracle_vs_reference_golden(golden, case):
    g = golden("index_max_golden.npz")
    out = ops_np.index_max_forward(g[case + "_data"], g[case + "_index"], int(g[case + "_K"]))
    np.testing.assert_array_equal(out, g[case + "_out"])


def test_index_max_vectorised_equals_literal_loops():
    rng = np.random.default_rng(0)
    data = np.maximum(rng.standard_normal((2, 3, 200)), 0).astype(np.float32)
    data[0, 0, 5] = np.nan
    index = rng.integers(0, 7, (2, 200)).astype(np.int32)
    np.testing.assert_array_equal(ops_np.index_max_forward(data, index, 9), ops_np.index_max_forward_loops(data, index, 9))


def test_index_max_against_live_reference_build():
    from oracle.ref_loader import load_ref_index_max
    m = load_ref_index_max()
    if m is None:
        pytest.skip("oracle/_ref not built (reference absent)")
    import torch
    rng = np.random.default_rng(3)
    data = rng.standard_normal((2, 4, 513)).astype(np.float32)
    index = rng.integers(0, 20, (2, 513)).astype(np.int32)
    ref = m.forward_cpu(torch.from_numpy(data), torch.from_numpy(index), 24).numpy()
    np.testing.assert_array_equal(ops_np.index_max_forward(data, index, 24), ref)


def test_ball_query_known_answers():
    """Hand-derived from ball_query_cuda.cu:23-47 (parity unpinned by the reference: no CPU variant, no test)."""
    d = np.array([[[0.5, 0.1, 0.9, 0.2, 0.1, 0.05],     # hits n=1,3,4,5 ; K=3 -> first three
                   [0.9, 0.9, 0.9, 0.9, 0.9, 0.9],      # no hit -> zeros
                   [0.9, 0.2, 0.9, 0.9, 0.9, 0.9],      # one hit -> repeated
                   [0.9, 0.2, 0.9, 0.2, 0.9, 0.9]]],    # two hits, K=5 -> 1,3,1,3,1
                 dtype=np.float32)
    out = ops_np.ball_query_forward(d, 0.2, 3)
    np.testing.assert_array_equal(out[0], [[1, 3, 4], [0, 0, 0], [1, 1, 1], [1, 3, 1]])
    out5 = ops_np.ball_query_forward(d, 0.2, 5)
    np.testing.assert_array_equal(out5[0, 3], [1, 3, 1, 3, 1])
    np.testing.assert_array_equal(out5[0, 0], [1, 3, 4, 5, 1])
    # dist == radius is a hit ( <= ), float comparison
    assert ops_np.ball_query_forward(np.full((1, 1, 2), 0.2, np.float32), 0.2, 1)[0, 0, 0] == 0
    assert ops_np.ball_query_forward(np.zeros((1, 1, 0), np.float32), 0.2, 2).tolist() == [[[0, 0]]]
