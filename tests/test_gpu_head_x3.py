"""GPU parity of di2p_point_head_x3 (the coarse per-point head as one wave-autonomous launch on the bf16 matrix instructions with exact
three-way fp32 splits, head_x3.hip) against an fp64 evaluation of the same three layers and against the fp32-MFMA fused head it replaces
(di2p_point_head, which is bit-identical to the reference-shaped chain of three pointwise layers).  Reference: per_point_pn of
models/networks_united.py:57-74, applied at :188-197."""
import pytest
import torch

from deepi2p_amd import _lib

pytestmark = pytest.mark.gpu


def _case(dev, B, N, nodes, P, seed, weights=True):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    d = dict(first=r(B, 32, N), second=r(B, 64, N), W0=r(96, 128) / 96 ** 0.5, W1=r(128, 128) / 128 ** 0.5, W2=r(128, P) / 128 ** 0.5,
             sc0=torch.rand(128, generator=g) + 0.5, sh0=r(128) * 0.1, sc1=torch.rand(128, generator=g) + 0.5, sh1=r(128) * 0.1, sh2=r(P) * 0.1,
             Ga=r(B, nodes[0], 128) * 0.5, Gb=r(B, nodes[1], 128) * 0.5,
             ia=torch.randint(0, nodes[0], (B, N, 3), generator=g, dtype=torch.int32), ib=torch.randint(0, nodes[1], (B, N, 3), generator=g, dtype=torch.int32),
             wa=torch.rand(B, N, 3, generator=g), wb=torch.rand(B, N, 3, generator=g) if weights else None)
    return {k: (v.to(dev) if v is not None else None) for k, v in d.items()}


def _ref64(d):
    f = lambda t: t.double()
    x = torch.cat((f(d["first"]), f(d["second"])), dim=1)                         # [B,96,N]
    B, _, N = x.shape
    y = torch.einsum("km,bkn->bmn", f(d["W0"]), x)
    for G, idx, w in ((d["Ga"], d["ia"], d["wa"]), (d["Gb"], d["ib"], d["wb"])):
        rows = torch.gather(f(G), 1, idx.long().reshape(B, N * 3, 1).expand(B, N * 3, 128)).reshape(B, N, 3, 128)
        ww = f(w) if w is not None else torch.ones(B, N, 3, dtype=torch.float64, device=G.device)
        y = y + (rows * ww.unsqueeze(3)).sum(2).transpose(1, 2)
    y = torch.relu(y * f(d["sc0"]).view(1, -1, 1) + f(d["sh0"]).view(1, -1, 1))
    y = torch.relu(torch.einsum("km,bkn->bmn", f(d["W1"]), y) * f(d["sc1"]).view(1, -1, 1) + f(d["sh1"]).view(1, -1, 1))
    return torch.einsum("kp,bkn->bpn", f(d["W2"]), y) + f(d["sh2"]).view(1, -1, 1)


def _run_x3(d, N):
    from deepi2p_amd import ops
    packed = {"W0p": ops.head_x3_pack(d["W0"].contiguous()), "W1p": ops.head_x3_pack(d["W1"].contiguous()),
              "ss": torch.stack((d["sc0"], d["sh0"], d["sc1"], d["sh1"])).contiguous(), "relu0": True, "relu1": True,
              "W2t": d["W2"].contiguous(), "sc2": None, "sh2": d["sh2"], "relu2": False}
    return ops.point_head_x3(d["first"], d["second"], packed, [(d["Ga"], d["ia"], d["wa"]), (d["Gb"], d["ib"], d["wb"])], N)


def _run_fp32(d, N):
    from deepi2p_amd import ops
    return ops.point_head([ops.Src(d["first"]), ops.Src(d["second"])], (d["W0"].contiguous(), d["sc0"], d["sh0"], True),
                          (d["W1"].contiguous(), d["sc1"], d["sh1"], True), (d["W2"].contiguous(), None, d["sh2"], False), N,
                          gathered=[(d["Ga"], d["ia"], d["wa"]), (d["Gb"], d["ib"], d["wb"])])


@pytest.mark.parametrize("B,N,nodes,P,tab", [(2, 20480, (128, 128), 2, 1), (3, 1000, (128, 128), 2, 1), (2, 4100, (64, 96), 4, 1),
                                              (2, 1000, (128, 128), 2, 0), (1, 33, (16, 16), 1, 1), (2, 2048, (700, 700), 2, 1),
                                              (2, 20480, (128, 128), 2, 2), (3, 1000, (128, 128), 4, 2)])
def test_point_head_x3_matches_fp64_and_is_as_accurate_as_the_fp32_head(dev, B, N, nodes, P, tab):
    """tab = 1: tables in LDS, eight waves per workgroup (the default); 2: four waves (one per SIMD); 0: the node tables are gathered from memory
    (knob head_x3_tab); 700 nodes do not fit the LDS: the same path by itself.
    N = 1000 / 4100 / 33: ragged last blocks."""
    d = _case(dev, B, N, nodes, P, 11 + N + P, weights=(N != 4100))
    ref = _ref64(d)
    with _lib.option("head_x3_tab", tab):
        y3 = _run_x3(d, N)
    e3 = (y3.double() - ref).abs()
    tol = 3e-6 * 128 ** 0.5 * float(ref.abs().max()) * 4 + 1e-6          # three chained contractions
    assert float(e3.max()) <= tol, (float(e3.max()), tol)
    if N % 4 == 0:                                                          # the fp32 fused head wants whole 4-column groups
        e1 = (_run_fp32(d, N).double() - ref).abs()
        assert float(e3.max()) <= 1.25 * float(e1.max()) + 1e-7 and float((e3 ** 2).mean().sqrt()) <= 1.1 * float((e1 ** 2).mean().sqrt()) + 1e-8, \
            (float(e3.max()), float(e1.max()), float((e3 ** 2).mean().sqrt()), float((e1 ** 2).mean().sqrt()))


def test_point_head_x3_is_deterministic_and_tables_in_lds_equal_tables_in_memory(dev):
    d = _case(dev, 2, 5000, (128, 128), 2, 5)
    a = _run_x3(d, 5000)
    assert torch.equal(a, _run_x3(d, 5000))
    for tab in (0, 2):
        with _lib.option("head_x3_tab", tab):
            b = _run_x3(d, 5000)
        assert torch.equal(a, b)          # the same arithmetic in the same order, whatever the tables are read from and however many waves share a SIMD


def test_network_logits_with_and_without_head_x3(dev):
    """The whole classifier with the head on the bf16x3 kernel against the fp32-MFMA fused head: logits agree to fp32 round-off."""
    from deepi2p_amd import synthetic as nt
    from deepi2p_amd.networks import KeypointDetector
    opt = nt.OptLike(20480, 160, 512, False)
    det = KeypointDetector(opt)
    det.load_state_dict(nt.synthetic_state_dict(opt))
    det = det.to(dev)
    b = nt.make_batch(77, 2, N=20480, H=160, W=512)
    x = [torch.from_numpy(b[k]).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    with _lib.option("head_x3", 1):
        a = det(*x).clone()
    with _lib.option("head_x3", 0):
        c = det(*x).clone()
    assert float((a - c).abs().max()) <= 1e-4 * float(c.abs().max()) + 1e-6
