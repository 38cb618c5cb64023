"""Thin torch-tensor front-ends of the C-ABI kernels (include/deepi2p_hip.h).

torch supplies device memory and the current HIP stream only.  Every function launches on
``torch.cuda.current_stream()`` (the reference launched on the legacy default stream,
index_max_cuda.cu:75,93) and never synchronises, so sequences of these calls can be captured in a
``torch.cuda.CUDAGraph``.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import EpilogueT, SrcT, call, ptr, require_cuda, stream

_f32, _i32, _f64 = torch.float32, torch.int32, torch.float64
MAX_GK = 16          # DI2P_MAX_GK (include/deepi2p_hip.h)


def _chk(t, dtype, name):
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))


# ------------------------------------------------------------------------------------ native ops
def index_max(data, index, K, return_values=False, mask=None):
    """data f32[B,C,N], index i32[B,N] -> max_idx i32[B,C,K] (and masked max values if asked)."""
    require_cuda(data, index, mask)
    _chk(data, _f32, "data")
    _chk(index, _i32, "index")
    B, C, N = data.shape
    K = int(K)
    if tuple(index.shape) != (B, N):
        raise RuntimeError("index must be i32[B, N] for data f32[B, C, N]")
    if mask is not None and tuple(mask.shape) != (B, K):
        raise RuntimeError("mask must be f32[B, K]")
    idx = torch.empty((B, C, K), dtype=_i32, device=data.device)
    ws = torch.empty((B * C * K,), dtype=torch.int64, device=data.device)
    if return_values:
        val = torch.empty((B, C, K), dtype=_f32, device=data.device)
        call("di2p_index_max_values", ptr(data), ptr(index), ptr(mask), ptr(val), ptr(idx), B, C, N, K, ptr(ws), stream())
        return idx, val
    call("di2p_index_max_forward", ptr(data), ptr(index), ptr(idx), B, C, N, K, ptr(ws), stream())
    return idx


def ball_query(node_to_point_dist, radius, K):
    require_cuda(node_to_point_dist)
    _chk(node_to_point_dist, _f32, "node_to_point_dist")
    B, M, N = node_to_point_dist.shape
    out = torch.empty((B, M, int(K)), dtype=_i32, device=node_to_point_dist.device)
    call("di2p_ball_query_forward", ptr(node_to_point_dist), ptr(out), float(radius), int(K), B, M, N, stream())
    return out


def knn_nodes(query, nodes, k, want_weights=False):
    """query f32[B,3,Nq], nodes f32[B,3,M] -> idx i32[B,Nq,k] (+ weights f32[B,Nq,k])."""
    require_cuda(query, nodes)
    B, _, Nq = query.shape
    M = nodes.shape[2]
    idx = torch.empty((B, Nq, k), dtype=_i32, device=query.device)
    w = torch.empty((B, Nq, k), dtype=_f32, device=query.device) if want_weights else None
    call("di2p_knn_nodes", ptr(query), ptr(nodes), ptr(idx), ptr(w), B, Nq, M, int(k), stream())
    return (idx, w) if want_weights else idx


def cluster_stats(pc, knn_idx, M):
    require_cuda(pc, knn_idx)
    B, _, N = pc.shape
    mean = torch.empty((B, 3, M), dtype=_f32, device=pc.device)
    mask = torch.empty((B, M), dtype=_f32, device=pc.device)
    min_idx = torch.empty((B, N), dtype=_i32, device=pc.device)
    call("di2p_cluster_stats", ptr(pc), ptr(knn_idx), knn_idx.shape[2], ptr(mean), ptr(mask), ptr(min_idx), B, N, M, stream())
    return mean, mask, min_idx


def build_point_input(pc, intensity, sn, cluster_mean, min_idx):
    require_cuda(pc, intensity, sn, cluster_mean, min_idx)
    B, _, N = pc.shape
    centers = torch.empty_like(pc)
    aug = torch.empty((B, 7, N), dtype=_f32, device=pc.device)
    call("di2p_build_point_input", ptr(pc), ptr(intensity), ptr(sn), ptr(cluster_mean), ptr(min_idx), ptr(centers),
         ptr(aug), B, N, cluster_mean.shape[2], stream())
    return centers, aug


def interpolate(feats, idx, weights):
    require_cuda(feats, idx, weights)
    B, C, M = feats.shape
    Nq, k = idx.shape[1], idx.shape[2]
    out = torch.empty((B, C, Nq), dtype=_f32, device=feats.device)
    call("di2p_interpolate", ptr(feats), ptr(idx), ptr(weights), ptr(out), B, C, M, Nq, k, stream())
    return out


def gather_neighbors(database, query, idx):
    require_cuda(database, query, idx)
    B, _, Md = database.shape
    Mq, K = idx.shape[1], idx.shape[2]
    out = torch.empty((B, 3, Mq * K), dtype=_f32, device=database.device)
    call("di2p_gather_neighbors", ptr(database), ptr(query), ptr(idx), ptr(out), B, Md, Mq, K, stream())
    return out


def argmax_channels(scores):
    """scores f32[B,C,N] (a channel slice of a larger [B,C',N] tensor is fine) -> i32[B,N]."""
    if not scores.is_cuda:
        raise RuntimeError("tensor must be a CUDA tensor/variable")
    B, C, N = scores.shape
    if scores.stride(2) != 1 or scores.stride(1) != N:
        raise RuntimeError("scores rows must be contiguous")
    out = torch.empty((B, N), dtype=_i32, device=scores.device)
    call("di2p_argmax_channels", ptr(scores), ptr(out), B, C, N, scores.stride(0), stream())
    return out


def channel_max(x):
    require_cuda(x)
    B, C, N = x.shape
    out = torch.empty((B, C), dtype=_f32, device=x.device)
    call("di2p_channel_max", ptr(x), ptr(out), B, C, N, stream())
    return out


# ------------------------------------------------------------------------------------ contractions
class Src:
    """One operand source of the virtual concatenation (see di2p_src_t)."""

    def __init__(self, t, mode=_lib.SRC_DENSE, gidx=None, group=1):
        require_cuda(t, gidx)
        assert t.dim() == 3
        self.t, self.mode, self.gidx, self.group = t, mode, gidx, group


class X3Planes:
    """An activation f32[B,C,N] held ALREADY SPLIT for the bf16x3 kernel that contracts over its channels: u16[B,3,C/4,N,4], the three exact bf16
    terms of every value, four consecutive channels of one column per 8 bytes (di2p_epilogue_t.planes_out; include/deepi2p_hip.h).  Written
    by pointwise_gemm(..., planes_out=True), read by pointwise_gemm([X3Planes], ...); float() gives the fp32 values back (tests)."""

    def __init__(self, t, C, N):
        self.t, self.C, self.N = t, C, N
        self.shape = (t.shape[0], C, N)

    @staticmethod
    def ok(C, N):
        """Can a layer with C output rows and N columns hand its output on as planes (to a consumer with K = C)?"""
        return C % 32 == 0 and N % 128 == 0 and 3 * C * N * 2 < (1 << 31)

    def float(self):
        B = self.t.shape[0]
        w = self.t.view(torch.int16).view(B, 3, self.C // 4, self.N, 4).to(torch.int32) << 16
        f = w.view(torch.float32)
        return ((f[:, 2] + f[:, 1]) + f[:, 0]).permute(0, 1, 3, 2).reshape(B, self.C, self.N)


def _fill_srcs(srcs):
    arr = (SrcT * len(srcs))()
    for i, s in enumerate(srcs):
        arr[i].ptr = ptr(s.t)
        arr[i].gidx = ptr(s.gidx)
        arr[i].batch_stride = s.t.stride(0)
        arr[i].row_stride = s.t.stride(1)
        arr[i].channels = s.t.shape[1]
        arr[i].mode = s.mode
        arr[i].group = s.group
    return arr


def _fill_gathered(e, gathered, B, M, N):
    """Up to two gathered tables (table f32[B,nodes,M] node-major, idx i32[B,N,k_t], w f32[B,N,k_t] | None); each
    table carries its OWN k (opt.k_interp_point_a and k_interp_point_b may differ)."""
    e.g_k[0] = e.g_k[1] = 0
    if not gathered:
        return
    if len(gathered) > 2:
        raise RuntimeError("at most two gathered tables")
    for t, (tab, gi, gw) in enumerate(gathered):
        require_cuda(tab, gi, gw)
        _chk(tab, _f32, "gathered table")
        _chk(gi, _i32, "gathered index")
        if tab.dim() != 3 or tab.shape[0] != B or tab.shape[2] != M:
            raise RuntimeError("gathered table must be f32[B, nodes, M]")
        if gi.dim() != 3 or gi.shape[0] != B or gi.shape[1] != N:
            raise RuntimeError("gathered index must be i32[B, N, k]")
        k = gi.shape[2]
        if not 1 <= k <= MAX_GK:
            raise RuntimeError("gathered k must be in [1, %d]" % MAX_GK)
        if gw is not None:
            _chk(gw, _f32, "gathered weights")
            if gw.shape != gi.shape:
                raise RuntimeError("gathered weights must have the index's shape")
        e.g_table[t], e.g_idx[t], e.g_w[t] = ptr(tab), ptr(gi), ptr(gw)
        e.g_nodes[t] = tab.shape[1]
        e.g_k[t] = k


# bf16x3 operand cache: packed (split) weights of the GEMM-shaped layers, keyed by the fp32 operand they were made from.  An entry lives exactly
# as long as the fp32 operand's BASE tensor (a derived tensor owned by one networks._PackedModule's `_packed` container): a weak-reference
# finaliser drops it when that tensor is freed, so (i) a recycled address can never return a stale split, (ii) invalidating ONE module (its
# .to(), load_state_dict, an optimiser step) never touches another module's splits -- a holder of raw pointers (a captured hipGraph) keeps the
# module's packed operands referenced and thereby their splits (pipeline.RegistrationExecutor._packed_refs / _x3_refs).
# An entry is published only when it is COMPLETE: the packing stream is synchronised before the entry becomes visible, so another stream
# that finds it needs no ordering; under stream capture nothing can be waited for -- the pack becomes a node of THAT graph and is not
# cached (the executor runs one eager step before it captures, so this is the cold path only).
_X3_CACHE = {}            # (address, K, M, device) -> (weak reference to the fp32 operand's base tensor, split operand)
X3_MIN_K = 128


def x3_invalidate():
    """Drop every cached split (tests; never needed for correctness: entries die with the operand they were made from)."""
    _X3_CACHE.clear()


def x3_live_operands(owners=None):
    """The split operands currently cached (a holder of raw pointers to them keeps this list).  owners: packed-operand containers
    (networks._PackedModule.packed_operands()) -- only the splits whose fp32 operand lives in one of them, so that a holder pins the
    splits of ITS model and not those of every other model (or test) that happens to be alive in the process."""
    if owners is None:
        return [e[1] for e in _X3_CACHE.values()]
    mine = set()

    def walk(o):
        if isinstance(o, torch.Tensor):
            mine.add(id(o._base if o._base is not None else o))
        elif isinstance(o, dict):
            for v in o.values():
                walk(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
        elif hasattr(o, "__dict__"):
            for v in vars(o).values():
                walk(v)

    walk(owners)
    out = []
    for ref, wp in list(_X3_CACHE.values()):
        base = ref()
        if base is not None and id(base) in mine:
            out.append(wp)
    return out


def bf16x3_pack(Wt):
    """Wt f32[K,M] (contiguous) -> the split operand of di2p_pointwise_gemm_x3 (uint8 storage)."""
    require_cuda(Wt)
    K, M = Wt.shape
    Wp = torch.empty((_lib.load().di2p_bf16x3_packed_bytes(K, M),), dtype=torch.uint8, device=Wt.device)
    call("di2p_bf16x3_pack", ptr(Wt), K, M, ptr(Wp), stream())
    return Wp


def bf16x3_pack_conv3x3(W, dgrad=False):
    """The split operand of conv3x3_x3 straight from a filter bank W f32[Cout,Cin,3,3] (contiguous): the forward filter, or (dgrad=True) the
    filter whose convolution with dY is the layer's input gradient -- the same bytes as bf16x3_pack of the permuted (and flipped, transposed)
    matrix, in one launch."""
    require_cuda(W)
    _chk(W, _f32, "bf16x3_pack_conv3x3 filter")
    if W.dim() != 4 or W.shape[2:] != (3, 3) or not W.is_contiguous():
        raise RuntimeError("bf16x3_pack_conv3x3 needs a contiguous f32[Cout,Cin,3,3]")
    Cout, Cin = int(W.shape[0]), int(W.shape[1])
    K, M = (9 * Cout, Cin) if dgrad else (9 * Cin, Cout)
    Wp = torch.empty((_lib.load().di2p_bf16x3_packed_bytes(K, M),), dtype=torch.uint8, device=W.device)
    call("di2p_bf16x3_pack_conv3x3", ptr(W), Cout, Cin, int(bool(dgrad)), ptr(Wp), stream())
    return Wp


def _x3_auto(Wt, M, N):
    """The automatic rule of _x3_operand: does the layer with weights Wt [K, M] on N columns per frame run on the bf16x3 kernel?"""
    return (Wt.shape[0] >= X3_MIN_K and M % 128 == 0 and N % 4 == 0 and ((N + 127) // 128) * (M // 128) >= 8 and not Wt.requires_grad
            and _lib.get_option("pw_x3") != 0)


def _x3_operand(Wt, B, M, N, x3):
    """The packed operand if this layer runs on the bf16x3 kernel, else None.  x3: None = by shape and the library knob `pw_x3`; True / False force.
    Automatic rule: K >= 128, M % 128 == 0 and at least eight 128 x 128 workgroups PER FRAME -- a rule on the layer's shape only: a frame's
    result must not depend on the batch it is in (measured per layer of a 32-frame step,
    tools/call_times.py: 256x256x2048 135 -> 124 us, 512x256x2048 198 -> 139, 256x512x2048 211 -> 166, 1024x768x128 70 -> 55; the node-level
    layers with 128 workgroups or fewer are FASTER on the fp32 kernel's 64 x 64 tiles: 512x1024x128 48 vs 63 us, 128x512x128 19 vs 35)."""
    if x3 is None:
        x3 = _x3_auto(Wt, M, N)
    if not x3:
        return None
    K = Wt.shape[0]
    if x3 == "step":
        # training: the operand is this step's transposed copy of a parameter -- split it on the stream, use it once, cache nothing (no
        # host synchronisation: the split, the launch that reads it and the allocator's reuse of its memory are ordered by the stream)
        if M % 4 or N % 4 or not Wt.is_contiguous():
            raise RuntimeError("bf16x3 needs M % 4 == 0, N % 4 == 0 and a contiguous [K,M] weight")
        return bf16x3_pack(Wt)
    if M % 4 or N % 4 or not Wt.is_contiguous():
        raise RuntimeError("bf16x3 needs M % 4 == 0, N % 4 == 0 and a contiguous [K,M] weight")
    # an address is only an identity while the tensor that owns it is alive: the entry remembers (weakly) the base tensor it was made from --
    # a deleted module's operand whose memory the allocator hands to the next module must not find the old split
    base = Wt._base if Wt._base is not None else Wt
    key = (Wt.data_ptr(), K, M, Wt.device.index)
    ent = _X3_CACHE.get(key)
    if ent is not None and ent[0]() is base:
        return ent[1]
    Wp = bf16x3_pack(Wt)
    if torch.cuda.is_current_stream_capturing():
        return Wp                                    # a node of this graph only: never visible to another stream or graph
    torch.cuda.current_stream(Wt.device).synchronize()   # complete before it is published (once per layer and weights version)

    def _expired(ref, key=key):
        ent = _X3_CACHE.get(key)
        if ent is not None and ent[0] is ref:
            del _X3_CACHE[key]

    _X3_CACHE[key] = (weakref.ref(base, _expired), Wp)
    return Wp


def x3_planes_link(producer, consumer_Wt, B, N):
    """Should `producer` (a (Wt, scale, shift, act) layer) hand its [B, M, N] output to the layer with weights consumer_Wt [K = M, M2] as
    split planes?  Both must run on the bf16x3 kernel by the automatic rule, the shapes must fit X3Planes, and the knob `pw_x3_planes` be on."""
    Wt = producer[0]
    M = Wt.shape[1]
    if _lib.get_option("pw_x3_planes") == 0 or consumer_Wt.shape[0] != M or not X3Planes.ok(M, N):
        return False
    return all(_x3_auto(W, W.shape[1], N) and W.is_contiguous() for W in (Wt, consumer_Wt))


def pointwise_gemm(srcs, Wt, M, N, scale=None, shift=None, batch_bias=None, relu=False, group_max=1, gathered=None,
                   transpose_out=False, also_full=False, x3=None, planes_out=False):
    """Y[b] = epi(Wt^T @ concat(srcs)[b]).  srcs: list of Src; Wt f32[K,M].  gathered: optional list of up
    to two (table f32[B,nodes,M] node-major, idx i32[B,N,k], w f32[B,N,k]).  transpose_out: Y is f32[B,N,M].
    group_max > 1 returns the group maxima; with also_full=True it returns (full Y, maxima) from the same launch.
    x3: run the contraction on the bf16x3 kernel (exact three-way bf16 split of both operands, fp32 accumulation; None = for the GEMM-shaped
    layers, K >= 128 and M % 128 == 0, unless the knob `pw_x3` is 0; "step" = yes, with the weights split for this call only -- the
    training step, whose operand is a fresh transposed copy of a parameter every time).
    planes_out (bf16x3 layers only): the full-size output comes back as X3Planes -- already split for the bf16x3 layer that consumes it as
    `srcs=[planes]` (bit-identical to handing the fp32 output on; the split leaves the consumer's K loop)."""
    B = srcs[0].t.shape[0]
    K = Wt.shape[0]
    from_planes = any(isinstance(s, X3Planes) for s in srcs)
    if from_planes:
        if len(srcs) != 1 or srcs[0].C != K or srcs[0].N != N or srcs[0].t.device != Wt.device:
            raise RuntimeError("a split-planes source must be the only source, with K channels and N columns, on the weights' device")
        x3 = True
    if planes_out and (transpose_out or not X3Planes.ok(M, N)):
        raise RuntimeError("planes_out needs M % 32 == 0, N % 128 == 0 and no transpose_out")
    Wp = _x3_operand(Wt, B, M, N, x3)
    if planes_out and Wp is None:
        raise RuntimeError("planes_out: only the bf16x3 kernels write split planes")
    arr = None if from_planes else _fill_srcs(srcs)
    e = EpilogueT()
    e.scale, e.shift, e.batch_bias = ptr(scale), ptr(shift), ptr(batch_bias)
    e.relu, e.group_max = int(bool(relu)), int(group_max)
    _fill_gathered(e, gathered, B, M, N)
    e.transpose_out = int(bool(transpose_out))
    Nout = N // group_max if group_max > 1 else N
    Ymax = None
    if planes_out:
        # the full-size output as split planes; with group_max > 1 the maxima come out of the same launch (also_full or not: there is no
        # maxima-only planes output)
        if group_max > 1 and not also_full:
            raise RuntimeError("planes_out with group_max > 1 returns (planes, maxima): pass also_full=True")
        Y = X3Planes(torch.empty((_lib.load().di2p_bf16x3_planes_bytes(B, M, N),), dtype=torch.uint8, device=Wt.device).view(B, -1), M, N)
        e.planes_out = ptr(Y.t)
        if group_max > 1:
            Ymax = torch.empty((B, M, Nout), dtype=_f32, device=Wt.device)
            e.group_max_out = ptr(Ymax)
        y_ptr = None
    elif also_full and group_max > 1:
        Y = torch.empty((B, M, N), dtype=_f32, device=Wt.device)
        Ymax = torch.empty((B, M, Nout), dtype=_f32, device=Wt.device)
        e.group_max_out = ptr(Ymax)
        y_ptr = ptr(Y)
    else:
        Y = torch.empty((B, Nout, M) if transpose_out else (B, M, Nout), dtype=_f32, device=Wt.device)
        y_ptr = ptr(Y)
    if Wp is not None:
        if _lib.WORK is not None:
            _lib.WORK["di2p_pointwise_gemm_x3"] = _lib.WORK.get("di2p_pointwise_gemm_x3", 0) + B * M * K * N
        if from_planes:
            call("di2p_pointwise_gemm_x3p", ptr(srcs[0].t), ptr(Wp), y_ptr, B, M, K, N, ctypes.byref(e), stream())
        else:
            call("di2p_pointwise_gemm_x3", arr, len(srcs), ptr(Wp), y_ptr, B, M, K, N, ctypes.byref(e), stream())
        return (Y, Ymax) if Ymax is not None else Y
    if _lib.WORK is not None:
        _lib.WORK["di2p_pointwise_gemm"] = _lib.WORK.get("di2p_pointwise_gemm", 0) + B * M * K * N
    call("di2p_pointwise_gemm", arr, len(srcs), ptr(Wt), ptr(Y), B, M, K, N, ctypes.byref(e), stream())
    return (Y, Ymax) if Ymax is not None else Y


def point_head(srcs, layer0, layer1, layer2, N, batch_bias=None, gathered=None):
    """Fused three-layer per-point head (di2p_point_head): layers are (Wt[K,M], scale, shift, relu) tuples with hidden
    width 128 and at most 4 outputs; layer0's Wt holds only the rows of the dense `srcs`.  -> f32[B, P, N]."""
    Wt0, sc0, sh0, act0 = layer0
    Wt1, sc1, sh1, act1 = layer1
    Wt2, sc2, sh2, act2 = layer2
    B = srcs[0].t.shape[0]
    M, P = Wt0.shape[1], Wt2.shape[1]
    require_cuda(Wt0, Wt1, Wt2, sc0, sh0, sc1, sh1, sc2, sh2, batch_bias)
    e = EpilogueT()
    e.scale, e.shift, e.batch_bias = ptr(sc0), ptr(sh0), ptr(batch_bias)
    e.relu, e.group_max, e.transpose_out = int(bool(act0)), 1, 0
    _fill_gathered(e, gathered, B, M, N)
    out = torch.empty((B, P, N), dtype=_f32, device=Wt0.device)
    if _lib.WORK is not None:
        _lib.WORK["di2p_point_head"] = _lib.WORK.get("di2p_point_head", 0) + B * N * (Wt0.shape[0] * M + M * M + M * P)
    call("di2p_point_head", _fill_srcs(srcs), len(srcs), ptr(Wt0), Wt0.shape[0], ctypes.byref(e), ptr(Wt1), ptr(sc1), ptr(sh1),
         int(bool(act1)), ptr(Wt2), ptr(sc2), ptr(sh2), int(bool(act2)), ptr(out), B, M, P, N, stream())
    return out


def head_x3_pack(Wt):
    """Wt f32[K,128] (contiguous, K % 16 == 0) -> the fragment-ordered split operand of di2p_point_head_x3 (uint8 storage)."""
    require_cuda(Wt)
    K, M = Wt.shape
    if M != 128 or K % 16 or not Wt.is_contiguous():
        raise RuntimeError("head_x3_pack needs a contiguous f32[K,128] with K % 16 == 0")
    Wp = torch.empty((_lib.load().di2p_head_x3_packed_bytes(K),), dtype=torch.uint8, device=Wt.device)
    call("di2p_head_x3_pack", ptr(Wt), K, ptr(Wp), stream())
    return Wp


def point_head_x3(src0, src1, packed, gathered, N):
    """The coarse per-point head in one launch on the bf16 matrix instructions (di2p_point_head_x3).  src0 / src1: dense f32[B,ch,N];
    packed: dict W0p, W1p (head_x3_pack), ss f32[4,128] (scale0, shift0, scale1, shift1), relu0, relu1, W2t f32[128,P], sc2, sh2, relu2;
    gathered: [(table f32[B,nodes,128], idx i32[B,N,3], w f32[B,N,3] | None)] x 2.  -> f32[B,P,N]."""
    require_cuda(src0, src1, packed["W0p"], packed["W1p"], packed["ss"], packed["W2t"], packed["sc2"], packed["sh2"])
    B = src0.shape[0]
    P = packed["W2t"].shape[1]
    # raw pointers below: the packs must be the ones made for THIS head (layer 0's dense rows, layer 1), the small operands fp32 of the right size
    k0 = int(src0.shape[1]) + int(src1.shape[1])
    hb = _lib.load().di2p_head_x3_packed_bytes
    if k0 % 16 or packed["W0p"].numel() * packed["W0p"].element_size() != hb(k0) or packed["W1p"].numel() * packed["W1p"].element_size() != hb(128):
        raise RuntimeError("point_head_x3: W0p / W1p must be head_x3_pack of the [%d,128] dense rows and of the [128,128] second layer" % k0)
    for name, n in (("ss", 4 * 128), ("W2t", 128 * P), ("sc2", P), ("sh2", P)):
        if packed[name] is None and name in ("sc2", "sh2"):          # the output layer may come without scale / shift
            continue
        _chk(packed[name], _f32, "point_head_x3 " + name)
        if packed[name].numel() < n:
            raise RuntimeError("point_head_x3: %s holds %d values, needs %d" % (name, packed[name].numel(), n))
    if tuple(packed["W2t"].shape) != (128, P) or not 1 <= P <= 4:
        raise RuntimeError("point_head_x3: W2t must be f32[128,P] with P <= 4")
    h = _lib.HeadX3T()
    for t, x in enumerate((src0, src1)):
        _chk(x, _f32, "head source")
        if x.dim() != 3 or x.shape[2] != N or x.stride(2) != 1:
            raise RuntimeError("head sources must be [B, ch, N] with unit column stride")
        h.src[t], h.batch_stride[t], h.row_stride[t], h.channels[t] = ptr(x), x.stride(0), x.stride(1), x.shape[1]
    for t, (tab, gi, gw) in enumerate(gathered):
        require_cuda(tab, gi, gw)
        _chk(tab, _f32, "gathered table")
        _chk(gi, _i32, "gathered index")
        if tuple(tab.shape[0::2]) != (B, 128) or tuple(gi.shape) != (B, N, 3) or (gw is not None and tuple(gw.shape) != (B, N, 3)):
            raise RuntimeError("the head takes node tables [B,nodes,128] with three neighbours per point")
        h.tab[t], h.idx[t], h.w[t], h.nodes[t] = ptr(tab), ptr(gi), ptr(gw), tab.shape[1]
    h.W0p, h.W1p, h.scale_shift = ptr(packed["W0p"]), ptr(packed["W1p"]), ptr(packed["ss"])
    h.relu0, h.relu1, h.relu2, h.P = int(bool(packed["relu0"])), int(bool(packed["relu1"])), int(bool(packed["relu2"])), P
    h.W2t, h.scale2, h.shift2 = ptr(packed["W2t"]), ptr(packed["sc2"]), ptr(packed["sh2"])
    out = torch.empty((B, P, N), dtype=_f32, device=src0.device)
    if _lib.WORK is not None:
        _lib.WORK["di2p_point_head_x3"] = _lib.WORK.get("di2p_point_head_x3", 0) + B * N * ((src0.shape[1] + src1.shape[1]) * 128 + 128 * 128 + 128 * P)
    call("di2p_point_head_x3", ctypes.byref(h), ptr(out), B, N, stream())
    return out


def point_chain_ok(srcs, layers, N):
    """Can di2p_point_chain run these layers ((Wt, scale, shift, relu) tuples; layers[0].Wt holds the rows of the dense `srcs`)?"""
    if len(layers) not in (2, 3) or len(srcs) != 1 or _lib.get_option("pw_nochain"):
        return False
    M = layers[0][0].shape[1]
    if M not in (32, 64) or layers[0][0].shape[0] > M or any(tuple(l[0].shape) != (M, M) for l in layers[1:]):
        return False
    if any(l[0].requires_grad or not l[0].is_contiguous() for l in layers):
        return False
    return srcs[0].mode == _lib.SRC_DENSE and srcs[0].t.stride(2) == 1


def point_chain(srcs, layers, N, batch_bias=None, gathered=None):
    """Fused narrow PointNet chain (di2p_point_chain): two or three (Wt[K,M], scale, shift, relu) layers of one width
    M in {32, 64}; layers[0]'s Wt holds only the rows of the dense `srcs`.  -> f32[B, M, N], bit-identical to the
    separate pointwise_gemm calls."""
    Wt0, sc0, sh0, act0 = layers[0]
    Wt1, sc1, sh1, act1 = layers[1]
    Wt2, sc2, sh2, act2 = layers[2] if len(layers) == 3 else (None, None, None, False)
    B = srcs[0].t.shape[0]
    M = Wt0.shape[1]
    require_cuda(Wt0, Wt1, Wt2, sc0, sh0, sc1, sh1, sc2, sh2, batch_bias)
    e = EpilogueT()
    e.scale, e.shift, e.batch_bias = ptr(sc0), ptr(sh0), ptr(batch_bias)
    e.relu, e.group_max, e.transpose_out = int(bool(act0)), 1, 0
    _fill_gathered(e, gathered, B, M, N)
    out = torch.empty((B, M, N), dtype=_f32, device=Wt0.device)
    if _lib.WORK is not None:
        _lib.WORK["di2p_point_chain"] = _lib.WORK.get("di2p_point_chain", 0) + B * N * (Wt0.shape[0] * M + (len(layers) - 1) * M * M)
    call("di2p_point_chain", _fill_srcs(srcs), len(srcs), ptr(Wt0), Wt0.shape[0], ctypes.byref(e), ptr(Wt1), ptr(sc1), ptr(sh1),
         int(bool(act1)), ptr(Wt2), ptr(sc2), ptr(sh2), int(bool(act2)), ptr(out), B, M, N, stream())
    return out


def batch_gemv(Wt, k0, v):
    """out[b,m] = sum_k Wt[k0+k, m] v[b,k]."""
    require_cuda(Wt, v)
    B, Kv = v.shape
    M = Wt.shape[1]
    out = torch.empty((B, M), dtype=_f32, device=Wt.device)
    call("di2p_batch_gemv", ptr(Wt), M, int(k0), ptr(v), Kv, ptr(out), B, stream())
    return out


def batch_gemv2(Wt, k0, v0, k1, v1):
    """out[b,m] = sum_k Wt[k0+k, m] v0[b,k] + sum_k Wt[k1+k, m] v1[b,k]  (one launch)."""
    require_cuda(Wt, v0, v1)
    B = v0.shape[0]
    M = Wt.shape[1]
    out = torch.empty((B, M), dtype=_f32, device=Wt.device)
    call("di2p_batch_gemv2", ptr(Wt), M, int(k0), ptr(v0), v0.shape[1], int(k1), ptr(v1), v1.shape[1], ptr(out), B, stream())
    return out


def attention_pool(feat, score):
    """feat f32[B,C,HW], score f32[B,HW,Mn] -> f32[B,C,Mn] = feat@score / HW."""
    require_cuda(feat, score)
    B, C, HW = feat.shape
    Mn = score.shape[2]
    out = torch.empty((B, C, Mn), dtype=_f32, device=feat.device)
    call("di2p_attention_pool", ptr(feat), ptr(score), ptr(out), B, C, HW, Mn, stream())
    return out


def conv2d(x, Wt, scale, shift, KH, KW, stride, pad, relu, residual=None, tap_major=False):
    require_cuda(x, Wt, scale, shift, residual)
    B, Cin, H, W = x.shape
    Cout = Wt.shape[1]
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - KW) // stride + 1
    y = torch.empty((B, Cout, OH, OW), dtype=_f32, device=x.device)
    args = (ptr(x), ptr(Wt), ptr(scale), ptr(shift), ptr(residual), ptr(y), B, Cin, H, W, Cout, KH, KW,
            stride, pad, int(bool(relu)), int(bool(tap_major)))
    nws = _lib.load().di2p_conv2d_workspace_bytes(B, Cin, H, W, Cout, KH, KW, stride, pad, int(bool(tap_major)))
    if nws > 0:       # split-K scratch (stream-ordered: the caching allocator hands it out per stream)
        ws = torch.empty((nws,), dtype=torch.uint8, device=x.device)
        call("di2p_conv2d_ws", *args, ptr(ws), nws, stream())
    else:
        call("di2p_conv2d", *args, stream())
    return y


def conv3x3_x3_supported(x_shape, Cout, stride):
    """True if di2p_conv3x3_x3 has a kernel instance for this layer shape (x_shape = (B, Cin, H, W))."""
    B, Cin, H, W = (int(v) for v in x_shape)
    return bool(_lib.load().di2p_conv3x3_x3_supported(B, Cin, H, W, int(Cout), int(stride)))


def conv3x3_x3(x, Wp, Cout, scale, shift, stride, relu, residual=None, downsample=None):
    """3x3 / pad 1 convolution on the bf16 matrix instructions with exact three-way fp32 splits (di2p_conv3x3_x3):
    y = relu?(scale * conv(x) + shift + residual).  Wp = bf16x3_pack(tap-major Wt[9 Cin, Cout]).  stride 2 takes
    downsample = (Wp_ds, scale_ds, shift_ds) -- the BasicBlock's 1x1 / stride-2 branch of the same input (resnet.py:160-164) -- and
    returns (y, y_ds)."""
    require_cuda(x, Wp, scale, shift, residual)
    _chk(x, _f32, "conv3x3_x3 input")
    if x.dim() != 4:
        raise RuntimeError("conv3x3_x3 takes x f32[B,Cin,H,W]")
    B, Cin, H, W = x.shape
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    # the kernel reads the operands through raw pointers: a pack made for another layer shape or a smaller residual would be read out of bounds
    packed_bytes = _lib.load().di2p_bf16x3_packed_bytes
    if Wp.numel() * Wp.element_size() != packed_bytes(9 * Cin, int(Cout)):
        raise RuntimeError("conv3x3_x3: Wp is not bf16x3_pack of a [9*%d, %d] matrix (%d bytes, expected %d)" % (Cin, Cout, Wp.numel() * Wp.element_size(), packed_bytes(9 * Cin, int(Cout))))
    for name, v in (("scale", scale), ("shift", shift)):
        _chk(v, _f32, "conv3x3_x3 " + name)
        if v.numel() < Cout:
            raise RuntimeError("conv3x3_x3: %s holds %d values, needs %d" % (name, v.numel(), Cout))
    if residual is not None:
        _chk(residual, _f32, "conv3x3_x3 residual")
        if tuple(residual.shape) != (B, Cout, OH, OW):
            raise RuntimeError("conv3x3_x3: residual must be f32%s, got %s" % ((B, Cout, OH, OW), tuple(residual.shape)))
    y = torch.empty((B, Cout, OH, OW), dtype=_f32, device=x.device)
    Wd = sd = hd = yd = None
    if downsample is not None:
        Wd, sd, hd = downsample
        require_cuda(Wd, sd, hd)
        if Wd.numel() * Wd.element_size() != packed_bytes(Cin, int(Cout)):
            raise RuntimeError("conv3x3_x3: the downsample pack is not bf16x3_pack of a [%d, %d] matrix" % (Cin, Cout))
        _chk(sd, _f32, "conv3x3_x3 downsample scale"); _chk(hd, _f32, "conv3x3_x3 downsample shift")
        if sd.numel() < Cout or hd.numel() < Cout:
            raise RuntimeError("conv3x3_x3: downsample scale / shift need %d values" % Cout)
        yd = torch.empty((B, Cout, OH, OW), dtype=_f32, device=x.device)
    if _lib.WORK is not None:
        _lib.WORK["di2p_conv3x3_x3"] = _lib.WORK.get("di2p_conv3x3_x3", 0) + B * Cout * Cin * (9 + (1 if downsample is not None else 0)) * OH * OW
    call("di2p_conv3x3_x3", ptr(x), ptr(Wp), ptr(scale), ptr(shift), ptr(residual), ptr(y), B, Cin, H, W, Cout, int(stride), int(bool(relu)),
         ptr(Wd), ptr(sd), ptr(hd), ptr(yd), stream())
    return (y, yd) if downsample is not None else y


def stem_weights(weight):
    """weight f32[64,3,7,7] -> Wp f32[168,64]: the packed filter bank of conv_stem."""
    require_cuda(weight)
    if tuple(weight.shape) != (64, 3, 7, 7) or weight.dtype != _f32:
        raise RuntimeError("stem_weights needs the f32[64,3,7,7] ResNet stem filter")
    w = weight.detach().contiguous()
    Wp = torch.empty((168, 64), dtype=_f32, device=w.device)
    call("di2p_stem_pack", ptr(w), ptr(Wp), stream())
    return Wp


def conv_stem(x, Wp, scale, shift, relu=True):
    """ResNet stem: y = relu?(scale * conv7x7/2(x) + shift); x f32[B,3,H,W] -> f32[B,64,OH,OW]."""
    require_cuda(x, Wp, scale, shift)
    B, C, H, W = x.shape
    if C != 3:
        raise RuntimeError("the stem takes 3 input channels")
    y = torch.empty((B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=_f32, device=x.device)
    call("di2p_conv7x7s2_stem", ptr(x), ptr(Wp), ptr(scale), ptr(shift), ptr(y), B, H, W, int(bool(relu)), stream())
    return y


def stem_x3_weights(weight):
    """weight f32[64,3,7,7] -> the split, fragment-ordered operand of stem_x3 (uint8 storage)."""
    require_cuda(weight)
    if tuple(weight.shape) != (64, 3, 7, 7) or weight.dtype != _f32:
        raise RuntimeError("stem_x3_weights needs the f32[64,3,7,7] ResNet stem filter")
    w = weight.detach().contiguous()
    Wp = torch.empty((_lib.load().di2p_stem_x3_packed_bytes(),), dtype=torch.uint8, device=w.device)
    call("di2p_stem_x3_pack", ptr(w), ptr(Wp), stream())
    return Wp


def stem_x3_supported(H, W):
    return bool(_lib.load().di2p_stem_x3_supported(int(H), int(W)))


def stem_x3(x, Wp, scale, shift):
    """conv1 + bn1 + relu + maxpool of the image branch in one launch (di2p_stem_x3): x f32[B,3,H,W] -> f32[B,64,H/4,W/4]."""
    require_cuda(x, Wp, scale, shift)
    _chk(x, _f32, "stem_x3 input"); _chk(scale, _f32, "stem_x3 scale"); _chk(shift, _f32, "stem_x3 shift")
    if x.dim() != 4:
        raise RuntimeError("stem_x3 takes x f32[B,3,H,W]")
    B, C, H, W = x.shape
    if C != 3:
        raise RuntimeError("the stem takes 3 input channels")
    if Wp.numel() * Wp.element_size() != _lib.load().di2p_stem_x3_packed_bytes() or scale.numel() < 64 or shift.numel() < 64:
        raise RuntimeError("stem_x3: Wp must be stem_x3_weights(...) and scale / shift hold 64 values")
    y = torch.empty((B, 64, H // 4, W // 4), dtype=_f32, device=x.device)
    if _lib.WORK is not None:
        _lib.WORK["di2p_stem_x3"] = _lib.WORK.get("di2p_stem_x3", 0) + B * 64 * 147 * (H // 2) * (W // 2)
    call("di2p_stem_x3", ptr(x), ptr(Wp), ptr(scale), ptr(shift), ptr(y), B, H, W, stream())
    return y


def winograd_weights(weight):
    """weight f32[Cout,Cin,3,3] -> U f32[16,Cin,Cout] = G g G^T per (ci, co): the operand of conv3x3_winograd."""
    Cout, Cin, KH, KW = weight.shape
    if (KH, KW) != (3, 3) or weight.dtype != _f32:
        raise RuntimeError("winograd_weights needs an f32 3x3 filter bank")
    w = weight.detach().contiguous()          # (a flipped / transposed view is fine: the training path passes one)
    require_cuda(w)
    U = torch.empty((16, Cin, Cout), dtype=_f32, device=w.device)
    call("di2p_winograd_weight_transform", ptr(w), ptr(U), Cin, Cout, stream())
    return U


def winograd_weights_dgrad(weight):
    """weight f32[Cout,Cin,3,3] of a stride-1 3x3 layer -> U f32[16,Cout,Cin]: the transformed filter of the layer's INPUT GRADIENT (the
    convolution of dY with the flipped, channel-transposed filter), in one launch."""
    Cout, Cin, KH, KW = weight.shape
    if (KH, KW) != (3, 3) or weight.dtype != _f32:
        raise RuntimeError("winograd_weights_dgrad needs an f32 3x3 filter bank")
    w = weight.detach().contiguous()
    require_cuda(w)
    U = torch.empty((16, Cout, Cin), dtype=_f32, device=w.device)
    call("di2p_winograd_weight_transform_dgrad", ptr(w), ptr(U), Cout, Cin, stream())
    return U


def conv3x3_winograd(x, U, scale, shift, relu, residual=None):
    """3x3 / stride 1 / pad 1 convolution via Winograd F(2x2,3x3): y = relu?(scale * conv(x) + shift + residual)."""
    require_cuda(x, U, scale, shift, residual)
    B, Cin, H, W = x.shape
    Cout = U.shape[2]
    if U.shape[0] != 16 or U.shape[1] != Cin:
        raise RuntimeError("U must be f32[16, Cin, Cout]")
    y = torch.empty((B, Cout, H, W), dtype=_f32, device=x.device)
    if _lib.WORK is not None:
        _lib.WORK["di2p_conv3x3_winograd"] = _lib.WORK.get("di2p_conv3x3_winograd", 0) + B * Cout * Cin * 16 * ((H + 1) // 2) * ((W + 1) // 2)
    call("di2p_conv3x3_winograd", ptr(x), ptr(U), ptr(scale), ptr(shift), ptr(residual), ptr(y), B, Cin, H, W, Cout, int(bool(relu)), stream())
    return y


def maxpool3x3s2(x):
    require_cuda(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=_f32, device=x.device)
    call("di2p_maxpool3x3s2", ptr(x), ptr(y), B, C, H, W, stream())
    return y


def global_avgpool(x):
    require_cuda(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, 1, 1), dtype=_f32, device=x.device)
    call("di2p_global_avgpool", ptr(x), ptr(y), B, C, H * W, stream())
    return y


# ------------------------------------------------------------------------------------ solver
def _dbl3(v):
    a = (ctypes.c_double * 3)()
    for i in range(3):
        a[i] = float(v[i])      # IndexError on short lists, like the reference's .at(i)
    return a


def initial_guess(points64, labels):
    require_cuda(points64, labels)
    F, _, N = points64.shape
    yaw0 = torch.empty((F,), dtype=_f64, device=points64.device)
    labels_out = torch.empty_like(labels)
    has_inside = torch.empty((F,), dtype=_i32, device=points64.device)
    call("di2p_initial_guess", ptr(points64), ptr(labels), ptr(yaw0), ptr(labels_out), ptr(has_inside), F, N, stream())
    return yaw0, labels_out, has_inside


def solve_batched(points, labels, K, init_y, init_T, H, W, lb, ub, max_iter, is_2d, yaw0=None, sweeps=None):
    """points f64|f32 [F,3,N], labels i32[F,N], K f64[F,3,3], init_y f64[F,R], init_T f64[F,R,3]
    -> params f64[F,R,np], cost f64[F,R], iters i32[F,R]."""
    require_cuda(points, labels, K, init_y, init_T, yaw0)
    _chk(labels, _i32, "labels")
    _chk(K, _f64, "K")
    _chk(init_y, _f64, "init_y")
    _chk(init_T, _f64, "init_T")
    F, _, N = points.shape
    R = init_y.shape[1]
    npar = 4 if is_2d else 6
    params = torch.empty((F, R, npar), dtype=_f64, device=points.device)
    cost = torch.empty((F, R), dtype=_f64, device=points.device)
    iters = torch.empty((F, R), dtype=_i32, device=points.device)
    name = "di2p_solve_batched" if points.dtype == _f64 else "di2p_solve_batched_f32"
    if points.dtype not in (_f64, _f32):
        raise RuntimeError("points must be float64 or float32")
    ws = torch.empty((_lib.load().di2p_solve_workspace_bytes(F, R, N),), dtype=torch.uint8, device=points.device)
    call(name, ptr(points), ptr(labels), ptr(K), ptr(init_y), ptr(init_T), ptr(yaw0), float(H), float(W), _dbl3(lb),
         _dbl3(ub), int(max_iter), int(bool(is_2d)), F, R, N, ptr(params), ptr(cost), ptr(iters), ptr(sweeps), ptr(ws), stream())
    return params, cost, iters


def select_best(params, cost, is_2d, has_inside=None):
    require_cuda(params, cost, has_inside)
    F, R = cost.shape
    best = torch.empty((F,), dtype=_i32, device=cost.device)
    P = torch.empty((F, 4, 4), dtype=_f64, device=cost.device)
    bc = torch.empty((F,), dtype=_f64, device=cost.device)
    call("di2p_select_best", ptr(params), ptr(cost), ptr(has_inside), int(bool(is_2d)), F, R, ptr(best), ptr(P), ptr(bc), stream())
    return best, P, bc


def solver_residuals(points64, labels, K, params, H, W, is_2d):
    require_cuda(points64, labels, K, params)
    F, _, N = points64.shape
    res = torch.empty((F, 3 * N), dtype=_f64, device=points64.device)
    counts = torch.empty((F,), dtype=_i32, device=points64.device)
    cost = torch.empty((F,), dtype=_f64, device=points64.device)
    call("di2p_solver_residuals", ptr(points64), ptr(labels), ptr(K), ptr(params), float(H), float(W), int(bool(is_2d)),
         F, N, ptr(res), ptr(counts), ptr(cost), stream())
    return res, counts, cost
