"""Train-mode forward and backward of the classifier on the HIP kernels.

The reference trains with ``self.detector.train(); loss.backward(); optimizer.step()``
(models/multimodal_classifier.py:213-218): BatchNorm uses batch statistics and updates its running
buffers, per_point_pn applies Dropout(0.5) after its first two layers (networks_united.py:57-74,
layers_pc.py:300-303,339-340), and torch.autograd derives the backward of every op.  Here every op
that touches a feature tensor is a torch.autograd.Function whose forward AND backward are kernels of
libdeepi2p_hip.so; autograd is only the tape that orders them (plus views / concatenations / the
broadcast of the two global feature vectors, which carry no arithmetic of their own).

    op (reference)                                   forward kernel                backward kernels
    nn.Conv1d / MyConv2d 1x1 (layers_pc.py:110-342)  di2p_pointwise_gemm           di2p_pointwise_gemm (dX), di2p_bmm_rc (dW), di2p_channel_sum (db)
    nn.BatchNorm1d/2d, train (+ReLU, +residual)      di2p_bn_train_forward         di2p_bn_train_backward
    nn.Conv2d (resnet.py)                            di2p_conv2d                   di2p_conv2d_dgrad, di2p_conv2d_wgrad
    max_pool2d 3x3/2, adaptive_avg_pool2d            di2p_maxpool3x3s2, _avgpool   di2p_maxpool3x3s2_backward, broadcast
    index_max + gather + mask (networks_pc.py:88-104) di2p_index_max_values        di2p_segment_max_backward
    torch.gather along the point axis                di2p_gather_points            di2p_gather_backward (k = 1)
    torch.max over neighbours / nodes                di2p_group_max_forward        di2p_group_max_backward
    upsample_by_interpolation (networks_united.py)   di2p_interpolate              di2p_gather_backward (k = 3, weights)
    attention map mean (networks_united.py:139-150)  di2p_attention_pool           di2p_bmm_rc (d feat), di2p_bmm_km (d score)
    nn.Dropout                                       di2p_dropout_mask/_apply_mask di2p_apply_mask

Parameters are addressed by the reference's state_dict keys (a dict name -> tensor: the
``named_parameters`` / ``named_buffers`` of deepi2p_amd.networks.KeypointDetector, or any dict with
the same keys).  tests/test_gpu_training.py compares loss and every gradient with torch.autograd on
oracle/network_torch.py in train mode, and with the imported reference's own gradients through the
committed fixture.
"""
import torch
from torch.autograd import Function

from . import _lib, ops, prep
from ._lib import call, ptr, stream
from .ops import Src

BN_EPS = 1e-5
_f32, _i32 = torch.float32, torch.int32


def _sink(param):
    """Where a Function's backward writes the gradient of `param` DIRECTLY (ClassifierTrainer points every parameter at its slice
    of the flat gradient buffer): the Function then returns None for it and autograd launches no accumulation kernel."""
    return getattr(param, "_di2p_grad", None)


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 16),), dtype=torch.uint8, device=device)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------ contractions
def _x3_step(K, M, N):
    """Forward and input-gradient contractions of the big point layers on the bf16x3 kernel (exact three-way splits: as accurate as the
    fp32-MFMA kernel, tests/test_gpu_contractions.py; the weights are split per call, ops.pointwise_gemm x3="step"): the layers with at
    least 128 rows on both sides of the contraction and 2048 columns per frame -- the per-point head and the kNN-fusion layers, 2.2 of the
    3.3 ms a step spends in di2p_pointwise_gemm.  Knob `pw_x3` = 0: fp32-MFMA kernels everywhere (rounds 3-5)."""
    return "step" if (K >= 128 and M >= 128 and M % 4 == 0 and N % 4 == 0 and N >= 2048 and _lib.get_option("pw_x3") != 0) else False


class _Linear(Function):
    """y[b,m,n] = sum_k W[m,k] x[b,k,n] + bias[m]; x f32[B,K,N]."""

    @staticmethod
    def forward(ctx, x, W, bias):
        x = _c(x)
        B, K, N = x.shape
        M = W.shape[0]
        W2 = W.reshape(M, K)
        y = ops.pointwise_gemm([Src(x)], W2.t().contiguous(), M, N, shift=bias, x3=_x3_step(K, M, N))
        ctx.save_for_backward(x, W2)
        ctx.has_bias = bias is not None
        ctx.wshape = W.shape
        ctx.sinks = (_sink(W), _sink(bias) if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W2 = ctx.saved_tensors
        dy = _c(dy)
        B, K, N = x.shape
        M = W2.shape[0]
        lib = _lib.load()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.pointwise_gemm([Src(dy)], _c(W2), K, N, x3=_x3_step(M, K, N))          # W[m][k] is the k-major operand of the reduction over m
        if ctx.needs_input_grad[1]:
            dW = ctx.sinks[0] if ctx.sinks[0] is not None else torch.empty((M, K), dtype=_f32, device=x.device)
            nb = lib.di2p_bmm_rc_workspace_bytes(B, M, K, N)
            ws = _ws(nb, x.device)
            call("di2p_bmm_rc", ptr(dy), N, M * N, ptr(x), N, K * N, ptr(dW), B, M, K, N, 1.0, 1, ptr(ws), nb, stream())
            dW = None if ctx.sinks[0] is not None else dW.view(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ctx.sinks[1] if ctx.sinks[1] is not None else torch.empty((M,), dtype=_f32, device=x.device)
            ws = _ws(lib.di2p_channel_reduce_workspace_bytes(B, M, N), x.device)
            call("di2p_channel_sum", ptr(dy), ptr(db), B, M, N, ptr(ws), stream())
            db = None if ctx.sinks[1] is not None else db
        return dx, dW, db


def linear(x, W, bias):
    """1x1 convolution over any trailing shape: x f32[B,K,*] -> f32[B,M,*]."""
    shp = x.shape
    y = _Linear.apply(x.reshape(shp[0], shp[1], -1), W, bias)
    return y.view(shp[0], W.shape[0], *shp[2:])


_UNIT = {}


def _unit_affine(C, device):
    """(ones[C], zeros[C]): the identity epilogue of di2p_conv2d (the BatchNorm after it runs on batch statistics, not folded)."""
    key = (C, str(device))
    if key not in _UNIT:
        _UNIT[key] = (torch.ones((C,), dtype=_f32, device=device), torch.zeros((C,), dtype=_f32, device=device))
    return _UNIT[key]


def _use_conv_x3(xshape, Cout):
    """Does the stride-1 3 x 3 layer x[B, Cin, H, W] -> Cout run on di2p_conv3x3_x3 in the training step?  From 16 frames on: every stage the knob
    `conv_x3` names.  Below (the reference's training batch of 8): only the 256- and 512-channel stages -- the kernel's workgroup tiles are
    sized for whole rounds of the chip at 32 frames; at 8 frames the Winograd kernel's smaller tiles win on the two large-image stages
    (tools/bench_conv_x3.py, B=8: 30.8 / 38.7 us against 29.2 / 42.7) and lose on the small-image ones (48.2 / 73.2 against 29.2 / 46.0 us with
    the smallest tile configuration, which _conv_forward then forces)."""
    if not (_lib.get_option("conv_x3") & (1 << max(0, min(3, Cout.bit_length() - 7)))) or not ops.conv3x3_x3_supported(xshape, Cout, 1):
        return False
    return xshape[0] >= 16 or Cout >= 256


def _conv3x3_x3(x, W, dgrad):
    """The stride-1 3 x 3 layer with filter bank W[Cout, Cin, 3, 3] on di2p_conv3x3_x3 (exact three-way splits; the filters change every step: their
    split is ONE small launch per call, straight from W): dgrad False: y = conv(x, W); True: the input gradient = conv(x = dY, W flipped and
    channel-transposed) -- the filter the pack kernel builds itself."""
    Cdst = W.shape[1] if dgrad else W.shape[0]
    one, zero = _unit_affine(Cdst, x.device)
    Wp = ops.bf16x3_pack_conv3x3(_c(W), dgrad=dgrad)
    small = x.shape[0] < 16 and _lib.get_option("conv_x3_cfg") < 0
    # (the knob is process-wide and set for the duration of ONE launch call: a training step and an inference executor must not run on two
    #  threads of one process at the same time -- the reference trains and evaluates in turn, kitti/train_classifier.py:54-154)
    if small:        # few frames: the smallest tile configuration fills more of the chip (the library prices its choice for 32 frames)
        _lib.set_option("conv_x3_cfg", 3)
        if not ops.conv3x3_x3_supported(x.shape, Cdst, 1):        # ... where it runs the shape at all
            _lib.set_option("conv_x3_cfg", -1)
            small = False
    try:
        return ops.conv3x3_x3(x, Wp, Cdst, one, zero, 1, False)
    finally:
        if small:
            _lib.set_option("conv_x3_cfg", -1)


def _conv_forward(x, W, stride, pad):
    """conv2d through the inference engine with an identity epilogue; Cin % 16 == 0 takes its tap-major (16-byte staged) path."""
    Cout, Cin, KH, KW = W.shape
    one, zero = _unit_affine(Cout, x.device)
    if (KH, KW, stride, pad) == (3, 3, 1, 1) and Cin % 16 == 0 and _use_conv_x3(x.shape, Cout):
        return _conv3x3_x3(x, W, False)
    if (KH, KW, stride, pad) == (3, 3, 1, 1) and Cin % 16 == 0 and Cout % 32 == 0 and x.shape[3] % 2 == 0 and x.shape[3] >= 4:
        # the fused Winograd kernel (the filters change every step: their transform is one small launch per call)
        return ops.conv3x3_winograd(x, ops.winograd_weights(W), one, zero, False)
    if Cin % 16 == 0:
        return ops.conv2d(x, W.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous(), one, zero, KH, KW, stride, pad, False, tap_major=True)
    return ops.conv2d(x, W.reshape(Cout, -1).t().contiguous(), one, zero, KH, KW, stride, pad, False)


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, W, stride, pad):
        x = _c(x)
        Cout, Cin, KH, KW = W.shape
        y = _conv_forward(x, W, stride, pad)
        ctx.save_for_backward(x, W)
        ctx.cfg = (stride, pad)
        ctx.sink = _sink(W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = _c(dy)
        B, Cin, H, Wd = x.shape
        Cout, _, KH, KW = W.shape
        lib = _lib.load()
        dx = dW = None
        if ctx.needs_input_grad[0]:
            if stride == 1 and KH == KW and 2 * pad == KH - 1:
                # a stride-1 "same" convolution's input gradient is the convolution of dY with the flipped, channel-transposed filter:
                # the forward engine does it (the strided layers go through the generic gather-form kernel)
                one, zero = _unit_affine(Cin, dy.device)
                if ((KH, pad) == (3, 1) and Cout % 16 == 0 and Cin % 32 == 0 and dy.shape[3] % 2 == 0 and dy.shape[3] >= 4
                        and not _use_conv_x3(dy.shape, Cin)):
                    # round 6: the Winograd kernel with the gradient filter transformed straight from W (one launch instead of flip + transpose +
                    # copy + transform); the same U, the same kernel, the same bits as the path below
                    dx = ops.conv3x3_winograd(dy, ops.winograd_weights_dgrad(W), one, zero, False)
                elif (KH, pad) == (3, 1) and Cout % 16 == 0 and _use_conv_x3(dy.shape, Cin):
                    dx = _conv3x3_x3(dy, W, True)
                else:
                    dx = _conv_forward(dy, W.flip(2, 3).transpose(0, 1), 1, pad)
            else:
                dx = torch.empty_like(x)
                call("di2p_conv2d_dgrad", ptr(dy), ptr(_c(W)), ptr(dx), B, Cin, H, Wd, Cout, KH, KW, stride, pad, stream())
        if ctx.needs_input_grad[1]:
            dW = ctx.sink if ctx.sink is not None else torch.empty_like(W, memory_format=torch.contiguous_format)
            nb = lib.di2p_conv2d_wgrad_workspace_bytes(B, Cin, H, Wd, Cout, KH, KW, stride, pad)
            ws = _ws(nb, x.device)
            call("di2p_conv2d_wgrad", ptr(x), ptr(dy), ptr(dW), B, Cin, H, Wd, Cout, KH, KW, stride, pad, ptr(ws), nb, stream())
            if ctx.sink is not None:
                dW = None
        return dx, dW, None, None


# ------------------------------------------------------------------------------------------------ BatchNorm (train mode)
class _BatchNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, relu, residual):
        x = _c(x)
        B, C = x.shape[0], x.shape[1]
        N = x.numel() // (B * C)
        y = torch.empty_like(x)
        mean = torch.empty((C,), dtype=_f32, device=x.device)
        invstd = torch.empty((C,), dtype=_f32, device=x.device)
        res = _c(residual) if residual is not None else None
        ws = _ws(_lib.load().di2p_channel_reduce_workspace_bytes(B, C, N), x.device)
        call("di2p_bn_train_forward", ptr(x), ptr(gamma), ptr(beta), ptr(res), ptr(y), ptr(mean), ptr(invstd), ptr(running_mean),
             ptr(running_var), float(momentum), BN_EPS, int(bool(relu)), B, C, N, ptr(ws), stream())
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.relu, ctx.has_res = bool(relu), residual is not None
        ctx.sinks = (_sink(gamma), _sink(beta))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy = _c(dy)
        B, C = x.shape[0], x.shape[1]
        N = x.numel() // (B * C)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        sunk = ctx.sinks[0] is not None and ctx.sinks[1] is not None
        dgamma = ctx.sinks[0] if sunk else torch.empty_like(gamma)
        dbeta = ctx.sinks[1] if sunk else torch.empty_like(gamma)
        ws = _ws(_lib.load().di2p_channel_reduce_workspace_bytes(B, C, N), x.device)
        call("di2p_bn_train_backward", ptr(x), ptr(y), ptr(dy), ptr(gamma), ptr(mean), ptr(invstd), int(ctx.relu), ptr(dx), ptr(dres),
             ptr(dgamma), ptr(dbeta), B, C, N, ptr(ws), stream())
        return dx, (None if sunk else dgamma), (None if sunk else dbeta), None, None, None, None, dres


# `num_batches_tracked += 1` is a one-element launch per BatchNorm layer (64 per step, 3 us each plus the gap around it).  Inside a whole
# forward (keypoint_detector) the counters are collected and bumped by ONE multi-tensor launch at its end; a lone batch_norm call bumps at once.
_NBT_PENDING = None


def _flush_batch_counters():
    global _NBT_PENDING
    pending, _NBT_PENDING = _NBT_PENDING, None
    if pending:
        torch._foreach_add_(pending, 1)


def batch_norm(P, key, x, relu, residual=None, momentum=0.1):
    """nn.BatchNorm{1,2}d of state-dict prefix `key` in train mode (+ fused residual add and ReLU)."""
    rm, rv = P.get(key + ".running_mean"), P.get(key + ".running_var")
    y = _BatchNorm.apply(x, P[key + ".weight"], P[key + ".bias"], rm, rv, momentum, relu, residual)
    nbt = P.get(key + ".num_batches_tracked")
    if nbt is not None:
        if _NBT_PENDING is not None:
            _NBT_PENDING.append(nbt)
        else:
            nbt += 1
    return y


# ------------------------------------------------------------------------------------------------ routers
class _SegmentMax(Function):
    """index_max + gather + empty-cluster mask (networks_pc.py:88-93, 101-104): data f32[B,C,N], index i32[B,N] -> f32[B,C,Ma]."""

    @staticmethod
    def forward(ctx, data, index, Ma, mask):
        data = _c(data)
        idx, val = ops.index_max(data, index, Ma, return_values=True, mask=mask)
        ctx.save_for_backward(idx, mask)
        ctx.N = data.shape[2]
        return val

    @staticmethod
    def backward(ctx, dval):
        idx, mask = ctx.saved_tensors
        dval = _c(dval)
        B, C, M = dval.shape
        dx = torch.empty((B, C, ctx.N), dtype=_f32, device=dval.device)
        call("di2p_segment_max_backward", ptr(dval), ptr(idx), ptr(mask), ptr(dx), B, C, ctx.N, M, stream())
        return dx, None, None, None


class _GatherCols(Function):
    """out[b,c,j] = x[b,c,idx[b,j]]  (torch.gather along the last axis with an index shared by the channels)."""

    @staticmethod
    def forward(ctx, x, idx):
        x = _c(x)
        ctx.save_for_backward(idx)
        ctx.M = x.shape[2]
        return prep.gather_points(x, idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        B, C, J = dy.shape
        dx = torch.empty((B, C, ctx.M), dtype=_f32, device=dy.device)
        nb = _lib.load().di2p_gather_backward_workspace_bytes(B, C, J, ctx.M)
        ws = _ws(nb, dy.device)
        call("di2p_gather_backward", ptr(dy), ptr(idx), None, 1, ptr(dx), B, C, J, ctx.M, ptr(ws), nb, stream())
        return dx, None


class _Interpolate(Function):
    """upsample_by_interpolation given the 3-NN indices and weights (functions of the coordinates only)."""

    @staticmethod
    def forward(ctx, feats, idx, w):
        feats = _c(feats)
        ctx.save_for_backward(idx, w)
        ctx.M = feats.shape[2]
        return ops.interpolate(feats, idx, w)

    @staticmethod
    def backward(ctx, dy):
        idx, w = ctx.saved_tensors
        dy = _c(dy)
        B, C, J = dy.shape
        dx = torch.empty((B, C, ctx.M), dtype=_f32, device=dy.device)
        nb = _lib.load().di2p_gather_backward_workspace_bytes(B, C, J, ctx.M)
        ws = _ws(nb, dy.device)
        call("di2p_gather_backward", ptr(dy), ptr(idx), ptr(w), 3, ptr(dx), B, C, J, ctx.M, ptr(ws), nb, stream())
        return dx, None, None


class _GroupMax(Function):
    """max over the last axis; the gradient goes to the first arg-max."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        K = x.shape[-1]
        rows = x.numel() // K
        y = torch.empty(x.shape[:-1], dtype=_f32, device=x.device)
        arg = torch.empty(x.shape[:-1], dtype=_i32, device=x.device)
        call("di2p_group_max_forward", ptr(x), ptr(y), ptr(arg), rows, K, stream())
        ctx.save_for_backward(arg)
        ctx.K = K
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty(tuple(dy.shape) + (ctx.K,), dtype=_f32, device=dy.device)
        call("di2p_group_max_backward", ptr(dy), ptr(arg), ptr(dx), dy.numel(), ctx.K, stream())
        return dx


class _AttentionPool(Function):
    """out[b,c,m] = (1/HW) sum_hw feat[b,c,hw] * score[b,hw,m]  (networks_united.py:139-150 without the B x C x HW x M tensor)."""

    @staticmethod
    def forward(ctx, feat, score):
        feat, score = _c(feat), _c(score)
        ctx.save_for_backward(feat, score)
        return ops.attention_pool(feat, score)

    @staticmethod
    def backward(ctx, dout):
        feat, score = ctx.saved_tensors
        dout = _c(dout)
        B, C, HW = feat.shape
        Mn = score.shape[2]
        dfeat = dscore = None
        if ctx.needs_input_grad[0]:
            dfeat = torch.empty_like(feat)
            nb = _lib.load().di2p_bmm_rc_workspace_bytes(B, C, HW, Mn)
            ws = _ws(nb, feat.device)
            call("di2p_bmm_rc", ptr(dout), Mn, C * Mn, ptr(score), Mn, HW * Mn, ptr(dfeat), B, C, HW, Mn, 1.0 / HW, 0, ptr(ws), nb, stream())
        if ctx.needs_input_grad[1]:
            dscore = torch.empty_like(score)
            call("di2p_bmm_km", ptr(feat), HW, C * HW, ptr(dout), Mn, C * Mn, ptr(dscore), B, HW, Mn, C, 1.0 / HW, stream())
        return dfeat, dscore


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.maxpool3x3s2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        call("di2p_maxpool3x3s2_backward", ptr(x), ptr(dy), ptr(dx), B, C, H, W, stream())
        return dx


class _AvgPool(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.shape = x.shape
        return ops.global_avgpool(x)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        return (dy.reshape(B, C, 1, 1) / float(H * W)).expand(B, C, H, W).contiguous()


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, mask, scale):
        x = _c(x)
        y = torch.empty_like(x)
        call("di2p_apply_mask", ptr(x), ptr(mask), float(scale), ptr(y), x.numel(), stream())
        ctx.save_for_backward(mask)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        call("di2p_apply_mask", ptr(dy), ptr(mask), ctx.scale, ptr(dx), dy.numel(), stream())
        return dx, None, None


def dropout_mask(shape, p, seed, stream_id, device):
    """u8 keep-mask of nn.Dropout(p) from the library's counter-based generator."""
    mask = torch.empty(shape, dtype=torch.uint8, device=device)
    call("di2p_dropout_mask", int(seed), int(stream_id), float(p), mask.numel(), ptr(mask), stream())
    return mask


# ------------------------------------------------------------------------------------------------ the network, train mode
def _equivariant(P, q, x, dropout=None):
    x = linear(x, P[q + ".conv.weight"], P.get(q + ".conv.bias"))
    if (q + ".norm.weight") in P:
        x = batch_norm(P, q + ".norm", x, relu=True)
    if dropout is not None:
        x = _Dropout.apply(x, dropout, 2.0)
    return x


def pointnet(P, p, x, dropouts=None):
    i = 0
    while (p + ".layers.%d.conv.weight" % i) in P:
        x = _equivariant(P, p + ".layers.%d" % i, x, dropouts[i] if dropouts is not None and i < len(dropouts) else None)
        i += 1
    return x


def _myconv2d(P, q, x):
    x = linear(x, P[q + ".conv.weight"], P.get(q + ".conv.bias"))
    return batch_norm(P, q + ".norm", x, relu=True)


def pc_encoder(P, opt, pc, intensity, sn, node_a, node_b, p="pc_encoder"):
    B, N, Ma, Mb = pc.shape[0], pc.shape[2], node_a.shape[2], node_b.shape[2]
    idx_a, w_a = ops.knn_nodes(pc, node_a, opt.k_interp_point_a, want_weights=True)
    cluster_mean, mask, min_idx = ops.cluster_stats(pc, idx_a, Ma)
    _, aug = ops.build_point_input(pc, intensity, sn, cluster_mean, min_idx)
    first = pointnet(P, p + ".first_pointnet", aug)
    first_max = _SegmentMax.apply(first, min_idx, Ma, mask)
    scattered = _GatherCols.apply(first_max, min_idx)
    second = pointnet(P, p + ".second_pointnet", torch.cat((first, scattered), dim=1))
    node_a_features = _SegmentMax.apply(second, min_idx, Ma, mask)
    # GeneralKNNFusionModule (layers_pc.py:779-818)
    K = opt.k_ab
    knn_I = ops.knn_nodes(node_b, cluster_mean, K)
    coord = ops.gather_neighbors(cluster_mean, node_b, knn_I)                       # [B,3,Mb*K], decentred
    feat = _GatherCols.apply(node_a_features, knn_I.view(B, Mb * K))
    q = p + ".knnlayer"
    y = torch.cat((coord, feat), dim=1)
    i = 0
    while (q + ".layers_before.%d.conv.weight" % i) in P:
        y = _myconv2d(P, q + ".layers_before.%d" % i, y)
        i += 1
    C = y.shape[1]
    fmax = _GroupMax.apply(y.view(B, C, Mb, K))
    y = torch.cat((fmax.unsqueeze(3).expand(B, C, Mb, K).reshape(B, C, Mb * K), y), dim=1)
    i = 0
    while (q + ".layers_after.%d.conv.weight" % i) in P:
        y = _myconv2d(P, q + ".layers_after.%d" % i, y)
        i += 1
    node_b_features = _GroupMax.apply(y.view(B, y.shape[1], Mb, K))
    final = pointnet(P, p + ".final_pointnet", torch.cat((node_b, node_b_features), dim=1))
    global_feature = _GroupMax.apply(final).unsqueeze(2)
    return dict(first=first, second=second, node_a_features=node_a_features, node_b_features=node_b_features,
                global_feature=global_feature, idx_a=idx_a, w_a=w_a)


_SIDE = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def _conv_bn(P, pc, pb, x, stride, pad, relu, residual=None):
    y = _Conv2d.apply(x, P[pc + ".weight"], stride, pad)
    return batch_norm(P, pb, y, relu=relu, residual=residual)


def resnet34(P, x, p="img_encoder.backbone"):
    outs = []
    x = _conv_bn(P, p + ".conv1", p + ".bn1", x, 2, 3, True)
    x = _MaxPool.apply(x)
    for li, nblocks in enumerate((3, 4, 6, 3), start=1):
        for bi in range(nblocks):
            q = "%s.layer%d.%d" % (p, li, bi)
            stride = 2 if (bi == 0 and li > 1) else 1
            identity = x
            y = _conv_bn(P, q + ".conv1", q + ".bn1", x, stride, 1, True)
            if (q + ".downsample.0.weight") in P:
                identity = _conv_bn(P, q + ".downsample.0", q + ".downsample.1", x, stride, 0, False)
            x = _conv_bn(P, q + ".conv2", q + ".bn2", y, 1, 1, True, residual=identity)     # relu(bn2(conv2(y)) + identity)
        outs.append(x)
    return outs[2], outs[3], _AvgPool.apply(x)


def keypoint_detector(P, opt, pc, intensity, sn, node_a, node_b, img, dropouts=None, branch_streams=False):
    """KeypointDetector.forward (networks_united.py:105-210) in train mode -> scores f32[B, 2 (+L), N].
    dropouts: the two u8 keep-masks [B, C, N] of per_point_pn layers 0 and 1 (None: no dropout, i.e. p = 0).
    branch_streams: run the image branch on a second HIP stream (same results: no kernel changes its summation order)."""
    global _NBT_PENDING
    _NBT_PENDING = []                       # the BatchNorm counters of this forward: one launch at its end (the streams are joined by then)
    try:
        return _keypoint_detector(P, opt, pc, intensity, sn, node_a, node_b, img, dropouts, branch_streams)
    finally:
        _flush_batch_counters()


def _keypoint_detector(P, opt, pc, intensity, sn, node_a, node_b, img, dropouts, branch_streams):
    B, N, Ma, Mb = pc.shape[0], pc.shape[2], node_a.shape[2], node_b.shape[2]
    if branch_streams and img.is_cuda:
        # The image branch and the point branch are independent up to the attention layers, and at the training batch (8 frames) neither
        # fills the chip: the ResNet runs on a second stream next to the point encoder.  autograd runs every backward node on the stream
        # of its forward, so the two backward halves overlap the same way, and the engine joins the streams at the end of backward().
        main, side = torch.cuda.current_stream(img.device), _side_stream(img.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            s16, s32, iglob = resnet34(P, img)
        e = pc_encoder(P, opt, pc, intensity, sn, node_a, node_b)
        main.wait_stream(side)
        for t in (s16, s32, iglob):
            t.record_stream(main)          # allocated on the side stream, read on the main one
    else:
        e = pc_encoder(P, opt, pc, intensity, sn, node_a, node_b)
        s16, s32, iglob = resnet34(P, img)
    C_img = iglob.shape[1]
    s16f = s16.reshape(B, s16.shape[1], -1)
    s32f = s32.reshape(B, s32.shape[1], -1)
    ig = iglob.reshape(B, C_img, 1)
    ig_a, ig_b = ig.expand(B, C_img, Ma), ig.expand(B, C_img, Mb)

    score_b = pointnet(P, "node_b_attention_pn", torch.cat((e["node_b_features"], ig_b), dim=1))
    w_s32 = _AttentionPool.apply(s32f, score_b)
    up_b = pointnet(P, "node_b_pn", torch.cat((e["node_b_features"], e["global_feature"].expand(B, -1, Mb), w_s32, ig_b), dim=1))
    idx_pb, w_pb = ops.knn_nodes(pc, node_b, opt.k_interp_point_b, want_weights=True)
    interp_pb = _Interpolate.apply(up_b, idx_pb, w_pb)

    score_a = pointnet(P, "node_a_attention_pn", torch.cat((e["node_a_features"], ig_a), dim=1))
    w_s16 = _AttentionPool.apply(s16f, score_a)
    idx_ab, w_ab = ops.knn_nodes(node_a, node_b, opt.k_interp_ab, want_weights=True)
    interp_ab = _Interpolate.apply(up_b, idx_ab, w_ab)
    up_a = pointnet(P, "node_a_pn", torch.cat((e["node_a_features"], interp_ab, w_s16), dim=1))
    interp_pa = _Interpolate.apply(up_a, e["idx_a"], e["w_a"])

    return pointnet(P, "per_point_pn", torch.cat((interp_pa, interp_pb, e["first"], e["second"]), dim=1), dropouts)


def head_widths(P):
    """Output widths of per_point_pn layers 0 and 1 (the dropout mask shapes)."""
    return P["per_point_pn.layers.0.conv.weight"].shape[0], P["per_point_pn.layers.1.conv.weight"].shape[0]
