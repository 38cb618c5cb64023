"""Host-side mirror of the reference's network modules, running on the HIP kernels.

Same constructor arguments, call signatures, return tuples and ``state_dict`` keys (361 tensors)
as the reference, so a released checkpoint loads unchanged:

  PointNet / EquivariantLayer      models/layers_pc.py:259-408
  GeneralKNNFusionModule           models/layers_pc.py:756-818
  PCEncoder                        models/networks_pc.py:15-124
  ImageEncoder (ResNet-34)         models/networks_img.py:12-28, models/resnet.py:125-216
  KeypointDetector                 models/networks_united.py:14-210
  MMClassifer / MMClassiferCoarse  models/multimodal_classifier.py:25-117, :380-469 (inference part)

The module forwards here are the inference (eval-mode) path: BatchNorm uses running statistics, dropout is
identity; the train-mode forward and the backward live in train_net.py / training.py.  The modules hold the reference's parameters verbatim; ``_pack()`` derives the kernel operands once
per load (weights transposed to [K,M]; BN folded to scale/shift; concatenated inputs that are
broadcasts folded into per-frame bias vectors; per_point_pn layer 0 split so that the interpolated
inputs are contracted per NODE instead of per POINT -- W*sum_k w_k f_k == sum_k w_k (W f_k)).
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .ops import Src

BN_EPS = 1e-5


# ------------------------------------------------------------------ parameter tree with reference keys
class _Holder(nn.Module):
    pass


def _register(root, key, tensor, is_buffer):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def _pn_spec(prefix, cin, couts, norm_last):
    spec, c = [], cin
    for i, co in enumerate(couts):
        q = "%s.layers.%d" % (prefix, i)
        spec += [(q + ".conv.weight", (co, c, 1)), (q + ".conv.bias", (co,))]
        if i < len(couts) - 1 or norm_last:
            spec += [(q + ".norm." + s, (co,)) for s in ("weight", "bias", "running_mean", "running_var")]
            spec += [(q + ".norm.num_batches_tracked", ())]
        c = co
    return spec


def _c2d_spec(prefix, cin, co):
    return ([(prefix + ".conv.weight", (co, cin, 1, 1)), (prefix + ".conv.bias", (co,))]
            + [(prefix + ".norm." + s, (co,)) for s in ("weight", "bias", "running_mean", "running_var")]
            + [(prefix + ".norm.num_batches_tracked", ())])


def _bn_spec(prefix, c):
    return [(prefix + "." + s, (c,)) for s in ("weight", "bias", "running_mean", "running_var")] + \
           [(prefix + ".num_batches_tracked", ())]


def _materialise(module, spec):
    """Create parameters/buffers with reference names; default init = identity BN, zero bias, He weights."""
    g = torch.Generator().manual_seed(0)
    for key, shape in spec:
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            _register(module, key, torch.zeros((), dtype=torch.long), True)
        elif leaf in ("running_mean",):
            _register(module, key, torch.zeros(shape), True)
        elif leaf in ("running_var",):
            _register(module, key, torch.ones(shape), True)
        elif leaf == "bias":
            _register(module, key, torch.zeros(shape), False)
        elif len(shape) == 1:   # norm weight
            _register(module, key, torch.ones(shape), False)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            _register(module, key, torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5, False)


class _PackedModule(nn.Module):
    """nn.Module whose kernel operands are re-derived after every load_state_dict / device move."""

    def __init__(self):
        super().__init__()
        self._packed = None
        self._weights_version = 0
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    def _invalidate(self):
        """Drop the derived kernel operands (weights changed or moved) and bump the version every holder of those operands checks --
        a captured hipGraph keeps RAW pointers to them (pipeline.RegistrationExecutor re-captures when the version moves)."""
        for m in self.modules():
            if isinstance(m, _PackedModule):
                m._packed = None
                m._weights_version += 1
        # the split (bf16x3) copies of these operands expire with them (ops._X3_CACHE entries are weakly tied to the operand they were
        # made from): nothing global is cleared here -- another model's splits, which its captured graphs point to, stay where they are

    @property
    def weights_version(self):
        """Changes whenever load_state_dict / .to() / an optimiser step invalidated the packed operands of this module or a sub-module."""
        return sum(m._weights_version for m in self.modules() if isinstance(m, _PackedModule))

    def packed_operands(self):
        """The live packed-operand containers of this module tree (a holder of raw pointers keeps them referenced)."""
        return [m._packed for m in self.modules() if isinstance(m, _PackedModule) and m._packed is not None]

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def _sd(self):
        return {k: v for k, v in self.state_dict().items()}

    def prepack(self):
        """Derive the kernel operands of this module and its sub-modules NOW, on the current stream, and wait for them:
        afterwards forward() may be issued from any stream (or captured in a graph) without an ordering hazard on the
        lazily packed weights."""
        for m in self.modules():
            if isinstance(m, _PackedModule) and hasattr(m, "_pack"):
                m._pack()
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        return self


def _fold(sd, conv_key, norm_key, tap_major=False):
    """(Wt [K,M], scale [M]|None, shift [M]) for conv(+bias) followed by optional BN(eval).
    tap_major: rows ordered (kh,kw,ci) instead of the weight's (ci,kh,kw) (see di2p_conv2d)."""
    w = sd[conv_key + ".weight"]
    if tap_major:
        Wt = w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous()
    else:
        Wt = w.reshape(w.shape[0], -1).t().contiguous()
    bias = sd.get(conv_key + ".bias")
    if norm_key is not None and (norm_key + ".weight") in sd:
        scale = sd[norm_key + ".weight"] / torch.sqrt(sd[norm_key + ".running_var"] + BN_EPS)
        shift = sd[norm_key + ".bias"] - sd[norm_key + ".running_mean"] * scale
        if bias is not None:
            shift = shift + bias * scale
        return Wt, scale.contiguous(), shift.contiguous(), True
    shift = bias.contiguous() if bias is not None else torch.zeros(w.shape[0], device=w.device)
    return Wt, None, shift, False


def _pn_pack(sd, prefix):
    layers, i = [], 0
    while (prefix + ".layers.%d.conv.weight" % i) in sd:
        q = prefix + ".layers.%d" % i
        layers.append(_fold(sd, q + ".conv", q + ".norm"))
        i += 1
    return layers


def _run_layer(srcs, layer, N, **kw):
    Wt, scale, shift, act = layer
    return ops.pointwise_gemm(srcs, Wt, Wt.shape[1], N, scale=scale, shift=shift, relu=act, **kw)


def _run_split_layer(dense_srcs, dense_rows, node_feats, node_rows, idx, layer, N, **kw):
    """W @ cat(dense, node_feats[:, :, idx]) == W_d @ dense + (W_n @ node_feats)[:, :, idx]: the gathered (or
    group-broadcast) half of a concatenation is contracted ONCE PER NODE ([B,nodes,M], node-major) and added per column
    by the epilogue (unit-weight gathered add), so the big GEMM only sees the dense channels."""
    Wt, scale, shift, act = layer
    M = Wt.shape[1]
    B = node_feats.shape[0]
    G = ops.pointwise_gemm([Src(node_feats)], Wt[node_rows[0]:node_rows[1]], M, node_feats.shape[2], transpose_out=True)
    return ops.pointwise_gemm(dense_srcs, Wt[dense_rows[0]:dense_rows[1]], M, N, scale=scale, shift=shift, relu=act,
                              gathered=[(G, idx.reshape(B, N, 1), None)], **kw)


def _run_pn(x, layers, **kw):
    N = x.shape[2]
    if not kw and ops.point_chain_ok([Src(x)], layers, N):       # narrow chains of one width: one launch, hidden activations in LDS
        return ops.point_chain([Src(x)], layers, N)
    for layer in layers:
        x = _run_layer([Src(x)], layer, N, **kw)
    return x


def _run_split_pn(dense_srcs, dense_rows, node_feats, node_rows, idx, layers, N):
    """_run_split_layer for layers[0] followed by the rest of the chain -- in one launch when the chain is narrow."""
    Wt, scale, shift, act = layers[0]
    M = Wt.shape[1]
    B = node_feats.shape[0]
    head = (Wt[dense_rows[0]:dense_rows[1]], scale, shift, act)
    chain = [head] + list(layers[1:])
    if ops.point_chain_ok(dense_srcs, chain, N):
        G = ops.pointwise_gemm([Src(node_feats)], Wt[node_rows[0]:node_rows[1]], M, node_feats.shape[2], transpose_out=True)
        return ops.point_chain(dense_srcs, chain, N, gathered=[(G, idx.reshape(B, N, 1), None)])
    x = _run_split_layer(dense_srcs, dense_rows, node_feats, node_rows, idx, layers[0], N)
    return _run_pn(x, layers[1:])


# ------------------------------------------------------------------ point-cloud encoder
class PCEncoder(_PackedModule):
    def __init__(self, opt, Ca: int, Cb: int, Cg: int, prefix=""):
        super().__init__()
        self.opt, self.Ca, self.Cb, self.Cg = opt, Ca, Cb, Cg
        spec = (_pn_spec("first_pointnet", 7, [Ca // 2] * 3, True)
                + _pn_spec("second_pointnet", Ca, [Ca, Ca], True)
                + _c2d_spec("knnlayer.layers_before.0", 3 + Ca, Cb) + _c2d_spec("knnlayer.layers_before.1", Cb, Cb)
                + _c2d_spec("knnlayer.layers_after.0", 2 * Cb, 2 * Cb) + _c2d_spec("knnlayer.layers_after.1", 2 * Cb, Cb)
                + _pn_spec("final_pointnet", 3 + Cb, [Cg // 2, Cg], True))
        _materialise(self, spec)

    def _pack(self):
        if self._packed is None:
            sd = self._sd()
            p = {"first": _pn_pack(sd, "first_pointnet"), "second": _pn_pack(sd, "second_pointnet"),
                 "final": _pn_pack(sd, "final_pointnet")}
            for name in ("layers_before.0", "layers_before.1", "layers_after.0", "layers_after.1"):
                q = "knnlayer." + name
                p[name] = _fold(sd, q + ".conv", q + ".norm")
            self._packed = p
        return self._packed

    def _group_index(self, B, Mb, K, device):
        """i32[B, Mb*K]: column n of the neighbour-expanded tensor belongs to node n // K (cached constant)."""
        key = (B, Mb, K, str(device))
        cache = self.__dict__.setdefault("_gidx_cache", {})
        if key not in cache:
            cache[key] = (torch.arange(Mb * K, dtype=torch.int32, device=device) // K).unsqueeze(0).expand(B, -1).contiguous()
            torch.cuda.current_stream().synchronize()     # created once per batch shape: visible to every stream from here on
        return cache[key]

    def forward_device(self, pc, intensity, sn, node_a, node_b):
        """Returns the reference 8-tuple plus the device-side extras the fusion head re-uses."""
        if pc.size(2) != self.opt.input_pt_num:
            raise RuntimeError("N must equal opt.input_pt_num (the reference bakes it in, networks_pc.py:44-45)")
        p = self._pack()
        B, N, Ma, Mb = pc.size(0), pc.size(2), node_a.size(2), node_b.size(2)
        idx_a, w_a = ops.knn_nodes(pc, node_a, self.opt.k_interp_point_a, want_weights=True)
        cluster_mean, mask, min_idx = ops.cluster_stats(pc, idx_a, Ma)
        pc_centers, aug = ops.build_point_input(pc, intensity, sn, cluster_mean, min_idx)
        first = _run_pn(aug, p["first"])
        _, first_max = ops.index_max(first, min_idx, Ma, return_values=True, mask=mask)
        Ch = first.shape[1]
        second = _run_split_pn([Src(first)], (0, Ch), first_max, (Ch, 2 * Ch), min_idx, p["second"], N)   # cat(first, first_max[min_idx])
        _, node_a_features = ops.index_max(second, min_idx, Ma, return_values=True, mask=mask)
        # GeneralKNNFusionModule (layers_pc.py:779-818)
        K = self.opt.k_ab
        knn_I = ops.knn_nodes(node_b, cluster_mean, K)
        coord = ops.gather_neighbors(cluster_mean, node_b, knn_I)
        y = _run_split_layer([Src(coord)], (0, 3), node_a_features, (3, 3 + node_a_features.shape[1]), knn_I.view(B, Mb * K),
                             p["layers_before.0"], Mb * K)
        # consecutive bf16x3 layers hand their activations on already split (ops.X3Planes: the same bits, the split out of the consumers' K loops)
        pow2 = K & (K - 1) == 0 and K <= 32
        Cb1 = p["layers_before.1"][0].shape[1]
        pl = [pow2 and ops.x3_planes_link(p["layers_before.1"], p["layers_after.0"][0][Cb1:2 * Cb1], B, Mb * K),
              pow2 and ops.x3_planes_link(p["layers_after.0"], p["layers_after.1"][0], B, Mb * K)]
        if pow2:                              # the max over the K neighbours comes out of the same launch as y itself
            y, fmax = _run_layer([Src(y)], p["layers_before.1"], Mb * K, group_max=K, also_full=True, planes_out=pl[0])
            Cb = y.shape[1]
        else:
            y = _run_layer([Src(y)], p["layers_before.1"], Mb * K)
            Cb = y.shape[1]
            fmax = ops.channel_max(y.view(B, Cb * Mb, K)).view(B, Cb, Mb)
        y = _run_split_layer([y if pl[0] else Src(y)], (Cb, 2 * Cb), fmax, (0, Cb), self._group_index(B, Mb, K, pc.device), p["layers_after.0"], Mb * K,
                             planes_out=pl[1])
        if pow2:
            node_b_features = _run_layer([y if pl[1] else Src(y)], p["layers_after.1"], Mb * K, group_max=K)
        else:
            y = _run_layer([Src(y)], p["layers_after.1"], Mb * K)
            node_b_features = ops.channel_max(y.view(B, y.shape[1] * Mb, K)).view(B, y.shape[1], Mb)
        final = _run_layer([Src(node_b), Src(node_b_features)], p["final"][0], Mb)
        final = _run_pn(final, p["final"][1:])
        global_feature = ops.channel_max(final).unsqueeze(2)
        ref_tuple = (pc_centers, cluster_mean, idx_a, first, second, node_a_features, node_b_features, global_feature)
        return ref_tuple, dict(w_a=w_a, min_idx=min_idx, mask=mask, knn_I=knn_I)

    def forward(self, pc, intensity, sn, node_a, node_b):
        t, _ = self.forward_device(pc, intensity, sn, node_a, node_b)
        return t[0], t[1], t[2].long(), t[3], t[4], t[5], t[6], t[7]


# ------------------------------------------------------------------ image encoder
class ImageEncoder(_PackedModule):
    LAYERS = (3, 4, 6, 3)

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        r = "backbone"
        spec = [(r + ".conv1.weight", (64, 3, 7, 7))] + _bn_spec(r + ".bn1", 64)
        inpl = 64
        for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), self.LAYERS), start=1):
            for bi in range(nb):
                q = "%s.layer%d.%d" % (r, li, bi)
                spec += [(q + ".conv1.weight", (planes, inpl, 3, 3))] + _bn_spec(q + ".bn1", planes)
                spec += [(q + ".conv2.weight", (planes, planes, 3, 3))] + _bn_spec(q + ".bn2", planes)
                if bi == 0 and li > 1:
                    spec += [(q + ".downsample.0.weight", (planes, inpl, 1, 1))] + _bn_spec(q + ".downsample.1", planes)
                inpl = planes
        spec += [(r + ".fc.weight", (1000, 512)), (r + ".fc.bias", (1000,))]   # present, unused (resnet.py:152)
        _materialise(self, spec)

    def _pack(self):
        if self._packed is None:
            sd = self._sd()
            r = "backbone"
            p = {"stem": _fold(sd, r + ".conv1", r + ".bn1"), "blocks": []}
            w_stem = sd[r + ".conv1.weight"]
            if w_stem.is_cuda and tuple(w_stem.shape) == (64, 3, 7, 7):
                p["stem_p"] = ops.stem_weights(w_stem)
                p["stem_x3"] = ops.stem_x3_weights(w_stem)      # conv1 + bn1 + relu + maxpool as one launch on the bf16 matrix instructions
            for li, nb in enumerate(self.LAYERS, start=1):
                for bi in range(nb):
                    q = "%s.layer%d.%d" % (r, li, bi)
                    blk = {"c1": _fold(sd, q + ".conv1", q + ".bn1", True), "c2": _fold(sd, q + ".conv2", q + ".bn2", True),
                           "stride": 2 if (bi == 0 and li > 1) else 1, "last_of_stage": bi == nb - 1, "stage": li}
                    if (q + ".downsample.0.weight") in sd:
                        blk["ds"] = _fold(sd, q + ".downsample.0", q + ".downsample.1", True)
                    # 3x3 stride-1 layers: Winograd-domain weights U[16][Cin][Cout] for di2p_conv3x3_winograd (device tensors only)
                    for name in ("conv1", "conv2"):
                        w = sd[q + "." + name + ".weight"]
                        if w.is_cuda and (name == "conv2" or blk["stride"] == 1):
                            blk["U" + name[-1]] = ops.winograd_weights(w)
                        # every 3x3 layer: the split (bf16x3) copy of the tap-major matrix for di2p_conv3x3_x3; owned by this container,
                        # so it is dropped and re-derived with the weights it was made from
                        if w.is_cuda and w.shape[1] % 16 == 0:
                            blk["X" + name[-1]] = ops.bf16x3_pack(blk["c" + name[-1]][0])
                    if "ds" in blk and blk["ds"][0].is_cuda and blk["ds"][0].shape[0] % 16 == 0:
                        blk["Xd"] = ops.bf16x3_pack(blk["ds"][0])
                    p["blocks"].append(blk)
            self._packed = p
        return self._packed

    def forward(self, x):
        p = self._pack()
        ops.require_cuda(x)
        Wt, sc, sh, _ = p["stem"]
        if "stem_x3" in p and _lib.get_option("stem_x3") and not _lib.get_option("conv_nostem") and ops.stem_x3_supported(x.shape[2], x.shape[3]):
            x = ops.stem_x3(x, p["stem_x3"], sc, sh)
        else:
            if "stem_p" in p and not _lib.get_option("conv_nostem"):
                x = ops.conv_stem(x, p["stem_p"], sc, sh, True)
            else:
                x = ops.conv2d(x, Wt, sc, sh, 7, 7, 2, 3, True)
            x = ops.maxpool3x3s2(x)
        stage_out = {}
        wino = not _lib.get_option("conv_nowinograd")
        # bit s-1: the stride-1 layers of stage s, bit 4: the stride-2 layers (+ their 1x1 downsample branch) on di2p_conv3x3_x3
        x3_mask = _lib.get_option("conv_x3")
        for blk in p["blocks"]:
            identity = x
            Wt, sc, sh, _ = blk["c1"]
            Cout, stride = Wt.shape[1], blk["stride"]
            x3_bit = 16 if stride == 2 else 1 << (blk["stage"] - 1)
            y = None
            if (x3_mask & x3_bit) and "X1" in blk and (stride == 1 or "Xd" in blk) and ops.conv3x3_x3_supported(x.shape, Cout, stride):
                if stride == 2:
                    Wd, sd_, shd, _ = blk["ds"]
                    y, identity = ops.conv3x3_x3(x, blk["X1"], Cout, sc, sh, 2, True, downsample=(blk["Xd"], sd_, shd))
                else:
                    y = ops.conv3x3_x3(x, blk["X1"], Cout, sc, sh, 1, True)
            if y is None:
                wino_ok = wino and x.shape[3] % 2 == 0 and x.shape[3] >= 4       # the kernel wants an even width (else: direct kernel)
                if wino_ok and "U1" in blk:
                    y = ops.conv3x3_winograd(x, blk["U1"], sc, sh, True)
                else:
                    y = ops.conv2d(x, Wt, sc, sh, 3, 3, stride, 1, True, tap_major=True)
                if "ds" in blk:
                    Wd, sd_, shd, _ = blk["ds"]
                    identity = ops.conv2d(x, Wd, sd_, shd, 1, 1, stride, 0, False, tap_major=True)
            Wt, sc, sh, _ = blk["c2"]
            x3_bit = 1 << (blk["stage"] - 1)
            if (x3_mask & x3_bit) and "X2" in blk and ops.conv3x3_x3_supported(y.shape, Cout, 1):
                x = ops.conv3x3_x3(y, blk["X2"], Cout, sc, sh, 1, True, residual=identity)
            elif wino and "U2" in blk and y.shape[3] % 2 == 0 and y.shape[3] >= 4:
                x = ops.conv3x3_winograd(y, blk["U2"], sc, sh, True, residual=identity)
            else:
                x = ops.conv2d(y, Wt, sc, sh, 3, 3, 1, 1, True, residual=identity, tap_major=True)
            if blk["last_of_stage"]:
                stage_out[blk["stage"]] = x
        return stage_out[3], stage_out[4], ops.global_avgpool(stage_out[4])


# ------------------------------------------------------------------ fusion classifier
class KeypointDetector(_PackedModule):
    fuse_head = True      # coarse per_point_pn as ONE launch (ops.point_head); False: three pointwise_gemm launches (tests compare)

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.pc_encoder = PCEncoder(opt, Ca=64, Cb=256, Cg=512)
        self.img_encoder = ImageEncoder(opt)
        self.H_fine_res = int(round(opt.img_H / opt.img_fine_resolution_scale))
        self.W_fine_res = int(round(opt.img_W / opt.img_fine_resolution_scale))
        L = self.H_fine_res * self.W_fine_res
        spec = (_pn_spec("node_b_attention_pn", 256 + 512, [256, L], False)
                + _pn_spec("node_b_pn", 256 + 512 + 512 + 512, [1024, 512, 512], False)
                + _pn_spec("node_a_attention_pn", 64 + 512, [256, L * 4], False)
                + _pn_spec("node_a_pn", 64 + 256 + 512, [512, 128, 128], False))
        if opt.is_fine_resolution:
            spec += _pn_spec("per_point_pn", 736, [256, 256, 2 + L], False)
        else:
            spec += _pn_spec("per_point_pn", 736, [128, 128, 2], False)
        _materialise(self, spec)

    def _pack(self):
        if self._packed is None:
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith(("pc_encoder.", "img_encoder."))}
            p = {n: _pn_pack(sd, n) for n in ("node_b_attention_pn", "node_b_pn", "node_a_attention_pn", "node_a_pn", "per_point_pn")}
            # node_b_pn layer 0: channel order [node_b_feat 256 | global 512 (bcast) | w_s32 512 | img_global 512 (bcast)]
            Wt = p["node_b_pn"][0][0]
            p["node_b_pn_dense_Wt"] = torch.cat((Wt[0:256], Wt[768:1280]), dim=0).contiguous()
            # the coarse head on the bf16 matrix instructions (di2p_point_head_x3): the dense rows of layer 0 and layer 1, split once into
            # fragment order; the folded scale / shift rows as one [4,128] block.  Owned by this container like every derived operand.
            l0, l1, l2 = p["per_point_pn"]
            if (l0[0].is_cuda and tuple(l0[0].shape) == (736, 128) and tuple(l1[0].shape) == (128, 128) and l2[0].shape[0] == 128
                    and l2[0].shape[1] <= 4 and not l0[0].requires_grad):
                ones = torch.ones(128, device=l0[0].device)
                p["head_x3"] = {"W0p": ops.head_x3_pack(l0[0][640:736].contiguous()), "W1p": ops.head_x3_pack(l1[0]),
                                "ss": torch.stack((l0[1] if l0[1] is not None else ones, l0[2], l1[1] if l1[1] is not None else ones, l1[2])).contiguous(),
                                "relu0": l0[3], "relu1": l1[3], "W2t": l2[0], "sc2": l2[1], "sh2": l2[2], "relu2": l2[3]}
            self._packed = p
        return self._packed

    def forward(self, pc, intensity, sn, node_a, node_b, img):
        ops.require_cuda(pc, intensity, sn, node_a, node_b, img)
        p = self._pack()
        B, N, Ma, Mb = pc.size(0), pc.size(2), node_a.size(2), node_b.size(2)
        (pc_center, cluster_mean, idx_a, first, second, node_a_features, node_b_features, global_feature), ex = \
            self.pc_encoder.forward_device(pc, intensity, sn, node_a, node_b)
        s16, s32, iglob = self.img_encoder(img)
        s16f = s16.view(B, s16.size(1), -1)
        s32f = s32.view(B, s32.size(1), -1)
        ig = iglob.view(B, -1)                      # [B,512]
        gf = global_feature.view(B, -1)             # [B,512]

        # node_b attention over the /32 map (networks_united.py:139-150); broadcast channels -> per-frame bias
        Wt, sc, sh, act = p["node_b_attention_pn"][0]
        h = ops.pointwise_gemm([Src(node_b_features)], Wt[0:256], Wt.shape[1], Mb, scale=sc, shift=sh, relu=act,
                               batch_bias=ops.batch_gemv(Wt, 256, ig))
        score_b = _run_pn(h, p["node_b_attention_pn"][1:])
        w_s32 = ops.attention_pool(s32f, score_b)
        # node_b_pn (:152-155)
        Wt, sc, sh, act = p["node_b_pn"][0]
        bias = ops.batch_gemv2(Wt, 256, gf, 1280, ig)
        h = ops.pointwise_gemm([Src(node_b_features), Src(w_s32)], p["node_b_pn_dense_Wt"], Wt.shape[1], Mb,
                               scale=sc, shift=sh, relu=act, batch_bias=bias)
        up_b = _run_pn(h, p["node_b_pn"][1:])
        # point <- node_b neighbours (:158-161)
        idx_pb, w_pb = ops.knn_nodes(pc, node_b, self.opt.k_interp_point_b, want_weights=True)
        # node_a attention over the /16 map (:170-174)
        Wt, sc, sh, act = p["node_a_attention_pn"][0]
        h = ops.pointwise_gemm([Src(node_a_features)], Wt[0:64], Wt.shape[1], Ma, scale=sc, shift=sh, relu=act,
                               batch_bias=ops.batch_gemv(Wt, 64, ig))
        score_a = _run_pn(h, p["node_a_attention_pn"][1:])
        w_s16 = ops.attention_pool(s16f, score_a)
        # node_a <- node_b interpolation (:176-182), node_a_pn (:184-187)
        idx_ab, w_ab = ops.knn_nodes(node_a, node_b, self.opt.k_interp_ab, want_weights=True)
        interp_ab = ops.interpolate(up_b, idx_ab, w_ab)
        h = _run_layer([Src(node_a_features), Src(interp_ab), Src(w_s16)], p["node_a_pn"][0], Ma)
        up_a = _run_pn(h, p["node_a_pn"][1:])
        # per-point head (:188-197): layer 0 contracts the interpolated inputs per node
        Wt, sc, sh, act = p["per_point_pn"][0]
        M0 = Wt.shape[1]
        G_a = ops.pointwise_gemm([Src(up_a)], Wt[0:128], M0, Ma, transpose_out=True)     # [B,Ma,M0] node-major
        G_b = ops.pointwise_gemm([Src(up_b)], Wt[128:640], M0, Mb, transpose_out=True)
        gathered = [(G_a, idx_a, ex["w_a"]), (G_b, idx_pb, w_pb)]
        l1, l2 = p["per_point_pn"][1], p["per_point_pn"][2]
        if ("head_x3" in p and self.fuse_head and _lib.get_option("head_x3") and first.shape[1] == 32 and second.shape[1] == 64
                and idx_a.shape[-1] == 3 and idx_pb.shape[-1] == 3):
            # coarse head, round 5: one wave-autonomous launch on the bf16 matrix instructions (exact three-way splits), node tables in LDS
            scores = ops.point_head_x3(first, second, p["head_x3"],
                                       [(G_a, idx_a.reshape(B, N, 3), ex["w_a"]), (G_b, idx_pb.reshape(B, N, 3), w_pb)], N)
        elif M0 == 128 and l1[0].shape[1] == 128 and l2[0].shape[1] <= 4 and N % 4 == 0 and self.fuse_head:
            # coarse head: the three layers in one launch, hidden activations stay in LDS (bit-identical to the chain below)
            scores = ops.point_head([Src(first), Src(second)], (Wt[640:736], sc, sh, act), l1, l2, N, gathered=gathered)
        else:
            h = ops.pointwise_gemm([Src(first), Src(second)], Wt[640:736], M0, N, scale=sc, shift=sh, relu=act, gathered=gathered)
            # the coarse chain is the bit-exact reference of the fused kernel (fp32 matrix instructions): not on the bf16x3 kernel
            scores = _run_pn(h, p["per_point_pn"][1:], x3=False if M0 == 128 else None)
        coarse = scores[:, 0:2, :]
        if self.opt.is_fine_resolution:
            return coarse, scores[:, 2:, :]
        return coarse

    def intermediates(self, pc, intensity, sn, node_a, node_b, img):
        """Test hook: the materialised per-stage tensors the reference would produce."""
        p = self._pack()
        B, N, Ma, Mb = pc.size(0), pc.size(2), node_a.size(2), node_b.size(2)
        t, ex = self.pc_encoder.forward_device(pc, intensity, sn, node_a, node_b)
        s16, s32, iglob = self.img_encoder(img)
        return dict(pc_center=t[0], cluster_mean=t[1], a_min_k_idx=t[2], first_pn_out=t[3], second_pn_out=t[4],
                    node_a_features=t[5], node_b_features=t[6], global_feature=t[7], s16=s16, s32=s32, img_global=iglob,
                    w_a=ex["w_a"], knn_I=ex["knn_I"])


def model_state_dict_convert_auto(sd):
    """util/pytorch_helper.py:24-33: accept DataParallel ('module.'-prefixed) or bare checkpoints."""
    if len(sd) and all(k.startswith("module.") for k in sd):
        return {k[len("module."):]: v for k, v in sd.items()}
    return sd


class MMClassifer:
    """models/multimodal_classifier.py:25-225 (coarse + fine): inference_pass, and optimize / test_model through
    deepi2p_amd.training.ClassifierTrainer (train-mode forward + backward on the HIP kernels, Adam)."""

    fine = True

    def __init__(self, opt, writer=None):
        self.opt = opt
        self.writer = writer
        opt.is_fine_resolution = self.fine if not hasattr(opt, "is_fine_resolution") else opt.is_fine_resolution
        self.device = getattr(opt, "device", torch.device("cuda", 0))
        self.detector = KeypointDetector(opt).to(self.device).eval()
        self.pc = self.intensity = self.sn = self.node_a = self.node_b = self.P = self.img = self.K = None

    def load_model(self, model_path):
        self.detector.load_state_dict(model_state_dict_convert_auto(torch.load(model_path, map_location="cpu")))

    def set_input(self, pc, intensity, sn, node_a, node_b, P, img, K):
        """H2D copies into persistent device tensors (multimodal_classifier.py:82-93)."""
        def put(name, t):
            cur = getattr(self, name)
            if cur is None or cur.shape != t.shape or cur.dtype != t.dtype:
                cur = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                setattr(self, name, cur)
            cur.copy_(t, non_blocking=True)
        for n, t in (("pc", pc), ("intensity", intensity), ("sn", sn), ("node_a", node_a), ("node_b", node_b),
                     ("P", P), ("img", img), ("K", K)):
            put(n, t)

    def forward(self, pc, intensity, sn, node_a, node_b, img):
        return self.detector(pc, intensity, sn, node_a, node_b, img)

    def _trainer(self):
        if getattr(self, "_trainer_obj", None) is None:
            from .training import ClassifierTrainer
            self._trainer_obj = ClassifierTrainer(self.detector, self.opt, lr=self._lr())
        return self._trainer_obj

    def _lr(self):
        if getattr(self, "old_lr_detector", None) is None:
            self.old_lr_detector = float(getattr(self.opt, "lr", 1e-3))
        return self.old_lr_detector

    def save_network(self, network, save_filename):
        """multimodal_classifier.py:263-265: the reference's 361-key state_dict (tensors copied out of the trainer's flat buffer)."""
        import os
        save_path = os.path.join(getattr(self.opt, "checkpoints_dir", "."), save_filename)
        torch.save({k: v.detach().clone().cpu() for k, v in network.state_dict().items()}, save_path)

    def update_learning_rate(self, ratio):
        """multimodal_classifier.py:267-277: lr = max(old_lr * ratio, 1e-5) (kitti/train_classifier.py:95-101 calls it with lr_decay_scale)."""
        lr_detector = max(self._lr() * ratio, 0.00001)
        if getattr(self, "_trainer_obj", None) is not None:
            self._trainer_obj.adam.lr = lr_detector
        print('update detector learning rate: %f -> %f' % (self.old_lr_detector, lr_detector))
        self.old_lr_detector = lr_detector

    def _record(self, L, prefix):
        setattr(self, prefix + "_loss_dict", {"loss": L["loss"], "coarse": L["coarse"], "fine": L["fine"]})
        setattr(self, prefix + "_accuracy", {"coarse_accuracy": L["coarse_accuracy"], "fine_accuracy": L["fine_accuracy"]})

    def optimize(self):
        """multimodal_classifier.py:213-218: train(), zero_grad, foraward_pass, backward, optimizer step."""
        self.detector.train()
        self._record(self._trainer().optimize(self.pc, self.intensity, self.sn, self.node_a, self.node_b, self.img, self.K, self.P), "train")

    def test_model(self):
        """multimodal_classifier.py:220-223: eval-mode losses and accuracies."""
        self._record(self._trainer().test_model(self.pc, self.intensity, self.sn, self.node_a, self.node_b, self.img, self.K, self.P), "test")
        self.detector.eval()

    def inference_labels(self):
        """Device-resident variant of inference_pass: coarse argmax as i32 [B,N] (what the solver consumes)."""
        out = self.forward(self.pc, self.intensity, self.sn, self.node_a, self.node_b, self.img)
        return ops.argmax_channels(out[0] if self.opt.is_fine_resolution else out)

    def inference_pass(self):
        out = self.forward(self.pc, self.intensity, self.sn, self.node_a, self.node_b, self.img)
        if self.opt.is_fine_resolution:
            coarse, fine = out
            return ops.argmax_channels(coarse).long(), ops.argmax_channels(fine).long()
        return ops.argmax_channels(out).long()


class MMClassiferCoarse(MMClassifer):
    """models/multimodal_classifier.py:380-469: coarse head only, inference_pass -> coarse_prediction."""

    fine = False

    def __init__(self, opt, writer=None):
        opt.is_fine_resolution = False
        super().__init__(opt, writer)
