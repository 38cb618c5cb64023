"""Oracle: plain-PyTorch (CPU, fp32) restatement of the DeepI2P classification network.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Also timed as the CPU baseline
(``cpu_baseline.kind == "port"``) by bench.py.

Functional style: every function takes the reference's ``state_dict`` (same 361 keys,
see SURVEY.md section 5) and evaluates the eval-mode forward with stock torch ops and no
custom kernels.  It follows the reference's op graph so that, on CPU, it agrees with the
imported reference to float round-off:

  pointnet / equivariant layer   models/layers_pc.py:259-342, 345-408
  MyConv2d (1x1 conv+BN+ReLU)    models/layers_pc.py:110-190
  GeneralKNNFusionModule         models/layers_pc.py:779-818
  PCEncoder.forward              models/networks_pc.py:47-124
  ResNet-34 forward              models/resnet.py:56-72, 195-216
  ImageEncoder.forward           models/networks_img.py:26-28
  KeypointDetector.forward       models/networks_united.py:76-210
  inference_pass (argmax)        models/multimodal_classifier.py:100-117, 458-469

PINNED: tests/test_oracle_network.py compares it with the imported reference (when
/root/reference is present) and with tests/golden/network_*.npz (generated from the
imported reference by tests/golden/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm default, never overridden by the reference


def strip_module_prefix(sd):
    """util/pytorch_helper.py:24-33 -- accept DataParallel ('module.') or bare keys."""
    if all(k.startswith("module.") for k in sd):
        return {k[len("module."):]: v for k, v in sd.items()}
    return dict(sd)


# Train mode (models/multimodal_classifier.py:213-218 `self.detector.train()`): BatchNorm normalises with batch statistics and
# updates the running buffers of `sd` in place; per_point_pn applies its Dropout(0.5) layers (networks_united.py:57-74) with the
# keep-masks given here (torch's own dropout stream cannot be reproduced on the device).  Off by default: eval mode.
_TRAIN = {"on": False, "momentum": 0.1, "dropouts": None}


class train_mode:
    def __init__(self, momentum=0.1, dropouts=None):
        self.cfg = {"on": True, "momentum": momentum, "dropouts": dropouts}

    def __enter__(self):
        self.saved = dict(_TRAIN)
        _TRAIN.update(self.cfg)

    def __exit__(self, *a):
        _TRAIN.update(self.saved)


def _batch_norm(sd, p, x):
    if _TRAIN["on"]:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            True, _TRAIN["momentum"], BN_EPS)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _equivariant(sd, p, x):
    """Conv1d(k=1) -> [BN1d] -> [ReLU]; norm/act present iff the keys exist (layers_pc.py:325-342)."""
    x = F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    if (p + ".norm.weight") in sd:
        x = _batch_norm(sd, p + ".norm", x)
        x = F.relu(x)
    return x


def pointnet(sd, p, x, dropouts=None):
    """dropouts: per-layer keep-masks (train mode, nn.Dropout(0.5) after the activation: layers_pc.py:339-340)."""
    i = 0
    while (p + ".layers.%d.conv.weight" % i) in sd:
        x = _equivariant(sd, p + ".layers.%d" % i, x)
        if dropouts is not None and i < len(dropouts) and dropouts[i] is not None:
            x = x * dropouts[i].to(x.dtype) * 2.0
        i += 1
    return x


def _myconv2d(sd, p, x):
    x = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    x = _batch_norm(sd, p + ".norm", x)
    return F.relu(x)


def knn_fusion(sd, p, query, database, database_features, K):
    """GeneralKNNFusionModule.forward (layers_pc.py:779-818)."""
    B, M, N, C = query.size(0), query.size(2), database.size(2), database_features.size(1)
    q = query.unsqueeze(3)
    norm = torch.norm(q - database.unsqueeze(2), dim=1)                     # B x M x N
    _, knn_I = torch.topk(norm, k=K, dim=2, largest=False, sorted=True)
    idx3 = knn_I.unsqueeze(1).expand(B, 3, M, K).reshape(B, 3, M * K)
    idxC = knn_I.unsqueeze(1).expand(B, C, M, K).reshape(B, C, M * K)
    coord = torch.gather(database, 2, idx3).view(B, 3, M, K) - q
    feat = torch.gather(database_features, 2, idxC).view(B, C, M, K)
    y = torch.cat((coord, feat), dim=1)
    i = 0
    while (p + ".layers_before.%d.conv.weight" % i) in sd:
        y = _myconv2d(sd, p + ".layers_before.%d" % i, y)
        i += 1
    fmax, _ = torch.max(y, dim=3, keepdim=True)
    y = torch.cat((fmax.expand_as(y), y), dim=1)
    i = 0
    while (p + ".layers_after.%d.conv.weight" % i) in sd:
        y = _myconv2d(sd, p + ".layers_after.%d" % i, y)
        i += 1
    out, _ = torch.max(y, dim=3)
    return out, knn_I


def index_max_torch(data, index, K):
    """Segment arg-max with the semantics of index_max.cpp:73-112, vectorised in torch
    (first n attaining the max; floor -1000; empty cluster -> 0)."""
    B, C, N = data.shape
    neg = torch.full((B, C, K), -1000.0, dtype=data.dtype)
    idx = index.long().unsqueeze(1).expand(B, C, N)
    vmax = neg.scatter_reduce(2, idx, data, reduce="amax", include_self=True)   # B x C x K
    at_max = (data == torch.gather(vmax, 2, idx)) & (data > -1000.0)
    n_ids = torch.arange(N).view(1, 1, N).expand(B, C, N)
    cand = torch.where(at_max, n_ids, torch.full_like(n_ids, N))
    first = torch.full((B, C, K), N, dtype=torch.long).scatter_reduce(2, idx, cand, reduce="amin",
                                                                     include_self=True)
    return torch.where(first == N, torch.zeros_like(first), first)


def pc_encoder(sd, opt, pc, intensity, sn, node_a, node_b, p="pc_encoder"):
    """PCEncoder.forward (networks_pc.py:47-124) -> the reference's 8-tuple."""
    B, N, Ma = pc.size(0), pc.size(2), node_a.size(2)
    diff = torch.norm(pc.unsqueeze(3) - node_a.unsqueeze(2), dim=1, p=2)      # B x N x Ma
    _, min_k_idx = torch.topk(diff, k=opt.k_interp_point_a, dim=2, largest=False, sorted=True)
    min_idx = min_k_idx[:, :, 0]
    mask = torch.eq(min_idx.unsqueeze(2), torch.arange(Ma).view(1, 1, Ma))    # B x N x Ma
    mask_row_max = mask.any(dim=1).unsqueeze(1).float()                       # B x 1 x Ma
    mask_f = mask.unsqueeze(1).float()
    mask_row_sum = mask_f.sum(dim=2)
    cluster_mean = torch.sum(pc.unsqueeze(3) * mask_f, dim=2) / (mask_row_sum + 1e-5)
    pc_centers = torch.gather(cluster_mean, 2, min_idx.unsqueeze(1).expand(B, 3, N))
    x = torch.cat((pc - pc_centers, intensity, sn), dim=1)
    first = pointnet(sd, p + ".first_pointnet", x)
    gi = index_max_torch(first, min_idx, Ma)
    first_max = first.gather(2, gi) * mask_row_max
    scattered = torch.gather(first_max, 2, min_idx.unsqueeze(1).expand(B, first.size(1), N))
    second = pointnet(sd, p + ".second_pointnet", torch.cat((first, scattered), dim=1))
    gi2 = index_max_torch(second, min_idx, Ma)
    node_a_features = second.gather(2, gi2) * mask_row_max
    node_b_features, _ = knn_fusion(sd, p + ".knnlayer", node_b, cluster_mean, node_a_features, opt.k_ab)
    final = pointnet(sd, p + ".final_pointnet", torch.cat((node_b, node_b_features), dim=1))
    global_feature, _ = torch.max(final, dim=2, keepdim=True)
    return (pc_centers, cluster_mean, min_k_idx, first, second,
            node_a_features, node_b_features, global_feature)


def _conv_bn(sd, pc, pb, x, stride, padding, relu):
    x = F.conv2d(x, sd[pc + ".weight"], None, stride=stride, padding=padding)
    x = _batch_norm(sd, pb, x)
    return F.relu(x) if relu else x


def resnet34(sd, x, p="img_encoder.backbone"):
    """ResNet.forward (resnet.py:195-216) with BasicBlock (:56-72), layers [3,4,6,3]."""
    outs = []
    x = _conv_bn(sd, p + ".conv1", p + ".bn1", x, 2, 3, True)
    outs.append(x)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, nblocks in enumerate((3, 4, 6, 3), start=1):
        for bi in range(nblocks):
            q = "%s.layer%d.%d" % (p, li, bi)
            stride = 2 if (bi == 0 and li > 1) else 1
            identity = x
            y = _conv_bn(sd, q + ".conv1", q + ".bn1", x, stride, 1, True)
            y = _conv_bn(sd, q + ".conv2", q + ".bn2", y, 1, 1, False)
            if (q + ".downsample.0.weight") in sd:
                identity = _conv_bn(sd, q + ".downsample.0", q + ".downsample.1", x, stride, 0, False)
            x = F.relu(y + identity)
        outs.append(x)
    outs.append(F.adaptive_avg_pool2d(x, (1, 1)))
    return outs


def image_encoder(sd, img):
    o = resnet34(sd, img)
    return o[3], o[4], o[5]


def _gather_topk(idx, feats):
    B, N, k = idx.shape
    C, M = feats.size(1), feats.size(2)
    return torch.gather(feats.unsqueeze(3).expand(B, C, M, k), 2, idx.unsqueeze(1).expand(B, C, N, k))


def upsample_by_interpolation(idx, query, nodes, node_feats):
    """networks_united.py:90-103 -- NB the weights 1 - d/sum(d) sum to k-1 = 2, not 1."""
    nb = _gather_topk(idx, nodes)
    d = torch.norm(query.unsqueeze(3) - nb, dim=1, p=2)
    w = 1 - d / torch.sum(d, dim=2, keepdim=True)
    return torch.sum(w.unsqueeze(1) * _gather_topk(idx, node_feats), dim=3)


def keypoint_detector(sd, opt, pc, intensity, sn, node_a, node_b, img, return_intermediates=False):
    """KeypointDetector.forward (networks_united.py:105-210)."""
    B, N, Ma, Mb = pc.size(0), pc.size(2), node_a.size(2), node_b.size(2)
    (pc_center, cluster_mean, a_min_k_idx, first, second,
     node_a_features, node_b_features, global_feature) = pc_encoder(sd, opt, pc, intensity, sn, node_a, node_b)
    s16, s32, iglob = image_encoder(sd, img)
    C_img = iglob.size(1)
    s16f = s16.reshape(B, s16.size(1), -1)
    s32f = s32.reshape(B, s32.size(1), -1)
    ig_a = iglob.squeeze(3).expand(B, C_img, Ma)
    ig_b = iglob.squeeze(3).expand(B, C_img, Mb)

    score_b = pointnet(sd, "node_b_attention_pn", torch.cat((node_b_features, ig_b), dim=1))   # B x HW32 x Mb
    # mean over HW of feat[b,c,hw]*score[b,hw,m]  ==  (feat @ score) / HW   (networks_united.py:147-150)
    w_s32 = torch.bmm(s32f, score_b) / s32f.size(2)
    up_b = pointnet(sd, "node_b_pn", torch.cat((node_b_features, global_feature.expand(B, -1, Mb),
                                                 w_s32, ig_b), dim=1))
    d_pb = torch.norm(pc.unsqueeze(3) - node_b.unsqueeze(2), p=2, dim=1)
    _, pb_idx = torch.topk(d_pb, k=opt.k_interp_point_b, dim=2, largest=False, sorted=True)
    interp_pb = upsample_by_interpolation(pb_idx, pc, node_b, up_b)

    score_a = pointnet(sd, "node_a_attention_pn", torch.cat((node_a_features, ig_a), dim=1))
    w_s16 = torch.bmm(s16f, score_a) / s16f.size(2)
    d_ab = torch.norm(node_a.unsqueeze(3) - node_b.unsqueeze(2), p=2, dim=1)
    _, ab_idx = torch.topk(d_ab, k=opt.k_interp_ab, dim=2, largest=False, sorted=True)
    interp_ab = upsample_by_interpolation(ab_idx, node_a, node_b, up_b)
    up_a = pointnet(sd, "node_a_pn", torch.cat((node_a_features, interp_ab, w_s16), dim=1))
    interp_pa = upsample_by_interpolation(a_min_k_idx, pc, node_a, up_a)

    scores = pointnet(sd, "per_point_pn", torch.cat((interp_pa, interp_pb, first, second), dim=1),
                      _TRAIN["dropouts"] if _TRAIN["on"] else None)
    coarse = scores[:, 0:2, :]
    fine = scores[:, 2:, :] if opt.is_fine_resolution else None
    if return_intermediates:
        inter = dict(pc_center=pc_center, cluster_mean=cluster_mean, a_min_k_idx=a_min_k_idx,
                     first_pn_out=first, second_pn_out=second, node_a_features=node_a_features,
                     node_b_features=node_b_features, global_feature=global_feature,
                     s16=s16, s32=s32, img_global=iglob, score_b=score_b, w_s32=w_s32, up_b=up_b,
                     pb_idx=pb_idx, interp_pb=interp_pb, score_a=score_a, w_s16=w_s16, ab_idx=ab_idx,
                     interp_ab=interp_ab, up_a=up_a, interp_pa=interp_pa)
        return coarse, fine, inter
    return (coarse, fine) if opt.is_fine_resolution else coarse


def inference_pass(sd, opt, pc, intensity, sn, node_a, node_b, img):
    """multimodal_classifier.py:100-117 / :458-469 -- argmax, first max wins."""
    out = keypoint_detector(sd, opt, pc, intensity, sn, node_a, node_b, img)
    if opt.is_fine_resolution:
        return torch.max(out[0], dim=1)[1], torch.max(out[1], dim=1)[1]
    return torch.max(out, dim=1)[1]


# closed-form ("synthetic") weights + option bag: data generators shared with the product's bench/tests live in
# deepi2p_amd/synthetic.py (no compute there); re-exported here for the fixture generator and the tests.
from deepi2p_amd.synthetic import OptLike, random_state_dict, state_dict_spec, synthetic_state_dict  # noqa: E402,F401
