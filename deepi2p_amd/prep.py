"""Input preparation and label hand-off on the device ("next" rows).

FarthestSampler.sample             data/kitti_helper.py:224-243
node sampling of __getitem__       data/kitti_pc_img_pose_loader.py:416-423
downsample_np index gather         data/kitti_pc_img_pose_loader.py:158-171
GT label projection / accuracies   evaluation/visualize_and_save_data.py:100-147
7 x N pc_label hand-off record     evaluation/visualize_and_save_data.py:174-186 -> evaluation/registration_lsq.py:291-302
"""
import numpy as np
import torch

from . import ops
from ._lib import call, ptr, require_cuda, stream


def farthest_point_sampling(pts, k, init_idx=None):
    """pts f32[B,3,M] (device) -> (nodes f32[B,3,k], idx i32[B,k])."""
    require_cuda(pts, init_idx)
    B, _, M = pts.shape
    idx = torch.empty((B, k), dtype=torch.int32, device=pts.device)
    nodes = torch.empty((B, 3, k), dtype=torch.float32, device=pts.device)
    call("di2p_farthest_point_sampling", ptr(pts), ptr(init_idx), ptr(idx), ptr(nodes), B, M, int(k), stream())
    return nodes, idx


class FarthestSampler:
    """Drop-in for data/kitti_helper.py FarthestSampler (numpy in / numpy out, 3-D points)."""

    def __init__(self, dim=3):
        if dim != 3:
            raise NotImplementedError("only dim=3 is on the registration path")
        self.dim = dim

    def sample(self, pts, k, init_idx=None):
        # the reference draws np.random.randint(len(pts)) with pts of shape (3, M), i.e. an index in {0,1,2}
        if init_idx is None:
            init_idx = np.random.randint(len(pts))
        dev = torch.device("cuda", torch.cuda.current_device())
        t = torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32), device=dev).unsqueeze(0)
        ii = torch.tensor([int(init_idx)], dtype=torch.int32, device=dev)
        nodes, idx = farthest_point_sampling(t, k, ii)
        return nodes[0].double().cpu().numpy(), idx[0].cpu().numpy().astype(np.int64)


def gather_points(src, idx):
    """out[b,c,n] = src[b,c,idx[b,n]] ; src f32[B,C,Nsrc], idx i32[B,Nout]."""
    require_cuda(src, idx)
    B, C, Nsrc = src.shape
    Nout = idx.shape[1]
    out = torch.empty((B, C, Nout), dtype=torch.float32, device=src.device)
    call("di2p_gather_points", ptr(src), ptr(idx), ptr(out), B, C, Nsrc, Nout, stream())
    return out


def sample_nodes(pc, node_num, cand_idx, init_idx=None):
    """node_a / node_b of one batch: FPS over the `cand_idx` (i32[B, 8*node_num], host-drawn random subset) of pc."""
    cand = gather_points(pc, cand_idx)
    nodes, _ = farthest_point_sampling(cand, node_num, init_idx)
    return nodes


def project_labels(pc, P, K, H, W, fine_scale=32, want_pxpy=False):
    """-> (coarse i32[B,N], fine i32[B,N][, pxpy f32[B,2,N]])."""
    require_cuda(pc, P, K)
    B, _, N = pc.shape
    coarse = torch.empty((B, N), dtype=torch.int32, device=pc.device)
    fine = torch.empty((B, N), dtype=torch.int32, device=pc.device)
    pxpy = torch.empty((B, 2, N), dtype=torch.float32, device=pc.device) if want_pxpy else None
    call("di2p_project_labels", ptr(pc), ptr(P), P.shape[1], ptr(K), float(H), float(W), float(fine_scale), ptr(coarse),
         ptr(fine), ptr(pxpy), B, N, stream())
    return (coarse, fine, pxpy) if want_pxpy else (coarse, fine)


def label_accuracy(coarse_pred, coarse_gt, fine_pred=None, fine_gt=None):
    """-> f32[B,2] = (coarse accuracy, fine accuracy over gt-inside points)."""
    require_cuda(coarse_pred, coarse_gt, fine_pred, fine_gt)
    B, N = coarse_pred.shape
    out = torch.empty((B, 2), dtype=torch.float32, device=coarse_pred.device)
    call("di2p_label_accuracy", ptr(coarse_pred), ptr(coarse_gt), ptr(fine_pred), ptr(fine_gt), ptr(out), B, N, stream())
    return out


def pack_pc_label(pc, coarse_pred, coarse_gt, fine_pred=None, fine_gt=None):
    """-> f64[B,7,N]: the in-memory equivalent of the reference's %06d_%02d_pc_label.npy."""
    require_cuda(pc, coarse_pred, coarse_gt, fine_pred, fine_gt)
    B, _, N = pc.shape
    out = torch.empty((B, 7, N), dtype=torch.float64, device=pc.device)
    call("di2p_pack_pc_label", ptr(pc), ptr(coarse_pred), ptr(coarse_gt), ptr(fine_pred), ptr(fine_gt), ptr(out), B, N, stream())
    return out


def random_choice(seed, B, n_src, n_out, device, stream_id=0):
    """-> i32[B, n_out]: n_out of n_src indices without replacement, uniformly random subset in uniformly random order, drawn
    ON THE DEVICE from (seed, stream_id, frame, index) (np.random.choice(n_src, n_out, replace=False) of the loaders)."""
    from . import _lib
    out = torch.empty((B, n_out), dtype=torch.int32, device=device)
    ws = torch.empty((_lib.load().di2p_random_choice_workspace_bytes(B, n_src),), dtype=torch.uint8, device=device)
    call("di2p_random_choice", int(seed), int(stream_id), B, int(n_src), int(n_out), ptr(out), ptr(ws), stream())
    return out


def downsample(pc, intensity, sn, input_pt_num, seed):
    """Device counterpart of KittiLoader.downsample_np (data/kitti_pc_img_pose_loader.py:158-171) for a batch of equally long
    raw scans: pc f32[B,3,Nraw] (+ intensity [B,1,Nraw], sn [B,3,Nraw]) -> the same with input_pt_num points.
    Nraw >= input_pt_num: a random subset; else every point floor(input_pt_num / Nraw) times plus a random remainder."""
    require_cuda(pc, intensity, sn)
    B, _, Nraw = pc.shape
    if Nraw >= input_pt_num:
        idx = random_choice(seed, B, Nraw, input_pt_num, pc.device)
    else:
        reps = input_pt_num // Nraw
        fix = torch.arange(Nraw, dtype=torch.int32, device=pc.device).repeat(reps).unsqueeze(0).expand(B, -1)
        rem = input_pt_num - reps * Nraw
        idx = fix if rem == 0 else torch.cat((fix, random_choice(seed, B, Nraw, rem, pc.device)), dim=1)
        idx = idx.contiguous()
    return gather_points(pc, idx), gather_points(intensity, idx), gather_points(sn, idx), idx


def sample_nodes_device(pc, node_num, seed, stream_id):
    """node_a / node_b entirely on the device: node_num * 8 random candidates (:416-423) + FPS from candidate 0."""
    B, _, N = pc.shape
    cand_idx = random_choice(seed, B, N, min(N, node_num * 8), pc.device, stream_id=stream_id)
    return sample_nodes(pc, node_num, cand_idx)
