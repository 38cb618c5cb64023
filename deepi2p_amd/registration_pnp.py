"""PnP registration front-end (BASELINE config 3) on the HIP RANSAC kernel.

solve_PnP                evaluation/registration_pnp.py:95-148   (same signature, same return pair)
camera_matrix_scaling    evaluation/registration_pnp.py:58-61

cv2.solvePnPRansac's sampling is internal and unseeded; here the RANSAC draws are an explicit input (or come from a
seeded numpy Generator), like the solver's restart list.  OpenCV is absent: parity unpinned.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, require_cuda, stream

SAMPLE_SIZE = 6


def camera_matrix_scaling(K, s):
    K_scale = s * np.asarray(K, dtype=np.float64)
    K_scale[2, 2] = 1
    return K_scale


def draw_samples(rng, F, iters):
    """RANSAC draws: i32[F, iters, 6] INDEPENDENT uniform integers; the device reduces each modulo the frame's correspondence
    count.  A sample may therefore repeat a correspondence (likely only for frames with a handful of them); such samples are
    rejected as degenerate by the estimators, which lowers the effective iteration count for those frames but is never wrong."""
    return rng.integers(0, 2 ** 30, size=(F, iters, SAMPLE_SIZE), dtype=np.int64).astype(np.int32)


def pack_correspondences(pc, coarse, fine, W_fine, pixels=None):
    """What solve_PnP hands to cv2.solvePnPRansac (registration_pnp.py:97-110,125-127), packed on the device.
    pc f32[F,3,N], coarse/fine i32[F,N] -> (corr f32[F,N,8] = {x, y, z, u, v, 0, 0, 0}, n_corr i32[F]); the first n_corr[f] records are valid."""
    require_cuda(pc, coarse, fine, pixels)
    F, _, N = pc.shape
    corr = torch.zeros((F, N, 8), dtype=torch.float32, device=pc.device)
    n_corr = torch.empty((F,), dtype=torch.int32, device=pc.device)
    call("di2p_pnp_pack", ptr(pc), ptr(coarse), ptr(fine), ptr(pixels), int(W_fine), F, N, ptr(corr), ptr(n_corr), stream())
    return corr, n_corr


def pnp_ransac(pc, coarse, fine, K_scaled, W_fine, samples, reproj_err=0.6, refine_rounds=20, refine_iters=5, pixels=None,
               method="epnp"):
    """Batched device entry.  pc f32[F,3,N], coarse/fine i32[F,N], K_scaled f64[F,3,3], samples i32[F,iters,6]
    -> dict(P f64[F,4,4], outlier_ratio f64[F], n_inliers, n_corr, best i32[F]).
    method "epnp" (default -- what the reference asks OpenCV for, registration_pnp.py:125-132): EPnP minimal-sample hypotheses (5 points,
    or 4 when a frame has only 4) + one EPnP re-fit on the inliers, the estimator of cv2.solvePnPRansac(flags=SOLVEPNP_EPNP);
    "dlt_lo": the builder's 6-point DLT hypotheses + locally optimised best model (more robust on cell-quantised observations, but not the
    reference's algorithm; refine_rounds / refine_iters apply to it only)."""
    require_cuda(pc, coarse, fine, K_scaled, samples, pixels)
    F, _, N = pc.shape
    iters = samples.shape[1]
    dev = pc.device
    P = torch.empty((F, 4, 4), dtype=torch.float64, device=dev)
    ratio = torch.empty((F,), dtype=torch.float64, device=dev)
    n_in = torch.empty((F,), dtype=torch.int32, device=dev)
    n_corr = torch.empty((F,), dtype=torch.int32, device=dev)
    best = torch.empty((F,), dtype=torch.int32, device=dev)
    ws = torch.empty((_lib.load().di2p_pnp_workspace_bytes(F, N, iters),), dtype=torch.uint8, device=dev)
    if method == "epnp":
        call("di2p_pnp_ransac_epnp", ptr(pc), ptr(coarse), ptr(fine), ptr(pixels), ptr(K_scaled), int(W_fine), ptr(samples), iters,
             float(reproj_err), F, N, ptr(P), ptr(ratio), ptr(n_in), ptr(n_corr), ptr(best), ptr(ws), stream())
    elif method == "dlt_lo":
        call("di2p_pnp_ransac", ptr(pc), ptr(coarse), ptr(fine), ptr(pixels), ptr(K_scaled), int(W_fine), ptr(samples), iters,
             float(reproj_err), int(refine_rounds), int(refine_iters), F, N, ptr(P), ptr(ratio), ptr(n_in), ptr(n_corr), ptr(best), ptr(ws), stream())
    else:
        raise ValueError("method must be 'epnp' or 'dlt_lo'")
    return dict(P=P, outlier_ratio=ratio, n_inliers=n_in, n_corr=n_corr, best=best)


def solve_PnP(pc_np, coarse_predictions_np, fine_predictions_np, K_np, H, W, fine_resolution_scale, iterationsCount,
              method=None, rng=None, samples=None):
    """Reference signature (registration_pnp.py:95-96).  `fine_resolution_scale` is the reference's 1/32 factor
    (it multiplies H, W and K, :101-104).  `method`: None / cv2.SOLVEPNP_EPNP (the reference's choice) -> the EPnP RANSAC;
    the string "dlt_lo" selects the builder's 6-point DLT + locally optimised variant."""
    if not torch.cuda.is_available():
        raise RuntimeError("deepi2p_amd.registration_pnp needs a HIP device (there is no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    Ws = W * fine_resolution_scale
    K_fine = camera_matrix_scaling(K_np, fine_resolution_scale)
    if int((np.asarray(coarse_predictions_np) == 1).sum()) < 4:          # :123 (points.shape[1] >= 4)
        return np.identity(4), 1
    if samples is None:
        rng = rng if rng is not None else np.random.default_rng()
        samples = draw_samples(rng, 1, int(iterationsCount))
    pc = torch.as_tensor(np.ascontiguousarray(pc_np, dtype=np.float32), device=dev).unsqueeze(0)
    co = torch.as_tensor(np.ascontiguousarray(coarse_predictions_np).astype(np.int32), device=dev).unsqueeze(0)
    fi = torch.as_tensor(np.ascontiguousarray(fine_predictions_np).astype(np.int32), device=dev).unsqueeze(0)
    Kt = torch.as_tensor(K_fine, device=dev).reshape(1, 3, 3)
    out = pnp_ransac(pc, co, fi, Kt, int(round(Ws)), torch.as_tensor(np.ascontiguousarray(samples, dtype=np.int32), device=dev),
                     method="dlt_lo" if method == "dlt_lo" else "epnp")
    ratio = float(out["outlier_ratio"][0])
    return out["P"][0].cpu().numpy(), (1 if ratio == 1.0 else ratio)
