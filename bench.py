#!/usr/bin/env python3
"""Headline benchmark: end-to-end frames/s of the DeepI2P registration hot path on MI355X.

One "step" = one batch of BASELINE.json configs[1]: 32 synthetic KITTI-shaped frames (20480 points,
160x512 image), coarse frustum classification (image + point + fusion network, fp32) -> argmax labels ->
initial guess -> 60-restart Gauss-Newton/LM pose solve (fp64) -> argmin.  Inputs are resident in HBM when
the timed region starts; weights are random-init closed-form (no checkpoints available).  N>1: one process
per GPU (torchrun), frames sharded across ranks, no data-path collective (weak scaling).

    python bench.py --gpus 1 --steps 10 --warmup 2
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA peak (same guide)
FP64_VALU_PEAK_TFLOPS = 78.6


def conv_flops_per_frame(H, W):
    """2*MACs of the ResNet-34 convolutions actually executed (models/resnet.py [3,4,6,3])."""
    def out(h, k, s, p):
        return (h + 2 * p - k) // s + 1
    total = 0
    h, w = out(H, 7, 2, 3), out(W, 7, 2, 3)
    total += 64 * 3 * 49 * h * w
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for bi in range(nb):
            s = 2 if (bi == 0 and li > 1) else 1
            oh, ow = out(h, 3, s, 1), out(w, 3, s, 1)
            total += planes * inpl * 9 * oh * ow + planes * planes * 9 * oh * ow
            if bi == 0 and li > 1:
                total += planes * inpl * oh * ow
            h, w, inpl = oh, ow, planes
    return 2 * total


def conv_bytes_per_frame(H, W):
    """Compulsory HBM bytes of the same convolutions: every input / residual read once, every output written once
    (fp32); the weights (85 MB, shared by the batch) are added once per step by the caller."""
    def out(h, k, s, p):
        return (h + 2 * p - k) // s + 1
    h, w = out(H, 7, 2, 3), out(W, 7, 2, 3)
    elems = 3 * H * W + 64 * h * w
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for bi in range(nb):
            s = 2 if (bi == 0 and li > 1) else 1
            oh, ow = out(h, 3, s, 1), out(w, 3, s, 1)
            elems += inpl * h * w + planes * oh * ow                 # conv1: in, out
            elems += planes * oh * ow * 3                            # conv2: in, residual, out
            if bi == 0 and li > 1:
                elems += inpl * h * w + planes * oh * ow             # 1x1 downsample: in, out
            h, w, inpl = oh, ow, planes
    return 4 * elems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=20480)
    ap.add_argument("--restarts", type=int, default=60)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams = batches in flight (1 = fully serial)")
    ap.add_argument("--priority", type=int, default=0, help="1: classifier on a high-priority stream, solves on --streams low-priority streams")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # DI2P_BENCH_BACKEND=gloo + DI2P_BENCH_ONE_DEVICE=1 let the multi-rank code path be exercised on a 1-GPU box
    backend = os.environ.get("DI2P_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    if os.environ.get("DI2P_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from deepi2p_amd import _lib, ops, synthetic
    from deepi2p_amd.networks import MMClassiferCoarse
    from deepi2p_amd.registration import RegistrationPipeline

    B, N, H, W, R = args.batch, args.points, 160, 512, args.restarts
    opt = synthetic.OptLike(N, H, W, False)
    opt.device = dev
    sd = synthetic.synthetic_state_dict(opt)
    mm = MMClassiferCoarse(opt)
    mm.detector.load_state_dict(sd)
    if world > 1:  # weights broadcast once over RCCL/xGMI (stands in for nn.DataParallel's per-step replicate)
        for t in mm.detector.state_dict().values():
            if backend == "nccl":
                dist.broadcast(t, 0)
            else:                       # gloo (test mode): via host memory
                h = t.cpu()
                dist.broadcast(h, 0)
                t.copy_(h)
        mm.detector._invalidate()
    batch = synthetic.make_batch(1000 + rank, B, N=N, H=H, W=W)
    t = {k: torch.from_numpy(batch[k]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    mm.set_input(t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], torch.zeros(B, 3, 4), t["img"],
                 torch.from_numpy(batch["K"]).float())
    K64 = torch.from_numpy(batch["K"]).to(dev)
    pipe = RegistrationPipeline(H, W, R=R, seed=rank)
    restarts = pipe.draw(B, dev)
    torch.cuda.synchronize()

    # The weights are random-init (no checkpoints exist here), so the network's argmax labels carry no frustum
    # information (typically "all outside", for which the reference skips the solver, registration_lsq.py:329-332).
    # Per SURVEY.md 8(d) the solver therefore consumes the synthetic labels = exact frustum labels with 5 % random
    # flips; the network forward + argmax still run in full inside the timed step and their output is kept.
    solver_labels = torch.from_numpy(batch["labels"]).to(dev)

    # S HIP streams, round-robin: whole steps (classifier -> pose solve of one batch) are independent, so several are
    # kept in flight.  The solver's tail (a few long-running hypotheses keep a handful of CUs busy while the rest of
    # the chip idles) is filled by the next batches' kernels.  Each step is still one full batch through the whole
    # path on its own stream, and all K steps complete inside the timed region.
    n_streams = max(1, args.streams)
    overlap = n_streams > 1
    prio = args.priority and overlap
    if prio:
        # one HIGH-priority stream for the classifier (short MFMA kernels) running ahead, S low-priority streams for
        # the pose solves: network workgroups are dispatched first whenever a CU frees up, the long fp64 solver
        # workgroups fill the rest and overlap each other's tails
        s_net = torch.cuda.Stream(priority=-1)
        streams = [torch.cuda.Stream(priority=0) for _ in range(n_streams)]
    else:
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
    step_no = [0]

    def step():
        if not overlap:
            pred = mm.inference_labels()                   # image + point + fusion network, argmax (i32 [B,N])
            o = pipe(mm.pc, solver_labels, K64, restarts)
            o["pred"] = pred
            return o
        st = streams[step_no[0] % n_streams]
        step_no[0] += 1
        if prio:
            with torch.cuda.stream(s_net):
                pred = mm.inference_labels()
                ev = torch.cuda.Event()
                ev.record()
            with torch.cuda.stream(st):
                st.wait_event(ev)                          # the pose solve of a batch follows its classification
                o = pipe(mm.pc, solver_labels, K64, restarts)
                pred.record_stream(st)
        else:
            with torch.cuda.stream(st):
                pred = mm.inference_labels()
                o = pipe(mm.pc, solver_labels, K64, restarts)   # same stream: the pose solve of a batch follows its classification
        o["pred"] = pred
        return o

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    dt = time.perf_counter() - t0
    # per-kernel-family durations: HIP events on the launch stream, in a SERIAL pass right after the timed region
    # (with several batches in flight, events inside the timed region would measure contention, not the kernels)
    timed_names = ("di2p_conv2d", "di2p_conv2d_ws", "di2p_pointwise_gemm", "di2p_point_head", "di2p_index_max_values", "di2p_solve_batched_f32", "di2p_knn_nodes")
    prof_steps = 2
    overlap_saved, overlap = overlap, False
    _lib.TIMED = {n: [] for n in timed_names}
    for _ in range(prof_steps):
        out = step()
    torch.cuda.synchronize()
    timed = _lib.TIMED
    _lib.TIMED = None
    overlap = overlap_saved
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    frames = B * args.steps * world
    ms_per_step = dt / args.steps * 1e3

    # ---- per-kernel-family time from the events recorded inside the timed region
    fam_ms = {n: sum(e0.elapsed_time(e1) for e0, e1, _ in v) / prof_steps for n, v in timed.items()}
    launches = {n: len(v) // prof_steps for n, v in timed.items()}
    # the convolution family = the plain entry point + the split-K one (stage-4 layers: K-slice kernel + ordered reduce pass)
    fam_ms["di2p_conv2d"] += fam_ms.pop("di2p_conv2d_ws")
    launches["di2p_conv2d"] += launches.pop("di2p_conv2d_ws")
    # pointwise family = the single-layer launches + the fused three-layer point head
    fam_ms["di2p_pointwise_gemm"] += fam_ms.pop("di2p_point_head")
    launches["di2p_pointwise_gemm"] += launches.pop("di2p_point_head")
    conv_flops = conv_flops_per_frame(H, W) * B
    idx_bytes = B * (4 * 32 * N + 4 * N + 2 * 4 * 32 * 128) + B * (4 * 64 * N + 4 * N + 2 * 4 * 64 * 128)
    iters = out["iters"].float()
    n_active = float((out["labels_front"] >= 0).float().sum(dim=1).mean())
    solver_sweeps = float(iters.sum()) * 1.0
    roofs = {
        "conv2d_kernel(implicit-GEMM fp32 MFMA)": {
            "bound": "mfma", "achieved": conv_flops / (fam_ms["di2p_conv2d"] * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "ms_per_step": fam_ms["di2p_conv2d"], "launches_per_step": launches["di2p_conv2d"]},
        "index_max_kernel": {
            "bound": "hbm", "achieved": idx_bytes / (fam_ms["di2p_index_max_values"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "ms_per_step": fam_ms["di2p_index_max_values"], "launches_per_step": launches["di2p_index_max_values"]},
        "solve_kernel(fp64 VALU, not hbm/mfma bound)": {
            "ms_per_step": fam_ms["di2p_solve_batched_f32"], "mean_iters": float(iters.mean()), "max_iters": float(iters.max()),
            "points_per_sweep": n_active, "sweeps_lower_bound": solver_sweeps},
        "pointwise_gemm_kernel(+point_head)": {"ms_per_step": fam_ms["di2p_pointwise_gemm"], "launches_per_step": launches["di2p_pointwise_gemm"]},
        "knn_nodes_kernel": {"ms_per_step": fam_ms["di2p_knn_nodes"], "launches_per_step": launches["di2p_knn_nodes"]},
    }
    for r in roofs.values():
        if "achieved" in r:
            r["frac"] = r["achieved"] / r["peak"]
    dom = roofs["conv2d_kernel(implicit-GEMM fp32 MFMA)"]
    # HBM traffic per convolution call from the committed PMC passes of the same kernels on the same shapes (separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH x2 for the 16-byte/lane loads as MI355X_MICROARCH.md prescribes;
    # profiles/r01_pmc_traffic.json); None if the file is absent
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            pmc = json.load(fh)
        if B == 32 and (H, W) == (160, 512):
            traffic = pmc["conv2d_resnet34_B32_160x512"]["hbm_bytes_per_call_corrected"]
            roofs["index_max_kernel"]["traffic_C64"] = pmc["index_max_C64_B32_N20480_K128"]["hbm_bytes_corrected"]
            roofs["index_max_kernel"]["algorithmic_C64"] = pmc["index_max_C64_B32_N20480_K128"]["algorithmic_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"kernel": "conv2d_kernel", "bound": "mfma", "achieved": dom["achieved"], "peak": dom["peak"],
                "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic,
                "algorithmic_flop_per_launch": conv_flops / 36.0,
                "compulsory_bytes_per_launch": (conv_bytes_per_frame(H, W) * B + 85.1e6) / 36.0,
                "note": "algorithmic 2*MAC of the 36 ResNet-34 convolution calls of one step / their summed HIP-event time "
                        "(events on the launch stream, serial pass of %d steps directly after the timed region)" % prof_steps}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(batch, sd, opt, H, W, R)

    if rank == 0:
        line = {
            "metric": "frames/sec (img+pc infer + 60-restart GN pose) KITTI 20k-pt, 1/2/4/8 GPU",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 network (fp32-input MFMA) + f64 solver",
            "data": "synthetic frames, random-init closed-form weights; solver labels = GT frustum labels with 5% flips (SURVEY 8d)",
            "config": {"workload": "BASELINE configs[1]: KITTI 20480-pt / 160x512, batch %d per GPU, coarse classification "
                                   "+ %d-restart 2D GN/LM solver, max_iter 500" % (B, R),
                       "frames_per_gpu_per_step": B, "points": N, "image": [H, W], "restarts": R, "parallelism": "dp%d" % world,
                       "streams": n_streams, "priority_streams": bool(prio)},
            "roofline": roofline, "kernels": roofs, "cpu_baseline": cpu_baseline,
            "pose_check": pose_check(out, batch),
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def pose_check(out, batch):
    """Sanity (not parity): the network has random weights, so its labels are meaningless; report only that the
    solver ran to completion on them."""
    from deepi2p_amd.registration import get_P_diff
    P = out["P"].cpu().numpy()
    errs = [get_P_diff(P[i], batch["P_gt"][i]) for i in range(P.shape[0])]
    ok = sum(1 for t, r in errs if t < 2.0 and r < 5.0)
    return {"mean_iters": float(out["iters"].float().mean()), "frames_with_inside_points": int((out["best"] >= 0).sum()),
            "frames_within_2m_5deg_of_gt": ok, "frames": int(P.shape[0]),
            "network_pred_inside_fraction": float((out["pred"] == 1).float().mean())}


def run_cpu_baseline(batch, sd, opt, H, W, R):
    """The oracle ("port" of the reference's CPU path) timed on this box's host cores on a bounded sample."""
    from oracle import frustum_lm as flm
    from oracle import network_torch as nt
    cores = os.cpu_count() or 1
    net_threads = min(cores, 32)      # torch CPU ops stop scaling (and regress) far below the box's 256 threads
    torch.set_num_threads(net_threads)
    nb = 2
    t = {k: torch.from_numpy(batch[k][:nb]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
    with torch.no_grad():
        nt.keypoint_detector(sd, opt, *[t[k][:1] for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")])   # warm-up
        t0 = time.perf_counter()
        logits = nt.keypoint_detector(sd, opt, t["pc"], t["intensity"], t["sn"], t["node_a"], t["node_b"], t["img"])
        t_net = (time.perf_counter() - t0) / nb
    labels = logits.argmax(1).numpy().astype(np.int32)
    pc = batch["pc"][0].astype(np.float64)
    lab = batch["labels"][0]          # same synthetic solver labels as the GPU leg
    _, y0, pcf, labf = flm.get_initial_guess(pc, lab)
    rng = np.random.default_rng(0)
    ys, Ts = flm.draw_restarts(rng, R, y0, 10 * math.pi / 180, 10)
    t0 = time.perf_counter()
    _, _, iters, _, _ = flm.solve_restarts(pcf, labf, batch["K"][0], ys, Ts, H, W, [-5, -0.1, -10], [5, 0.1, 10], 500, True, nthreads=cores)
    t_sol = time.perf_counter() - t0
    return {"value": 1.0 / (t_net + t_sol), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "network: %d frames (torch fp32, %d threads, %.2f s/frame); solver: 1 frame x %d restarts over %d threads "
                      "(%.2f s, mean %.1f LM iterations)" % (nb, net_threads, t_net, R, cores, t_sol, float(iters.mean()))}


if __name__ == "__main__":
    main()
