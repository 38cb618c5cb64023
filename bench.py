#!/usr/bin/env python3
"""Headline benchmark: end-to-end frames/s of the DeepI2P registration hot path on MI355X.

One "step" = one batch of BASELINE.json configs[1]: 32 synthetic KITTI-shaped frames (20480 points,
160x512 image), coarse frustum classification (image + point + fusion network, fp32) -> argmax labels ->
initial guess -> 60-restart Gauss-Newton/LM pose solve (fp64) -> argmin.  Inputs are resident in HBM when
the timed region starts (`value`); the same loop with the per-step H2D copy of the batch from pinned host
memory inside the step (async copies on a second stream per slot) is timed right after it and reported as `value_with_h2d` (SURVEY.md 8d's definition).
Weights are random-init closed-form (no checkpoints available).

Multi-GPU: one process per GPU over RCCL.  `python bench.py --gpus N` launches the N ranks itself
(torch.distributed.run, 127.0.0.1 rendezvous) unless it already runs under a launcher (WORLD_SIZE set).
  --mode frames (default)  frames sharded across ranks, no data-path collective, weak scaling (configs[1]/[3])
  --mode hyp               BASELINE configs[4]: every rank classifies the same frames, the R=256 pose hypotheses
                           of each frame are sharded across ranks, ONE all_gather of (cost, params) + identical
                           argmin on every rank; strong scaling, collective latency reported

    python bench.py --gpus 1 --steps 12 --warmup 3
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# One hardware queue per HIP stream: the runtime's default of 4 makes stream counts above 3 share queues (4 streams are slower than 3,
# see DESIGN.md section 4); must be set before the HIP runtime initialises (torch is imported inside main()).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA peak (same guide)
MFMA_BF16_PEAK_TFLOPS = 2516.6  # dense bf16 MFMA peak (same guide: 16 x the fp32-input rate)
FP64_VALU_PEAK_TFLOPS = 78.6   # fp64 vector peak (256 CUs x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
SOLVER_FLOP_PER_POINT = 150.0  # flop per point per hypothesis-iteration (rotate, project, residual, J, J^T J)
# reference-algorithmic MACs per frame of the pointwise (Conv1d / 1x1 Conv2d) layers, SURVEY.md 8(d) [probed]:
# knnlayer, per_point_pn (coarse), node_b_pn, second_pointnet, node_a_pn, first_pointnet, attention PNs, final_pointnet
POINTWISE_REF_GMAC_COARSE = 0.975 + 2.270 + 0.336 + 0.168 + 0.065 + 0.047 + 0.057 + 0.025


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 48 timed steps (0.3 s) behind 8 warm-up steps -- at 24 / 4 the ratio of the two timed loops moved by +-2 % from run to run
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--mode", choices=("frames", "hyp", "train"), default="frames",
                    help="frames: the headline (BASELINE configs[1]); hyp: configs[4], hypotheses sharded over the ranks; "
                         "train: one optimisation step of the classifier (SURVEY 8f rank 4), data-parallel with one gradient all-reduce")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (frames: 32, hyp: 16 in total)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--restarts", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-graph", action="store_true", help="--mode train, one GPU: forward + losses + backward of a step replayed from one hipGraph (ClassifierTrainer.optimize_graphed)")
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams = batches in flight (1 = fully serial); default 8 for the frames mode (with GPU_MAX_HW_QUEUES=16, "
                         "set below unless the environment already has it), 3 otherwise")
    ap.add_argument("--no-h2d-pass", action="store_true", help="skip the second timed loop with the H2D copy inside the step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one captured hipGraph per stream")
    ap.add_argument("--split-solver", action="store_true", help="experiment: classifier and pose solve of a step as two graphs on a high- and a normal-priority stream")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="only launch the ranks, rendezvous, run one barrier + all_reduce and print n_gpus (no GPU work)")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_spawn(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def init_dist(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # DI2P_BENCH_BACKEND=gloo + DI2P_BENCH_ONE_DEVICE=1 let the multi-rank code path be exercised on a 1-GPU box
    backend = os.environ.get("DI2P_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    if os.environ.get("DI2P_BENCH_ONE_DEVICE"):
        local_rank = 0
    elif not args.launch_selftest:
        # one process per GPU: a rank without a device of its own would silently share (or fail much later inside RCCL)
        ndev = torch.cuda.device_count()
        if local_rank >= ndev:
            raise SystemExit("bench.py: local rank %d of %d ranks, but this node exposes %d GPU(s) -- start at most one rank per GPU "
                             "(or set DI2P_BENCH_ONE_DEVICE=1 with DI2P_BENCH_BACKEND=gloo to exercise the multi-rank path on one device)"
                             % (local_rank, world, ndev))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and not args.launch_selftest:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
    return rank, local_rank, world, backend, dist


def launch_selftest(args):
    """Rendezvous + one barrier + one all_reduce on the configured backend; no HIP kernels (CPU-testable with gloo)."""
    import torch
    rank, local_rank, world, backend, dist = init_dist(args)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        if backend == "nccl":
            t = t.cuda()
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_selftest": True, "n_gpus": world, "requested_gpus": args.gpus, "backend": backend if world > 1 else None,
                          "max_rank_plus_one": float(t.item()),
                          # what the ranks run with: the launcher only fills these in when the caller's environment has not set them
                          "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "MASTER_ADDR", "GPU_MAX_HW_QUEUES")}}))


def broadcast_weights(detector, dist, backend):
    """Weights broadcast ONCE as one flat fp32 buffer over RCCL/xGMI (stands in for nn.DataParallel's per-step replicate,
    models/multimodal_classifier.py:37-38): 26 M parameters = 105 MB, one collective instead of 361."""
    import torch
    tensors = [t for t in detector.state_dict().values() if t.dtype == torch.float32]
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if backend == "nccl":
        dist.broadcast(flat, 0)
    else:                       # gloo (test mode): via host memory
        h = flat.cpu()
        dist.broadcast(h, 0)
        flat.copy_(h)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    detector._invalidate()
    return flat.numel() * 4


def main():
    args = parse_args()
    maybe_spawn(args)
    if args.launch_selftest:
        launch_selftest(args)
        return
    if args.mode == "train":
        main_train(args)
        return
    import numpy as np
    import torch
    rank, local_rank, world, backend, dist = init_dist(args)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from deepi2p_amd import _lib, ops, synthetic
    from deepi2p_amd.distributed import shard_range, solve_hypotheses_sharded
    from deepi2p_amd.networks import MMClassiferCoarse
    from deepi2p_amd.registration import RegistrationPipeline

    hyp = args.mode == "hyp"
    if hyp:      # BASELINE configs[4]: Oxford-shaped 40960 pts / 384x640, 256 hypotheses per frame, same frames on every rank
        B, N, H, W, R = args.batch or 16, args.points or 40960, 384, 640, args.restarts or 256
    else:        # BASELINE configs[1]
        B, N, H, W, R = args.batch or 32, args.points or 20480, 160, 512, args.restarts or 60
    opt = synthetic.OptLike(N, H, W, False)
    opt.device = dev
    sd = synthetic.synthetic_state_dict(opt)
    mm = MMClassiferCoarse(opt)
    mm.detector.load_state_dict(sd)
    bcast_bytes = broadcast_weights(mm.detector, dist, backend) if world > 1 else 0
    mm.detector.prepack()                 # derive the kernel operands now, on this stream, before other streams use them
    batch = synthetic.make_batch(1000 + (0 if hyp else rank), B, N=N, H=H, W=W)
    names = ("pc", "intensity", "sn", "node_a", "node_b", "img")
    host = {k: torch.from_numpy(batch[k]).pin_memory() for k in names}
    mm.set_input(host["pc"], host["intensity"], host["sn"], host["node_a"], host["node_b"], torch.zeros(B, 3, 4), host["img"],
                 torch.from_numpy(batch["K"]).float())
    K64 = torch.from_numpy(batch["K"]).to(dev)
    pipe = RegistrationPipeline(H, W, R=R, seed=0 if hyp else rank)
    restarts = pipe.draw(B, dev)
    torch.cuda.synchronize()

    # The weights are random-init (no checkpoints exist here), so the network's argmax labels carry no frustum
    # information (typically "all outside", for which the reference skips the solver, registration_lsq.py:329-332).
    # Per SURVEY.md 8(d) the solver therefore consumes the synthetic labels = exact frustum labels with 5 % random
    # flips; the network forward + argmax still run in full inside the timed step and their output is kept.
    solver_labels = torch.from_numpy(batch["labels"]).to(dev)

    # The stream / graph executor is the product's (deepi2p_amd/pipeline.py): S HIP streams, one slot of input buffers per stream, a
    # whole step (H2D copies -> classifier -> argmax -> initial guess -> R-restart solve -> argmin) captured once per slot as ONE
    # hipGraph and replayed; every step is still one full batch through the whole path on its own stream, and all K steps complete
    # inside the timed region.  This file only feeds it, times it and prices the kernels.
    from deepi2p_amd.pipeline import RegistrationExecutor
    n_streams = max(1, args.streams if args.streams is not None else (3 if hyp else 8))
    coll_events = []
    step_fn = post_fn = None
    if hyp:
        # hypothesis fan-out (BASELINE configs[4]): rank r solves restarts [lo, hi) of every frame -- that part is the captured graph --
        # then ONE all_gather of (cost, params) and the identical argmin on every rank, launched eagerly behind the graph
        lo, hi = shard_range(R, rank, world)

        def step_fn(slot, d):
            pred = ops.argmax_channels(mm.detector(d["pc"], d["intensity"], d["sn"], d["node_a"], d["node_b"], d["img"]))
            pts64 = torch.empty((B, 3, N), dtype=torch.float64, device=dev)
            ops.call("di2p_f32_to_f64", ops.ptr(d["pc"]), ops.ptr(pts64), B * 3 * N, ops.stream())
            yaw0, lab_front, has_inside = ops.initial_guess(pts64, solver_labels)
            p, c, it = ops.solve_batched(d["pc"], lab_front, K64, restarts[0][:, lo:hi].contiguous(), restarts[1][:, lo:hi].contiguous(),
                                         H, W, pipe.lb, pipe.ub, pipe.max_iter, True, yaw0=yaw0)
            return dict(pred=pred, yaw0=yaw0, labels_front=lab_front, has_inside=has_inside, params_local=p, cost_local=c, iters=it)

        def post_fn(slot, o):
            ev = None
            if coll_events is not None and world > 1 and backend == "nccl":
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            best, bp, bc, allc = solve_hypotheses_sharded(lambda iy, iT: (o["params_local"], o["cost_local"]), restarts[0], restarts[1],
                                                          gather_events=ev)
            if ev is not None:
                coll_events.append(ev)
            _, P, _ = ops.select_best(bp.view(B, 1, -1).contiguous(), bc.view(B, 1).contiguous(), True, has_inside=o["has_inside"])
            return dict(o, P=P, cost=bc, best=best.int(), costs=allc)
    ex = RegistrationExecutor(mm, pipe, K64, host, n_streams=n_streams, use_graph=not args.no_graph, restarts=restarts,
                              labels_override=solver_labels, step_fn=step_fn, post_fn=post_fn, h2d_mode=os.environ.get("DI2P_H2D_MODE", "copy_stream"), split_solver=args.split_solver, double_buffer=bool(os.environ.get("DI2P_DOUBLE_BUFFER")))

    def barrier():
        if world > 1:
            dist.barrier()

    rank_ms = {}

    def timed_loop(with_h2d):
        dt, o, lat = ex.throughput(args.steps, args.warmup, with_h2d, barrier=barrier)
        if world > 1:
            # the contract's time is the MAX over ranks; every rank's own time is kept too, so that a scaling run explains itself
            mine = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rank_ms["with_h2d" if with_h2d else "resident"] = [float(t.item()) / args.steps * 1e3 for t in every]
            dt = max(float(t.item()) for t in every)
        return dt, o, lat

    # the two timed loops run back to back on a chip whose clock follows its power / thermal state: DI2P_BENCH_H2D_FIRST=1 swaps their order
    # (a probe: how much of `value_with_h2d / value` is the second loop running on a warmer chip)
    dt_h2d = lat_h2d = None
    if os.environ.get("DI2P_BENCH_H2D_FIRST") and not args.no_h2d_pass:
        dt_h2d, _, lat_h2d = timed_loop(True)
        dt, out, lat = timed_loop(False)
    else:
        dt, out, lat = timed_loop(False)
        if not args.no_h2d_pass:
            dt_h2d, _, lat_h2d = timed_loop(True)
    use_graph = ex.use_graph
    if ex.graph_error:
        print("hipGraph capture failed (%s); ran eagerly" % ex.graph_error, file=sys.stderr)
    # per-batch latency with ONE step in flight (the other end of the throughput / latency trade: `value` keeps n_streams in flight)
    lat1 = None
    if not hyp and n_streams > 1:
        ex.synchronize()
        lat1 = []
        for _ in range(3):
            tk = ex.submit(None, with_h2d=False)
            ex.result(tk)
            lat1.append(ex.latency_ms(tk))

    # per-kernel-family durations: HIP events on the launch stream, in a SERIAL pass right after the timed region
    # (with several batches in flight, events inside the timed region would measure contention, not the kernels)
    timed_names = ("di2p_conv2d", "di2p_conv2d_ws", "di2p_conv3x3_winograd", "di2p_conv3x3_x3", "di2p_conv7x7s2_stem", "di2p_stem_x3", "di2p_pointwise_gemm", "di2p_pointwise_gemm_x3", "di2p_pointwise_gemm_x3p", "di2p_point_head", "di2p_point_head_x3", "di2p_point_chain", "di2p_index_max_values",
                   "di2p_solve_batched_f32", "di2p_knn_nodes")
    prof_steps = 2
    _lib.TIMED = {n: [] for n in timed_names}
    _lib.WORK = {}
    coll_events = None
    for _ in range(prof_steps):
        out = ex.step_eager(0, False)
    torch.cuda.synchronize()
    timed, work = _lib.TIMED, _lib.WORK
    _lib.TIMED = _lib.WORK = None
    frames_per_step = B if hyp else B * world
    frames = frames_per_step * args.steps
    ms_per_step = dt / args.steps * 1e3

    fam_ms = {n: sum(e0.elapsed_time(e1) for e0, e1, _ in v) / prof_steps for n, v in timed.items()}
    launches = {n: len(v) // prof_steps for n, v in timed.items()}
    # the convolution family = the plain entry point + the split-K one (K-slice kernel + ordered reduce pass)
    # and the fused Winograd kernel that runs the 3x3 stride-1 layers
    direct_ms, wino_ms, wino_calls = fam_ms["di2p_conv2d"] + fam_ms["di2p_conv2d_ws"], fam_ms["di2p_conv3x3_winograd"], launches["di2p_conv3x3_winograd"]
    stem_ms = fam_ms["di2p_conv7x7s2_stem"] + fam_ms["di2p_stem_x3"]      # (di2p_stem_x3: conv1 + bn1 + relu AND the max-pool, one launch)
    # ... and the 3x3 layers that run as direct convolutions on the bf16 matrix instructions with exact three-way fp32 splits (conv_x3.hip)
    # (the stem on the same instructions, di2p_stem_x3, is priced with them: by its 147 algorithmic taps per output -- it executes 176, the
    #  kx pad and the 22nd (channel, ky) pair -- and its time includes the max-pool it carries)
    cx_ms, cx_calls = fam_ms["di2p_conv3x3_x3"] + fam_ms["di2p_stem_x3"], launches["di2p_conv3x3_x3"] + launches["di2p_stem_x3"]
    cx_mac = (work.get("di2p_conv3x3_x3", 0) + work.get("di2p_stem_x3", 0)) / prof_steps
    for n in ("di2p_conv2d_ws", "di2p_conv3x3_winograd", "di2p_conv7x7s2_stem", "di2p_stem_x3", "di2p_conv3x3_x3"):
        fam_ms["di2p_conv2d"] += fam_ms.pop(n)
        launches["di2p_conv2d"] += launches.pop(n)
    wino_exec_flops = 2.0 * work.get("di2p_conv3x3_winograd", 0) / prof_steps
    # pointwise family = the single-layer launches + the fused three-layer point head
    # ... and the GEMM-shaped layers that run on the bf16 matrix instructions with exact three-way fp32 splits (priced separately below)
    # (di2p_pointwise_gemm_x3p = the same kernel fed with the split planes of the previous bf16x3 layer; its work is booked with the other's)
    x3_ms, x3_calls = fam_ms.pop("di2p_pointwise_gemm_x3") + fam_ms.pop("di2p_pointwise_gemm_x3p"), launches.pop("di2p_pointwise_gemm_x3") + launches.pop("di2p_pointwise_gemm_x3p")
    x3_mac = work.get("di2p_pointwise_gemm_x3", 0) / prof_steps
    chain_ms, chain_calls = fam_ms.pop("di2p_point_chain"), launches.pop("di2p_point_chain")     # fused narrow PointNet chains (HBM-bound)
    hx_ms, hx_calls, hx_mac = fam_ms.pop("di2p_point_head_x3"), launches.pop("di2p_point_head_x3"), work.get("di2p_point_head_x3", 0) / prof_steps
    fam_ms["di2p_pointwise_gemm"] += fam_ms.pop("di2p_point_head") + x3_ms + chain_ms + hx_ms
    launches["di2p_pointwise_gemm"] += launches.pop("di2p_point_head") + x3_calls + chain_calls + hx_calls
    pw_exec_flops = 2.0 * (work.get("di2p_pointwise_gemm", 0) + work.get("di2p_point_head", 0) + work.get("di2p_point_chain", 0)) / prof_steps + 2.0 * x3_mac + 2.0 * hx_mac
    conv_flops = conv_flops_per_frame(H, W) * B
    knn_bytes = 2 * B * (12 * N + 24 * N + 3 * 4 * 128)          # the two point-level calls (pc -> node_a, pc -> node_b); node-level calls are negligible
    idx_bytes = B * (4 * 32 * N + 4 * N + 2 * 4 * 32 * 128) + B * (4 * 64 * N + 4 * N + 2 * 4 * 64 * 128)
    iters = out["iters"].float()
    sweeps = out.get("sweeps")
    n_front = float((out["labels_front"] >= 0).float().sum(dim=1).mean())
    n_hyp = int(out["iters"].numel())
    sol_ms = fam_ms["di2p_solve_batched_f32"]
    sol_flop_iters = SOLVER_FLOP_PER_POINT * n_front * float(iters.sum())
    roofs = {
        "solve_kernel": {
            "bound": "valu-fp64 (the frame's records are L2-resident: neither hbm nor mfma applies)",
            "achieved": sol_flop_iters / (sol_ms * 1e-3) / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "ms_per_step": sol_ms, "launches_per_step": launches["di2p_solve_batched_f32"], "hypotheses": n_hyp,
            "mean_iters": float(iters.mean()), "max_iters": float(iters.max()), "points_per_sweep": n_front,
            "algorithmic_flop_per_launch": sol_flop_iters,
            "note": "SURVEY 8(d) unit: 150 flop x front-filtered points x LM iterations of all hypotheses of the launch"},
        "conv2d_kernel": {
            "bound": "mfma", "achieved": conv_flops / (fam_ms["di2p_conv2d"] * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "ms_per_step": fam_ms["di2p_conv2d"], "launches_per_step": launches["di2p_conv2d"],
            "algorithmic_flop_per_launch": conv_flops / 36.0,
            "compulsory_bytes_per_launch": (conv_bytes_per_frame(H, W) * B + 85.1e6) / 36.0,
            "winograd": {"calls_per_step": wino_calls, "ms_per_step": wino_ms, "executed_mfma_tflops": wino_exec_flops / max(wino_ms, 1e-9) / 1e9,
                         "direct_kernel_ms_per_step": direct_ms, "stem_kernel_ms_per_step": stem_ms},
            "bf16x3": {"calls_per_step": cx_calls, "ms_per_step": cx_ms, "fp32_equivalent_tflops": 2.0 * cx_mac / max(cx_ms, 1e-9) / 1e9,
                       "frac_of_fp32_mfma_peak": 2.0 * cx_mac / max(cx_ms, 1e-9) / 1e9 / MFMA_F32_PEAK_TFLOPS,
                       "executed_bf16_tflops": 12.0 * cx_mac / max(cx_ms, 1e-9) / 1e9, "bf16_mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                       "frac_of_bf16_mfma_peak": 12.0 * cx_mac / max(cx_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                       "note": "3x3 layers (and the 1x1 / stride-2 branches fused into the stride-2 ones) as direct implicit GEMMs on the bf16 matrix "
                               "instructions, both fp32 operands split EXACTLY into three bf16 terms (six products per fp32 product, fp32 accumulation): "
                               "fp32_equivalent = 2*MAC / time, executed = 6 x that, priced against the 2.5 PFLOP/s dense bf16 peak"},
            "note": "achieved_reference_algorithmic = reference 2*MAC of the 36 convolutions (SURVEY 8d) / time of the whole family; frac / achieved / peak: see "
                    "frac_note (utilisation of the matrix pipes the launches run on).  Launches on the bf16 matrix instructions (exact three-way splits: "
                    "`bf16x3`) execute 6 bf16 products per fp32 product at 16x the fp32-input rate; launches that run as Winograd F(2x2,3x3) (shapes / knob "
                    "settings without a bf16x3 instance) execute 16 instead of 36 multiplications: `winograd.executed_mfma_tflops`"},
        "pointwise_gemm_kernel(+point_head)": {
            "bound": "mfma", "achieved": 2e9 * POINTWISE_REF_GMAC_COARSE * B / (fam_ms["di2p_pointwise_gemm"] * 1e-3) / 1e12,
            "achieved_executed": pw_exec_flops / (fam_ms["di2p_pointwise_gemm"] * 1e-3) / 1e12,
            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "ms_per_step": fam_ms["di2p_pointwise_gemm"],
            "launches_per_step": launches["di2p_pointwise_gemm"],
            "fused_chains": {"calls_per_step": chain_calls, "ms_per_step": chain_ms,
                             "hbm_tb_per_s": B * N * 4.0 * ((7 + 32) + (32 + 64)) / max(chain_ms, 1e-9) / 1e9,
                             "note": "first_pointnet (7->32->32->32) and second_pointnet (32+32->64->64) as one launch each: compulsory "
                                     "traffic = inputs once + outputs once, hidden activations in LDS"},
            "head_bf16x3": {"calls_per_step": hx_calls, "ms_per_step": hx_ms, "fp32_equivalent_tflops": 2.0 * hx_mac / max(hx_ms, 1e-9) / 1e9,
                            "executed_bf16_tflops": 12.0 * hx_mac / max(hx_ms, 1e-9) / 1e9, "frac_of_bf16_mfma_peak": 12.0 * hx_mac / max(hx_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                            "note": "the coarse per-point head (three layers, gathered node products) as ONE wave-autonomous launch on the bf16 matrix instructions "
                                    "(exact three-way splits): di2p_point_head_x3"},
            "bf16x3": {"calls_per_step": x3_calls, "ms_per_step": x3_ms, "fp32_equivalent_tflops": 2.0 * x3_mac / max(x3_ms, 1e-9) / 1e9,
                       "frac_of_fp32_mfma_peak": 2.0 * x3_mac / max(x3_ms, 1e-9) / 1e9 / MFMA_F32_PEAK_TFLOPS,
                       "executed_bf16_tflops": 12.0 * x3_mac / max(x3_ms, 1e-9) / 1e9, "bf16_mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                       "frac_of_bf16_mfma_peak": 12.0 * x3_mac / max(x3_ms, 1e-9) / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                       "note": "the GEMM-shaped layers (K >= 128, M % 128 == 0: kNN-fusion layers, node-level PointNets) run on "
                               "v_mfma_f32_32x32x16_bf16 with both fp32 operands split EXACTLY into three bf16 terms: six bf16 products per fp32 "
                               "product, fp32 accumulation -- as accurate against fp64 as the fp32-MFMA kernels (tests assert it).  "
                               "fp32_equivalent = 2*MAC / time (what the layer computes), executed = 6 x that (what the bf16 units issue)"},
            "note": "achieved_reference_algorithmic = reference 2*MAC (SURVEY 8d) / time; achieved = achieved_executed = fp32-equivalent flops actually "
                    "issued (per-node premultiply of per_point_pn.layers.0 and split concatenations execute fewer); frac: see frac_note"},
        "index_max_kernel": {
            "bound": "hbm", "achieved": idx_bytes / (fam_ms["di2p_index_max_values"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "ms_per_step": fam_ms["di2p_index_max_values"], "launches_per_step": launches["di2p_index_max_values"],
            "note": "in-pipeline durations (both calls: C=32 and C=64), not a cache-warm microbenchmark"},
        "knn_nodes_kernel": {
            # VALU-bound: per query and node 26 wave-instructions (83 VALU + 19 SALU + 3 other per 4 nodes in the ISA of knn_nodes_kernel<3>:
            # three subtractions, the squared distance, one 64-bit key compare per kept candidate and a branch-free chain of selects)
            "bound": "valu", "achieved": 2 * B * N * 128 * 26.0 / (fam_ms["di2p_knn_nodes"] * 1e-3) / 1e12, "peak": 78.6, "unit": "T lane-instructions/s",
            "ms_per_step": fam_ms["di2p_knn_nodes"], "launches_per_step": launches["di2p_knn_nodes"],
            "hbm_gbs_if_priced_as_survey_8d": knn_bytes / (fam_ms["di2p_knn_nodes"] * 1e-3) / 1e9,
            "note": "3-NN assignment of every point against 128 nodes (two point-level calls + two node-level ones, whose work is negligible): "
                    "achieved = 26 wave-instructions x 64 lanes per (query, node) pair / time, peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz; SURVEY 8(d) prices "
                    "it against HBM (36 N bytes per frame and call), which it does not touch: `hbm_gbs_if_priced_as_survey_8d`"},
    }
    if sweeps is not None:
        roofs["solve_kernel"]["mean_sweeps"] = float(sweeps.float().mean())
        roofs["solve_kernel"]["max_sweeps"] = float(sweeps.max())
    for r in roofs.values():
        if "achieved" in r:
            r["frac"] = r["achieved"] / r["peak"]
    # Schema 6 (round 6): the two matrix-pipe families run a MIX of instructions (bf16 products of exact three-way splits at 2.5 PFLOP/s, fp32-input
    # products at 157.3 TFLOP/s).  Their `frac` is the UTILISATION OF THE PIPES THEY RUN ON: the time the matrix pipes would need at peak for the
    # products the launches actually execute (6 bf16 products per fp32 product on the bf16x3 launches, the executed fp32 products elsewhere) over
    # the family's measured time; `achieved` = executed fp32-equivalent TFLOP/s, `peak` = the same flops over that peak-rate time (the best the
    # pipes could do on this mix), so that frac = achieved / peak as everywhere else.  The figures of schema <= 5 keep their own names:
    # `frac_reference_algorithmic_fp32_peak` (reference 2*MAC over the fp32-MFMA peak: above 1 once the work moved to the bf16 pipe -- a speed-up
    # statement, not a utilisation).
    def pipe_mix(fam, exec_f32_flops, x3_macs):
        """exec_f32_flops: fp32-equivalent flops the family executes per step (all launches); x3_macs: the MACs of its bf16x3 launches"""
        ms = fam["ms_per_step"]
        other = max(exec_f32_flops - 2.0 * x3_macs, 0.0)
        peak_ms = (12.0 * x3_macs / (MFMA_BF16_PEAK_TFLOPS * 1e12) + other / (MFMA_F32_PEAK_TFLOPS * 1e12)) * 1e3
        fam["frac_reference_algorithmic_fp32_peak"] = fam["achieved"] / MFMA_F32_PEAK_TFLOPS
        fam["achieved_reference_algorithmic"] = fam["achieved"]
        fam["achieved"] = exec_f32_flops / (ms * 1e-3) / 1e12
        fam["peak"] = exec_f32_flops / max(peak_ms * 1e-3, 1e-12) / 1e12
        fam["frac"] = peak_ms / max(ms, 1e-9)
        fam["matrix_pipe_ms_at_peak"] = peak_ms
        fam["bf16x3_share_of_executed_flops"] = 2.0 * x3_macs / max(exec_f32_flops, 1.0)
        fam["frac_note"] = ("frac = matrix-pipe time at peak for the EXECUTED products (bf16x3 launches: 6 bf16 products per fp32 product against 2.5 PFLOP/s; "
                            "other launches: fp32-input products against 157.3 TFLOP/s) / measured time; achieved = executed fp32-equivalent TFLOP/s")
    cf = roofs["conv2d_kernel"]
    # executed fp32-equivalent flops of the convolution family: the bf16x3 launches execute their algorithmic MACs, Winograd launches 16/36
    # of theirs (wino_exec_flops), the remaining direct launches their algorithmic MACs
    conv_other_flops = max(conv_flops - 2.0 * cx_mac - (wino_exec_flops * 36.0 / 16.0), 0.0)
    pipe_mix(cf, 2.0 * cx_mac + wino_exec_flops + conv_other_flops, cx_mac)
    pwf = roofs["pointwise_gemm_kernel(+point_head)"]
    pwf["achieved"] = pwf["achieved"]            # reference-algorithmic until pipe_mix renames it
    pipe_mix(pwf, pw_exec_flops, x3_mac + hx_mac)
    pwf["achieved_executed"] = pwf["achieved"]
    # HBM traffic per launch from the committed PMC passes of the same kernels on the same shapes (separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE runs, corrected as MI355X_MICROARCH.md prescribes); None if absent
    pmc = {}
    for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as fh:
                pmc = json.load(fh)
            break
        except (OSError, ValueError):
            pass
    if B == 32 and (H, W) == (160, 512):
        roofs["conv2d_kernel"]["traffic"] = pmc.get("conv2d_resnet34_B32_160x512", {}).get("hbm_bytes_per_call_corrected")
        roofs["index_max_kernel"]["traffic_C64"] = pmc.get("index_max_C64_B32_N20480_K128", {}).get("hbm_bytes_corrected")
        # (the solver's traffic comes from the SAME file as its instruction counters, below: one source)
    # executed fp64 flop of the solver: from the committed instruction-counter pass of the same kernel on the same workload shape
    # (tools/prof_solver_counters.sh -> profiles/r05_solver_counters.json, taken on the SHIPPED instantiation of the kernel:
    # 64 lanes x (2 FMA + ADD + MUL) wave-instructions per launch)
    try:
        sc = None
        for fn in ("r06_solver_counters.json", "r05_solver_counters.json", "r04_solver_counters.json", "r03_solver_counters.json"):
            path = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(path):
                with open(path) as fh:
                    sc = json.load(fh)
                roofs["solve_kernel"]["counters_file"] = "profiles/" + fn
                break
        if sc is not None and B == 32 and R == 60 and N == 20480:
            ex_flop = 64.0 * (2 * sc["SQ_INSTS_VALU_FMA_F64"] + sc["SQ_INSTS_VALU_ADD_F64"] + sc["SQ_INSTS_VALU_MUL_F64"])
            roofs["solve_kernel"]["executed_fp64_flop_per_launch"] = ex_flop
            if "FETCH_SIZE" in sc and "WRITE_SIZE" in sc:       # KiB per launch; FETCH x2 on gfx950 (16-byte streaming reads), WRITE raw
                roofs["solve_kernel"]["traffic"] = (2.0 * sc["FETCH_SIZE"] + sc["WRITE_SIZE"]) * 1024.0
                roofs["solve_kernel"]["traffic_note"] = "HBM-side bytes per launch from `counters_file` (FETCH_SIZE x 2 + WRITE_SIZE): the records are cache resident, the writes are the classification-cache entries (every store leaves the write-through L2)"
            roofs["solve_kernel"]["achieved_executed"] = ex_flop / (sol_ms * 1e-3) / 1e12
            roofs["solve_kernel"]["frac_executed"] = roofs["solve_kernel"]["achieved_executed"] / FP64_VALU_PEAK_TFLOPS
            roofs["solve_kernel"]["note"] += ("; achieved_executed = fp64 flop actually issued (counter pass on the same workload shape: the cluster test, the "
                                              "fp32 pre-filter and the active-set compaction skip most of the 150 flop x points of the unit)")
    except (OSError, ValueError, KeyError):
        pass
    # the line's roofline object = the TIME-DOMINANT kernel family of the step
    dom_name = max((n for n in roofs if "frac" in roofs[n]), key=lambda n: roofs[n]["ms_per_step"])
    dom = roofs[dom_name]
    roofline = {"kernel": dom_name, "bound": "mfma" if dom["bound"] == "mfma" else ("hbm" if dom["bound"] == "hbm" else "valu-fp64"),
                "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                "traffic": dom.get("traffic"), "ms_per_step": dom["ms_per_step"], "achieved_executed": dom.get("achieved_executed"),
                "frac_executed": dom.get("frac_executed"),
                "algorithmic_per_launch": dom.get("algorithmic_flop_per_launch"),
                "note": "time-dominant kernel family of the step; HIP events on the launch stream, serial pass of %d steps "
                        "directly after the timed region; every family's roofline is under `kernels`" % prof_steps}

    cpu_baseline = None
    if rank == 0 and world == 1 and not hyp and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(batch, sd, opt, H, W, R)

    collective = None
    if world > 1:
        # hyp mode: the exchange step of the path.  frames mode: there is NO data-path collective (frames are independent); the same small
        # all_gather is timed anyway as a probe of the RCCL / xGMI set-up the ranks run on
        collective = measure_collective(dist, backend, dev, B, R, world)
        collective["in_data_path"] = bool(hyp)
    placement = [{"rank": rank, "device": local_rank, "device_count": torch.cuda.device_count(), "name": torch.cuda.get_device_name(local_rank)}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, placement[0])
        placement = gathered
    if rank == 0:
        line = {
            "metric": "frames/sec (img+pc infer + 60-restart GN pose) KITTI 20k-pt, 1/2/4/8 GPU",
            "schema": 6,
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if hyp else "weak", "vs_baseline": None,
            "dtype": "f32 network (fp32-input MFMA; 3x3 convolutions, GEMM-shaped point layers and the per-point head: bf16 MFMA on exact three-way fp32 splits, fp32 accumulation) + f64 solver",
            "data": "synthetic frames, random-init closed-form weights; solver labels = GT frustum labels with 5% flips (SURVEY 8d)",
            "config": {"workload": ("BASELINE configs[4]: Oxford-shaped %d-pt / %dx%d, %d frames per step on every rank, coarse classification "
                                    "+ %d 2D GN/LM hypotheses per frame sharded over the ranks, all_gather + argmin" % (N, H, W, B, R)) if hyp else
                                   ("BASELINE configs[1]: KITTI 20480-pt / 160x512, batch %d per GPU, coarse classification "
                                    "+ %d-restart 2D GN/LM solver, max_iter 500" % (B, R)),
                       "mode": args.mode, "frames_per_gpu_per_step": B, "points": N, "image": [H, W], "restarts": R,
                       "parallelism": ("hyp%d" if hyp else "dp%d") % world, "streams": n_streams, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "hip_graph": bool(use_graph),
                       "weights_broadcast_bytes": bcast_bytes, "placement": placement},
            "value_resident": frames / dt,
            "value_with_h2d": (frames / dt_h2d) if dt_h2d else None,
            "ms_per_step_with_h2d": (dt_h2d / args.steps * 1e3) if dt_h2d else None,
            "value_definition": "`value` = `value_resident`: every step's inputs are already in HBM when the timed region starts (the bench "
                                "contract: inputs resident; a PCIe-inclusive rate is never `value`).  `value_with_h2d` = SURVEY.md 8(d)'s "
                                "definition of the metric: the same loop with the host->device copy of every batch (50 MB from pinned host "
                                "memory) inside the step.",
            "per_rank_ms_per_step": rank_ms or None,
            "latency_ms_per_batch": {"streams_%d" % n_streams: (sum(lat) / len(lat)) if lat else None,
                                     "streams_%d_with_h2d" % n_streams: (sum(lat_h2d) / len(lat_h2d)) if lat_h2d else None,
                                     "one_step_in_flight": (sum(lat1) / len(lat1)) if lat1 else None,
                                     "note": "device time of one batch, first launch to last: with all streams busy (what `value` is measured "
                                             "under) and with a single step in flight"},
            "roofline": roofline, "kernels": roofs, "cpu_baseline": cpu_baseline, "collective": collective,
            "pose_check": pose_check(out, batch),
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main_train(args):
    """--mode train: K optimisation steps of the reference's training configuration (kitti/options.py: batch 8 per GPU, 20480 points,
    160x512, coarse + fine heads, Adam lr 1e-3) -- train-mode forward, losses, backward, ONE all-reduce of the flat gradient buffer,
    Adam -- all on the HIP kernels (deepi2p_amd/train_net.py).  Not the headline metric: a separate line for SURVEY.md 8f rank 4."""
    import numpy as np
    import torch
    rank, local_rank, world, backend, dist = init_dist(args)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from deepi2p_amd import _lib, synthetic
    from deepi2p_amd.networks import KeypointDetector
    from deepi2p_amd.training import ClassifierTrainer

    B, N, H, W = args.batch or 8, args.points or 20480, 160, 512
    opt = synthetic.OptLike(N, H, W, True)
    opt.lr, opt.coarse_loss_alpha = 1e-3, 50.0
    det = KeypointDetector(opt)
    det.load_state_dict(synthetic.random_state_dict(opt, 0))
    det = det.to(dev)
    bcast_bytes = broadcast_weights(det, dist, backend) if world > 1 else 0
    tr = ClassifierTrainer(det, opt, seed=0)          # the trainer folds the rank into its dropout seed
    b = synthetic.make_batch(2000 + rank, B, N=N, H=H, W=W)
    t = [torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
    K = torch.from_numpy(b["K"]).float().to(dev)
    Pgt = torch.from_numpy(np.ascontiguousarray(b["P_gt"][:, :3, :])).float().to(dev)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    step_fn = tr.optimize_graphed if (getattr(args, "train_graph", False) and world == 1) else tr.optimize
    for _ in range(args.warmup):
        losses.append(step_fn(*t, K, Pgt)["loss"].clone())
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step_fn(*t, K, Pgt)["loss"].clone())
    t_enqueue = time.perf_counter() - t0          # host time to enqueue the steps (close to dt: the step is launch-bound)
    sync_all()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # the gradient exchange on its own (one all-reduce of the flat fp32 buffer), and a serial pass with events around every C-ABI call
    ar_ms = None
    if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        e0.record()
        from deepi2p_amd.training import allreduce_gradients
        for _ in range(5):
            allreduce_gradients(tr.flat_grad)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 5
    _lib.TIMED = {name: [] for name in _lib._SIGS}
    tr.optimize(*t, K, Pgt)
    torch.cuda.synchronize()
    fam = {k: (sum(a.elapsed_time(c) for a, c, _ in v), len(v)) for k, v in _lib.TIMED.items() if v}
    _lib.TIMED = None
    # roofline of the time-dominant family of a training step: the MFMA contractions (forward, input gradients, weight gradients).
    # Algorithmic work = what autograd of the reference's graph does: forward + dgrad + wgrad = 3 x the forward multiply-accumulates of
    # SURVEY.md 8(d) (coarse + fine model: 13.284 GMAC per frame at 20480 points / 160x512), minus the input gradient of the stem (the
    # image needs none).  Time = HIP events around every contraction call of one serial step.
    mfma_calls = ("di2p_pointwise_gemm", "di2p_pointwise_gemm_x3", "di2p_point_head", "di2p_point_chain", "di2p_conv3x3_winograd", "di2p_conv3x3_x3", "di2p_conv2d",
                  "di2p_conv2d_ws", "di2p_conv7x7s2_stem", "di2p_conv2d_wgrad", "di2p_conv2d_dgrad", "di2p_bmm_rc", "di2p_bmm_km", "di2p_gather_backward",
                  "di2p_winograd_weight_transform", "di2p_winograd_weight_transform_dgrad", "di2p_bf16x3_pack", "di2p_bf16x3_pack_conv3x3")
    mfma_ms = sum(fam[k][0] for k in mfma_calls if k in fam)
    roofline = None
    if (N, H, W) == (20480, 160, 512) and mfma_ms > 0:
        gmac_fwd = 13.284
        stem_gmac = 3 * 64 * 49 * (H // 2) * (W // 2) / 1e9
        algo = 2e9 * (3 * gmac_fwd - stem_gmac) * B
        roofline = {"kernel": "mfma contraction family of the training step (forward + dgrad + wgrad: %s)" % ", ".join(k for k in mfma_calls if k in fam),
                    "bound": "mfma", "achieved": algo / (mfma_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": algo / (mfma_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": None, "ms_per_step": mfma_ms,
                    "algorithmic_per_launch": algo / max(sum(fam[k][1] for k in mfma_calls if k in fam), 1),
                    "step_fraction": mfma_ms / (1e3 * dt / args.steps),
                    "note": "achieved = reference-algorithmic flops (3 x forward 2*MAC of SURVEY 8d, coarse+fine, minus the stem's input gradient) / "
                            "summed event time of the contraction calls of one serial step (fp32- and bf16x3-MFMA launches and their filter transforms / "
                            "splits alike); priced against the fp32-input MFMA peak; the rest of the step is "
                            "BatchNorm statistics / normalisation passes (HBM-bound), routers and the optimiser"}
    if rank == 0:
        top = sorted(fam.items(), key=lambda kv: -kv[1][0])[:10]
        line = {"metric": "training frames/sec (train-mode fwd + focal/CE loss + bwd + gradient all-reduce + Adam) KITTI 20k-pt 160x512, batch %d per GPU" % B,
                "value": B * world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * dt / args.steps, "host_enqueue_ms_per_step": 1e3 * t_enqueue / args.steps,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32 (fp32-input MFMA contractions; the big point layers and the 256- / 512-channel 3x3 layers, forward and input gradient: bf16 MFMA on exact "
                         "three-way fp32 splits, fp32 accumulation; fp64 BatchNorm/bias reductions)", "data": "synthetic frames, seeded He-normal initial weights",
                "config": {"workload": "reference training configuration kitti/options.py:20-60 (batch 8, 20480 pts, 160x512, coarse+fine, Adam 1e-3)",
                           "mode": "train", "frames_per_gpu_per_step": B, "points": N, "image": [H, W], "parallelism": "dp%d" % world,
                           "step": "optimize_graphed (forward + losses + backward replayed from one hipGraph)" if step_fn is not tr.optimize else "optimize (eager launches)",
                           "weights_broadcast_bytes": bcast_bytes},
                "gradient_allreduce": {"bytes": int(tr.flat_grad.numel() * 4), "ms": ar_ms, "launches_per_step": 1},
                "roofline": roofline,
                "loss_first_last": [float(losses[0]), float(losses[-1])],
                "calls_ms_per_step": {k: {"ms": round(v[0], 3), "calls": v[1]} for k, v in top},
                "note": "calls_ms_per_step: HIP events around every C-ABI call of one extra serial step (ATen views/concatenations excluded)"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measure_collective(dist, backend, dev, F, R, world):
    """Latency of the config-5 all_gather alone (payload R/world x 5 doubles per frame per rank), 50 calls."""
    import torch
    width = -(-R // world)
    buf = torch.zeros((F, width, 5), dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    outs = [torch.empty_like(buf) for _ in range(world)]
    for _ in range(5):
        dist.all_gather(outs, buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        dist.all_gather(outs, buf)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    return {"op": "all_gather", "backend": "rccl" if backend == "nccl" else backend, "bytes_per_rank": buf.numel() * 8,
            "latency_us": us}


def pose_check(out, batch):
    """Sanity (not parity): the network has random weights, so its labels are meaningless; report only that the
    solver ran to completion on the synthetic labels."""
    from deepi2p_amd.registration import get_P_diff
    P = out["P"].cpu().numpy()
    errs = [get_P_diff(P[i], batch["P_gt"][i]) for i in range(P.shape[0])]
    ok = sum(1 for t, r in errs if t < 2.0 and r < 5.0)
    return {"mean_iters": float(out["iters"].float().mean()), "frames_with_inside_points": int((out["best"] >= 0).sum()),
            "frames_within_2m_5deg_of_gt": ok, "frames": int(P.shape[0]),
            "network_pred_inside_fraction": float((out["pred"] == 1).float().mean())}


def run_cpu_baseline(batch, sd, opt, H, W, R):
    """The oracle ("port" of the reference's CPU path) timed on this box's host cores on a BOUNDED sample of the same workload, after
    the protocol of BASELINE.md section 3: network 3 warm-up + 10 timed iterations, solver one thread per restart across the host
    cores on the seeded hypothesis list -- in THROUGHPUT mode (several frames' restarts in flight together, so that all cores have
    work: one frame's 60 restarts can occupy at most 60 threads) -- plus a single-thread figure of both parts."""
    import threading
    import numpy as np
    import torch
    from oracle import frustum_lm as flm
    from oracle import network_torch as nt
    names = ("pc", "intensity", "sn", "node_a", "node_b", "img")
    cores = os.cpu_count() or 1
    lb, ub = [-5, -0.1, -10], [5, 0.1, 10]

    def net_time(nb, threads, warm, timed):
        torch.set_num_threads(threads)
        t = [torch.from_numpy(batch[k][:nb]) for k in names]
        with torch.no_grad():
            for _ in range(warm):
                nt.keypoint_detector(sd, opt, *t)
            t0 = time.perf_counter()
            for _ in range(timed):
                nt.keypoint_detector(sd, opt, *t)
        return (time.perf_counter() - t0) / (timed * nb)

    def restart_list(i, n):
        pc = batch["pc"][i].astype(np.float64)
        _, y0, pcf, labf = flm.get_initial_guess(pc, batch["labels"][i])          # same synthetic solver labels as the GPU leg
        ys, Ts = flm.draw_restarts(np.random.default_rng(i), n, y0, 10 * math.pi / 180, 10)
        return pcf, labf, ys, Ts

    # network, all cores, two ways -- the better one counts: (a) one batch-2 forward on 32 intra-op threads (torch CPU ops stop scaling,
    # and regress, far below the box's 256 threads), 3 warm-up + 10 timed; (b) throughput mode: `nt_conc` single-thread forwards of one
    # frame each in flight together (the ops release the GIL)
    net_threads = min(cores, 32)
    nb = 2
    t_net_a = net_time(nb, net_threads, 3, 10)
    nt_conc = min(cores, 32)
    torch.set_num_threads(1)
    frames1 = [[torch.from_numpy(batch[k][i % batch["pc"].shape[0]:i % batch["pc"].shape[0] + 1]) for k in names] for i in range(nt_conc)]

    def fwd(i):
        with torch.no_grad():
            nt.keypoint_detector(sd, opt, *frames1[i])

    def net_round():
        th = [threading.Thread(target=fwd, args=(i,)) for i in range(nt_conc)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return time.perf_counter() - t0
    net_round()
    t_net_b = sum(net_round() for _ in range(3)) / (3 * nt_conc)
    t_net = min(t_net_a, t_net_b)
    # solver: one thread per restart across the host cores, two ways -- the better one counts: one frame at a time (R restarts on R
    # threads), and throughput mode with `conc` frames' restarts in flight together
    def solve_frames(conc, per, rounds):
        jobs = [restart_list(i, R) for i in range(conc)]
        its = [None] * conc

        def one(i):
            pcf, labf, ys, Ts = jobs[i]
            its[i] = flm.solve_restarts(pcf, labf, batch["K"][i], ys, Ts, H, W, lb, ub, 500, True, nthreads=per)[2]

        def rnd():
            th = [threading.Thread(target=one, args=(i,)) for i in range(conc)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            return time.perf_counter() - t0
        rnd()                                                    # warm-up
        return sum(rnd() for _ in range(rounds)) / (rounds * conc), float(np.mean([x.mean() for x in its]))
    t_sol_1, mean_iters = solve_frames(1, min(cores, R), 3)
    conc = max(1, min(4, cores // max(R, 1)))
    t_sol_c = solve_frames(conc, max(1, cores // conc), 2)[0] if conc > 1 else t_sol_1
    t_sol = min(t_sol_1, t_sol_c)
    rounds, per = 3, min(cores, R)
    # single-thread figures on a smaller sample: one frame through the network, two restarts of one frame (scaled to R)
    t_net1 = net_time(1, 1, 0, 1)
    pcf, labf, ys, Ts = restart_list(0, 2)
    t0 = time.perf_counter()
    flm.solve_restarts(pcf, labf, batch["K"][0], ys, Ts, H, W, lb, ub, 500, True, nthreads=1)
    t_sol1 = (time.perf_counter() - t0) / 2 * R
    torch.set_num_threads(net_threads)
    return {"value": 1.0 / (t_net + t_sol), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "network: better of [batch %d on %d threads, 3 warm-up + 10 timed forwards: %.2f s/frame] and [%d single-thread forwards in "
                      "flight, 1 warm-up + 3 timed rounds: %.2f s/frame]; solver (one thread per restart, %d restarts, 1 warm-up + timed rounds): better "
                      "of [one frame at a time: %.2f s/frame] and [%d frames in flight: %.2f s/frame], mean %.1f LM iterations; "
                      "value = 1 / (network + solver) seconds per frame" % (nb, net_threads, t_net_a, nt_conc, t_net_b, R, t_sol_1, conc, t_sol_c, mean_iters),
            "network_s_per_frame": t_net, "solver_s_per_frame": t_sol,
            "one_thread": {"value": 1.0 / (t_net1 + t_sol1), "unit": "frames/s", "cores": 1,
                           "sample": "network: 1 frame, 1 forward on 1 thread (%.1f s); solver: 2 restarts of 1 frame on 1 thread, "
                                     "scaled to %d restarts (%.1f s/frame)" % (t_net1, R, t_sol1)}}


if __name__ == "__main__":
    main()
