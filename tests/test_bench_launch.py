"""bench.py's multi-rank entry point.

CPU (here): `python bench.py --gpus N` without a launcher must start N ranks itself, rendezvous on 127.0.0.1 and report
n_gpus == N; the same under an external launcher (torch.distributed.run, the driver's command shape).  These runs use
--launch-selftest (rendezvous + barrier + all_reduce on gloo, no HIP work).
GPU box (one device): the full frames / hyp modes with two gloo ranks sharing device 0 (DI2P_BENCH_ONE_DEVICE=1)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(DI2P_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", **kw)
    return env


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gpus_flag_spawns_the_ranks_itself():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["launch_selftest"] and line["n_gpus"] == 2 and line["requested_gpus"] == 2
    assert line["max_rank_plus_one"] == 2.0            # the all_reduce really spanned both ranks


@pytest.mark.parametrize("user_value", [None, "1"])
def test_ipc_mode_is_a_default_not_an_override(user_value):
    """The launcher exports HSA_ENABLE_IPC_MODE_LEGACY=0 (this host driver only does dmabuf IPC; RCCL needs it) ONLY when the caller's
    environment does not set it: a user's own value reaches the ranks untouched."""
    env = _env()
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    if user_value is not None:
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = user_value
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = _last_json(r.stdout)["env"]
    assert seen["HSA_ENABLE_IPC_MODE_LEGACY"] == (user_value if user_value is not None else "0")
    assert seen["MASTER_ADDR"] == "127.0.0.1"


def test_under_external_launcher_uses_its_world():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), BENCH, "--gpus", "2", "--launch-selftest"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 2


def test_single_rank_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH, "--launch-selftest"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["frames", "hyp"])
def test_two_ranks_full_path_on_one_device(mode):
    """Both modes end to end with 2 ranks (gloo, shared device): frames = weak scaling (2 x batch), hyp = the same frames on
    both ranks, hypotheses sharded, all_gather + argmin."""
    args = [sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--mode", mode]
    args += ["--batch", "4"] if mode == "frames" else ["--batch", "2", "--points", "8192", "--restarts", "24"]
    r = subprocess.run(args, env=_env(DI2P_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["mode"] == mode
    assert line["scaling"] == ("weak" if mode == "frames" else "strong")
    assert line["pose_check"]["frames_with_inside_points"] == line["pose_check"]["frames"]
    assert line["roofline"]["kernel"] in line["kernels"] and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_two_ranks_training_step_on_one_device():
    """--mode train with 2 ranks (gloo, shared device): data-parallel optimisation steps with one all-reduce of the flat gradient
    buffer per step; both ranks keep identical weights, the loss falls."""
    args = [sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "train", "--batch", "2", "--points", "2048"]
    r = subprocess.run(args, env=_env(DI2P_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["mode"] == "train" and line["value"] > 0
    assert line["gradient_allreduce"]["bytes"] > 1e8 and line["gradient_allreduce"]["ms"] > 0
    assert line["loss_first_last"][1] < line["loss_first_last"][0]
