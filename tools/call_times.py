"""Per-call device time of ONE serial step of the config-2 path (GPU box): HIP events around every C-ABI call, in launch order, with the
shapes of the pointwise contractions.   python tools/call_times.py [min_us]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepi2p_amd import _lib, ops, synthetic
from deepi2p_amd.networks import MMClassiferCoarse
from deepi2p_amd.pipeline import RegistrationExecutor
from deepi2p_amd.registration import RegistrationPipeline

B, N, H, W, R = int(os.environ.get("B", 32)), 20480, 160, 512, 60
dev = torch.device("cuda", 0)
opt = synthetic.OptLike(N, H, W, bool(int(os.environ.get("FINE", "0"))))
opt.device = dev
mm = MMClassiferCoarse(opt)
mm.detector.load_state_dict(synthetic.synthetic_state_dict(opt))
batch = synthetic.make_batch(1000, B, N=N, H=H, W=W)
host = {k: torch.from_numpy(batch[k]) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")}
pipe = RegistrationPipeline(H, W, R=R, seed=0)
ex = RegistrationExecutor(mm, pipe, torch.from_numpy(batch["K"]), host, n_streams=1, use_graph=False,
                          labels_override=torch.from_numpy(batch["labels"]).to(dev))
_pg = ops.pointwise_gemm


def tagged(srcs, Wt, M, Nn, **kw):
    _lib.TIMED_TAG = "M=%d K=%d N=%d srcs=%s%s%s%s" % (M, Wt.shape[0], Nn, "+".join("%s%d" % ("dgG"[s.mode] if hasattr(s, "mode") else "p", getattr(s, "C", s.t.shape[1])) for s in srcs),
                                                      " gmax%d" % kw["group_max"] if kw.get("group_max", 1) > 1 else "", " gathered" if kw.get("gathered") else "",
                                                      (" T" if kw.get("transpose_out") else "") + (" ->planes" if kw.get("planes_out") else ""))
    try:
        return _pg(srcs, Wt, M, Nn, **kw)
    finally:
        _lib.TIMED_TAG = None


ops.pointwise_gemm = tagged
import deepi2p_amd.networks as nw
nw.ops.pointwise_gemm = tagged
for _ in range(2):
    ex.step_eager(0, False)
torch.cuda.synchronize()
reps = 3
order = []
_lib.TIMED = {n: [] for n in _lib._SIGS}
_call = _lib.call


def call(name, *a):
    order.append(name)
    return _call(name, *a)


_lib.call = call
ops.call = call
for m in (nw, __import__("deepi2p_amd.registration", fromlist=["x"])):
    if hasattr(m, "call"):
        m.call = call
for _ in range(reps):
    ex.step_eager(0, False)
torch.cuda.synchronize()
per = len(order) // reps
idx = {n: 0 for n in _lib.TIMED}
rows = []
for i in range(per):
    n = order[i]
    v = _lib.TIMED.get(n)
    if not v:
        continue
    k = idx[n]; idx[n] += 1
    cnt = len(v) // reps
    ms = sum(v[k + r * cnt][0].elapsed_time(v[k + r * cnt][1]) for r in range(reps)) / reps
    rows.append((i, n, ms * 1e3, v[k][2]))
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 15.0
tot = {}
for i, n, us, tag in rows:
    tot[n] = tot.get(n, 0.0) + us
    if us >= thr:
        print("%4d %-28s %8.1f us  %s" % (i, n.replace("di2p_", ""), us, tag or ""))
print("--- per entry point (us per step):")
for n, us in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-30s %9.1f  (%d calls)" % (n.replace("di2p_", ""), us, sum(1 for r in rows if r[1] == n)))
print("sum %.1f us" % sum(tot.values()))
