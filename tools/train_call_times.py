"""One serial training step (batch 8) with HIP events around every C-ABI call, listed per call with its integer arguments (the shape), largest
first.   python tools/train_call_times.py [names-substring]     (profiles/r06_train_call_times.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepi2p_amd import _lib, synthetic
from deepi2p_amd.networks import KeypointDetector
from deepi2p_amd.training import ClassifierTrainer
import deepi2p_amd.ops as ops
import deepi2p_amd.train_net as tn

dev = torch.device("cuda", 0)
B, N, H, W = 8, 20480, 160, 512
opt = synthetic.OptLike(N, H, W, True)
opt.lr, opt.coarse_loss_alpha = 1e-3, 50.0
det = KeypointDetector(opt)
det.load_state_dict(synthetic.random_state_dict(opt, 0))
det = det.to(dev)
tr = ClassifierTrainer(det, opt, seed=0)
b = synthetic.make_batch(2000, B, N=N, H=H, W=W)
t = [torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("pc", "intensity", "sn", "node_a", "node_b", "img")]
K = torch.from_numpy(b["K"]).float().to(dev)
Pgt = torch.from_numpy(np.ascontiguousarray(b["P_gt"][:, :3, :])).float().to(dev)
for _ in range(2):
    tr.optimize(*t, K, Pgt)
torch.cuda.synchronize()
rec = []
_call = _lib.call


def call(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _call(name, *args)
    e1.record()
    rec.append((name, [a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 24)], e0, e1))


_lib.call = ops.call = tn.call = call
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "__name__", "").startswith("deepi2p_amd") and getattr(m, "call", None) is _call:
        m.call = call
tr.optimize(*t, K, Pgt)
torch.cuda.synchronize()
filt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = [(e0.elapsed_time(e1) * 1e3, n, a) for n, a, e0, e1 in rec if filt in n]
tot = {}
for us, n, a in rows:
    tot[n] = tot.get(n, 0.0) + us
print("per entry point (us per step):", ", ".join("%s %.0f" % (n, v) for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]))
for us, n, a in sorted(rows, key=lambda r: -r[0])[:int(os.environ.get("TOP", 70))]:
    print("%8.1f us  %-28s %s" % (us, n.replace("di2p_", ""), " ".join(str(x) for x in a)))
