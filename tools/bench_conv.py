"""Image-encoder-only driver for profiling the conv kernels (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd.networks import ImageEncoder
from deepi2p_amd import synthetic as nt

B = int(os.environ.get("B", 32))
dev = torch.device("cuda", 0)
opt = nt.OptLike(20480, 160, 512, False)
sd = {k[len("img_encoder."):]: v for k, v in nt.synthetic_state_dict(opt).items() if k.startswith("img_encoder.")}
enc = ImageEncoder(opt)
enc.load_state_dict(sd)
enc = enc.to(dev)
img = torch.rand(B, 3, 160, 512, device=dev) * 255
for _ in range(2):
    enc(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    enc(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("image encoder B=%d: %.3f ms  -> %.1f TFLOP/s algorithmic" % (B, dt * 1e3, 2 * 5.981e9 * B / dt / 1e12))
