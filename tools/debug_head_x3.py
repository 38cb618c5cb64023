import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from deepi2p_amd import _lib, ops
import test_gpu_head_x3 as T
dev = torch.device("cuda", 0)
B, N, P = 1, 64, 4
d = T._case(dev, B, N, (128, 128), P, 1)
one, zero = torch.ones(128, device=dev), torch.zeros(128, device=dev)
I = torch.eye(128, device=dev)
def run(d, relu=False):
    packed = {"W0p": ops.head_x3_pack(d["W0"].contiguous()), "W1p": ops.head_x3_pack(d["W1"].contiguous()),
              "ss": torch.stack((d["sc0"], d["sh0"], d["sc1"], d["sh1"])).contiguous(), "relu0": relu, "relu1": relu,
              "W2t": d["W2"].contiguous(), "sc2": None, "sh2": d["sh2"], "relu2": False}
    return ops.point_head_x3(d["first"], d["second"], packed, [(d["Ga"], d["ia"], d["wa"]), (d["Gb"], d["ib"], d["wb"])], N)
x = torch.cat((d["first"], d["second"]), 1)
base = dict(d, sc0=one, sh0=zero, sc1=one, sh1=zero, sh2=torch.zeros(P, device=dev), Ga=torch.zeros_like(d["Ga"]), Gb=torch.zeros_like(d["Gb"]))
# P2: W0 routes input channel k to hidden channel k; W1 = I; W2 selects hidden channels c0..c0+3
W0 = torch.zeros(96, 128, device=dev); W0[torch.arange(96), torch.arange(96)] = 1.0
for c0 in (0, 4, 8, 12, 16, 28, 32, 40, 64, 92):
    W2 = torch.zeros(128, P, device=dev)
    for p in range(P): W2[c0 + p, p] = 1.0
    y = run(dict(base, W0=W0, W1=I.clone(), W2=W2))
    ref = x[:, c0:c0 + P, :]
    print("route channel %3d..: max err %.3g" % (c0, float((y - ref).abs().max())), "" if float((y - ref).abs().max()) < 1e-5 else ("got %s want %s" % (y[0, :, 0].tolist(), ref[0, :, 0].tolist())))
# P1: random W0, W1 = I
W2 = torch.zeros(128, P, device=dev)
for p in range(P): W2[p * 33, p] = 1.0
y = run(dict(base, W1=I.clone(), W2=W2))
ref = torch.einsum("km,bkn->bmn", d["W0"], x)[:, [0, 33, 66, 99], :]
print("layer 0 alone: max err %.3g" % float((y - ref).abs().max()))
# layer 1 alone: W0 routes, random W1
y = run(dict(base, W0=W0, W2=W2))
h = torch.zeros(B, 128, N, device=dev); h[:, :96] = x
ref = torch.einsum("km,bkn->bmn", d["W1"], h)[:, [0, 33, 66, 99], :]
print("layer 1 alone: max err %.3g" % float((y - ref).abs().max()))
# P3: gather alone
y = run(dict(base, W0=torch.zeros(96, 128, device=dev), W1=I.clone(), W2=W2, Ga=d["Ga"], Gb=d["Gb"]))
ref = T._ref64(dict(base, W0=torch.zeros(96, 128, device=dev), W1=I.clone(), W2=W2, Ga=d["Ga"], Gb=d["Gb"]))
print("gather alone (relu on in ref!): max err %.3g" % float((torch.relu(y.double()) - ref).abs().max()))
