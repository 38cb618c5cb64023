"""bf16x3 kernels on operands far from 1: scaled by 2^+-60 / 2^+-100, tiny activations whose second / third split terms are bf16 denormals,
a +inf activation (pointwise x3 and conv x3 against the fp32-MFMA kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
B, K, M, N = 1, 256, 256, 2048
x = torch.randn(B, K, N, generator=g).to(dev)
Wt = (torch.randn(K, M, generator=g) / K ** 0.5).to(dev)
ref = ops.pointwise_gemm([ops.Src(x)], Wt, M, N, x3=False)
for e in (0, 60, 100, 120, 126):
    xs, ws = x * 2.0 ** -e, Wt * 2.0 ** e
    y3 = ops.pointwise_gemm([ops.Src(xs)], ws.contiguous(), M, N, x3=True)
    y1 = ops.pointwise_gemm([ops.Src(xs)], ws.contiguous(), M, N, x3=False)
    print("pointwise: x * 2^-%d, W * 2^%d: x3 vs unscaled fp32 rel err %.3g ; fp32 kernel itself %.3g" % (
        e, e, float((y3 - ref).abs().max() / ref.abs().max()), float((y1 - ref).abs().max() / ref.abs().max())))
xi = x.clone(); xi[0, 5, 7] = float("inf")
y3 = ops.pointwise_gemm([ops.Src(xi)], Wt, M, N, x3=True)
y1 = ops.pointwise_gemm([ops.Src(xi)], Wt, M, N, x3=False)
print("pointwise +inf at column 7: fp32 non-finite columns", torch.nonzero(~torch.isfinite(y1).all(dim=1)[0]).flatten().tolist()[:5],
      "x3 non-finite columns", torch.nonzero(~torch.isfinite(y3).all(dim=1)[0]).flatten().tolist()[:5],
      "fp32 has inf:", bool(torch.isinf(y1).any()), "x3 has nan:", bool(torch.isnan(y3).any()), "x3 has inf:", bool(torch.isinf(y3).any()))
Cin, H, W, Cout = 64, 8, 64, 64
xc = torch.randn(1, Cin, H, W, generator=g).to(dev)
w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
Wtc = w.permute(2, 3, 1, 0).reshape(-1, Cout).contiguous()
one, zero = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
refc = ops.conv2d(xc, Wtc, one, zero, 3, 3, 1, 1, False, tap_major=True)
for e in (0, 60, 100, 120, 126):
    y3 = ops.conv3x3_x3(xc * 2.0 ** -e, ops.bf16x3_pack((Wtc * 2.0 ** e).contiguous()), Cout, one, zero, 1, False)
    print("conv: x * 2^-%d, W * 2^%d: x3 vs unscaled fp32 rel err %.3g" % (e, e, float((y3 - refc).abs().max() / refc.abs().max())))
xci = xc.clone(); xci[0, 3, 4, 5] = float("inf")
y3 = ops.conv3x3_x3(xci, ops.bf16x3_pack(Wtc), Cout, one, zero, 1, False)
y1 = ops.conv2d(xci, Wtc, one, zero, 3, 3, 1, 1, False, tap_major=True)
print("conv +inf at (4,5): fp32 non-finite pixels %d, x3 non-finite pixels %d, same set: %s" % (
    int((~torch.isfinite(y1)).any(dim=1).sum()), int((~torch.isfinite(y3)).any(dim=1).sum()), bool(torch.equal((~torch.isfinite(y1)).any(dim=1), (~torch.isfinite(y3)).any(dim=1)))))
