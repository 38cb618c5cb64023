"""Cycles of a wave's lifetime in conv3x3_x3_kernel, by stage, per ResNet-34 layer shape.  Needs the diagnostic build of conv_x3.hip:
    python tools/build_variant.py conv_x3.hip cx_clk -DDI2P_CX_CLK=1
    DI2P_LIB=$PWD/deepi2p_amd/lib/variants/cx_clk/libdeepi2p_hip.so python tools/exp_cx_clk.py        (profiles/r06_c35_cx_clk.txt)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepi2p_amd import ops, _lib
dev = torch.device("cuda", 0)
B = 32
lib = _lib.load()
lib.di2p_cx_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (Cin, H, W, Cout, s) in ((64, 40, 128, 64, 1), (128, 20, 64, 128, 1), (256, 10, 32, 256, 1), (512, 5, 16, 512, 1), (64, 40, 128, 128, 2), (128, 20, 64, 256, 2), (256, 10, 32, 512, 2)):
    x = torch.randn(B, Cin, H, W, device=dev)
    Wp = ops.bf16x3_pack(torch.randn(9 * Cin, Cout, device=dev) * 0.05)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Cout, OH, OW, device=dev)
    ds = (ops.bf16x3_pack(torch.randn(Cin, Cout, device=dev) * 0.05), sc, sh) if s == 2 else None
    f = lambda: ops.conv3x3_x3(x, Wp, Cout, sc, sh, s, True, residual=res if s == 1 else None, downsample=ds)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    import numpy as np
    nw = 2048 if (Cin, s) == (64, 1) else 1024
    out = (ctypes.c_ulonglong * (nw * 4))()
    lib.di2p_cx_clk(out, nw)
    a = np.array(out[:], dtype=np.float64).reshape(nw, 4)
    m = a.mean(0)
    print("%3d,%3d,%3d,%3d,s%d: %.1f us per launch | cycles per wave (mean; total max): prologue %6.0f  taps %6.0f  chunk barriers %6.0f  epilogue %6.0f  | total %6.0f  max %6.0f"
          % (Cin, H, W, Cout, s, e0.elapsed_time(e1) / 5 * 1e3, m[0], m[1], m[2], m[3], m.sum(), a.sum(1).max()), flush=True)
