"""bf16 conv forward WITH a residual (and bias) at the shapes of the transformer / conditioner blocks, alone on the GPU: the epilogue's residual
round trips are what this measures (SET_AMD_LIB=<other build> for an A/B).   python tools/conv_res_probe.py"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = [(16, 256, 256, 1, 800), (16, 192, 192, 1, 800), (16, 256, 768, 1, 800), (16, 1024, 256, 1, 800), (16, 256, 1024, 9, 800),
          (32, 192, 384, 5, 100), (32, 384, 192, 1, 100), (32, 192, 192, 1, 800), (32, 256, 512, 1, 800)]
ops.set_compute_dtype("bf16")
for (B, Cin, Cout, K, T) in SHAPES:
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    res = torch.randn(B, Cout, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    cw = ops.ConvWeight(lambda: w, Cout, Cin, K)
    pad = (K - 1) // 2
    y = torch.empty(B, Cout, T, device=dev)
    out = []
    for r in (None, res):
        fn = lambda: ops.conv1d(x, cw, b, dil=1, pad=pad, impl="bf16", out=y, res=r)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 50)
    chk = float(y.double().abs().sum())
    print("B%2d %4d->%4d k%d T%4d | no residual %6.1f us | with residual %6.1f us | checksum %.6e" % (B, Cin, Cout, K, T, out[0], out[1], chk), flush=True)
