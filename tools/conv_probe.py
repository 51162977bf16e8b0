"""Diagnostic: time single conv shapes (v1 mfma vs v2) at the DiffNet training shapes."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
SHAPES = [(32, 256, 512, 3, 1, 800), (32, 512, 256, 3, -1, 800), (32, 256, 512, 1, 1, 800), (32, 512, 256, 1, 1, 800),
          (32, 192, 512, 1, 1, 800), (32, 512, 192, 1, 1, 800), (32, 80, 256, 1, 1, 800), (32, 192, 192, 5, 1, 800),
          (32, 192, 192, 5, 1, 100), (32, 192, 384, 9, 1, 800), (16, 256, 256, 1, 1, 800), (16, 256, 1024, 9, 1, 800),
          (16, 256, 256, 3, 1, 6400), (16, 256, 256, 7, 3, 6400), (16, 256, 256, 11, 5, 6400),
          (16, 128, 128, 3, 1, 51200), (16, 128, 128, 7, 1, 51200), (16, 128, 128, 11, 1, 51200)]
for (B, Cin, Cout, K, dil, T) in SHAPES:
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)).to(dev)
    b = torch.zeros(Cout, device=dev)
    cw = ops.ConvWeight(lambda: w, Cout, Cin, K)
    pad = abs(dil) * (K - 1) // 2 * (1 if dil > 0 else -1)
    for impl in ("mfma", "mfma2"):
        if impl == "mfma2" and Cout < 96:
            continue
        for _ in range(2):
            y = ops.conv1d(x, cw, b, dil=dil, pad=pad, impl=impl, T_out=T, T_iter=T)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = ops.conv1d(x, cw, b, dil=dil, pad=pad, impl=impl, T_out=T, T_iter=T)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("B%d %4d->%4d k%d dil%2d T%5d %-6s %8.1f us  %6.1f TF/s" % (B, Cin, Cout, K, dil, T, impl, ms * 1e3, 2.0 * B * Cin * Cout * K * T / ms / 1e9))
