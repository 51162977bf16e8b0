"""Per-shape times of the bf16 conv forward / input-gradient kernel and the bf16 weight-gradient path at the shapes of the CampNet
step (B = 16, T = 800: 12,800 frames) and of the spec_denoiser conditioner (B = 32: 25,600 frames), next to the time their HBM
bytes take at 5 TB/s (fp32 activations in, fp32 out; for the weight gradient: both operands in)."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops, autograd_ops as ao, _lib
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
# (B, Cin, Cout, K, T)
SHAPES = [(16, 192, 192, 1, 800), (16, 192, 576, 1, 800), (16, 576, 192, 1, 800), (16, 768, 192, 1, 800), (16, 192, 768, 9, 800),
          (16, 768, 192, 9, 800), (16, 192, 384, 5, 800), (16, 80, 192, 1, 800),
          (32, 256, 256, 1, 800), (32, 256, 768, 1, 800), (32, 1024, 256, 1, 800), (32, 256, 1024, 9, 800), (32, 1024, 256, 9, 800)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


ops.set_compute_dtype("bf16")
if True:
    for (B, Cin, Cout, K, T) in SHAPES:
        x = torch.randn(B, Cin, T, generator=g).to(dev)
        gy = torch.randn(B, Cout, T, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)).to(dev)
        b = torch.zeros(Cout, device=dev)
        cw = ops.ConvWeight(lambda: w, Cout, Cin, K)
        pad = (K - 1) // 2
        y = torch.empty(B, Cout, T, device=dev)
        dw = torch.zeros(Cout, Cin, K, device=dev)
        us_f = timed(lambda: ops.conv1d(x, cw, b, dil=1, pad=pad, impl="bf16", out=y))
        us_w = timed(lambda: ao.conv_wgrad(gy, x, None, dw, B, Cin, Cout, K, 1, pad, T, T, dtype=_lib.DTYPE_BF16))
        flop = 2.0 * B * T * Cin * Cout * K
        by_f = 4.0 * B * T * (Cin + Cout)
        print("B%2d %4d->%4d k%d T%4d | fwd %7.1f us %6.1f TF/s (HBM floor %5.1f us) | wgrad+reduce %7.1f us %6.1f TF/s (floor %5.1f us)" % (
            B, Cin, Cout, K, T, us_f, flop / us_f / 1e6, by_f / 5e6, us_w, flop / us_w / 1e6, by_f / 5e6), flush=True)
