"""Times of the three grouped weight-gradient launches of the DiffNet training backward (20 layers, B = 32, T = 800: dilated conv,
conditioner projection, output projection; bf16 output gradients saved by the layer kernels) with hipEvents, per variant switch:
SET_AMD_WGRAD3_UNITS=0|1 (three-copy / one-copy 3-tap kernel).  Prints TFLOP/s and checks that the variants agree bit for bit."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import autograd_ops as ao, _lib
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
L_, B, C_, H, T = int(os.environ.get("L", 20)), int(os.environ.get("B", 32)), 256, 192, int(os.environ.get("T", 800))
DIL = int(os.environ.get("DIL", 1))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def q4(t):  # bf16 operands are channel-quad interleaved in memory: [..][C / 4][T][4] (csrc/diffnet_bf16.hip)
    s = t.shape
    return t.reshape(*s[:-2], s[-2] // 4, 4, s[-1]).transpose(-1, -2).contiguous().view(s)


dy16 = q4(torch.randn(L_, B, 2 * C_, T, generator=g).to(dev).to(torch.bfloat16))
do16 = q4(torch.randn(L_, B, 2 * C_, T, generator=g).to(dev).to(torch.bfloat16))
z16 = q4(torch.randn(L_, B, C_, T, generator=g).to(dev).to(torch.bfloat16))
x_all = torch.randn(L_, B, C_, T, generator=g).to(dev)
dl_all = torch.randn(L_, B, C_, generator=g).to(dev)
cond = torch.randn(B, H, T, generator=g).to(dev)
n_act = B * 2 * C_ * T
dw_d = torch.zeros(L_, 2 * C_, C_, 3, device=dev)
dw_c = torch.zeros(L_, 2 * C_, H, 1, device=dev)
dw_o = torch.zeros(L_, 2 * C_, C_, 1, device=dev)
G16, GX16 = _lib.DTYPE_BF16_G16, _lib.DTYPE_BF16_G16_X16


def run_d(): ao.conv_wgrad_grouped(dy16, x_all, dl_all, dw_d.data_ptr(), L_, n_act, B * C_ * T, B * C_, dw_d[0].numel(), B, C_, 2 * C_, 3, DIL, DIL, T, T, G16)
def run_c(): ao.conv_wgrad_grouped(dy16, cond, None, dw_c.data_ptr(), L_, n_act, 0, 0, dw_c[0].numel(), B, H, 2 * C_, 1, 1, 0, T, T, G16)
def run_o(): ao.conv_wgrad_grouped(do16, z16, None, dw_o.data_ptr(), L_, n_act, B * C_ * T, 0, dw_o[0].numel(), B, C_, 2 * C_, 1, 1, 0, T, T, GX16)


res = {}
for units in ("0", "1", "0", "1"):
    os.environ["SET_AMD_WGRAD3_UNITS"] = units
    dw_d.zero_(); run_d(); torch.cuda.synchronize()
    res.setdefault(units, dw_d.clone())
    us = timed(run_d)
    print("dilated conv (3 taps, dil %d) units=%s: %8.1f us  %6.1f TFLOP/s" % (DIL, units, us, 2.0 * L_ * B * T * C_ * 2 * C_ * 3 / us / 1e6), flush=True)
print("one-copy == three-copy (bit for bit):", bool(torch.equal(res["0"], res["1"])))
us = timed(run_c)
print("conditioner projection (192 -> 512, 1 tap): %8.1f us  %6.1f TFLOP/s" % (us, 2.0 * L_ * B * T * H * 2 * C_ / us / 1e6))
us = timed(run_o)
print("output projection (256 -> 512, 1 tap, bf16 z): %8.1f us  %6.1f TFLOP/s" % (us, 2.0 * L_ * B * T * C_ * 2 * C_ / us / 1e6))
