"""Per-stage phase times of one block of conv1d_bf16_kernel (measurement build: tools/build_exp.sh convprobe bf16.hip -DSET_CONV_PROBE=1, run with
SET_AMD_LIB=build/exp/libset_amd_convprobe.so).  Phases per stage: barrier 1 | wait for the stage's loads + LDS writes | barrier 2 | issue of the
next stage's loads | fragment reads + MFMAs."""
import ctypes as C, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops, _lib
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
L = _lib.lib()
L.set_debug_conv_phase_buffer.argtypes = [C.c_void_p]
buf = torch.zeros(8, dtype=torch.int64, device=dev)
g = torch.Generator().manual_seed(0)
SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "16x768x192x9x800,16x192x768x9x800,32x192x192x5x800").split(",")]
for (B, Cin, Cout, K, T) in SHAPES:
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)).to(dev)
    b = torch.zeros(Cout, device=dev)
    cw = ops.ConvWeight(lambda: w, Cout, Cin, K)
    y = torch.empty(B, Cout, T, device=dev)
    for _ in range(3):
        ops.conv1d(x, cw, b, dil=1, pad=(K - 1) // 2, impl="bf16", out=y)
    torch.cuda.synchronize()
    assert L.set_debug_conv_phase_buffer(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv1d(x, cw, b, dil=1, pad=(K - 1) // 2, impl="bf16", out=y); e1.record(); torch.cuda.synchronize()
    L.set_debug_conv_phase_buffer(None)
    st = buf.cpu().tolist()
    n = max(st[6], 1)
    us = [v / 100.0 for v in st[:6]]
    print("B%d %d->%d k%d T%d: kernel %.1f us, block (1,1,1): %d stages, %.1f us in the stage loop | per stage: barrier1 %.2f  loads+LDS writes %.2f  barrier2 %.2f  "
          "issue %.2f  reads+MFMA %.2f us | prologue %.1f us" % (B, Cin, Cout, K, T, e0.elapsed_time(e1) * 1e3, n, sum(us[1:]), us[1] / n, us[2] / n, us[3] / n, us[4] / n,
                                                                   us[5] / n, us[0]), flush=True)
