"""Phase timing of one block of the fused bf16 layer forward kernel (s_memtime stamps, 100 MHz reference clock)."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib
dev = torch.device("cuda:0")
B, T, Cc, H = 32, 800, 256, 192
L = _lib.lib()
n = L.set_diffnet_layer_bf16_image_size()
img = torch.empty(n, dtype=torch.bfloat16, device=dev)
wd, wc, wo = torch.randn(512, 256, 3, device=dev) * 0.03, torch.randn(512, 192, device=dev) * 0.05, torch.randn(512, 256, device=dev) * 0.05
_lib.check(L.set_pack_diffnet_layer_bf16(wd.data_ptr(), wc.data_ptr(), wo.data_ptr(), img.data_ptr(), None), "pack")
x, xo, sk = (torch.randn(B, Cc, T, device=dev) for _ in range(3))
cond = torch.randn(B, H, T, device=dev)
dst = torch.randn(B, Cc, device=dev)
bias = [torch.zeros(512, device=dev) for _ in range(3)]
buf = torch.zeros(8, dtype=torch.int64, device=dev)
_lib.check(L.set_debug_bf16_phase_buffer(buf.data_ptr()), "dbg")
a = _lib.SetDiffnetLayerBf16Args()
a.x_in, a.x_out, a.skip, a.cond, a.dstep, a.img = x.data_ptr(), xo.data_ptr(), sk.data_ptr(), cond.data_ptr(), dst.data_ptr(), img.data_ptr()
a.b_dil, a.b_cond, a.b_out = (b.data_ptr() for b in bias)
a.d_bs, a.d_cs, a.B, a.T, a.dil, a.first = Cc, 1, B, T, 1, 0
if os.environ.get("TRAIN"):  # the training form: y / z saved in bf16 (quad-interleaved)
    y16, z16 = torch.empty(B, 2 * Cc, T, dtype=torch.bfloat16, device=dev), torch.empty(B, Cc, T, dtype=torch.bfloat16, device=dev)
    a.y16, a.z16 = y16.data_ptr(), z16.data_ptr()
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(L.set_diffnet_layer_fwd_bf16(C.byref(a), None), "fwd")
    e1.record(); torch.cuda.synchronize()
    st = buf.cpu().tolist()
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(5)]
    print("tile %s: kernel %.1f us | stage %.1f gemm1 %.1f gate %.1f gemm2 %.1f epilogue %.1f (us, block (1,1))" % (
        os.environ.get("SET_AMD_BF16_TILE", "128"), e0.elapsed_time(e1) * 1000 / 20, *d))
L.set_debug_bf16_phase_buffer(None)
