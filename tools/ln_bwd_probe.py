"""Solo timing of set_layernorm_ch_bwd(_add) at the shapes of the training steps (nothing else on the GPU): how much of the 29 us per launch in the
step profiles is the kernel itself and how much is contention with the leaf stream's weight-gradient GEMMs.
  python tools/ln_bwd_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib
from set_amd.ops import _p, _stream
dev = torch.device("cuda:0")
L = _lib.lib()
for name, B, C, T in (("CampNet B=16 C=256 T=800", 16, 256, 800), ("spec_denoiser text B=32 C=192 T=100", 32, 192, 100),
                      ("spec_denoiser frames B=32 C=192 T=800", 32, 192, 800)):
    x, dy, add = (torch.randn(B, C, T, device=dev) for _ in range(3))
    gamma = torch.randn(C, device=dev)
    dx, dg, db = torch.empty_like(x), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    part = torch.empty(L.set_layernorm_ch_bwd_scratch(B, C, T) + 4096, device=dev)
    def run():
        _lib.check(L.set_layernorm_ch_bwd_add(_p(x), _p(gamma), None, _p(dy), _p(add), _p(dx), _p(dg), _p(db), _p(part), B, C, T, 1e-5, _stream()), "ln")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    mb = 4 * B * C * T * 4 / 1e6
    print("%-40s %.1f us per call (2 launches), %.0f MB -> %.2f TB/s" % (name, us, mb, mb / us))
