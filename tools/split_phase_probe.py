"""Phase timing of one block of the row-split stack kernel (s_memtime ticks at 100 MHz, summed over the layers) and the
kernel time per 20-layer launch, for small batches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib, ops
dev = torch.device("cuda:0")
L = 20
NAMES = ("wait_prev", "stage", "gemm1", "gate+zpub", "wait_z", "zload", "gemm2", "epi+pub")
g = torch.Generator().manual_seed(1)
w1 = torch.empty(L, 512 * 768, device=dev); w2 = torch.empty(L, 512 * 256, device=dev)
for l in range(L):
    ops.pack_diffnet_layer((torch.randn(512, 256, 3, generator=g) / 27.7).to(dev), (torch.randn(512, 256, 1, generator=g) / 16).to(dev), w1[l], w2[l])
bd = torch.zeros(L, 512, device=dev); bo = torch.zeros(L, 512, device=dev)
wx3 = None
if os.environ.get("X2", "1") != "0":  # the two-piece fp16 variant of the row-split kernel
    wx3 = ops.SplitOperandImages(L, 2, dev)
    g2 = torch.Generator().manual_seed(1)
    for l in range(L):
        wx3.pack(l, (torch.randn(512, 256, 3, generator=g2) / 27.7).to(dev), (torch.randn(512, 256, 1, generator=g2) / 16).to(dev))
packs = (w1, w2, bd, bo, None, None) + ops.split_images(w1, w2) + (wx3,)
buf = torch.zeros(32, dtype=torch.int64, device=dev)
for B, T in [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "1x800,2x800,4x800").split(",")]:
    x0 = torch.randn(B, 256, T, device=dev); cp = torch.randn(B, L * 512, T, device=dev) * 0.5
    dtab = torch.randn(L * 256, 100, device=dev)
    xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
    for mode in ("2", "0"):
        os.environ["SET_AMD_SPLIT"] = mode
        ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
        torch.cuda.synchronize()
        buf.zero_()
        if mode == "2":
            _lib.check((_lib.lib().set_debug_x3_phase_buffer if wx3 is not None else _lib.lib().set_debug_split_phase_buffer)(buf.data_ptr()), "dbg")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
        e1.record(); torch.cuda.synchronize()
        _lib.lib().set_debug_split_phase_buffer(None); _lib.lib().set_debug_x3_phase_buffer(None)
        us = e0.elapsed_time(e1) * 1000 / n
        line = "B=%d T=%d split=%s: %.1f us per 20-layer launch (%.1f us / layer)" % (B, T, mode, us, us / L)
        if mode == "2":
            st = buf.cpu().tolist()
            line += " | per layer, us: " + " ".join("%s %.2f" % (nm, v / 100.0 / (n * L)) for nm, v in zip(NAMES, st))
        print(line)
