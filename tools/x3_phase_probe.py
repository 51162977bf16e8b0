"""Phase timing of block 0 of the split-operand stack kernel (s_memtime ticks, summed over its tasks) and the kernel time
per 20-layer launch at the benchmark shape."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib, ops
dev = torch.device("cuda:0")
L = 20
NAMES = ("claim+wait", "stage", "gemm1", "gate", "gemm2", "epi+publish")
g = torch.Generator().manual_seed(1)
w1 = torch.empty(L, 512 * 768, device=dev); w2 = torch.empty(L, 512 * 256, device=dev)
wx3 = ops.SplitOperandImages(L, ops.split_operand_mode(), dev)
for l in range(L):
    wd, wo = (torch.randn(512, 256, 3, generator=g) / 27.7).to(dev), (torch.randn(512, 256, 1, generator=g) / 16).to(dev)
    ops.pack_diffnet_layer(wd, wo, w1[l], w2[l]); wx3.pack(l, wd, wo)
bd = torch.zeros(L, 512, device=dev); bo = torch.zeros(L, 512, device=dev)
packs = (w1, w2, bd, bo, None, None, None, None, wx3)
buf = torch.zeros(32, dtype=torch.int64, device=dev)
os.environ["SET_AMD_X3"] = "2"
for B, T in [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "32x800").split(",")]:
    x0 = torch.randn(B, 256, T, device=dev); cp = torch.randn(B, L * 512, T, device=dev) * 0.5
    dtab = torch.randn(L * 256, 100, device=dev)
    xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
    ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
    torch.cuda.synchronize()
    buf.zero_()
    _lib.check(_lib.lib().set_debug_x3_phase_buffer(buf.data_ptr()), "dbg")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
    e1.record(); torch.cuda.synchronize()
    _lib.lib().set_debug_x3_phase_buffer(None)
    us = e0.elapsed_time(e1) * 1000 / n
    st = buf.cpu().tolist()
    tot = sum(st[:6]) + sum(st[8:13]); ntask = max(1, st[7])
    NAMES = ("claim+wait", "stage", "gemm1", "gate:tail-barrier", "gemm2", "epi+publish", "-", "-", "gate:xres+barrier", "gate:math+lds", "gate:init", "boundary:issue", "boundary:drain")
    if os.environ.get("X3_NAMES"):  # other kernels of the family stamp other phases (x3v: see x3v_main)
        NAMES = tuple(os.environ["X3_NAMES"].split(","))
    if tot == 0:  # the shipped library has no phase stamps (SET_X3_PROBE=0): time only
        print("B=%d T=%d: %.1f us per 20-layer launch (no phase stamps in this build: tools/build_exp.sh probe diffnet_x3.hip -DSET_X3_PROBE=1, "
              "then SET_AMD_LIB=build/exp/libset_amd_probe.so)" % (B, T, us))
        continue
    print("mode %d, ticks per us: %.1f" % (wx3.mode, tot / (us * n)))
    print("B=%d T=%d: %.1f us per 20-layer launch; block 0: %d tasks/launch, %.1f us per task | share: %s" % (
        B, T, us, ntask // n, us / (ntask / n), " ".join("%s %.1f%%" % (nm, 100.0 * v / tot) for nm, v in zip(NAMES, st) if nm != "-")))
