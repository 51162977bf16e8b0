"""Split-operand stack kernel, block shape A/B at the benchmark shape: one 8-wave block per CU against two 4-wave blocks
per CU (SET_AMD_X3_WAVES), over worker counts (SET_AMD_STACK_GRID).  Prints us per 20-layer launch, checks that every
configuration gives the same bits, and the phase shares of block 0 (s_memtime) for the two defaults."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib, ops
dev = torch.device("cuda:0")
L = 20
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
NAMES = ("claim+wait", "stage", "gemm1", "gate:tail-barrier", "gemm2", "epi+publish", "-", "-", "gate:xres+barrier", "gate:math+lds",
         "gate:init", "boundary:issue", "boundary:drain")


def run(xa, xb, skip, cp, dtab, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for B, T in [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SIZES", "32x800").split(",")]:
    x0 = torch.randn(B, 256, T, device=dev); cp = torch.randn(B, L * 512, T, device=dev) * 0.5
    dtab = torch.randn(L * 256, 100, device=dev)
    ref = None
    grids = {"8": os.environ.get("GRIDS8", "0,224,256").split(","), "4": os.environ.get("GRIDS4", "0,256,288,320,352,384,400,448,512").split(",")}
    for waves in ("8", "4"):
        os.environ["SET_AMD_X3_WAVES"] = waves
        for grid in grids[waves]:
            if grid == "0":
                os.environ.pop("SET_AMD_STACK_GRID", None)
            else:
                os.environ["SET_AMD_STACK_GRID"] = grid
            xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
            ws = ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
            torch.cuda.synchronize()
            out = (xa.clone(), skip.clone())
            if ref is None:
                ref = out
            same = torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1])
            us = min(run(xa, xb, skip, cp, dtab, 10) for _ in range(3))
            print("B=%d T=%d waves=%s grid=%-7s %8.1f us per 20-layer launch  err=%d  bits_equal=%s" % (
                B, T, waves, grid if grid != "0" else "default", us, int(ws[1]), same), flush=True)
        # phase shares of block 0 (default grid + PHASE_GRIDS)
        for grid in ["0"] + [g_ for g_ in os.environ.get("PHASE_GRIDS" + waves, "").split(",") if g_]:
            if grid == "0":
                os.environ.pop("SET_AMD_STACK_GRID", None)
            else:
                os.environ["SET_AMD_STACK_GRID"] = grid
            xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
            buf.zero_()
            _lib.check(_lib.lib().set_debug_x3_phase_buffer(buf.data_ptr()), "dbg")
            n = 10
            us = run(xa, xb, skip, cp, dtab, n)
            _lib.lib().set_debug_x3_phase_buffer(None)
            st = buf.cpu().tolist()
            tot = sum(st[:6]) + sum(st[8:13]); ntask = max(1, st[7])
            print("  waves=%s grid=%s: ticks per us %.1f; block 0: %.1f tasks/launch, %.1f us per task | %s" % (
                waves, grid, tot / (us * n), ntask / n, us / (ntask / n),
                " ".join("%s %.1f%%" % (nm, 100.0 * v / tot) for nm, v in zip(NAMES, st) if nm != "-")), flush=True)
        os.environ.pop("SET_AMD_STACK_GRID", None)
    os.environ.pop("SET_AMD_X3_WAVES", None)
