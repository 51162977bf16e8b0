"""Timeline of ONE task of every block of diffnet_stack_x3v_kernel (timeline build: tools/build_exp.sh tl diffnet_x3.hip -DSET_X3_PROBE=2, then
SET_AMD_LIB=build/exp/libset_amd_tl.so): every wave keeps s_memtime stamps of its X3V_TL_TASK-th task in SGPRs and waves 0 and 7 store them once,
after the task -- no memory access between the stamps, unlike the summed phase counters of tools/x3_phase_probe.py.  Prints the median (and
10 / 90 % quantiles) of every interval over the blocks, in us (ticks scaled by the median task period = next task's first stamp - this one's)."""
import os, sys
import numpy as np
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
B, T = 32, 800
x0 = torch.randn(B, 256, T, device=dev); cp = torch.randn(B, L * 512, T, device=dev) * 0.5
dtab = torch.randn(L * 256, 100, device=dev)
xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
buf = torch.zeros(256 * 2 * 16, dtype=torch.int64, device=dev)  # 512 rows of 32 dwords
for _ in range(20):  # warm: clocks and power settle
    ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
torch.cuda.synchronize()
_lib.check(_lib.lib().set_debug_x3_phase_buffer(buf.data_ptr()), "dbg")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 100, 256 * 100, packs, 1)
e1.record(); torch.cuda.synchronize()
_lib.lib().set_debug_x3_phase_buffer(None)
us = e0.elapsed_time(e1) * 1000 / n
rows = buf.cpu().numpy().view(np.uint32).reshape(512, 32).astype(np.int64)
ORDER = [(0, "loop top"), (1, "accumulator-start loads + claim issued"), (2, "vmcnt(0): previous stores drained, loads / claim back"), (3, "barrier (task slots)"),
         (4, "dependency flags seen"), (8, "x3v_main entered"), (20, "stage A done (planes 1, 2) + ring preload issued"),
         (9, "barrier"), (10, "GEMM planes 1, 2 done"), (21, "E / O combine + barrier"), (22, "stage B done (planes 0, 3) + preload issued"), (11, "barrier"),
         (12, "GEMM planes 0, 3 done"), (23, "residual loads issued + barrier"), (24, "gate + GEMM 2 accumulator start done"), (16, "barrier"),
         (17, "GEMM 2 done"), (25, "x' stores issued"), (18, "skip loads back, skip stores issued")]
print("B=%d T=%d: %.1f us per 20-layer launch (timeline build)" % (B, T, us))
for wsel, wname in ((0, "wave 0"), (1, "wave 7")):
    r = rows[wsel::2]
    ok = (r[:, 18] > 0) & (r[:, 31] > 0)
    r = r[ok]
    period = (r[:, 31] - r[:, 7]) % (1 << 32)
    tick_us = np.median(period) / (us / (5334 / 256.0))  # ticks per us from the mean task period of the launch
    spun = int((r[:, 6] == 2).sum())
    print("%s: %d blocks, task period median %.0f ticks (%.1f ticks per us by the launch time), dependency wait had to spin in %d" % (wname, len(r), np.median(period), tick_us, spun))
    prev = 0
    for k, name in ORDER:
        d = r[:, k] - (r[:, prev] if k else 0)
        q = np.percentile(d, [10, 50, 90]) / tick_us
        at = np.median(r[:, k]) / tick_us
        print("  +%6.2f us (10%% %6.2f, 90%% %6.2f)  at %6.2f  %s" % (q[1], q[0], q[2], at, name))
        prev = k
    d = period - r[:, 18]
    print("  +%6.2f us  to the next task's loop top" % (np.median(d) / tick_us))
