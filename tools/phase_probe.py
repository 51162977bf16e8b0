"""Diagnostic: per-phase s_memtime breakdown of diffnet_layer_kernel at the benchmark shape (B=32, T=800)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, T, C = int(os.environ.get("PB", 32)), int(os.environ.get("PT", 800)), 256
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, T, generator=g).to(dev)
cp = torch.randn(B, 512, T, generator=g).to(dev)
d = torch.randn(C, generator=g).to(dev)
wd = (torch.randn(512, 256, 3, generator=g) / 27.7).to(dev)
wo = (torch.randn(512, 256, 1, generator=g) / 16).to(dev)
bd, bo = torch.randn(512, generator=g).to(dev) * 0.1, torch.randn(512, generator=g).to(dev) * 0.1
w1p, w2p = ops.pack_diffnet_layer(wd, wo)
xo, sk = torch.empty_like(x), torch.zeros_like(x)
nblk = ((T + 63) // 64) * B
clk = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
for it in range(3):
    ops.diffnet_layer(x, cp.data_ptr(), 512 * T, d.data_ptr(), 0, 1, w1p, bd, w2p, bo, xo, sk, 1, True, dbg_clock=clk)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
ops.diffnet_layer(x, cp.data_ptr(), 512 * T, d.data_ptr(), 0, 1, w1p, bd, w2p, bo, xo, sk, 1, True, dbg_clock=clk)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1)
c = clk.cpu().numpy().astype(np.uint64)
d = (c[:, 1:7] - c[:, 0:6]).astype(np.float64)  # per-block deltas (s_memtime bases differ per XCD: only deltas are meaningful)
tot = (c[:, 6] - c[:, 0]).astype(np.float64)
labels = ["stage x", "GEMM1", "gate math", "barrier+z store", "GEMM2", "epilogue"]
print("kernel %.1f us (hipEvent);  s_memtime ticks below; tick/us if block==kernel: %.1f" % (ms * 1e3, tot.max() / (ms * 1e3)))
for i, n in enumerate(labels):
    print("%-16s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (
        n, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 50), np.percentile(d[:, i], 90), d[:, i].max()))
print("block total      mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  max %9.0f" % (
    tot.mean(), np.percentile(tot, 10), np.percentile(tot, 50), np.percentile(tot, 90), tot.max()))
print("histogram of block totals (ticks):", np.histogram(tot, bins=8))
