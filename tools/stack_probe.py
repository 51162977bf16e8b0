"""Diagnostic: persistent stack kernel vs L per-layer launches at B=32,T=800,L=20 + wait/fence tick totals."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, T, L = int(os.environ.get("PB", 32)), int(os.environ.get("PT", 800)), 20
g = torch.Generator().manual_seed(0)
x0 = torch.randn(B, 256, T, generator=g).to(dev)
cp = (torch.randn(B, L * 512, T, generator=g) * 0.5).to(dev)
dtab = torch.randn(L * 256, 1, generator=g).to(dev)
w1 = torch.randn(L, 512 * 768, generator=g).to(dev) / 27.7
w2 = torch.randn(L, 512 * 256, generator=g).to(dev) / 16
bd = torch.zeros(L, 512, device=dev); bo = torch.zeros(L, 512, device=dev)
xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
def stack():
    return ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 1, 256, (w1, w2, bd, bo), 1)
def layers():
    h, nxt = xa, xb
    for l in range(L):
        ops.diffnet_layer(h, cp[:, l * 512:(l + 1) * 512].data_ptr(), cp.stride(0), dtab.data_ptr() + 4 * l * 256, 0, 1,
                          w1[l], bd[l], w2[l], bo[l], nxt, skip, 1, l == 0)
        h, nxt = nxt, h
for fn, name in ((layers, "20 per-layer launches"), (stack, "persistent stack")):
    for _ in range(2): ws = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ws = fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-24s %.3f ms  (%.1f us per layer, %.1f TF/s)" % (name, ms, ms * 1e3 / L, 2 * 524288 * B * T * L / ms / 1e9))
    if ws is not None:
        w = ws[:4].cpu().tolist()
        nblk = int(os.environ.get('SET_AMD_STACK_GRID', 512))
        print("   tasks grabbed %d, abort %d, wait %.0f kticks/block, fence %.0f kticks/block (kernel ~ %.0f kticks)" % (
            w[0], w[1], w[2] / nblk, w[3] / nblk, ms * 2.1e3))
