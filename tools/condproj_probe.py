"""Round 6: the hoisted conditioner projections of the reverse loop (20 x 192 -> 512 1x1 convs at B = 32, T = 800) as 20 launches against ONE launch
on the stacked weight (192 -> 10240), fp32 MFMA kernel both; results must be bit-identical.  usage: python tools/condproj_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, H, T, L, C2 = 32, 192, 800, 20, 512
g = torch.Generator().manual_seed(3)
ws = [(torch.randn(C2, H, 1, generator=g) / 14).to(dev) for _ in range(L)]
bs = [torch.randn(C2, generator=g).to(dev) for _ in range(L)]
cond = torch.randn(B, H, T, generator=g).to(dev)
cws = [ops.ConvWeight(lambda w=w: w, C2, H, 1) for w in ws]
wall = torch.cat(ws, 0).contiguous()
ball = torch.cat(bs, 0).contiguous()
cwall = ops.ConvWeight(lambda: wall, L * C2, H, 1)
out_a = torch.empty(B, L * C2, T, device=dev)
out_b = torch.empty_like(out_a)


def per_layer():
    for l in range(L):
        ops.conv1d(cond, cws[l], bs[l], out=out_a[:, l * C2:(l + 1) * C2, :])


def stacked():
    ops.conv1d(cond, cwall, ball, out=out_b)


for f in (per_layer, stacked):
    for _ in range(3):
        f()
torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(out_a, out_b)))
for name, f in (("20 launches", per_layer), ("1 stacked launch", stacked), ("20 launches", per_layer), ("1 stacked launch", stacked)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    print("%-18s %.3f ms per set" % (name, e0.elapsed_time(e1) / 10))
