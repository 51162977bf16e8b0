"""Long soak of the persistent Winograd kernel: N launches per shape, every result compared bit for bit with the first."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
N = int(os.environ.get("SOAK_N", 600))
SHAPES = ((32, 800, 20, 1, "2"), (16, 800, 20, 1, "2"), (7, 1548, 20, 1, "2"), (32, 800, 8, 4, "2"), (24, 797, 20, 1, "2"),
          (8, 800, 20, 1, "0"), (32, 800, 20, 1, "0"), (3, 203, 20, 3, "0"))  # last column: SET_AMD_WINO (0 = direct kernels)
for (B, T, L, dcl, wino) in SHAPES:
    g = torch.Generator().manual_seed(B * 7 + T)
    x0 = torch.randn(B, 256, T, generator=g).to(dev)
    cp = (torch.randn(B, L * 512, T, generator=g) * 0.5).to(dev)
    dtab = torch.randn(L * 256, 1, generator=g).to(dev)
    w1 = torch.empty(L, 512 * 768, device=dev); w2 = torch.empty(L, 512 * 256, device=dev)
    w1w = torch.empty(L, 512 * 256 * 4, device=dev); w2w = torch.empty(L, 512 * 256, device=dev)
    bd = (torch.randn(L, 512, generator=g) * 0.1).to(dev); bo = (torch.randn(L, 512, generator=g) * 0.1).to(dev)
    for l in range(L):
        wd = (torch.randn(512, 256, 3, generator=g) / 27.7).to(dev); wo = (torch.randn(512, 256, 1, generator=g) / 16.0).to(dev)
        ops.pack_diffnet_layer(wd, wo, w1[l], w2[l]); ops.pack_diffnet_layer_wino(wd, wo, w1w[l], w2w[l])
    os.environ["SET_AMD_WINO"] = wino
    ref, bad, aborts = None, 0, 0
    for it in range(N):
        xa, xb, skip = x0.clone(), torch.empty_like(x0), torch.empty_like(x0)
        ws = ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 1, 256, (w1, w2, bd, bo, w1w, w2w), dcl)
        out = xa if L % 2 == 0 else xb
        if ref is None:
            ref = (out.clone(), skip.clone())
        else:
            bad += int(not (torch.equal(out, ref[0]) and torch.equal(skip, ref[1])))
        if it % 50 == 0:
            aborts += int(ws[1])
    print("B=%d T=%d L=%d dcl=%d %s: %d launches, %d mismatches, aborts %d" % (
        B, T, L, dcl, "winograd" if wino == "2" else "direct", N, bad, aborts))
