"""Diagnostic: Winograd F(2,3) persistent stack vs the direct persistent stack (same real weights): max |diff| + time."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, T, L = int(os.environ.get("PB", 32)), int(os.environ.get("PT", 800)), int(os.environ.get("PL", 20))
g = torch.Generator().manual_seed(0)
x0 = torch.randn(B, 256, T, generator=g).to(dev)
cp = (torch.randn(B, L * 512, T, generator=g) * 0.5).to(dev)
dtab = torch.randn(L * 256, 1, generator=g).to(dev)
wd = (torch.randn(L, 512, 256, 3, generator=g) / 27.7).to(dev)
wo = (torch.randn(L, 512, 256, generator=g) / 16).to(dev)
bd = (torch.randn(L, 512, generator=g) * 0.1).to(dev); bo = (torch.randn(L, 512, generator=g) * 0.1).to(dev)
w1 = torch.empty(L, 512 * 768, device=dev); w2 = torch.empty(L, 512 * 256, device=dev)
w1w = torch.empty(L, 512 * 256 * 4, device=dev); w2w = torch.empty(L, 512 * 256, device=dev)
for l in range(L):
    ops.pack_diffnet_layer(wd[l], wo[l], w1[l], w2[l])
    ops.pack_diffnet_layer_wino(wd[l], wo[l], w1w[l], w2w[l])
res = {}
for mode in ("0", "2"):
    os.environ["SET_AMD_WINO"] = mode
    xa, xb, skip = x0.clone(), torch.zeros_like(x0), torch.zeros_like(x0)
    def stack():
        return ops.diffnet_stack(xa, xb, skip, cp, dtab.data_ptr(), 0, 1, 256, (w1, w2, bd, bo, w1w, w2w), 1)
    ws = stack(); torch.cuda.synchronize()
    out = (xa if L % 2 == 0 else xb).clone(); res[mode] = (out, skip.clone())
    for _ in range(2): stack()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ws = stack()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    w = ws[:4].cpu().tolist()
    if mode == "2" and os.environ.get("WINO_PHASES"):
        nt = B * ((T + 63) // 64)
        ph = ws[4 + nt:4 + nt + 9].cpu().tolist(); tot = float(sum(ph))
        names = ["init+drain", "stage", "gemm1", "gate+prev", "gemm2", "epilogue", "barrier", "-", "flag/wait/claim"]
        print("   phases %%: %s" % "  ".join("%s %.1f" % (n, 100 * v / tot) for n, v in zip(names, ph) if n != "-"))
    print("WINO=%s  %.3f ms  (%.1f us per layer, %.1f algorithmic TF/s)  tasks %d abort %d wait %d fence %d" % (
        mode, ms, ms * 1e3 / L, 2 * 524288 * B * T * L / ms / 1e9, w[0], w[1], w[2], w[3]))
for k, name in ((0, "x"), (1, "skip")):
    a, b = res["0"][k], res["2"][k]
    print("%s: max|direct| %.3f  max|wino - direct| %.3e  finite %s" % (name, a.abs().max().item(), (a - b).abs().max().item(),
                                                                       bool(torch.isfinite(b).all())))
