"""set_ssim_filter / set_ssim_bwd alone on the GPU at the training shapes: time per launch and a bit-level checksum of every output (run it with
SET_AMD_LIB=<another build> and compare the checksums: the LDS-tiled kernels must reproduce the per-pixel gather bit for bit).
  python tools/ssim_probe.py"""
import hashlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib
from set_amd.ops import _p, _stream
dev = torch.device("cuda:0")
L = _lib.lib()
for B, T, M in ((32, 800, 80), (16, 800, 80), (3, 77, 80), (2, 21, 13)):
    g = torch.Generator().manual_seed(B * T + M)
    img1 = (torch.randn(B, T, M, generator=g) * 1.5 - 3.0).to(dev)
    img2 = (torch.randn(B, T, M, generator=g) * 1.5 - 3.0).to(dev)
    maps = [torch.empty(B, T, M, device=dev) for _ in range(5)]
    ups = [torch.randn(B, T, M, generator=g).to(dev) for _ in range(3)]
    dimg = torch.empty(B, T, M, device=dev)
    f = lambda: _lib.check(L.set_ssim_filter(_p(img1), _p(img2), 6.0, *[_p(m) for m in maps], B, T, M, _stream()), "f")
    bw = lambda: _lib.check(L.set_ssim_bwd(_p(img1), _p(img2), 6.0, _p(ups[0]), _p(ups[1]), _p(ups[2]), _p(dimg), B, T, M, _stream()), "b")
    res = []
    for fn in (f, bw):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    h = hashlib.sha256()
    for t in maps + [dimg]:
        h.update(t.cpu().numpy().tobytes())
    print("B=%2d T=%3d M=%2d | filter %6.1f us | backward %6.1f us | sha256 of the 6 outputs %s" % (B, T, M, res[0], res[1], h.hexdigest()[:16]), flush=True)
