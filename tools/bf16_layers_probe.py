"""Phase shares of the layers m >= 1 of a fused group (set_diffnet_layers_fwd_bf16; s_memtime ticks of thread 0 of block (1, 1)) and
the time per launch for several group sizes at B = 32, T = 800."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib
dev = torch.device("cuda:0")
NLS = tuple(int(v) for v in os.environ.get("NLS", "1,2,4,8").split(","))
B, T, NL = 32, 800, max(NLS)
L = _lib.lib()
n = L.set_diffnet_layer_bf16_image_size()
imgs = torch.empty(NL, n, dtype=torch.bfloat16, device=dev)
for l in range(NL):
    wd, wc, wo = torch.randn(512, 256, 3, device=dev) * 0.03, torch.randn(512, 192, device=dev) * 0.05, torch.randn(512, 256, device=dev) * 0.05
    _lib.check(L.set_pack_diffnet_layer_bf16(wd.data_ptr(), wc.data_ptr(), wo.data_ptr(), imgs[l].data_ptr(), None), "pack")
x, xo, sk = (torch.randn(B, 256, T, device=dev) for _ in range(3))
cond = torch.randn(B, 192, T, device=dev)
dst = torch.randn(NL, B, 256, device=dev)
bias = [torch.zeros(NL, 512, device=dev) for _ in range(3)]
buf = torch.zeros(16, dtype=torch.int64, device=dev)
NAMES = ("init+barrier", "gemm1", "gate", "gemm2", "epilogue")
for nl in NLS:
    n_ws = L.set_diffnet_layers_bf16_scratch_floats(B, T, 0, nl, 1)
    ws = torch.empty(n_ws, device=dev)
    a = _lib.SetDiffnetLayersBf16Args()
    a.x_in, a.x_out, a.skip, a.cond, a.dstep, a.img = x.data_ptr(), xo.data_ptr(), sk.data_ptr(), cond.data_ptr(), dst.data_ptr(), imgs.data_ptr()
    a.b_dil, a.b_cond, a.b_out = (b.data_ptr() for b in bias)
    a.scratch, a.scratch_floats = ws.data_ptr(), n_ws
    a.d_bs, a.d_cs, a.d_ls, a.B, a.T, a.l0, a.nl, a.dilation_cycle_length, a.first = 256, 1, B * 256, B, T, 0, nl, 1, 0
    _lib.check(L.set_diffnet_layers_fwd_bf16(C.byref(a), None), "warm")
    torch.cuda.synchronize()
    buf.zero_()
    _lib.check(L.set_debug_bf16_phase_buffer(buf.data_ptr()), "dbg")
    # sustained: ~0.6 s of back-to-back launches to bring the clocks up (the GPU idles at 94 MHz between probes: ten launches after a
    # synchronisation measured 43 - 49 us per layer where the sustained loop runs 40 - 41), then REPS timed launches
    import time
    REPS = int(os.environ.get("REPS", "2000"))
    t0 = time.time()
    while time.time() - t0 < 0.6:
        for _ in range(100):
            _lib.check(L.set_diffnet_layers_fwd_bf16(C.byref(a), None), "fwd")
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        _lib.check(L.set_diffnet_layers_fwd_bf16(C.byref(a), None), "fwd")
    e1.record(); torch.cuda.synchronize()
    L.set_debug_bf16_phase_buffer(None)
    us = e0.elapsed_time(e1) * 1000 / REPS
    st = buf.cpu().tolist()
    tot = max(1, sum(st[:5]))
    tot4 = max(1, sum(st[8:13]))
    print("variant=%s tile=%s nl=%d: %.1f us per launch = %.1f us per layer | layers m >= 1, wave 0: %s (%.0f ticks per layer) | wave 4: %s" % (
        os.environ.get("SET_AMD_BF16_REG_VARIANT", "-"), os.environ.get("SET_AMD_BF16_FUSE_TILE", "64"), nl, us, us / nl,
        " ".join("%s %.0f%%" % (nm, 100.0 * v / tot) for nm, v in zip(NAMES, st)) if st[7] else "-", tot / max(1, st[7]),
        " ".join("%s %.0f%%" % (nm, 100.0 * v / tot4) for nm, v in zip(NAMES, st[8:13])) if st[7] and sum(st[8:13]) else "-"))
