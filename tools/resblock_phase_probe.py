"""Phase shares of the fused ResBlock-pair kernel (s_memtime ticks of sampled blocks, set_debug_resblock_phase_buffer) and its
time per launch at the HiFi-GAN V1 stage shapes (B = 64), next to the pair's HBM floor (read x + write out) and its MFMA time at
the measured split-operand rate."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import _lib, ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.split_convs().__enter__()  # the scope the vocoder forward runs in
NAMES = ("x wait+split", "gemm1 issue", "gemm1 drain+epi1", "gemm2 issue", "gemm2 drain+epi2")
B = int(os.environ.get("B", "64"))
g = torch.Generator().manual_seed(0)
buf = torch.zeros(8, dtype=torch.int64, device=dev)
shapes = [(256, 6400), (128, 51200), (64, 102400), (32, 204800)]
for C, T in shapes:
    x = torch.randn(B, C, T, device=dev)
    out = torch.empty_like(x)
    for K in (3, 7, 11):
        for dil in (1, 5):
            if not ops.resblock_pair_eligible(C, K, dil, T):
                continue
            ws = [(torch.randn(C, C, K, generator=g) / math.sqrt(C * K)).to(dev) for _ in range(2)]
            cw = [ops.ConvWeight((lambda w=w: w), C, C, K) for w in ws]
            b1, b2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            for acc in (False, True):
                ops.resblock_pair(x, cw[0], b1, cw[1], b2, dil, out=out, accumulate=acc)
                torch.cuda.synchronize()
                buf.zero_()
                _lib.check(_lib.lib().set_debug_resblock_phase_buffer(buf.data_ptr()), "dbg")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 3
                e0.record()
                for _ in range(n):
                    ops.resblock_pair(x, cw[0], b1, cw[1], b2, dil, out=out, accumulate=acc)
                e1.record(); torch.cuda.synchronize()
                _lib.lib().set_debug_resblock_phase_buffer(None)
                us = e0.elapsed_time(e1) * 1e3 / n
                st = buf.cpu().tolist()
                tot, nb = sum(st[:5]), max(1, st[7])
                floor_us = (2 + (1 if acc else 0)) * B * C * T * 4 / 5.0e6  # at 5 TB/s
                mfma_us = 3 * 2 * 2.0 * B * T * C * C * K / 1.1e9            # 2 convs x 3 products at 1.1 PFLOP/s
                print("C=%3d T=%6d K=%2d dil=%d acc=%d: %7.1f us (HBM floor %6.1f, MFMA %6.1f) | block life %5.1f us (100 MHz ticks) | %s" % (
                    C, T, K, dil, acc, us, floor_us, mfma_us, tot / nb / 100.0,
                    " ".join("%s %.0f%%" % (nm, 100.0 * v / max(tot, 1)) for nm, v in zip(NAMES, st))), flush=True)
    del x, out
