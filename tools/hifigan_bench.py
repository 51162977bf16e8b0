"""HiFi-GAN V1 generator forward throughput at BASELINE config 4 shape (B=64, T=800) + per-kernel breakdown hint."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd.hifigan import HifiGanGenerator
V1 = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
      "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
      "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]}
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, T = int(os.environ.get("HB", 64)), int(os.environ.get("HT", 800))
torch.manual_seed(0)
g = HifiGanGenerator(V1).to(dev).eval()
mel = torch.randn(B, 80, T, device=dev)
for _ in range(2):
    wav = g(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    wav = g(mel)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flop = 2 * 307052544 * B * T
print("HiFi-GAN V1 B=%d T=%d: %.1f ms/forward, %.0f mel-frames/s, %.1f TFLOP/s (fp32), wav %s finite=%s" % (
    B, T, dt * 1e3, B * T / dt, flop / dt / 1e12, tuple(wav.shape), bool(torch.isfinite(wav).all())))

# ---- per-stage / per-resblock breakdown (set HSTAGES=1): where the forward time goes and at what MFMA rate
if os.environ.get("HSTAGES"):
    from set_amd import ops
    if os.environ.get("SET_AMD_VOCODER_SPLIT", "1") != "0":
        ops.split_convs().__enter__()  # the stages below on the kernels the forward uses (two-piece fp16 convs)
        print("(stages timed on the f16x2 conv kernel; TF/s = algorithmic fp32 FLOPs / time)")

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n, out

    x = ops.conv1d(mel, g._pre.conv_weight(), g.conv_pre.bias.data, pad=3)
    for i in range(g.num_upsamples):
        cin, cout, k, u, P = g._up_cfg[i]
        up = g._ups[i]
        up.folded()
        ms, x = timed(lambda: ops.conv_transpose1d(x, lambda up=up: up._w, g.ups[i].bias.data, cin, cout, k, u, P,
                                                    pro="lrelu", pro_param=0.1, cache=up._phases))
        Tn = x.shape[2]
        print("stage %d  up %4d->%4d k%-2d s%d  T=%6d : %7.2f ms  %6.1f TF/s" % (
            i, cin, cout, k, u, Tn, ms, 2.0 * B * cin * cout * k / u * Tn / ms / 1e9))
        xs = torch.empty_like(x)
        nk = g.num_kernels
        for j in range(nk):
            rb = g.resblocks[i * nk + j]
            # the MRF sum / mean rides in the epilogue of the block's final conv (timed 3x into the same buffer: the
            # accumulated values are garbage for j > 0, so the stage output is recomputed once below)
            ms, _ = timed(lambda: rb.run(x, out=xs, accumulate=j > 0, out_div=float(nk) if j == nk - 1 else 0.0))
            fl = 2.0 * B * cout * cout * rb.k * Tn * 6
            print("stage %d  resblock C=%3d k=%-2d dil=%s T=%6d : %7.2f ms  %6.1f TF/s  (%.2f ms/conv)" % (
                i, cout, rb.k, rb.dil, Tn, ms, fl / ms / 1e9, ms / 6))
        for j in range(nk):
            g.resblocks[i * nk + j].run(x, out=xs, accumulate=j > 0, out_div=float(nk) if j == nk - 1 else 0.0)
        x = xs
