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
