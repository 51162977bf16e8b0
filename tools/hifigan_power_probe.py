"""Round 6: socket power / shader clock while the HiFi-GAN V1 forward (B = 64, T = 800, shipped split-operand path) runs back to back --
is the vocoder at the package power cap like the DiffNet stack kernel?  usage: python tools/hifigan_power_probe.py [seconds]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import set_amd  # noqa
from set_amd.hifigan import HifiGanGenerator
from power_probe import Sampler, hwmon_files

V1 = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4], "upsample_initial_channel": 512,
      "resblock_kernel_sizes": [3, 7, 11], "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]}
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
torch.manual_seed(0)
g = HifiGanGenerator(V1).to(dev).eval()
mel = torch.randn(64, 80, 800, device=dev)
for _ in range(3):
    g(mel)
torch.cuda.synchronize()
smp = Sampler(hwmon_files())
smp.start()
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < secs:
    g(mel)
    n += 1
torch.cuda.synchronize()
wall = time.perf_counter() - t0
smp.stop_flag = True
smp.join()
rows = smp.rows[len(smp.rows) // 4:]
pw = [r.get("power1_average", r.get("power1_input", 0.0)) / 1e6 for r in rows]
fq = [r.get("freq1_input", 0.0) / 1e6 for r in rows]
print("HiFi-GAN V1 B=64 T=800: %.1f ms per forward over %d forwards; socket power %.0f W mean / %.0f max, sclk %.0f MHz mean" % (
    1e3 * wall / n, n, sum(pw) / max(1, len(pw)), max(pw or [0]), sum(fq) / max(1, len(fq))))
