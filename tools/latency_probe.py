"""Single-utterance / small-batch latency of the 100-step reverse loop (BASELINE config 1 shape on the GPU path)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import set_amd  # noqa
from set_amd.synthetic import synthetic_inputs
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model = bench.build_model(dev, 100)
SIZES = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("SIZES", "1x400,1x800,4x800,8x800,16x800,32x800,64x800").split(",")]
for B, T in SIZES:
    inp = {k: v.to(dev) for k, v in synthetic_inputs(B, T, 100, seed=1).items()}
    f = lambda s: model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"], inp["uv"], infer=True, seed=s)
    f(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3 if B <= 8 else 2
    for i in range(n): f(i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B=%-3d T=%-4d  %.1f ms per 100-step batch  %.0f frames/s  (%.2f ms per denoise step)" % (B, T, dt * 1e3, B * T / dt, dt * 10))
