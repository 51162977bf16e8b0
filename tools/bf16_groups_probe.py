import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
import set_amd
from set_amd import ops
from set_amd.synthetic import synthetic_inputs
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model = bench.build_model(dev, 100)
inp = {k: v.to(dev) for k, v in synthetic_inputs(32, 800, 100, seed=1234).items()}
ops.set_compute_dtype("bf16")
for G in (1, 2, 4):
    f = lambda s: model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"], inp["uv"], infer=True, seed=s, n_groups=G)
    f(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3): f(i + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("bf16 loop groups=%d: %.1f ms per 100-step batch, %.0f frames/s" % (G, dt * 1e3, 32 * 800 / dt), flush=True)
