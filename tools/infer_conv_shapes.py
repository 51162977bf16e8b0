"""Round 6: which generic conv launches the reverse loop of the metric's configuration issues outside the stack kernel (shape, count, time by hipEvents).
usage: python tools/infer_conv_shapes.py"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from set_amd import ops  # noqa: E402
from set_amd.synthetic import synthetic_inputs  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
model = bench.build_model(dev, bench.DIFF_STEPS)
inp = {k: v.to(dev) for k, v in synthetic_inputs(bench.B_PER_GPU, bench.T, bench.T_TXT, seed=1234).items()}


def step(seed):
    return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"], inp["uv"], infer=True, seed=seed)


for _ in range(2):
    step(1)
torch.cuda.synchronize()
ops.CONV_EVENTS = []
step(2)
torch.cuda.synchronize()
ev, ops.CONV_EVENTS = ops.CONV_EVENTS, None
agg = collections.OrderedDict()
for key, e0, e1 in ev:
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("generic MFMA conv launches per loop: %d, %.3f ms" % (len(ev), tot))
for (cin, cout, k, b, t, impl), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %4d -> %4d k=%d  B=%d T=%d %-6s x%3d  %.3f ms total, %.1f us each" % (cin, cout, k, b, t, impl, n, ms, 1e3 * ms / n))
