"""Round 6 A/B harness: the reverse loop of the metric's configuration (B = 32, T = 800, 100 steps, Philox noise) under several environment
variants of the same library build (or, with SET_AMD_LIB, of an experiment build), socket power / shader clock sampled over each sustained run
(hwmon, tools/power_probe.py), the mels of every variant compared with the first one.  Logs: profiles/r06_*_ab.log.
usage: python tools/loop_ab_probe.py [seconds per variant (default 6)] [only-extra] [env:<name>:K=V,K=V ...]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from power_probe import Sampler, hwmon_files  # noqa: E402
from set_amd.synthetic import synthetic_inputs  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
if os.environ.get("AB_B"):  # another batch size than the metric's (tile-form comparisons at part-filled shapes)
    bench.B_PER_GPU = int(os.environ["AB_B"])
model = bench.build_model(dev, bench.DIFF_STEPS)
inp = {k: v.to(dev) for k, v in synthetic_inputs(bench.B_PER_GPU, bench.T, bench.T_TXT, seed=1234).items()}


def step(seed, spans=False):
    return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"], inp["uv"], infer=True,
                 seed=seed, want_layer_spans=spans)


files = hwmon_files()
variants = [("direct_form", {"SET_AMD_X3_WINO": "0"}), ("x3_winograd_default", {})]
for a in sys.argv[2:]:
    if a.startswith("env:"):  # env:NAME:K=V,K=V -- any other variant of the same loop
        _, name, kv = a.split(":", 2)
        variants.append((name, dict(x.split("=", 1) for x in kv.split(","))))
    elif a == "only-extra":
        variants = variants[:1] + variants[2:]
# (round-6 history: SET_AMD_LOOP_LAUNCH selected the whole-loop kernel of commit 9e25393; the first variant's name was per_step_launches then)
mels = {}
for name, env in variants:
    for k in ("SET_AMD_LOOP_LAUNCH", "SET_AMD_STACK_GRID", "SET_AMD_X3_WINO"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for w in range(2):
        mels[name] = step(7)["mel_out"].clone()
    torch.cuda.synchronize()
    smp = Sampler(files)
    smp.start()
    t0, n, loop_ms = time.perf_counter(), 0, []
    while time.perf_counter() - t0 < secs:
        ret = step(100 + n, spans=True)
        loop_ms.append(ret["loop_ms"])
        n += 1
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    smp.stop_flag = True
    smp.join()
    rows = smp.rows[len(smp.rows) // 4:]  # drop the ramp
    pw = [r.get("power1_average", r.get("power1_input", 0.0)) / 1e6 for r in rows]
    fq = [r.get("freq1_input", 0.0) / 1e6 for r in rows]
    out = {"variant": name, "loops": n, "ms_per_100_steps_wall": 1e3 * wall / n, "loop_ms_events_mean": sum(loop_ms) / len(loop_ms),
           "frames_per_s_wall": bench.B_PER_GPU * bench.T * n / wall, "power_w_mean": sum(pw) / max(1, len(pw)), "power_w_max": max(pw or [0]),
           "sclk_mhz_mean": sum(fq) / max(1, len(fq)), "span_ms_mean": sum(ret["layer_span_ms"]) / len(ret["layer_span_ms"])}
    print(json.dumps(out), flush=True)
base = mels["direct_form"]
import hashlib  # (mels of runs under different library builds -- SET_AMD_LIB -- are compared through these)
for name, m in mels.items():
    print("%-28s sha256 of the mel: %s" % (name, hashlib.sha256(m.cpu().numpy().tobytes()).hexdigest()[:16]))
for name, m in mels.items():
    print("%-28s bit-identical to the per-step launches: %s  (max |d| %.3e)" % (name, bool(torch.equal(m, base)), float((m - base).abs().max())))
