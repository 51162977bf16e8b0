"""Host-side (Python) cost of enqueueing one training step: cProfile over a few steps, top functions by own / cumulative time.
  MODEL=campnet|spec_denoiser DTYPE=bf16|f32 python tools/host_profile.py"""
import cProfile, os, pstats, sys, io
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import set_amd  # noqa
from set_amd import hparams as HP, ops, parallel, tasks
from set_amd.synthetic import synthetic_inputs
from set_amd.training import FlatAdamW
dev = torch.device("cuda:0")
campnet = os.environ.get("MODEL", "campnet") == "campnet"
HP.hparams.clear()
if campnet:
    HP.hparams.update(yaml.safe_load(open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "campnet.yaml"))))
    HP.hparams.update(binary_data_dir="", vocoder_ckpt="")
    task = tasks.CampNetTask(80, 100, build_vocoder=False)
else:
    HP.hparams.update(bench.load_hparams())
    task = tasks.SpeechDenoiserTask(build_vocoder=False)
if os.environ.get("DTYPE", "bf16") == "bf16":
    ops.set_compute_dtype("bf16")
task.build_model()
task.model.to(dev).train()
opt = FlatAdamW(task.model, lr=2e-4, warmup_updates=8000)
B = 16 if campnet else 32
inp = {k: v.to(dev) for k, v in synthetic_inputs(B, 800, 100, seed=1234, pad_tail=True).items()}
sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], spk_embed=inp["spk_embed"],
              time_mel_masks=inp["time_mel_masks"].squeeze(-1).contiguous())
if not campnet:
    sample.update(mel2ph=inp["mel2ph"], f0=inp["f0"], uv=inp["uv"])
for w in range(3):
    task.training_step(sample, opt, seed=w)
torch.cuda.synchronize()
# host time of the four phases of ONE step issued into an EMPTY queue (after a synchronize: no back-pressure, so this is Python + HIP
# launch cost only), three times
import time
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    t1 = time.perf_counter()
    losses, _ = task.run_model(sample, infer=False, seed=100 + rep)
    with torch.enable_grad():
        total = sum(losses.values())
    t2 = time.perf_counter()
    total.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print("host phases (ms): zero_grad %.2f | forward %.2f | backward %.2f | optimizer %.2f | total enqueue %.2f | + drain %.2f"
          % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0), 1e3 * (t5 - t4)), flush=True)
if hasattr(task, "global_step"):
    pass
pr = cProfile.Profile()
pr.enable()
N = 5
for k in range(N):
    task.training_step(sample, opt, seed=10 + k)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[:45]))
print("(divide by %d for per-step numbers)" % N)
