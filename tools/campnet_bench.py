"""Secondary metric (BASELINE configs[4]): CampNet masked-mel transformer TRAINING samples/s on synthetic 80-mel batches
(B=16 per GPU = egs/campnet.yaml max_sentences, T=800, T_txt=100, fp32): forward + coarse/fine mel losses + backward +
all-reduce + clip + AdamW.  Also prints inference (forward only) frames/s.
  python tools/campnet_bench.py               # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/campnet_bench.py"""
import json, os, sys, time
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import hparams as H, parallel, tasks
from set_amd.synthetic import synthetic_inputs
from set_amd.training import FlatAdamW

B, T, TT, STEPS = int(os.environ.get("TB", 16)), int(os.environ.get("TT", 800)), 100, int(os.environ.get("TSTEPS", 5))
rank, world, local_rank = parallel.init_from_env()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "campnet.yaml")) as f:
    H.hparams.clear(); H.hparams.update(yaml.safe_load(f))
torch.manual_seed(1234)
task = tasks.CampNetTask(80, 100)
model = task.build_tts_model().to(dev)
with torch.no_grad():
    model.mask_emb.normal_(0, 0.5)
opt = FlatAdamW(model, lr=H.hparams["lr"], betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=1.0, warmup_updates=8000)
full = synthetic_inputs(B * world, T, TT, seed=1234, pad_tail=True)
inp = {k: v.to(dev) for k, v in parallel.shard_batch(full, rank, world).items()}
sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], time_mel_masks=inp["time_mel_masks"].squeeze(-1).contiguous())
for w in range(2):
    task.training_step(sample, opt)
torch.cuda.synchronize(); parallel.barrier()
t0 = time.perf_counter()
for k in range(STEPS):
    total, parts, lr = task.training_step(sample, opt)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
parallel.barrier()
tmax = parallel.max_over_ranks(dt, device=dev if world > 1 else "cpu")
task.run_model(sample, infer=True); torch.cuda.synchronize()
t1 = time.perf_counter()
for k in range(STEPS):
    task.run_model(sample, infer=True)
torch.cuda.synchronize()
di = time.perf_counter() - t1
if rank == 0:
    print(json.dumps({"metric": "CampNet training samples/s (B=%d/GPU, T=%d, fp32)" % (B, T), "value": B * world * STEPS / tmax,
                      "unit": "samples/s", "frames_per_s": B * world * T * STEPS / tmax, "n_gpus": world, "steps": STEPS,
                      "ms_per_step": 1e3 * tmax / STEPS, "infer_ms": 1e3 * di / STEPS,
                      "infer_frames_per_s": B * T * STEPS / di, "loss": float(total),
                      "losses": {k: float(v) for k, v in parts.items()}}))
