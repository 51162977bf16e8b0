"""Soak of the training path's leaf stream (autograd_ops.leaf_work): two identical replicas at the benchmark shape, one with the leaf stream,
one with SET_AMD_LEAF_STREAM=0, STEPS optimisation steps each on changing batches / seeds (dropout on); after every step the losses, the
flat parameter buffer and both Adam moments must be bit-identical.  A race between the two streams (a target written by both, an operand
freed or rewritten under a leaf kernel) shows up as the first differing step.
  MODEL=spec_denoiser|campnet DTYPE=bf16|f32 STEPS=60 python tools/leaf_soak_probe.py"""
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import set_amd  # noqa: E402,F401
from set_amd import hparams as HP, ops, tasks  # noqa: E402
from set_amd.synthetic import synthetic_inputs  # noqa: E402
from set_amd.training import FlatAdamW  # noqa: E402

dev = torch.device("cuda:0")
campnet = os.environ.get("MODEL", "spec_denoiser") == "campnet"
dtype = os.environ.get("DTYPE", "bf16")
steps = int(os.environ.get("STEPS", "60"))
HP.hparams.clear()
if campnet:
    HP.hparams.update(yaml.safe_load(open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "campnet.yaml"))))
    HP.hparams.update(binary_data_dir="", vocoder_ckpt="")
else:
    HP.hparams.update(bench.load_hparams())
ops.set_compute_dtype(dtype)
reps = []
for _ in range(2):
    torch.manual_seed(1234)
    task = tasks.CampNetTask(80, 100, build_vocoder=False) if campnet else tasks.SpeechDenoiserTask(build_vocoder=False)
    task.build_model()
    if not campnet:
        torch.nn.init.normal_(task.model.denoise_fn.output_projection.weight, std=1.0 / 16.0)
    task.model.to(dev).train()
    reps.append((task, FlatAdamW(task.model, lr=2e-4, betas=(0.9, 0.98), clip_grad_norm=1.0, warmup_updates=20)))
(ta, oa), (tb, ob) = reps
assert torch.equal(oa.flat_p, ob.flat_p)
B = 16 if campnet else 32
bad, t0 = None, time.time()
for it in range(steps):
    inp = {k: v.to(dev) for k, v in synthetic_inputs(B, 800, 100, seed=1234 + it, pad_tail=True).items()}
    sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], spk_embed=inp["spk_embed"],
                  time_mel_masks=inp["time_mel_masks"].squeeze(-1).contiguous())
    if not campnet:
        sample.update(mel2ph=inp["mel2ph"], f0=inp["f0"], uv=inp["uv"])
    kw = {"seed": 900 + 7 * it}
    if not campnet:  # the diffusion step ids are drawn from torch's global generator unless given: both replicas must see the same ones
        kw["t"] = torch.randint(0, HP.hparams["timesteps"] + 1, (B,), generator=torch.Generator().manual_seed(it)).to(dev)
    os.environ["SET_AMD_LEAF_STREAM"] = "1"
    la, pa, _ = ta.training_step(sample, oa, **kw)
    os.environ["SET_AMD_LEAF_STREAM"] = "0"
    lb, pb, _ = tb.training_step(sample, ob, **kw)
    torch.cuda.synchronize()
    same = torch.equal(la, lb) and all(torch.equal(pa[k], pb[k]) for k in pa) and torch.equal(oa.flat_p, ob.flat_p) and \
        torch.equal(oa.m, ob.m) and torch.equal(oa.v, ob.v)
    if not same and bad is None:
        bad = it
        print("FIRST DIFFERENCE at step %d: loss %r vs %r, max |dp| %.3e" % (it, float(la), float(lb), float((oa.flat_p - ob.flat_p).abs().max())))
print("%s %s: %d steps at B=%d, T=800, leaf stream vs single stream: %s (%.1f s)" % (
    "campnet" if campnet else "spec_denoiser", dtype, steps, B, "bit-identical after every step" if bad is None else "DIFFER from step %d" % bad,
    time.time() - t0))
sys.exit(0 if bad is None else 1)
