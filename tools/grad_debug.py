import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import base_hparams, load_golden
from oracle import weights as Wt
import set_amd
from set_amd import hparams as H, tasks
dev = torch.device("cuda:0")
g = load_golden("train_losses"); m = g["meta"]
H.hparams.clear(); H.hparams.update(base_hparams(timesteps=m["steps"]))
task = tasks.SpeechDenoiserTask(build_vocoder=False); task.build_model()
task.model.load_state_dict(Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"]), strict=False)
task.model.to(dev).eval()
inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=True)
sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], mel2ph=inp["mel2ph"], f0=inp["f0"], uv=inp["uv"],
              time_mel_masks=inp["time_mel_masks"].squeeze(-1), spk_embed=inp["spk_embed"])
sample = {k: v.to(dev) for k, v in sample.items()}
losses, out = task.run_model(sample, infer=False, t=torch.from_numpy(g["t"]).to(dev), noises=torch.from_numpy(g["eps"]).to(dev))
sum(losses.values()).backward()
norms = dict(zip(m["param_names"], g["grad_norms"]))
rows = []
for k, p in task.model.named_parameters():
    ref = norms[k]
    if ref < 0: continue
    got = float(p.grad.norm()) if p.grad is not None else float("nan")
    rows.append((abs(got - ref) / (ref + 1e-12), k, got, ref))
rows.sort(reverse=True)
for r in rows[:25]: print("%.3e  %-60s got %.4e ref %.4e" % r)
