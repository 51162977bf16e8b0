"""Which lines of the host side issue ATen (torch) kernels during one training step?  A TorchDispatchMode counts every non-view ATen op
of one step and attributes it to the innermost frame inside the package (forward lines, and the backward() of the tape nodes; ops of
torch's own autograd nodes -- AddBackward, MulBackward, AccumulateGrad -- have no package frame and are listed as '<autograd engine>').
  MODEL=campnet|spec_denoiser DTYPE=bf16|f32 python tools/aten_sites.py"""
import collections, os, sys, traceback
import torch, yaml
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import set_amd  # noqa
from set_amd import hparams as HP, ops, tasks
from set_amd.synthetic import synthetic_inputs
from set_amd.training import FlatAdamW
PKG = os.path.dirname(os.path.abspath(set_amd.__file__))
dev = torch.device("cuda:0")
campnet = os.environ.get("MODEL", "spec_denoiser") == "campnet"
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
VIEWS = ("view", "reshape", "as_strided", "transpose", "permute", "detach", "alias", "slice", "select", "unsqueeze", "squeeze", "expand",
         "_unsafe_view", "empty", "new_empty", "empty_like", "empty_strided", "t", "unbind", "split", "chunk", "narrow", "unfold", "lift_fresh",
         "_local_scalar_dense", "is_same_size", "sym_size", "stride", "storage_offset", "numel", "record_stream", "set_", "_to_copy_view")
sites = collections.Counter()


class Count(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in VIEWS:
            site = "<autograd engine>"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(PKG):
                    site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


with Count():
    task.training_step(sample, opt, seed=50)
torch.cuda.synchronize()
by_op = collections.Counter()
for (name, site), n in sites.items():
    by_op[name] += n
print("%s %s: %d non-view ATen ops in one step; by op: %s" % ("campnet" if campnet else "spec_denoiser", os.environ.get("DTYPE", "bf16"),
                                                              sum(sites.values()), dict(by_op.most_common())))
for (name, site), n in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d  %-22s %s" % (n, name, site))
