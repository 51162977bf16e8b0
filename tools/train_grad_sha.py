"""Bit-level signature of the bf16 spec_denoiser training path: sha256 of the flat parameter buffer after two optimisation steps from a
fixed state, on four cases (full-size rows, ragged T, per-layer weight gradients, dilation cycle 4).  A change that only re-lays-out
intermediate tensors or re-stages operands must print the same four lines (compare a run before with a run after)."""
import hashlib, os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import set_amd  # noqa
from set_amd import hparams as H, ops, tasks
from set_amd.synthetic import synthetic_inputs
from set_amd.training import FlatAdamW

dev = torch.device("cuda:0")
CASES = [("B4_T800", 4, 800, 100, {}, {}), ("B3_T77_ragged", 3, 77, 20, {}, {}),
         ("B2_T800_per_layer_wgrad", 2, 800, 100, {}, {"SET_AMD_GROUPED_WGRAD_MB": "0"}),
         ("B2_T264_dilation_cycle_4", 2, 264, 40, {"dilation_cycle_length": 4}, {})]
for name, B, T, TT, hp, env in CASES:
    for k, v in env.items():
        os.environ[k] = v
    with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser.yaml")) as f:
        H.hparams.clear(); H.hparams.update(yaml.safe_load(f)); H.hparams.update(hp)
    torch.manual_seed(1234)
    task = tasks.SpeechDenoiserTask(build_vocoder=False)
    task.build_model()
    torch.nn.init.normal_(task.model.denoise_fn.output_projection.weight, std=1.0 / 16.0)
    task.model.to(dev).train()
    opt = FlatAdamW(task.model, lr=H.hparams["lr"], betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=1.0, warmup_updates=8000)
    inp = {k: v.to(dev) for k, v in synthetic_inputs(B, T, TT, seed=1234, pad_tail=True).items()}
    sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], mel2ph=inp["mel2ph"], f0=inp["f0"], uv=inp["uv"],
                  time_mel_masks=inp["time_mel_masks"].squeeze(-1).contiguous(), spk_embed=inp["spk_embed"])
    ops.set_compute_dtype("bf16")
    try:
        tot = None
        for k in range(2):
            tot, parts, lr = task.training_step(sample, opt, seed=k)
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype("f32")
        for k in env:
            os.environ.pop(k, None)
    h = hashlib.sha256(opt.flat_p.detach().cpu().numpy().tobytes()).hexdigest()
    print("%-28s loss %.6f  params sha256 %s" % (name, float(tot), h[:32]), flush=True)
