"""Secondary metric (BASELINE configs[3]): FluentSpeech 100-step batched inference + HiFi-GAN V1 vocoder, B=64, T=800,
one GPU: mel-frames/s through BOTH stages (conditioner + 100 x (DiffNet + posterior), then mel -> 204,800-sample wav)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import set_amd  # noqa
from set_amd import ops
from set_amd.hifigan import HifiGanGenerator
from set_amd.synthetic import synthetic_inputs
V1 = {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
      "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
      "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]}
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
B, T = int(os.environ.get("EB", 64)), int(os.environ.get("ET", 800))
STEPS = int(os.environ.get("ESTEPS", 100))  # 8 = the shipped egs/spec_denoiser.yaml, 100 = the benchmark override
model = bench.build_model(dev, STEPS)
torch.manual_seed(0)
voc = HifiGanGenerator(V1).to(dev).eval()
inp = {k: v.to(dev) for k, v in synthetic_inputs(B, T, 100, seed=1).items()}


def run(seed):
    ret = model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"],
                inp["uv"], infer=True, seed=seed)
    mel = ops.blend_mask(inp["ref_mels"].contiguous(), ret["mel_out"].contiguous(),
                         inp["time_mel_masks"].reshape(B, T).contiguous(), 80)   # paste (tasks/.../spec_denoiser.py:53)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    wav = voc(ops.btc_to_bct(mel))
    torch.cuda.synchronize()
    return wav, t1


run(0)
t0 = time.perf_counter()
wav, t1 = run(1)
t2 = time.perf_counter()
print(json.dumps({"metric": "diffusion + HiFi-GAN mel-frames/s (B=%d, T=%d, %d steps, fp32)" % (B, T, STEPS),
                  "value": B * T / (t2 - t0), "unit": "mel-frames/s", "diffusion_ms": 1e3 * (t1 - t0),
                  "vocoder_ms": 1e3 * (t2 - t1), "wav_shape": list(wav.shape), "finite": bool(torch.isfinite(wav).all())}))
