"""Vocoder plug-in registry + HifiGAN wrapper (tasks/tts/vocoder_infer/base_vocoder.py:6-29,
tasks/tts/vocoder_infer/hifigan.py:11-31).  `spec2wav(mel[T,80]) -> np.float32[T*hop]`."""
import numpy as np
import torch

from . import ops
from .ckpt_utils import load_ckpt
from .hifigan import HifiGanGenerator
from .hparams import hparams, set_hparams

REGISTERED_VOCODERS = {}


def register_vocoder(name):
    def _f(cls):
        REGISTERED_VOCODERS[name] = cls
        return cls
    return _f


def get_vocoder_cls(vocoder_name):
    return REGISTERED_VOCODERS.get(vocoder_name)


class BaseVocoder:
    def spec2wav(self, mel):
        raise NotImplementedError


@register_vocoder("HifiGAN")
class HifiGAN(BaseVocoder):
    def __init__(self, device=None):
        base_dir = hparams["vocoder_ckpt"]
        self.config = config = set_hparams("%s/config.yaml" % base_dir, global_hparams=False, print_hparams=False)
        if not torch.cuda.is_available():
            raise RuntimeError("HifiGAN (set_amd) needs an MI355X: there is no CPU fallback for this path")
        self.device = torch.device(device or "cuda")
        self.model = HifiGanGenerator(config)
        load_ckpt(self.model, base_dir, "model_gen")
        self.model.to(self.device)
        self.model.eval()

    def spec2wav(self, mel, **kwargs):
        with torch.no_grad():
            c = torch.as_tensor(np.asarray(mel) if not isinstance(mel, torch.Tensor) else mel,
                                dtype=torch.float32).to(self.device)
            c = ops.btc_to_bct(c.unsqueeze(0).contiguous())  # [1,T,80] -> [1,80,T]
            y = self.model(c).view(-1)
        return y.cpu().numpy()
