"""Task-level operator surface of the hot path.

`DIFF_DECODERS` and `SpeechDenoiserTask` mirror tasks/speech_editing/spec_denoiser.py:13-62 (registry,
`build_tts_model`, `run_model` incl. the `mel_out*mask + target*(1-mask)` paste).  Scope of this build
(SURVEY.md section 8): the inference hot path.  The trainer / dataset / loss code of the reference's task
base classes (utils/commons/trainer.py, tasks/tts/speech_base.py) is NOT part of this path; `start()` only
dispatches inference, and training entry points raise NotImplementedError.
"""
import importlib
import json
import os

import torch

from . import ops
from .diffnet import DiffNet
from .hparams import hparams, set_hparams
from .spec_denoiser import GaussianDiffusion
from .vocoder_infer import get_vocoder_cls

DIFF_DECODERS = {
    "wavenet": lambda hp: DiffNet(hp["audio_num_mel_bins"], hp),
}

# dotted task paths of the reference's yaml files -> classes of this package
TASK_ALIASES = {
    "tasks.speech_editing.spec_denoiser.SpeechDenoiserTask": "set_amd.tasks.SpeechDenoiserTask",
}


class SpeechDenoiserTask:
    def __init__(self, build_vocoder=True):
        # phone set: <binary_data_dir>/phone_set.json (tasks/tts/speech_base.py:40-41); only its length is used
        ph_path = os.path.join(hparams.get("binary_data_dir", ""), "phone_set.json")
        if os.path.exists(ph_path):
            with open(ph_path) as f:
                self.token_encoder = json.load(f)
        else:
            self.token_encoder = list(range(int(hparams.get("dict_size", 80))))
        self.vocoder = None
        if build_vocoder and os.path.exists(os.path.join(hparams.get("vocoder_ckpt", ""), "config.yaml")):
            self.vocoder = get_vocoder_cls(hparams["vocoder"])()
        self.model = None

    def build_tts_model(self):
        self.model = GaussianDiffusion(
            phone_encoder=self.token_encoder, out_dims=hparams["audio_num_mel_bins"],
            denoise_fn=DIFF_DECODERS[hparams["diff_decoder_type"]](hparams),
            timesteps=hparams["timesteps"], time_scale=hparams["timescale"], loss_type=hparams["diff_loss_type"],
            spec_min=hparams["spec_min"], spec_max=hparams["spec_max"], hp=hparams)
        return self.model

    def build_model(self):
        self.build_tts_model()
        return self.model

    @torch.no_grad()
    def run_model(self, sample, infer=False, **kwargs):
        """tasks/speech_editing/spec_denoiser.py:39-62 (inference branch)."""
        if not infer:
            raise NotImplementedError("training losses / backward are outside the round-1 hot path (SURVEY.md 8f)")
        target = sample["mels"]
        tmask = sample["time_mel_masks"][:, :, None]
        spk = sample.get("spk_embed") if not hparams["use_spk_id"] else sample.get("spk_ids")
        output = self.model(sample["txt_tokens"], tmask, mel2ph=sample["mel2ph"], spk_embed=spk, ref_mels=target,
                            f0=sample["f0"], uv=sample["uv"], energy=None, infer=True, **kwargs)
        B, T, M = target.shape
        # mel_out*mask + target*(1-mask)   (spec_denoiser.py:53)
        output["mel_out"] = ops.blend_mask(target.contiguous(), output["mel_out"], tmask.reshape(B, T).contiguous(), M)
        return output

    @classmethod
    def start(cls):
        if not hparams.get("infer"):
            raise NotImplementedError("SpeechDenoiserTask.start(): training is outside the round-1 hot path")
        raise NotImplementedError("dataset-driven --infer needs the IndexedDataset reader (SURVEY.md 8f rank 3)")


def run_task():
    """tasks/run.py:9-14."""
    path = TASK_ALIASES.get(hparams["task_cls"], hparams["task_cls"])
    pkg, cls_name = path.rsplit(".", 1)
    getattr(importlib.import_module(pkg), cls_name).start()


if __name__ == "__main__":
    set_hparams()
    run_task()
