"""Caller side of inference: the edit bookkeeping in front of the hot path.

`SpecDenoiserInfer` mirrors inference/tts/spec_denoiser.py:31-149 and inference/tts/base_tts_infer.py:36-47 of the
reference for everything from the batched sample onwards: known head / tail durations laid over the edited phoneme
sequence, duration prediction for the new words, splice of alignment / reference mel / f0 / uv / time mask, masked
diffusion inference with predicted pitch, paste, vocode.  Text normalisation, G2P, MFA alignment, pitch extraction and
the speaker encoder sit before that boundary and are not part of this build: `input_to_batch` takes an item that
already carries `ph_token`, `mel2ph`, `f0`, `uv`, `spk_embed` ... (the keys `preprocess_input` produces upstream).

The integer bookkeeping runs on the host on [1,T] index vectors, exactly like upstream (it drives tensor sizes); every
float tensor lives on the GPU and all arithmetic goes through libset_amd.so.  Errors surface as Python exceptions with
the upstream meaning (e.g. no frame predicted for the new words -> RuntimeError, where upstream fails in `.max()`).
"""
import re

import numpy as np
import torch

from . import ops
from .ckpt_utils import load_ckpt
from .hifigan import HifiGanGenerator
from .hparams import set_hparams

PUNCS = "!,.?;:"


def is_sil_phoneme(p):
    """utils/text/text_encoder.py:262-263"""
    return p == "" or not p[0].isalpha()


def parse_region_list_from_str(region_str):
    """inference/tts/infer_utils.py:47-53: "[3,4][7,7]" -> [[3, 4], [7, 7]], sorted by start."""
    found = re.findall(r"\[([1-9]\d*),([1-9]\d*)]", region_str)
    return sorted([[int(a), int(b)] for a, b in found], key=lambda r: r[0])


def get_words_region_from_origintxt_region(words, region_list):
    """inference/tts/infer_utils.py:29-44: regions counted in real words -> 1-based positions in the word list that
    still holds the boundary tokens '|', '<BOS>', '<pad>'."""
    assert len(region_list) >= 1, "length of region_list is %d" % len(region_list)
    out = [[0, 0] for _ in region_list]
    word_id, rid = 0, 0
    for i, w in enumerate(words):
        if is_sil_phoneme(w) and w in ("|", "<BOS>", "<pad>"):
            continue
        word_id += 1
        if word_id == region_list[rid][0]:
            out[rid][0] = i + 1
        if word_id == region_list[rid][1]:
            out[rid][1] = i + 1
            rid += 1
        if rid == len(region_list):
            break
    return out


def _np_i64(x):
    return (x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)).astype(np.int64)


def plan_masked_dur(ph2word, edited_ph2word, dur, words_region):
    """inference/tts/spec_denoiser.py:86-89: durations of the untouched head / tail phonemes, positioned in the EDITED
    phoneme sequence (head from the front, tail from the back); 0 = to be predicted.  int64 [1,T_txt_edited]."""
    ph2word, edited_ph2word, dur = _np_i64(ph2word), _np_i64(edited_ph2word), _np_i64(dur)
    w0, w1 = words_region
    out = np.zeros_like(edited_ph2word)
    n_head = int((ph2word < w0).sum())
    out[:, :n_head] = dur[:, :n_head]
    if int(ph2word.max()) > w1:
        n_tail = int((ph2word > w1).sum())
        out[:, -n_tail:] = dur[:, -n_tail:]
    return out


def plan_splice(mel2ph, mel2word, edited_ph2word, pred_mel2ph, words_region, edited_words_region):
    """inference/tts/spec_denoiser.py:97-111,113-116 (integer part): where the new words' frames go.
    Returns dict(mel2ph int64 [1,T_new], head_idx, tail_idx, after (bool [T]: original frames behind the region),
    in_region (bool [T])).  Tail phoneme indices are re-based to max(new) + 2 as upstream (:110)."""
    mel2ph, mel2word = _np_i64(mel2ph), _np_i64(mel2word)
    e_ph2word, pred = _np_i64(edited_ph2word)[0], _np_i64(pred_mel2ph)
    w0, w1 = words_region
    c0, c1 = edited_words_region
    e_mel2word = e_ph2word[pred[0] - 1][None, :]  # p == 0 wraps to the last phoneme, as upstream's [p - 1]
    sel = (e_mel2word >= c0) & (e_mel2word <= c1)
    in_region = (mel2word >= w0) & (mel2word <= w1)
    delta = int(sel.sum()) - int(in_region.sum())
    head = int((mel2word < w0).sum())
    tail = int((mel2word <= w1).sum()) + delta
    T_new = mel2ph.shape[1] + delta
    if int(sel.sum()) == 0:
        raise RuntimeError("no frame was predicted for the edited words (upstream fails in .max() of an empty tensor)")
    if tail - head != int(sel.sum()):
        raise RuntimeError("shape mismatch: %d predicted frames for a %d-frame slot (non-contiguous word region)"
                           % (int(sel.sum()), tail - head))
    new = np.zeros((1, T_new), dtype=np.int64)
    new[:, :head] = mel2ph[:, :head]
    new[:, head:tail] = pred[sel]
    after = mel2word > w1
    if int(mel2word.max()) > w1:
        tv = mel2ph[after]
        new[:, tail:] = tv - tv.min() + pred[sel].max() + 2
    return {"mel2ph": new, "head_idx": head, "tail_idx": tail, "after": after[0], "in_region": in_region[0],
            "length_edited": delta}


def _splice_frames(x, head, tail, after_idx, T_new):
    """out[:, :head] = x[:, :head]; out[:, tail:] = x[:, after]; zeros between (x: [1,T,...] on the GPU)."""
    out = torch.zeros((1, T_new) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
    out[:, :head] = x[:, :head]
    if after_idx.numel():
        out[:, tail:] = x[:, after_idx]
    return out


class SpecDenoiserInfer:
    """inference/tts/spec_denoiser.py:31-149.  `model` / `vocoder` may be passed in (tests, embedding callers);
    otherwise they are built from hparams and loaded from `work_dir` / `vocoder_ckpt` like upstream."""

    def __init__(self, hparams, device=None, model=None, vocoder=None):
        if not torch.cuda.is_available():
            raise RuntimeError("SpecDenoiserInfer (set_amd) needs an MI355X: there is no CPU fallback for this path")
        self.hparams = hparams
        self.device = torch.device(device or "cuda")
        self.model = (model if model is not None else self.build_model()).to(self.device).eval()
        self.vocoder = (vocoder if vocoder is not None else self.build_vocoder()).to(self.device).eval()

    def build_model(self):
        from .tasks import DIFF_DECODERS
        from .spec_denoiser import GaussianDiffusion
        hp = self.hparams
        n_tokens = int(hp.get("dict_size", 80))
        model = GaussianDiffusion(
            phone_encoder=list(range(n_tokens)), out_dims=hp["audio_num_mel_bins"],
            denoise_fn=DIFF_DECODERS[hp["diff_decoder_type"]](hp), timesteps=hp["timesteps"],
            time_scale=hp["timescale"], loss_type=hp["diff_loss_type"], spec_min=hp["spec_min"],
            spec_max=hp["spec_max"], hp=hp)
        load_ckpt(model, hp["work_dir"], "model")
        return model

    def build_vocoder(self):
        """inference/tts/base_tts_infer.py:36-42"""
        base_dir = self.hparams["vocoder_ckpt"]
        config = set_hparams("%s/config.yaml" % base_dir, global_hparams=False, print_hparams=False)
        vocoder = HifiGanGenerator(config)
        load_ckpt(vocoder, base_dir, "model_gen")
        return vocoder

    def run_vocoder(self, c):
        """inference/tts/base_tts_infer.py:44-47: c [B,T,80] -> wav [B, T*hop]"""
        return self.vocoder(ops.btc_to_bct(c.contiguous()))[:, 0]

    def input_to_batch(self, item):
        """inference/tts/spec_denoiser.py:198-248 without the speaker encoder (`spk_embed` must be in the item)."""
        dev = self.device
        if "spk_embed" not in item:
            raise KeyError("spk_embed: the speaker encoder (resemblyzer) is outside this build; pass the embedding")

        def L(k):
            return torch.as_tensor(np.asarray(item[k]), dtype=torch.int64)[None, :].to(dev)

        def Fl(k):
            return torch.as_tensor(np.asarray(item[k]), dtype=torch.float32)[None, :].to(dev)

        batch = {"item_name": [item.get("item_name", "<ITEM_NAME>")], "text": [item.get("text", "")],
                 "ph": [item.get("ph", "")], "ph2word": L("ph2word"), "edited_ph2word": L("edited_ph2word"),
                 "mel2ph": L("mel2ph"), "mel2word": L("mel2word"), "dur": L("dur"), "txt_tokens": L("ph_token"),
                 "edited_txt_tokens": L("edited_ph_token"), "words_region": item["words_region"],
                 "edited_words_region": item["edited_words_region"], "mel": Fl("mel"), "spk_embed": Fl("spk_embed"),
                 "f0": Fl("f0"), "uv": Fl("uv")}
        batch["txt_lengths"] = torch.tensor([batch["txt_tokens"].shape[1]], dtype=torch.int64, device=dev)
        if "wav" in item:
            batch["wav"] = Fl("wav")
        return batch

    @torch.no_grad()
    def forward_model(self, inp, noises=None, seed=0, return_aux=False):
        """inference/tts/spec_denoiser.py:63-149.  `inp`: an item (see input_to_batch) or an already batched sample.
        `noises` (optional): explicit [steps+1,1,1,80,T_new] tensor or a callable T_new -> tensor (parity runs);
        default is the on-device Philox stream seeded with `seed`.  Returns the upstream 6-tuple of numpy arrays
        (wav_out, wav_gt, mel_out, mel_gt, masked_mel_out, masked_mel_gt); `return_aux` appends the integer plan."""
        sample = inp if "edited_txt_tokens" in inp else self.input_to_batch(inp)
        dev = self.device
        txt = sample["edited_txt_tokens"].to(dev)
        mel, f0, uv = sample["mel"].to(dev), sample["f0"].to(dev), sample["uv"].to(dev)
        spk = sample["spk_embed"].to(dev)
        w_reg, e_reg = sample["words_region"][0], sample["edited_words_region"][0]
        # durations: known head / tail, predicted for the new words
        masked_dur = plan_masked_dur(sample["ph2word"], sample["edited_ph2word"], sample["dur"], w_reg)
        dur, pred_mel2ph = self.model.fs.predict_alignment(txt, spk, torch.from_numpy(masked_dur).to(dev))
        plan = plan_splice(sample["mel2ph"], sample["mel2word"], sample["edited_ph2word"], pred_mel2ph, w_reg, e_reg)
        head, tail = plan["head_idx"], plan["tail_idx"]
        T_new = plan["mel2ph"].shape[1]
        after_idx = torch.from_numpy(np.nonzero(plan["after"])[0]).to(dev)
        edited_mel2ph = torch.from_numpy(plan["mel2ph"]).to(dev)
        ref_mels = _splice_frames(mel, head, tail, after_idx, T_new)
        edited_f0 = _splice_frames(f0, head, tail, after_idx, T_new)
        edited_uv = _splice_frames(uv, head, tail, after_idx, T_new)
        time_mel_masks = torch.zeros(1, T_new, 1, dtype=torch.float32, device=dev)
        time_mel_masks[:, head:tail] = 1.0
        extra = {}
        if noises is not None:
            extra["noises"] = (noises(T_new) if callable(noises) else noises).to(dev)
        else:
            extra["seed"] = int(seed)
        output = self.model(txt, time_mel_masks=time_mel_masks, mel2ph=edited_mel2ph, spk_embed=spk,
                            ref_mels=ref_mels, f0=edited_f0, uv=edited_uv, energy=None, infer=True,
                            use_pred_pitch=True, **extra)
        mel_out = ops.blend_mask(ref_mels, output["mel_out"].contiguous(), time_mel_masks.reshape(1, T_new).contiguous(),
                                 mel.shape[2])
        wav_out = self.run_vocoder(mel_out)
        wav_gt = self.run_vocoder(mel)
        region = torch.from_numpy(plan["in_region"].astype(np.float32)).to(dev)[None, :]
        masked_mel_gt = ops.blend_mask(torch.zeros_like(mel), mel.contiguous(), region.contiguous(), mel.shape[2])
        res = tuple(t[0].cpu().numpy() for t in (wav_out, wav_gt, mel_out, mel, ref_mels, masked_mel_gt))
        if return_aux:
            aux = {"masked_dur": masked_dur, "dur_pred": dur.cpu().numpy(), "pred_mel2ph": pred_mel2ph.cpu().numpy(),
                   "edited_mel2ph": plan["mel2ph"], "edited_f0": edited_f0.cpu().numpy(),
                   "edited_uv": edited_uv.cpu().numpy(), "time_mel_masks": time_mel_masks.cpu().numpy(),
                   "head_idx": head, "tail_idx": tail, "mel2ph_out": output["mel2ph"].cpu().numpy()}
            if "pitch" in output:  # absent with use_pitch_embed false (egs/spec_denoiser_libritts.yaml)
                aux["pitch"] = output["pitch"].cpu().numpy()
            return res + (aux,)
        return res
