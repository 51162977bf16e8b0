"""FastSpeech-style conditioner of FluentSpeech on the HIP kernels (conv encoder, duration / pitch
predictors, alignment gather) + MelEncoder.

Module tree and parameter names equal the reference's (modules/speech_editing/spec_denoiser/fs.py:49-81,
modules/commons/conv.py:24-139, modules/commons/nar_tts_modules.py:8-100,
modules/speech_editing/commons/mel_encoder.py:3-19) so `state_dict()` keys match and reference
checkpoints load with strict=True.  Only the configuration the spec_denoiser yaml selects is implemented
(encoder_type/decoder_type 'conv', enc_dec_norm 'ln', use_spk_embed, use_pitch_embed, pitch_type 'frame',
use_uv); anything else raises NotImplementedError.  Internally every activation is [B, C, T].
"""
import math

import torch
from torch import nn

from . import autograd_ops, ops


def _backend():
    """Kernels with (training) or without (inference) an autograd tape; same function names either way."""
    return autograd_ops if torch.is_grad_enabled() else ops


def _embedding(n, dim, padding_idx=None):
    m = nn.Embedding(n, dim, padding_idx=padding_idx)  # layers.py:45-50
    nn.init.normal_(m.weight, mean=0, std=dim ** -0.5)
    if padding_idx is not None:
        nn.init.constant_(m.weight[padding_idx], 0)
    return m


def _cw(conv_or_linear):
    """ConvWeight over an nn.Conv1d / nn.Linear parameter container."""
    w = conv_or_linear.weight
    if w.dim() == 3:
        cout, cin, k = w.shape
    else:
        (cout, cin), k = w.shape, 1
    return ops.ConvWeight((conv_or_linear, "weight"), cout, cin, k)


class ResidualBlock(nn.Module):
    """n x [LN(ch) -> conv k (C->2C) -> *k^-0.5 -> GELU -> conv 1x1 (2C->C)] + residual, x nonpadding
    (modules/commons/conv.py:24-65)."""

    def __init__(self, channels, kernel_size, dilation, n=2, c_multiple=2, ln_eps=1e-5):
        super().__init__()
        self.kernel_size, self.dilation, self.ln_eps = kernel_size, dilation, ln_eps
        self.blocks = nn.ModuleList([
            nn.Sequential(
                nn.LayerNorm(channels, eps=ln_eps),
                nn.Conv1d(channels, c_multiple * channels, kernel_size, dilation=dilation,
                          padding=(dilation * (kernel_size - 1)) // 2),
                nn.Identity(), nn.Identity(),  # [2] = scale lambda, [3] = GELU in the reference (no params)
                nn.Conv1d(c_multiple * channels, channels, 1, dilation=dilation),
            ) for _ in range(n)])
        self._cw = [(_cw(b[1]), _cw(b[4])) for b in self.blocks]

    def run(self, x):
        F = _backend()
        k, d = self.kernel_size, self.dilation
        nonpad = F.abs_sum_mask(x.detach())  # conv.py:58
        for b, (w1, w2) in zip(self.blocks, self._cw):
            # LN -> conv k * k^-0.5 -> GELU -> conv 1x1, + x, * nonpadding: one tape node in training (autograd_ops._PreLnFfnFn)
            x = F.preln_ffn(x, (b[0].weight, b[0].bias), w1, b[1].bias, w2, b[4].bias, dil=d, pad=(d * (k - 1)) // 2, alpha=k ** -0.5,
                            act="gelu", mask=nonpad, eps=self.ln_eps)
        return x


class ConvBlocks(nn.Module):
    """modules/commons/conv.py:68-116 (norm 'ln', is_BTC handled by the caller: this works on [B,C,T])."""

    def __init__(self, hidden_size, out_dims, dilations, kernel_size, layers_in_block=2, c_multiple=2,
                 ln_eps=1e-5, post_net_kernel=3, norm_type="ln"):
        super().__init__()
        if norm_type != "ln":
            raise NotImplementedError("enc_dec_norm=%r (spec_denoiser.yaml uses 'ln')" % norm_type)
        self.res_blocks = nn.Sequential(*[
            ResidualBlock(hidden_size, kernel_size, d, n=layers_in_block, c_multiple=c_multiple, ln_eps=ln_eps)
            for d in dilations])
        self.last_norm = nn.LayerNorm(hidden_size, eps=ln_eps)
        self.post_net1 = nn.Conv1d(hidden_size, out_dims, kernel_size=post_net_kernel, padding=post_net_kernel // 2)
        self.post_net_kernel = post_net_kernel
        self.ln_eps = ln_eps
        for m in self.modules():  # init_weights_func, conv.py:18-21
            if isinstance(m, nn.Conv1d):
                nn.init.xavier_uniform_(m.weight)
        self._w_post = _cw(self.post_net1)

    def run(self, x):
        F = _backend()
        nonpad = F.abs_sum_mask(x.detach())  # conv.py:108
        for rb in self.res_blocks:
            x = rb.run(x)
        x = F.add_chan_mask(x, None, nonpad)
        x = F.layernorm_ch(x, self.last_norm.weight, self.last_norm.bias, mask=nonpad, eps=self.ln_eps)
        return F.conv1d(x, self._w_post, self.post_net1.bias, pad=self.post_net_kernel // 2, mask=nonpad)


class TextConvEncoder(ConvBlocks):
    """modules/commons/conv.py:119-139."""

    def __init__(self, dict_size, hidden_size, out_dims, dilations, kernel_size, layers_in_block=2,
                 post_net_kernel=3, norm_type="ln"):
        super().__init__(hidden_size, out_dims, dilations, kernel_size, layers_in_block=layers_in_block,
                         post_net_kernel=post_net_kernel, norm_type=norm_type)
        self.embed_tokens = _embedding(dict_size, hidden_size, 0)
        self.embed_scale = math.sqrt(hidden_size)

    def run_tokens(self, txt_tokens):
        x = _backend().embedding_bct(txt_tokens, self.embed_tokens.weight, scale=self.embed_scale, padding_idx=0)
        return self.run(x)


class _PredictorStack(nn.Module):
    def __init__(self, idim, n_layers, n_chans, kernel_size, dropout_rate=0.0):
        super().__init__()
        self.kernel_size = kernel_size
        self.dropout_rate = dropout_rate
        self._drop_site = getattr(type(self), "_site", 0)  # distinguishes the Philox streams of the two predictors
        self.conv = nn.ModuleList()
        for idx in range(n_layers):
            cin = idim if idx == 0 else n_chans
            self.conv.append(nn.Sequential(
                nn.Conv1d(cin, n_chans, kernel_size, stride=1, padding=kernel_size // 2),
                nn.Identity(),  # ReLU
                nn.LayerNorm(n_chans),
                nn.Identity(),  # Dropout (eval)
            ))
        self._cws = [_cw(f[0]) for f in self.conv]

    def run_stack(self, x, nonpad=None, seed=0):
        """conv -> ReLU -> LN -> Dropout (-> * nonpadding); dropout only in train() mode with a tape
        (nar_tts_modules.py:16-21,82-87); mask and inverted-dropout scaling commute, so the mask stays in the LN."""
        F = _backend()
        k = self.kernel_size
        drop = self.dropout_rate if (self.training and torch.is_grad_enabled()) else 0.0
        for li, (f, w) in enumerate(zip(self.conv, self._cws)):
            x = F.conv1d(x, w, f[0].bias, pad=k // 2, act="relu")
            x = F.layernorm_ch(x, f[2].weight, f[2].bias, mask=nonpad)
            if drop > 0:
                # the counter offset is a function of (site, layer) only: with the per-update seed that makes every
                # mask a function of (seed, update, site, layer) -- a resumed run draws the masks the uninterrupted one did
                x = F.dropout(x, drop, seed, (self._drop_site * 16 + li + 1) * (1 << 28))
        return x


class DurationPredictor(_PredictorStack):
    """modules/commons/nar_tts_modules.py:8-34 (inference / eval: dropout is identity)."""
    _site = 1

    def __init__(self, idim, n_layers=2, n_chans=384, kernel_size=3, dropout_rate=0.1):
        super().__init__(idim, n_layers, n_chans, kernel_size, dropout_rate)
        self.linear = nn.Sequential(nn.Linear(n_chans, 1), nn.Identity())  # [1] = Softplus
        self._w_lin = _cw(self.linear[0])

    def run(self, x, src_nonpad, seed=0):
        x = self.run_stack(x, src_nonpad, seed)
        d = _backend().conv1d(x, self._w_lin, self.linear[0].bias, act="softplus", mask=src_nonpad)
        return d.reshape(d.shape[0], d.shape[2])


class PitchPredictor(_PredictorStack):
    """modules/commons/nar_tts_modules.py:75-100."""
    _site = 2

    def __init__(self, idim, n_layers=5, n_chans=384, odim=2, kernel_size=5, dropout_rate=0.1):
        super().__init__(idim, n_layers, n_chans, kernel_size, dropout_rate)
        self.linear = nn.Linear(n_chans, odim)
        self._w_lin = _cw(self.linear)

    def run(self, x, seed=0):
        x = self.run_stack(x, None, seed)
        return _backend().conv1d(x, self._w_lin, self.linear.bias)  # [B, odim, T]


class LengthRegulator(nn.Module):
    """modules/commons/nar_tts_modules.py:37-72 (alpha = 1; padding given by txt_tokens == 0)."""

    def forward(self, dur, txt_tokens):
        return ops.length_regulate(dur.contiguous(), txt_tokens)


class MelEncoder(nn.Module):
    """modules/speech_editing/commons/mel_encoder.py:3-19."""

    def __init__(self, input_dim=80, hidden_size=192):
        super().__init__()
        self.encoder = nn.Sequential(nn.Linear(input_dim, hidden_size), nn.Identity(),
                                     nn.Linear(hidden_size, hidden_size), nn.Identity())
        self.fc_out = nn.Linear(hidden_size, hidden_size)
        self._w0, self._w2, self._w_out = _cw(self.encoder[0]), _cw(self.encoder[2]), _cw(self.fc_out)

    def run(self, x_bct, res=None, mask=None):
        """x [B,80,T] -> fc_out(...) (+ res) (* mask), all fused in the last conv's epilogue."""
        F = _backend()
        h = F.conv1d(x_bct, self._w0, self.encoder[0].bias, act="relu")
        h = F.conv1d(h, self._w2, self.encoder[2].bias, act="relu")
        return F.conv1d(h, self._w_out, self.fc_out.bias, res=res, mask=mask)


class FastSpeech(nn.Module):
    """modules/speech_editing/spec_denoiser/fs.py:49-189 with skip_decoder=True (the only way
    GaussianDiffusion calls it, spec_denoiser.py:159-161).  `decoder` / `mel_out` exist as parameters
    (checkpoint compatibility) but are never run, exactly as in the reference."""
    masked_predictor = True    # dur_embed + masked ground-truth duration / pitch fed to the predictors
    pitch_dropout = 0.2        # fs.py:77

    def __init__(self, dict_size, hp, out_dims=None):
        super().__init__()
        self.hparams = dict(hp)
        H = self.hidden_size = hp["hidden_size"]
        if hp["encoder_type"] != "conv" or hp["decoder_type"] != "conv":
            raise NotImplementedError("only encoder_type/decoder_type 'conv' (spec_denoiser.yaml)")
        self.encoder = TextConvEncoder(dict_size, H, H, hp["enc_dilations"], hp["enc_kernel_size"],
                                       layers_in_block=hp["layers_in_block"], norm_type=hp["enc_dec_norm"],
                                       post_net_kernel=hp.get("enc_post_net_kernel", 3))
        self.decoder = ConvBlocks(H, H, hp["dec_dilations"], hp["dec_kernel_size"],
                                  layers_in_block=hp["layers_in_block"], norm_type=hp["enc_dec_norm"],
                                  post_net_kernel=hp.get("dec_post_net_kernel", 3))
        self.out_dims = hp["audio_num_mel_bins"] if out_dims is None else out_dims
        self.mel_out = nn.Linear(H, self.out_dims, bias=True)
        if hp["use_spk_id"]:
            raise NotImplementedError("use_spk_id (spec_denoiser.yaml uses use_spk_embed)")
        if not hp["use_spk_embed"]:
            raise NotImplementedError("spec_denoiser.yaml / spec_denoiser_libritts.yaml set use_spk_embed")
        if hp.get("dec_inp_add_noise"):
            raise NotImplementedError("dec_inp_add_noise")
        self.spk_embed_proj = nn.Linear(256, H, bias=True)
        ph = hp["predictor_hidden"] if hp["predictor_hidden"] > 0 else H
        if self.masked_predictor:
            self.dur_embed = _embedding(2000, H, 0)
        self.dur_predictor = DurationPredictor(H, n_chans=ph, n_layers=hp["dur_predictor_layers"],
                                               dropout_rate=hp["predictor_dropout"],
                                               kernel_size=hp["dur_predictor_kernel"])
        self.length_regulator = LengthRegulator()
        if hp["use_pitch_embed"]:  # fs.py:73-78; egs/spec_denoiser_libritts.yaml:169 turns it off
            self.pitch_embed = _embedding(300, H, 0)
            self.pitch_predictor = PitchPredictor(H, n_chans=ph, n_layers=5, dropout_rate=self.pitch_dropout, odim=2,
                                                  kernel_size=hp["predictor_kernel"])
        self._w_spk = _cw(self.spk_embed_proj)

    def predict_alignment(self, txt_tokens, spk_embed, masked_dur):
        """The duration half of the model on its own, as the inference caller uses it (inference/tts/
        spec_denoiser.py:81-96): `forward_dur(dur_inp, ..., masked_dur=masked_dur, use_pred_mel2ph=True)`
        (fs.py:123-151) with `dur_inp = (encoder(txt) + style) * nonpadding`.  `masked_dur` int64 [B,T_txt] holds
        the known durations (0 = to be predicted).  Returns (dur fp32 [B,T_txt], mel2ph int64 [B,T_pred])."""
        F = _backend()
        B = txt_tokens.shape[0]
        enc = self.encoder.run_tokens(txt_tokens)
        src_nonpad = F.index_mask(txt_tokens)
        style = F.conv1d(spk_embed.reshape(B, 256, 1).contiguous(), self._w_spk, self.spk_embed_proj.bias)
        dur_inp = F.add_chan_mask(enc, style.reshape(B, self.hidden_size), src_nonpad)
        dur_inp = F.embedding_bct(masked_dur.contiguous(), self.dur_embed.weight, out=dur_inp, accumulate=True,
                                  padding_idx=0)
        dur = self.dur_predictor.run(dur_inp, src_nonpad, 0)
        return dur, self.length_regulator(dur.detach(), txt_tokens)

    def forward(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, f0, uv, spk_id=None, skip_decoder=True,
                infer=False, use_pred_mel2ph=False, use_pred_pitch=False, **kwargs):
        """Returns ret with `decoder_inp_bct` [B,H,T] (internal layout) next to the reference's keys
        (`dur`, `mel2ph`, `pitch_pred`, `f0_denorm`, `f0_denorm_pred`); the caller finishes `decoder_inp`."""
        if not skip_decoder:
            raise NotImplementedError("FluentSpeech always calls fs(..., skip_decoder=True)")
        hp = self.hparams
        if hp["use_pitch_embed"] and not (hp.get("pitch_type") == "frame" and hp["use_uv"]):
            raise NotImplementedError("pitch_type 'frame' + use_uv only")
        F = _backend()
        ret = {}
        B, T_txt = txt_tokens.shape
        seed = int(kwargs.get("dropout_seed", 0))
        pg = hp["predictor_grad"]
        tmask = time_mel_masks.reshape(B, -1).contiguous()  # [B,T]
        enc = self.encoder.run_tokens(txt_tokens)  # [B,H,T_txt]
        src_nonpad = F.index_mask(txt_tokens)
        style = F.conv1d(spk_embed.reshape(B, 256, 1).contiguous(), self._w_spk, self.spk_embed_proj.bias)
        style = style.reshape(B, self.hidden_size)  # fs.py:114-121
        enc_d, enc = F.fanout(enc, 2)               # duration branch + alignment gather
        if hp["use_pitch_embed"]:
            style_d, style_p, style = F.fanout(style, 3)  # duration / pitch / decoder_inp
        else:
            style_d, style = F.fanout(style, 2)
        # ---- duration (fs.py:123-151)
        dur_inp = F.add_chan_mask(enc_d, style_d, src_nonpad)
        mdur = F.masked_dur(mel2ph, tmask, txt_tokens)
        ret["masked_dur"] = mdur
        dur_inp = F.embedding_bct(mdur, self.dur_embed.weight, out=dur_inp, accumulate=True, padding_idx=0)
        dur_inp = F.grad_scale(dur_inp, pg)  # fs.py:144-145
        ret["dur"] = dur = self.dur_predictor.run(dur_inp, src_nonpad, seed)
        if use_pred_mel2ph:
            mel2ph = self.length_regulator(dur.detach(), txt_tokens)
        fm = hp["frames_multiple"]
        if fm != 1:
            mel2ph = mel2ph[:, :mel2ph.shape[1] // fm * fm].contiguous()  # align_ops.py:15-18
        ret["mel2ph"] = mel2ph
        tgt_nonpad = F.index_mask(mel2ph)
        dec = F.expand_states(enc, mel2ph)  # align_ops.py:21-25
        ret["tgt_nonpad"] = tgt_nonpad
        if not hp["use_pitch_embed"]:  # fs.py:97-102 without the pitch block
            ret["decoder_inp_bct"] = F.add_chan_mask(dec, style, tgt_nonpad)
            return ret
        dec_p, dec = F.fanout(dec, 2)
        # ---- pitch (fs.py:153-189)
        pitch_inp = F.add_chan_mask(dec_p, style_p, tgt_nonpad)
        _, masked_pitch = F.pitch_coarse(f0, uv, tmask=tmask, mel2ph_pad=mel2ph, want_denorm=False)
        ret["masked_pitch"] = masked_pitch
        pitch_inp = F.embedding_bct(masked_pitch, self.pitch_embed.weight, out=pitch_inp, accumulate=True, padding_idx=0)
        pitch_inp = F.grad_scale(pitch_inp, pg)  # fs.py:167-169
        pp = self.pitch_predictor.run(pitch_inp, seed + 1)  # [B,2,T]
        ret["pitch_pred_bct"] = pp
        ret["pitch_pred"] = F.bct_to_btc(pp)
        pad_idx = mel2ph
        ppd = pp.detach()
        if use_pred_pitch:
            pad_idx = None  # fs.py:172
            pred_f0 = ppd[:, 0, :].contiguous()
            pred_uv = (ppd[:, 1, :] > 0).to(torch.float32).contiguous()
            res_f0 = F.blend_mask(f0, pred_f0, tmask, 1)
            res_uv = F.blend_mask(uv, pred_uv, tmask, 1)
        else:
            res_f0, res_uv = f0, uv
        f0_denorm, pitch = F.pitch_coarse(res_f0, res_uv, mel2ph_pad=pad_idx)
        ret["pitch"] = pitch
        ret["f0_denorm"] = f0_denorm
        ret["f0_denorm_pred"], _ = F.pitch_coarse(ppd[:, 0, :].contiguous(), ppd[:, 1, :].contiguous(),
                                                  mel2ph_pad=pad_idx, uv_from_logit=True, want_coarse=False)
        dec = F.embedding_bct(pitch, self.pitch_embed.weight, out=dec, accumulate=True, padding_idx=0)
        ret["decoder_inp_bct"] = F.add_chan_mask(dec, style, tgt_nonpad)
        return ret


class FastSpeechNormal(FastSpeech):
    """modules/tts/fs.py:49-175 with skip_decoder=True: the conditioner of the `wo_masked_predictor` ablation
    (egs/spec_denoiser_wo_masked_predictor.yaml -> modules/speech_editing/spec_denoiser/spec_denoiser_normal.py:11).
    Against the masked variant above: no `dur_embed`, the predictors see no masked ground truth (:121-138,140-151),
    pitch-predictor dropout 0.1 (:75), `mel2ph is None` / `f0 is None` select the predictions (:135,153-156)."""
    masked_predictor = False
    pitch_dropout = 0.1

    def predict_alignment(self, *a, **k):
        raise NotImplementedError("the edit caller (inference/tts/spec_denoiser.py) uses the masked predictor")

    def forward(self, txt_tokens, mel2ph=None, spk_embed=None, spk_id=None, f0=None, uv=None, skip_decoder=True,
                infer=False, **kwargs):
        if not skip_decoder:
            raise NotImplementedError("GaussianDiffusion always calls fs(..., skip_decoder=True)")
        hp = self.hparams
        if hp["use_pitch_embed"] and not (hp.get("pitch_type") == "frame" and hp["use_uv"]):
            raise NotImplementedError("pitch_type 'frame' + use_uv only")
        F = _backend()
        ret = {}
        B = txt_tokens.shape[0]
        seed = int(kwargs.get("dropout_seed", 0))
        pg = hp["predictor_grad"]
        enc = self.encoder.run_tokens(txt_tokens)
        src_nonpad = F.index_mask(txt_tokens)
        style = F.conv1d(spk_embed.reshape(B, 256, 1).contiguous(), self._w_spk, self.spk_embed_proj.bias)
        style = style.reshape(B, self.hidden_size)
        enc_d, enc = F.fanout(enc, 2)
        if hp["use_pitch_embed"]:
            style_d, style_p, style = F.fanout(style, 3)
        else:
            style_d, style = F.fanout(style, 2)
        # ---- duration (modules/tts/fs.py:121-138)
        dur_inp = F.grad_scale(F.add_chan_mask(enc_d, style_d, src_nonpad), pg)
        ret["dur"] = dur = self.dur_predictor.run(dur_inp, src_nonpad, seed)
        if mel2ph is None:
            mel2ph = self.length_regulator(dur.detach(), txt_tokens)
        fm = hp["frames_multiple"]
        if fm != 1:
            mel2ph = mel2ph[:, :mel2ph.shape[1] // fm * fm].contiguous()
        ret["mel2ph"] = mel2ph
        ret["tgt_nonpad"] = tgt_nonpad = F.index_mask(mel2ph)
        dec = F.expand_states(enc, mel2ph)
        if not hp["use_pitch_embed"]:
            ret["decoder_inp_bct"] = F.add_chan_mask(dec, style, tgt_nonpad)
            return ret
        # ---- pitch (modules/tts/fs.py:140-168)
        dec_p, dec = F.fanout(dec, 2)
        pitch_inp = F.grad_scale(F.add_chan_mask(dec_p, style_p, tgt_nonpad), pg)
        pp = self.pitch_predictor.run(pitch_inp, seed + 1)  # [B,2,T]
        ret["pitch_pred_bct"] = pp
        ret["pitch_pred"] = F.bct_to_btc(pp)
        ppd = pp.detach()
        if f0 is None:  # :153-156 (the coarse bins carry no gradient, so detaching changes nothing)
            f0_denorm, pitch = F.pitch_coarse(ppd[:, 0, :].contiguous(), ppd[:, 1, :].contiguous(), mel2ph_pad=mel2ph,
                                              uv_from_logit=True)
        else:
            f0_denorm, pitch = F.pitch_coarse(f0.contiguous(), None if uv is None else uv.contiguous(),
                                              mel2ph_pad=mel2ph)
        ret["pitch"] = pitch
        ret["f0_denorm"] = f0_denorm
        ret["f0_denorm_pred"], _ = F.pitch_coarse(ppd[:, 0, :].contiguous(), ppd[:, 1, :].contiguous(),
                                                  mel2ph_pad=mel2ph, uv_from_logit=True, want_coarse=False)
        dec = F.embedding_bct(pitch, self.pitch_embed.weight, out=dec, accumulate=True, padding_idx=0)
        ret["decoder_inp_bct"] = F.add_chan_mask(dec, style, tgt_nonpad)
        return ret
