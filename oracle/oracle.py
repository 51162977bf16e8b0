"""CPU oracle for the FluentSpeech spec_denoiser hot path + HiFi-GAN generator.

TEST INFRASTRUCTURE.  This file is the *checker*, never the product:
only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import it.  The product path (speech-editing-toolkit_amd/) never imports
anything under oracle/ and fails loudly when the HIP library is missing.

It is a functional restatement (plain torch CPU ops over a flat
{state_dict key -> tensor} weight dict) of the reference algorithm, written
from the equations in SURVEY.md section 8a; each function cites the reference
file:line it follows (paths relative to the upstream repo root).

Pinning: the reference has no tests and no golden vectors,
so this oracle is pinned against outputs of the reference itself, imported in
the development container by oracle/make_golden.py; the resulting vectors are
committed under tests/golden/ and re-checked by tests/test_oracle_golden.py.
Third-party arithmetic is torch (CPU, mkldnn); no reference test pins results
at that boundary, so beyond those fixtures parity is unpinned.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# a1/a2: noise schedule + diffusion tables
# --------------------------------------------------------------------------
def vpsde_betas(timesteps_plus_1, min_beta=0.1, max_beta=40.0):
    """modules/speech_editing/spec_denoiser/diffusion_utils.py:16-18,36-38.

    beta_t = 1 - exp(-min/T' - 0.5 (max-min) (2t-1)/T'^2), t = 1..T' (fp64)."""
    Tp = int(timesteps_plus_1)
    t = np.arange(1, Tp + 1, dtype=np.float64)
    return 1.0 - np.exp(-min_beta / Tp - 0.5 * (max_beta - min_beta) * (2.0 * t - 1.0) / (Tp ** 2))


def diffusion_tables(timesteps, betas=None):
    """spec_denoiser.py:26-69: fp64 cumprod tables -> fp32 buffers.

    NB the schedule has timesteps+1 entries (spec_denoiser.py:31) while
    num_timesteps = int(timesteps) (spec_denoiser.py:42)."""
    if betas is None:
        betas = vpsde_betas(int(timesteps) + 1)
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    tab64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1.0),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.maximum(pv, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in tab64.items()}, tab64


# --------------------------------------------------------------------------
# a8-a10: DiffNet
# --------------------------------------------------------------------------
def sinusoidal_pos_emb(t, dim):
    """diffnet.py:34-46 (always fp32; t int64[B])."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half) * -e)
    ang = t[:, None] * freq[None, :]
    return torch.cat((ang.sin(), ang.cos()), dim=-1)


def mish(x):
    """diffnet.py:14-16."""
    return x * torch.tanh(F.softplus(x))


def step_embedding(W, t, prefix="denoise_fn."):
    """diffnet.py:121-122,97-101: sinusoid -> Linear -> Mish -> Linear. [B,C]"""
    C = W[prefix + "mlp.2.weight"].shape[0]
    e = sinusoidal_pos_emb(t, C).to(W[prefix + "mlp.0.weight"].dtype)
    h = F.linear(e, W[prefix + "mlp.0.weight"], W[prefix + "mlp.0.bias"])
    return F.linear(mish(h), W[prefix + "mlp.2.weight"], W[prefix + "mlp.2.bias"])


def residual_block(W, p, x, cond, dstep, dilation):
    """diffnet.py:60-81 (one DiffNet layer).  x[B,C,T], cond[B,H,T], dstep[B,C]."""
    d = F.linear(dstep, W[p + "diffusion_projection.weight"], W[p + "diffusion_projection.bias"])[:, :, None]
    c = F.conv1d(cond, W[p + "conditioner_projection.weight"], W[p + "conditioner_projection.bias"])
    y = F.conv1d(x + d, W[p + "dilated_conv.weight"], W[p + "dilated_conv.bias"],
                 padding=dilation, dilation=dilation) + c
    gate, filt = torch.chunk(y, 2, dim=1)
    y = torch.sigmoid(gate) * torch.tanh(filt)
    y = F.conv1d(y, W[p + "output_projection.weight"], W[p + "output_projection.bias"])
    res, skip = torch.chunk(y, 2, dim=1)
    return (x + res) / math.sqrt(2.0), skip


def diffnet_forward(W, spec, t, cond, dilation_cycle_length=1, prefix="denoise_fn.", trace=None):
    """diffnet.py:110-132.  spec[B,1,M,T], t int64[B], cond[B,H,T] -> [B,1,M,T]."""
    x = F.relu(F.conv1d(spec[:, 0], W[prefix + "input_projection.weight"], W[prefix + "input_projection.bias"]))
    dstep = step_embedding(W, t, prefix)
    n_layers = 0
    while (prefix + "residual_layers.%d.dilated_conv.weight" % n_layers) in W:
        n_layers += 1
    skip = None
    for i in range(n_layers):
        dil = 2 ** (i % dilation_cycle_length)  # diffnet.py:102-105
        x, s = residual_block(W, prefix + "residual_layers.%d." % i, x, cond, dstep, dil)
        skip = s if skip is None else skip + s  # == torch.sum(torch.stack(skip), 0), diffnet.py:128
        if trace is not None:
            trace.append((x.clone(), s.clone()))
    x = skip / math.sqrt(n_layers)
    x = F.relu(F.conv1d(x, W[prefix + "skip_projection.weight"], W[prefix + "skip_projection.bias"]))
    x = F.conv1d(x, W[prefix + "output_projection.weight"], W[prefix + "output_projection.bias"])
    return x[:, None]


# --------------------------------------------------------------------------
# a3-a6: posterior / forward diffusion
# --------------------------------------------------------------------------
def extract(a, t, ndim):
    """diffusion_utils.py:59-62."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_posterior_sample(tab, x0, x_t, t, noise):
    """spec_denoiser.py:86-101.  noise is drawn even at t == 0 (masked out)."""
    mean = extract(tab["posterior_mean_coef1"], t, x_t.dim()) * x0 + \
        extract(tab["posterior_mean_coef2"], t, x_t.dim()) * x_t
    logvar = extract(tab["posterior_log_variance_clipped"], t, x_t.dim())
    nonzero = (1 - (t == 0).float()).reshape(t.shape[0], *((1,) * (x_t.dim() - 1)))
    return mean + nonzero * (0.5 * logvar).exp() * noise


def q_sample(tab, x_start, t, noise):
    """spec_denoiser.py:126-132."""
    return extract(tab["sqrt_alphas_cumprod"], t, x_start.dim()) * x_start + \
        extract(tab["sqrt_one_minus_alphas_cumprod"], t, x_start.dim()) * noise


def p_sample_loop(W, tab, cond, noises, timesteps, dilation_cycle_length=1, trace=None):
    """spec_denoiser.py:178-184.  noises[0] = x_T, noises[1+k] = eps of the k-th
    executed step (i = timesteps-1-k).  Returns x [B,1,M,T]."""
    x = noises[0]
    B = x.shape[0]
    for k, i in enumerate(reversed(range(0, timesteps))):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = diffnet_forward(W, x, t, cond, dilation_cycle_length)
        x = q_posterior_sample(tab, x0, x, t, noises[1 + k])
        if trace is not None:
            trace.append((x0.clone(), x.clone()))
    return x


# --------------------------------------------------------------------------
# a11-a16: conditioner (FastSpeech-style encoder, predictors, MelEncoder)
# --------------------------------------------------------------------------
def layer_norm_ch(x, w, b, eps=1e-5):
    """modules/commons/layers.py:5-24 with dim=1: LN over channels of [B,C,T]."""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), w, b, eps).transpose(1, -1)


def conv_blocks(W, p, x, n_blocks=4, layers_in_block=2, kernel_size=5, dilations=None, post_k=3):
    """modules/commons/conv.py:24-116 (ConvBlocks.forward, is_BTC=True, norm 'ln').
    x [B,T,H] -> [B,T,out]."""
    dilations = dilations or [1] * n_blocks
    x = x.transpose(1, 2)
    nonpadding = (x.abs().sum(1) > 0).float()[:, None, :]  # conv.py:108
    for bi in range(n_blocks):
        d = dilations[bi]
        np_b = (x.abs().sum(1) > 0).float()[:, None, :]  # conv.py:58
        for li in range(layers_in_block):
            q = "%sres_blocks.%d.blocks.%d." % (p, bi, li)
            h = layer_norm_ch(x, W[q + "0.weight"], W[q + "0.bias"])
            h = F.conv1d(h, W[q + "1.weight"], W[q + "1.bias"], dilation=d,
                         padding=(d * (kernel_size - 1)) // 2)
            h = h * kernel_size ** -0.5
            h = F.gelu(h)
            h = F.conv1d(h, W[q + "4.weight"], W[q + "4.bias"])
            x = (x + h) * np_b
    x = x * nonpadding
    x = layer_norm_ch(x, W[p + "last_norm.weight"], W[p + "last_norm.bias"]) * nonpadding
    x = F.conv1d(x, W[p + "post_net1.weight"], W[p + "post_net1.bias"], padding=post_k // 2) * nonpadding
    return x.transpose(1, 2)


def text_encoder(W, txt_tokens, p="fs.encoder."):
    """conv.py:119-139: sqrt(H) * Embedding -> ConvBlocks."""
    emb = W[p + "embed_tokens.weight"]
    x = math.sqrt(emb.shape[1]) * F.embedding(txt_tokens, emb, padding_idx=0)
    return conv_blocks(W, p, x)


def mel2token_to_dur(mel2token, T_txt):
    """utils/audio/align.py:71-90 (scatter_add count, drop slot 0)."""
    B = mel2token.shape[0]
    dur = mel2token.new_zeros(B, T_txt + 1).scatter_add(1, mel2token, torch.ones_like(mel2token))
    return dur[:, 1:]


def predictor_stack(W, p, x, n_layers, k, padding_mask=None):
    """nar_tts_modules.py:8-34 / 75-100: n x (conv k -> ReLU -> LN(ch) -> dropout[eval])."""
    x = x.transpose(1, -1)
    for i in range(n_layers):
        q = "%sconv.%d." % (p, i)
        x = F.conv1d(x, W[q + "0.weight"], W[q + "0.bias"], padding=k // 2)
        x = F.relu(x)
        x = layer_norm_ch(x, W[q + "2.weight"], W[q + "2.bias"])
        if padding_mask is not None:
            x = x * (1 - padding_mask.float())[:, None, :]
    return x.transpose(1, -1)


def duration_predictor(W, x, x_padding, p="fs.dur_predictor.", n_layers=3, k=5):
    """nar_tts_modules.py:24-34."""
    h = predictor_stack(W, p, x, n_layers, k, x_padding)
    h = F.softplus(F.linear(h, W[p + "linear.0.weight"], W[p + "linear.0.bias"]))
    h = h * (1 - x_padding.float())[:, :, None]
    return h[..., 0]


def length_regulator(dur, dur_padding):
    """nar_tts_modules.py:42-72 (alpha = 1)."""
    dur = torch.round(dur.float()).long()
    dur = dur * (1 - dur_padding.long())
    token_idx = torch.arange(1, dur.shape[1] + 1)[None, :, None]
    cs = torch.cumsum(dur, 1)
    cs_prev = F.pad(cs, [1, -1])
    pos = torch.arange(int(dur.sum(-1).max()))[None, None]
    mask = (pos >= cs_prev[:, :, None]) & (pos < cs[:, :, None])
    return (token_idx * mask.long()).sum(1)


def denorm_f0(f0, uv, pitch_padding=None, fmin=50, fmax=900):
    """utils/audio/pitch/utils.py:71-82 (pitch_norm='log'); returns a new tensor."""
    f0 = (2 ** f0).clamp(min=fmin, max=fmax)
    if uv is not None:
        f0 = torch.where(uv > 0, torch.zeros_like(f0), f0)
    if pitch_padding is not None:
        f0 = torch.where(pitch_padding, torch.zeros_like(f0), f0)
    return f0


def f0_to_coarse(f0, f0_bin=256, f0_max=900.0, f0_min=50.0):
    """utils/audio/pitch/utils.py:17-28 -> int64 bins 1..255."""
    mel_min = 1127 * np.log(1 + f0_min / 700)
    mel_max = 1127 * np.log(1 + f0_max / 700)
    m = 1127 * (1 + f0 / 700).log()
    m = torch.where(m > 0, (m - mel_min) * (f0_bin - 2) / (mel_max - mel_min) + 1, m)
    m = torch.where(m <= 1, torch.ones_like(m), m)
    m = torch.where(m > f0_bin - 1, torch.full_like(m, f0_bin - 1), m)
    return (m + 0.5).long()


def expand_states(h, mel2token):
    """modules/tts/commons/align_ops.py:21-25: index 0 -> zero row."""
    h = F.pad(h, [0, 0, 1, 0])
    idx = mel2token[..., None].repeat([1, 1, h.shape[-1]])
    return torch.gather(h, 1, idx)


def mel_encoder(W, x, p="mel_encoder."):
    """modules/speech_editing/commons/mel_encoder.py:15-19."""
    h = F.relu(F.linear(x, W[p + "encoder.0.weight"], W[p + "encoder.0.bias"]))
    h = F.relu(F.linear(h, W[p + "encoder.2.weight"], W[p + "encoder.2.bias"]))
    return F.linear(h, W[p + "fc_out.weight"], W[p + "fc_out.bias"])


def fastspeech_forward(W, txt_tokens, time_mel_masks, mel2ph, spk_embed, f0, uv,
                       use_pred_mel2ph=False, use_pred_pitch=False, p="fs.", predictor_grad=0.1,
                       use_pitch_embed=True):
    """fs.py:83-189 with skip_decoder=True, use_spk_embed, pitch_type 'frame', use_uv (egs/spec_denoiser.yaml);
    `use_pitch_embed=False` is egs/spec_denoiser_libritts.yaml:169 (the pitch block fs.py:97-99 is skipped and the
    model has no pitch_embed / pitch_predictor, fs.py:73-78).  Eval mode (no dropout).
    Returns the ret dict incl. integer intermediates."""
    ret = {}
    enc = text_encoder(W, txt_tokens, p + "encoder.")
    src_nonpad = (txt_tokens > 0).float()[:, :, None]
    style = F.linear(spk_embed, W[p + "spk_embed_proj.weight"], W[p + "spk_embed_proj.bias"])[:, None, :]
    # -- duration (fs.py:123-151)
    dur_inp = (enc + style) * src_nonpad
    T_txt = txt_tokens.shape[1]
    nonpad = (txt_tokens != 0).float()
    masked_dur = mel2token_to_dur(mel2ph * (1 - time_mel_masks).squeeze(-1).long(), T_txt) * nonpad
    ret["masked_dur"] = masked_dur.long()
    dur_inp = dur_inp + F.embedding(masked_dur.long(), W[p + "dur_embed.weight"], padding_idx=0)
    dur_inp = dur_inp.detach() + predictor_grad * (dur_inp - dur_inp.detach())  # fs.py:144-145
    src_padding = txt_tokens == 0
    dur = duration_predictor(W, dur_inp, src_padding, p + "dur_predictor.")
    ret["dur"] = dur
    if use_pred_mel2ph:
        mel2ph = length_regulator(dur, src_padding)
    ret["mel2ph"] = mel2ph
    tgt_nonpad = (mel2ph > 0).float()[:, :, None]
    dec_inp = expand_states(enc, mel2ph)
    if not use_pitch_embed:
        ret["decoder_inp"] = (dec_inp + style) * tgt_nonpad
        return ret
    # -- pitch (fs.py:153-189)
    pitch_inp = (dec_inp + style) * tgt_nonpad
    pitch_padding = mel2ph == 0
    m = time_mel_masks.squeeze(-1)
    masked_f0 = f0 * (1 - m)
    masked_uv = uv * (1 - m)
    masked_pitch = f0_to_coarse(denorm_f0(masked_f0, masked_uv, pitch_padding))
    ret["masked_pitch"] = masked_pitch
    pp_inp = pitch_inp + F.embedding(masked_pitch, W[p + "pitch_embed.weight"], padding_idx=0)
    pp_inp = pp_inp.detach() + predictor_grad * (pp_inp - pp_inp.detach())  # fs.py:167-169
    h = predictor_stack(W, p + "pitch_predictor.", pp_inp, 5, 5)
    pitch_pred = F.linear(h, W[p + "pitch_predictor.linear.weight"], W[p + "pitch_predictor.linear.bias"])
    ret["pitch_pred"] = pitch_pred
    if use_pred_pitch:
        pitch_padding = None
        pred_f0 = pitch_pred[:, :, 0]
        pred_uv = pitch_pred[:, :, 1] > 0
        res_f0 = f0 * (1 - m) + pred_f0 * m
        res_uv = uv * (1 - m) + pred_uv * m
    else:
        res_f0, res_uv = f0, uv
    f0_denorm = denorm_f0(res_f0, res_uv, pitch_padding)
    pitch = f0_to_coarse(f0_denorm)
    ret["pitch"] = pitch
    ret["f0_denorm"] = f0_denorm
    ret["f0_denorm_pred"] = denorm_f0(pitch_pred[:, :, 0], pitch_pred[:, :, 1] > 0, pitch_padding)
    dec_inp = dec_inp + F.embedding(pitch, W[p + "pitch_embed.weight"], padding_idx=0)
    ret["decoder_inp"] = (dec_inp + style) * tgt_nonpad
    return ret


def fastspeech_normal_forward(W, txt_tokens, mel2ph, spk_embed, f0, uv, p="fs.", predictor_grad=0.1,
                              use_pitch_embed=True):
    """modules/tts/fs.py:81-168 with skip_decoder=True (the plain FastSpeech: predictors without masked ground
    truth, `mel2ph is None` / `f0 is None` select the predictions).  Eval mode."""
    ret = {}
    enc = text_encoder(W, txt_tokens, p + "encoder.")
    src_nonpad = (txt_tokens > 0).float()[:, :, None]
    style = F.linear(spk_embed, W[p + "spk_embed_proj.weight"], W[p + "spk_embed_proj.bias"])[:, None, :]
    dur_inp = (enc + style) * src_nonpad
    dur_inp = dur_inp.detach() + predictor_grad * (dur_inp - dur_inp.detach())  # :132-133
    src_padding = txt_tokens == 0
    ret["dur"] = dur = duration_predictor(W, dur_inp, src_padding, p + "dur_predictor.")
    if mel2ph is None:
        mel2ph = length_regulator(dur, src_padding)
    ret["mel2ph"] = mel2ph
    tgt_nonpad = (mel2ph > 0).float()[:, :, None]
    dec_inp = expand_states(enc, mel2ph)
    if use_pitch_embed:  # :140-168, pitch_type 'frame'
        pitch_inp = (dec_inp + style) * tgt_nonpad
        pitch_padding = mel2ph == 0
        pitch_inp = pitch_inp.detach() + predictor_grad * (pitch_inp - pitch_inp.detach())
        h = predictor_stack(W, p + "pitch_predictor.", pitch_inp, 5, 5)
        pitch_pred = F.linear(h, W[p + "pitch_predictor.linear.weight"], W[p + "pitch_predictor.linear.bias"])
        ret["pitch_pred"] = pitch_pred
        if f0 is None:
            f0 = pitch_pred[:, :, 0]
            uv = pitch_pred[:, :, 1] > 0
        ret["f0_denorm"] = f0_denorm = denorm_f0(f0, uv, pitch_padding)
        ret["pitch"] = pitch = f0_to_coarse(f0_denorm)
        ret["f0_denorm_pred"] = denorm_f0(pitch_pred[:, :, 0], pitch_pred[:, :, 1] > 0, pitch_padding)
        dec_inp = dec_inp + F.embedding(pitch, W[p + "pitch_embed.weight"], padding_idx=0)
    ret["decoder_inp"] = (dec_inp + style) * tgt_nonpad
    return ret


def conditioner(W, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv,
                use_pred_mel2ph=False, use_pred_pitch=False, use_pitch_embed=True, variant="masked"):
    """spec_denoiser.py:159-167: fs(...) + mel_encoder(ref*(1-mask))*nonpad.
    Returns (ret, cond[B,H,T]).  variant 'normal' = spec_denoiser_normal.py:158-162, whose positional call
    `fs(txt_tokens, mel2ph, spk_embed, f0, uv, energy)` binds f0 -> spk_id (unused), uv -> f0, energy=None -> uv
    against modules/tts/fs.py:81-82; restated as executed."""
    if variant == "normal":
        assert not (use_pred_mel2ph or use_pred_pitch)
        ret = fastspeech_normal_forward(W, txt_tokens, mel2ph, spk_embed, f0=uv, uv=None,
                                        use_pitch_embed=use_pitch_embed)
    else:
        ret = fastspeech_forward(W, txt_tokens, time_mel_masks, mel2ph, spk_embed, f0, uv,
                                 use_pred_mel2ph, use_pred_pitch, use_pitch_embed=use_pitch_embed)
    tgt_nonpad = (ret["mel2ph"] > 0).float()[:, :, None]
    dec = ret["decoder_inp"] + mel_encoder(W, ref_mels * (1 - time_mel_masks)) * tgt_nonpad
    ret["decoder_inp"] = dec
    return ret, dec.transpose(1, 2)


def gaussian_diffusion_infer(W, timesteps, inputs, noises, dilation_cycle_length=1, trace=None, **flags):
    """spec_denoiser.py:154-185 with infer=True and explicit noise tensors."""
    tab, _ = diffusion_tables(timesteps)
    ret, cond = conditioner(W, inputs["txt_tokens"], inputs["time_mel_masks"], inputs["mel2ph"],
                            inputs["spk_embed"], inputs["ref_mels"], inputs["f0"], inputs["uv"], **flags)
    x = p_sample_loop(W, tab, cond, noises, timesteps, dilation_cycle_length, trace)
    ret["mel_out"] = x[:, 0].transpose(1, 2)
    ret["cond"] = cond
    return ret


def gaussian_diffusion_train(W, timesteps, inputs, t, noise, dilation_cycle_length=1, use_pitch_embed=True,
                             variant="masked"):
    """spec_denoiser.py:168-176 (infer=False) with explicit t and eps; eval-mode predictors."""
    tab, _ = diffusion_tables(timesteps)
    ret, cond = conditioner(W, inputs["txt_tokens"], inputs["time_mel_masks"], inputs["mel2ph"],
                            inputs["spk_embed"], inputs["ref_mels"], inputs["f0"], inputs["uv"],
                            use_pitch_embed=use_pitch_embed, variant=variant)
    nonpadding = (inputs["mel2ph"] != 0).float().unsqueeze(1).unsqueeze(1)
    x_start = inputs["ref_mels"].transpose(1, 2)[:, None]
    x_t = q_sample(tab, x_start, t, noise) * nonpadding
    x0 = diffnet_forward(W, x_t, t, cond, dilation_cycle_length) * nonpadding
    ret["mel_out"] = x0[:, 0].transpose(1, 2)
    ret["x_t"] = x_t
    return ret


# --------------------------------------------------------------------------
# a17: HiFi-GAN generator forward
# --------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# caller side of inference: the edit bookkeeping in front of the hot path
# ---------------------------------------------------------------------------------------------
def parse_region_list_from_str(region_str):
    """inference/tts/infer_utils.py:47-53: "[3,4][7,7]" -> [[3,4],[7,7]] sorted by start (1-based, no leading 0)."""
    import re
    found = re.findall(r"\[([1-9]\d*),([1-9]\d*)]", region_str)
    return sorted([[int(a), int(b)] for a, b in found], key=lambda r: r[0])


def words_region_from_text_region(words, region_list, is_skipped):
    """inference/tts/infer_utils.py:29-44: map regions counted in real words to positions in the word list that
    still contains boundary tokens (`is_skipped(word)`: silence tokens among '|', '<BOS>', '<pad>')."""
    assert len(region_list) >= 1
    out = [[0, 0] for _ in region_list]
    word_id, rid = 0, 0
    for i, w in enumerate(words):
        if is_skipped(w):
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


def edit_plan_durations(sample):
    """inference/tts/spec_denoiser.py:83-96 (integer part): ground-truth durations of the untouched head / tail
    phonemes laid over the EDITED phoneme sequence, and the frame mask of the edited words in the original."""
    ph2word, e_ph2word, dur = sample["ph2word"], sample["edited_ph2word"], sample["dur"]
    w0, w1 = sample["words_region"][0]
    masked_dur = torch.zeros_like(e_ph2word)
    n_head = int((ph2word < w0).sum())
    masked_dur[:, :n_head] = dur[:, :n_head]
    if int(ph2word.max()) > w1:
        n_tail = int((ph2word > w1).sum())
        masked_dur[:, -n_tail:] = dur[:, -n_tail:]
    in_region = (sample["mel2word"] >= w0) & (sample["mel2word"] <= w1)
    return masked_dur, in_region


def edit_splice(sample, pred_mel2ph):
    """inference/tts/spec_denoiser.py:97-135: splice the predicted alignment of the new words between the original
    head and tail (tail phoneme indices re-based to `max(new) + 2`, :110), and build the masked reference mel,
    f0 / uv and the time mask of the edited utterance.  All integer work; returns a dict of tensors."""
    w0, w1 = sample["words_region"][0]
    c0, c1 = sample["edited_words_region"][0]
    mel, mel2ph, mel2word = sample["mel"], sample["mel2ph"], sample["mel2word"]
    f0, uv = sample["f0"], sample["uv"]
    e_ph2word = sample["edited_ph2word"][0]
    e_mel2word = e_ph2word[pred_mel2ph[0] - 1][None, :]  # index -1 (p == 0) wraps to the last phoneme, as upstream
    sel = (e_mel2word >= c0) & (e_mel2word <= c1)
    in_region = (mel2word >= w0) & (mel2word <= w1)
    delta = int(sel.sum()) - int(in_region.sum())
    head = int((mel2word < w0).sum())
    tail = int((mel2word <= w1).sum()) + delta
    T_new = mel2ph.shape[1] + delta
    new_m2p = torch.zeros(1, T_new, dtype=torch.int64)
    new_m2p[:, :head] = mel2ph[:, :head]
    new_m2p[:, head:tail] = pred_mel2ph[sel]
    after = mel2word > w1
    if int(mel2word.max()) > w1:
        tv = mel2ph[after]
        new_m2p[:, tail:] = tv - tv.min() + pred_mel2ph[sel].max() + 2
    ref = torch.zeros(1, T_new, mel.shape[2])
    ref[:, :head] = mel[:, :head]
    ref[:, tail:] = mel[after]
    nf0, nuv = torch.zeros(1, T_new), torch.zeros(1, T_new)
    nf0[:, :head], nuv[:, :head] = f0[:, :head], uv[:, :head]
    nf0[:, tail:], nuv[:, tail:] = f0[after], uv[after]
    tmask = torch.zeros(1, T_new, 1)
    tmask[:, head:tail] = 1.0
    return {"mel2ph": new_m2p, "ref_mels": ref, "f0": nf0, "uv": nuv, "time_mel_masks": tmask,
            "head_idx": head, "tail_idx": tail, "length_edited": delta}


def edit_forward_model(W, Wv, h, timesteps, sample, noises):
    """SpecDenoiserInfer.forward_model (inference/tts/spec_denoiser.py:63-149) on an already batched sample, with
    explicit noise: duration prediction on the edited text with the head/tail ground-truth durations embedded,
    splice, masked diffusion inference with use_pred_pitch, paste, vocode.  `noises` is a list (x_T, eps...) or a
    callable T_new -> list.  Returns the upstream 6-tuple plus the integer intermediates."""
    p = "fs."
    txt = sample["edited_txt_tokens"]
    masked_dur, in_region = edit_plan_durations(sample)
    enc = text_encoder(W, txt, p + "encoder.")
    src_nonpad = (txt > 0).float()[:, :, None]
    style = F.linear(sample["spk_embed"], W[p + "spk_embed_proj.weight"], W[p + "spk_embed_proj.bias"])[:, None, :]
    dur_inp = (enc + style) * src_nonpad
    dur_inp = dur_inp + F.embedding(masked_dur, W[p + "dur_embed.weight"], padding_idx=0)  # fs.py:141-142
    src_padding = txt == 0
    dur = duration_predictor(W, dur_inp, src_padding, p + "dur_predictor.")
    pred_mel2ph = length_regulator(dur, src_padding)
    sp = edit_splice(sample, pred_mel2ph)
    inputs = {"txt_tokens": txt, "time_mel_masks": sp["time_mel_masks"], "mel2ph": sp["mel2ph"],
              "spk_embed": sample["spk_embed"], "ref_mels": sp["ref_mels"], "f0": sp["f0"], "uv": sp["uv"]}
    if callable(noises):
        noises = noises(sp["mel2ph"].shape[1])
    ret = gaussian_diffusion_infer(W, timesteps, inputs, noises, use_pred_pitch=True)
    m = sp["time_mel_masks"]
    mel_out = ret["mel_out"] * m + sp["ref_mels"] * (1 - m)
    wav_out = hifigan_forward(Wv, h, mel_out.transpose(1, 2))[:, 0]
    wav_gt = hifigan_forward(Wv, h, sample["mel"].transpose(1, 2))[:, 0]
    masked_mel_gt = sample["mel"] * in_region.long()[:, :, None]
    return {"wav_out": wav_out[0], "wav_gt": wav_gt[0], "mel_out": mel_out[0], "mel_gt": sample["mel"][0],
            "masked_mel_out": sp["ref_mels"][0], "masked_mel_gt": masked_mel_gt[0],
            "masked_dur": masked_dur, "dur_pred": dur, "pred_mel2ph": pred_mel2ph, "edited_mel2ph": sp["mel2ph"],
            "edited_f0": sp["f0"], "edited_uv": sp["uv"], "time_mel_masks": sp["time_mel_masks"],
            "head_idx": sp["head_idx"], "tail_idx": sp["tail_idx"], "model_mel_out": ret["mel_out"],
            "pitch": ret["pitch"]}


# ---------------------------------------------------------------------------------------------
# CampNet: masked-mel transformer, coarse + fine decoder
# ---------------------------------------------------------------------------------------------
def sinusoid_table(n, dim, padding_idx=0):
    """modules/speech_editing/commons/transformer.py:31-48 (tensor2tensor flavour: [sin | cos], row padding_idx = 0)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(n, 1)], dim=1)
    e[padding_idx, :] = 0
    return e


def make_positions(x, padding_idx=0):
    """utils/nn/seq_utils.py:6-18: 1,2,3.. over the entries != padding_idx, padding_idx elsewhere."""
    m = x.ne(padding_idx).int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + padding_idx


def positional_embedding(x_first, dim):
    """transformer.py:50-69: rows of the sinusoid table picked by make_positions(input[..., 0] or tokens)."""
    pos = make_positions(x_first)
    tab = sinusoid_table(max(2000, int(pos.max()) + 1), dim)
    return tab[pos.reshape(-1)].view(x_first.shape[0], x_first.shape[1], dim)


def mha(W, p, query, key, n_heads, key_padding_mask=None, fill=float("-inf")):
    """Multi-head attention, bias-free projections (transformer.py:138-419).  query [Tq,B,H], key=value [Tk,B,H].
    Self-attention takes torch's F.multi_head_attention_forward (-inf key-padding fill); encoder-decoder attention
    (static_kv=True) takes the module's own path (:283-410) with a -1e8 fill.  The arithmetic is the same:
    q = Wq x * d^-1/2, softmax_fp32(q k^T (+fill)) v, out_proj.  Returns (out [Tq,B,H], probs [B,heads,Tq,Tk])."""
    Tq, B, H = query.shape
    Tk = key.shape[0]
    d = H // n_heads
    w = W[p + "in_proj_weight"]
    q = F.linear(query, w[:H]) * d ** -0.5
    k = F.linear(key, w[H:2 * H])
    v = F.linear(key, w[2 * H:])
    q = q.contiguous().view(Tq, B * n_heads, d).transpose(0, 1)
    k = k.contiguous().view(Tk, B * n_heads, d).transpose(0, 1)
    v = v.contiguous().view(Tk, B * n_heads, d).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if key_padding_mask is not None:
        s = s.view(B, n_heads, Tq, Tk).masked_fill(key_padding_mask[:, None, None, :], fill).view(B * n_heads, Tq, Tk)
    pr = F.softmax(s, dim=-1, dtype=torch.float32)
    o = torch.bmm(pr, v).transpose(0, 1).contiguous().view(Tq, B, H)
    return F.linear(o, W[p + "out_proj.weight"]), pr.view(B, n_heads, Tq, Tk)


def transformer_ffn(W, p, x, k, left_pad):
    """transformer.py:76-113: conv k (H->4H, SAME or LEFT padding) * k^-1/2 -> GELU -> Linear(4H->H).  x [T,B,H]."""
    h = x.permute(1, 2, 0)
    if left_pad:
        h = F.conv1d(F.pad(h, (k - 1, 0)), W[p + "ffn_1.1.weight"], W[p + "ffn_1.1.bias"])
    else:
        h = F.conv1d(h, W[p + "ffn_1.weight"], W[p + "ffn_1.bias"], padding=k // 2)
    h = F.gelu(h.permute(2, 0, 1) * k ** -0.5)
    return F.linear(h, W[p + "ffn_2.weight"], W[p + "ffn_2.bias"])


def _ln(W, p, x):
    return F.layer_norm(x, (x.shape[-1],), W[p + "weight"], W[p + "bias"], 1e-5)


def campnet_text_encoder(W, txt, n_layers=3, n_heads=2, k=9, p="encoder."):
    """TransformerEncoder (transformer.py:712-747) on FFTBlocks (:684-709) of EncSALayer (:504-528); dropout 0."""
    H = W[p + "embed_tokens.weight"].shape[1]
    pad = txt.eq(0)
    x = math.sqrt(H) * F.embedding(txt, W[p + "embed_tokens.weight"], padding_idx=0) + positional_embedding(txt, H)
    keep = 1 - pad.transpose(0, 1).float()[:, :, None]
    x = x.transpose(0, 1) * keep
    for i in range(n_layers):
        q = "%slayers.%d.op." % (p, i)
        a, _ = mha(W, q + "self_attn.", _ln(W, q + "layer_norm1.", x), _ln(W, q + "layer_norm1.", x), n_heads, pad)
        x = (x + a) * keep
        x = (x + transformer_ffn(W, q + "ffn.", _ln(W, q + "layer_norm2.", x), k, False)) * keep
        x = x * keep
    x = _ln(W, p + "layer_norm.", x) * keep
    return x.transpose(0, 1)


def campnet_coarse_decoder(W, x, enc_out, n_layers=6, n_heads=2, k=9, p="decoder_coarse."):
    """TransformerDecoder (transformer.py:750-811) of DecSALayer (:549-609): self-attention WITHOUT any mask (the layer
    is called without self_attn_padding_mask), encoder-decoder attention with the -1e8 key-padding fill, causal-padded
    conv FFN.  Returns (x [B,T,H], head-averaged attention of the FIRST layer [B,T,T_txt])."""
    H = x.shape[-1]
    enc_pad = enc_out.abs().sum(-1).eq(0)
    pad = x.abs().sum(-1).eq(0)
    keep = 1 - pad.transpose(0, 1).float()[:, :, None]
    x = x + W[p + "pos_embed_alpha"] * positional_embedding(x[..., 0], H)
    x = x.transpose(0, 1) * keep
    enc = enc_out.transpose(0, 1)
    first = None
    for i in range(n_layers):
        q = "%slayers.%d.op." % (p, i)
        h = _ln(W, q + "layer_norm1.", x)
        a, _ = mha(W, q + "self_attn.", h, h, n_heads, None)
        x = x + a
        a, pr = mha(W, q + "encoder_attn.", _ln(W, q + "layer_norm2.", x), enc, n_heads, enc_pad, -1e8)
        x = x + a
        x = x + transformer_ffn(W, q + "ffn.", _ln(W, q + "layer_norm3.", x), k, True)
        x = x * keep
        if first is None:
            first = pr.mean(dim=1)
    x = _ln(W, p + "layer_norm.", x) * keep
    return x.transpose(0, 1), first


def campnet_forward(W, txt, mels, time_mel_masks, k=9):
    """modules/speech_editing/campnet/campnet.py:42-69.  time_mel_masks [B,T,1]."""
    src_nonpad = (txt > 0).float()[:, :, None]
    enc = campnet_text_encoder(W, txt, k=k) * src_nonpad * src_nonpad
    m = time_mel_masks
    mel_nonpad = (mels.abs().sum(-1) > 0).float()[:, :, None]
    x = mel_encoder(W, mels * (1 - m) + W["mask_emb"] * m) * mel_nonpad
    h, attn = campnet_coarse_decoder(W, x, enc, k=k)
    coarse = F.linear(h * mel_nonpad, W["mel_out_coarse.weight"]) * mel_nonpad
    mel_coarse = mels * (1 - m) + coarse * m
    x = mel_encoder(W, mel_coarse) * mel_nonpad
    f = conv_blocks(W, "decoder_fine.", x, n_blocks=5, layers_in_block=2, kernel_size=5) * mel_nonpad
    fine = F.linear(f, W["mel_out_fine.weight"]) * mel_nonpad
    fine = mel_coarse + fine * m
    return {"mel_out_coarse": coarse, "mel_out_fine": fine, "attn": attn}


def campnet_losses(W, txt, mels, time_mel_masks, lambdas=None):
    """tasks/speech_editing/campnet.py:50-69: l1 + ssim (mel_losses l1:0.5|ssim:0.5) on the masked coarse and fine
    predictions; `mel_out` = fine prediction pasted into the original."""
    lam = lambdas or {"l1": 0.5, "ssim": 0.5}
    out = campnet_forward(W, txt, mels, time_mel_masks)
    m = time_mel_masks
    losses = {}
    for name in ("coarse", "fine"):
        pred, tgt = out["mel_out_" + name] * m, mels * m
        losses["l1_" + name] = l1_loss(pred, tgt) * lam["l1"]
        losses["ssim_" + name] = ssim_loss(pred, tgt) * lam["ssim"]
    out["mel_out"] = out["mel_out_fine"] * m + mels * (1 - m)
    return losses, out


def weight_norm_fold(g, v):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, norm over all dims but 0."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def _wn(W, p):
    if (p + "weight") in W:
        return W[p + "weight"]
    return weight_norm_fold(W[p + "weight_g"], W[p + "weight_v"])


def hifigan_forward(W, h, x, prefix=""):
    """modules/vocoder/hifigan/hifigan.py:126-142 (+ ResBlock1 :51-58, ResBlock2 :79-84).
    x [B,80,T] -> [B,1,T*prod(upsample_rates)]."""
    p = prefix
    x = F.conv1d(x, _wn(W, p + "conv_pre."), W[p + "conv_pre.bias"], padding=3)
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(W, p + "ups.%d." % i), W[p + "ups.%d.bias" % i],
                               stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            q = p + "resblocks.%d." % (i * nk + j)
            y = x
            if str(h["resblock"]) == "1":
                for m, d in enumerate(rd):
                    t = F.leaky_relu(y, 0.1)
                    t = F.conv1d(t, _wn(W, q + "convs1.%d." % m), W[q + "convs1.%d.bias" % m],
                                 dilation=d, padding=(rk * d - d) // 2)
                    t = F.leaky_relu(t, 0.1)
                    t = F.conv1d(t, _wn(W, q + "convs2.%d." % m), W[q + "convs2.%d.bias" % m],
                                 padding=(rk - 1) // 2)
                    y = t + y
            else:
                for m, d in enumerate(rd):
                    t = F.leaky_relu(y, 0.1)
                    t = F.conv1d(t, _wn(W, q + "convs.%d." % m), W[q + "convs.%d.bias" % m],
                                 dilation=d, padding=(rk * d - d) // 2)
                    y = t + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01, hifigan.py:138
    x = F.conv1d(x, _wn(W, p + "conv_post."), W[p + "conv_post.bias"], padding=3)
    return torch.tanh(x)


# --------------------------------------------------------------------------
# a20: training losses (tasks/tts/speech_base.py:219-257, tasks/speech_editing/speech_editing_base.py:58-108,
#      utils/metrics/ssim.py:12-44, utils/nn/seq_utils.py:33-37)
# --------------------------------------------------------------------------
def weights_nonzero_speech(target):
    return target.abs().sum(-1, keepdim=True).ne(0).float().repeat(1, 1, target.size(-1))


def l1_loss(pred, target):
    w = weights_nonzero_speech(target)
    return (F.l1_loss(pred, target, reduction="none") * w).sum() / w.sum()


def _ssim_map(img1, img2, window_size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float()[None, None]
    p = window_size // 2
    mu1, mu2 = F.conv2d(img1, window, padding=p), F.conv2d(img2, window, padding=p)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=p) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=p) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=p) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def ssim_loss(pred, target, bias=6.0):
    w = weights_nonzero_speech(target)
    m = _ssim_map(pred[:, None] + bias, target[:, None] + bias).mean(1)
    return ((1 - m) * w).sum() / w.sum()


def dur_losses(dur_pred, mel2ph, txt_tokens, sil_ids, lam_p, lam_w):
    B, T = txt_tokens.shape
    nonpad = (txt_tokens != 0).float()
    dur_gt = mel2token_to_dur(mel2ph, T).float() * nonpad
    is_sil = torch.zeros_like(txt_tokens).bool()
    for i in sil_ids:
        is_sil = is_sil | (txt_tokens == i)
    is_sil = is_sil.float()
    pd = F.mse_loss((dur_pred + 1).log(), (dur_gt + 1).log(), reduction="none")
    pd = (pd * nonpad).sum() / nonpad.sum() * lam_p
    word_id = (is_sil.cumsum(-1) * (1 - is_sil)).long()
    wp = dur_pred.new_zeros([B, int(word_id.max()) + 1]).scatter_add(1, word_id, dur_pred)[:, 1:]
    wg = dur_gt.new_zeros([B, int(word_id.max()) + 1]).scatter_add(1, word_id, dur_gt)[:, 1:]
    wl = F.mse_loss((wp + 1).log(), (wg + 1).log(), reduction="none")
    wm = (wg > 0).float()
    return pd, (wl * wm).sum() / wm.sum() * lam_w


def pitch_losses(pitch_pred, f0, uv, mel2ph, lam_uv, lam_f0):
    nonpad = (mel2ph != 0).float()
    uvl = (F.binary_cross_entropy_with_logits(pitch_pred[:, :, 1], uv, reduction="none") * nonpad).sum() \
        / nonpad.sum() * lam_uv
    nv = nonpad * (uv == 0).float()
    f0l = (F.l1_loss(pitch_pred[:, :, 0], f0, reduction="none") * nv).sum() / nv.sum() * lam_f0
    return uvl, f0l


def training_losses(W, timesteps, inputs, t, noise, sil_ids=(1, 2, 3), lambdas=None, dilation_cycle_length=1,
                    use_pitch_embed=True, variant="masked"):
    """tasks/speech_editing/spec_denoiser.py:39-62 (infer=False) on top of gaussian_diffusion_train; eval-mode
    predictors (no dropout).  Returns (losses dict, ret)."""
    lam = dict(l1=0.5, ssim=0.5, ph_dur=0.1, word_dur=1.0, uv=1.0, f0=1.0)
    lam.update(lambdas or {})
    ret = gaussian_diffusion_train(W, timesteps, inputs, t, noise, dilation_cycle_length, use_pitch_embed, variant)
    tm = inputs["time_mel_masks"]
    pred, target = ret["mel_out"] * tm, inputs["ref_mels"] * tm
    losses = {"l1_coarse": l1_loss(pred, target) * lam["l1"], "ssim_coarse": ssim_loss(pred, target) * lam["ssim"]}
    losses["pdur"], losses["wdur"] = dur_losses(ret["dur"], inputs["mel2ph"], inputs["txt_tokens"], sil_ids,
                                                lam["ph_dur"], lam["word_dur"])
    if use_pitch_embed:  # tasks/speech_editing/spec_denoiser.py:55-56
        losses["uv"], losses["f0"] = pitch_losses(ret["pitch_pred"], inputs["f0"], inputs["uv"], inputs["mel2ph"],
                                                  lam["uv"], lam["f0"])
    return losses, ret


# --------------------------------------------------------------------------
# metric: mel-level MCD (utils/eval/mcd.py:13-31,89-95,103-144 restated)
# --------------------------------------------------------------------------
def mel_mcd(mel_a, mel_b, n_mfcc=39):
    """DCT-II (unnormalised) over the mel axis of log10-mels [T,M], keep c1..c_n,
    /2, mean per-frame L2 of the difference."""
    from scipy.fft import dct
    a = dct(np.asarray(mel_a, dtype=np.float64), type=2, axis=-1, norm=None)[..., 1:n_mfcc + 1] / 2.0
    b = dct(np.asarray(mel_b, dtype=np.float64), type=2, axis=-1, norm=None)[..., 1:n_mfcc + 1] / 2.0
    return float(np.sqrt(((a - b) ** 2).sum(-1)).mean())
