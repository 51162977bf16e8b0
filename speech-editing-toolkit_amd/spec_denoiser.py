"""GaussianDiffusion (FluentSpeech spec_denoiser) on the HIP kernels.

Same constructor arguments, registered buffers (16), children (`denoise_fn`, `fs`, `mel_encoder`),
forward signature and return-dict keys as modules/speech_editing/spec_denoiser/spec_denoiser.py:16-185.
"""
from functools import partial

import numpy as np
import torch
from torch import nn

from . import autograd_ops, ops
from .diffusion_utils import get_noise_schedule_list
from .fs import FastSpeech, FastSpeechNormal, MelEncoder
from .hparams import hparams as _global_hparams


class GaussianDiffusion(nn.Module):
    fs_cls = FastSpeech
    # parameters no loss of this model ever reaches (the conditioner is always run with skip_decoder=True,
    # modules/speech_editing/spec_denoiser/spec_denoiser.py:159-161): the reference needs find_unused_parameters for them
    # (utils/commons/trainer.py:475-479); here they are laid out behind the exchanged part of the flat gradient buffer and never sent
    # (training.FlatAdamW: 95.35 -> 80.7 MB per step)
    unused_parameter_prefixes = ("fs.decoder.", "fs.mel_out.")

    def __init__(self, phone_encoder, out_dims, denoise_fn, timesteps=1000, time_scale=1, loss_type="l1",
                 betas=None, spec_min=None, spec_max=None, hp=None):
        super().__init__()
        hp = hp if hp is not None else _global_hparams
        self.hp = hp
        self.denoise_fn = denoise_fn
        self.fs = self.fs_cls(len(phone_encoder), hp)  # only len() is used, spec_denoiser.py:21
        self.mel_encoder = MelEncoder(hidden_size=self.fs.hidden_size)
        self.mel_bins = out_dims
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else betas
        else:
            betas = get_noise_schedule_list(schedule_mode=hp["schedule_type"], timesteps=timesteps + 1,
                                            min_beta=0.1, max_beta=40, s=0.008)  # spec_denoiser.py:29-35
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.time_scale = time_scale
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        f32 = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("timesteps", f32(self.num_timesteps))
        self.register_buffer("timescale", f32(self.time_scale))
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(ac_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - ac)))
        self.register_buffer("log_one_minus_alphas_cumprod", f32(np.log(1.0 - ac)))
        self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1.0 / ac)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1.0 / ac - 1)))
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.register_buffer("posterior_variance", f32(pv))
        self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(pv, 1e-20))))
        self.register_buffer("posterior_mean_coef1", f32(betas * np.sqrt(ac_prev) / (1.0 - ac)))
        self.register_buffer("posterior_mean_coef2", f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)))
        keep = hp["keep_bins"]
        self.register_buffer("spec_min", torch.FloatTensor(spec_min)[None, None, :keep])
        self.register_buffer("spec_max", torch.FloatTensor(spec_max)[None, None, :keep])

    # ---- diffusion pieces -------------------------------------------------------------------------------
    def posterior_coef(self, t):
        """[n,4] rows {c1, c2, logvar, nonzero} for step ids t (int64 [n])  (spec_denoiser.py:86-101)."""
        c1 = self.posterior_mean_coef1[t]
        c2 = self.posterior_mean_coef2[t]
        lv = self.posterior_log_variance_clipped[t]
        nz = (t != 0).to(torch.float32)
        return torch.stack([c1, c2, lv, nz], dim=-1).contiguous()

    def q_posterior_sample(self, x_start, x_t, t, noise=None, seed=0, offset=0):
        """spec_denoiser.py:95-101.  x_start/x_t [B,1,M,T]; explicit `noise` or on-device Philox."""
        out = torch.empty_like(x_t)
        ops.posterior_step(x_start.contiguous(), x_t.contiguous(), self.posterior_coef(t), eps=noise, out=out,
                           seed=seed, offset=offset)
        return out

    @torch.no_grad()
    def p_sample(self, x_t, t, cond, noise=None):
        """spec_denoiser.py:103-108: x0 = denoise_fn(x_t, t, cond) (no clamp) -> posterior sample."""
        x0 = self.denoise_fn(x_t, t, cond)
        return self.q_posterior_sample(x0.contiguous(), x_t, t, noise)

    def q_sample(self, x_start, t, noise):
        """spec_denoiser.py:126-132."""
        ab = torch.stack([self.sqrt_alphas_cumprod[t], self.sqrt_one_minus_alphas_cumprod[t]], dim=-1).contiguous()
        return ops.q_sample(x_start.contiguous(), noise.contiguous(), ab)

    def norm_spec(self, x):
        return x

    def denorm_spec(self, x):
        return x

    # ---- conditioner --------------------------------------------------------------------------------------
    def conditioner(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, infer=False,
                    use_pred_mel2ph=False, use_pred_pitch=False, dropout_seed=0):
        """spec_denoiser.py:159-167.  Returns (ret, cond [B,H,T])."""
        F = autograd_ops if torch.is_grad_enabled() else ops
        ret = self.fs(txt_tokens, time_mel_masks, mel2ph, spk_embed, f0, uv, None, skip_decoder=True, infer=infer,
                      use_pred_mel2ph=use_pred_mel2ph, use_pred_pitch=use_pred_pitch, dropout_seed=dropout_seed)
        B, T, M = ref_mels.shape
        tmask = time_mel_masks.reshape(B, T).contiguous()
        masked = ops.mul_one_minus_mask(ref_mels.contiguous(), tmask, M)  # ref_mels*(1-mask)
        cond = self.mel_encoder.run(ops.btc_to_bct(masked), res=ret.pop("decoder_inp_bct"), mask=ret["tgt_nonpad"])
        ret["decoder_inp"] = F.bct_to_btc(cond)  # [B,T,H] as in the reference ret dict
        return ret, cond

    # ---- reference forward --------------------------------------------------------------------------------
    def forward(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, energy=None, infer=False,
                use_pred_mel2ph=False, use_pred_pitch=False, *, noises=None, t=None, seed=None,
                want_layer_spans=False, n_groups=None, persistent=None):
        """Keyword-only extras (not in the reference): `noises` = explicit [steps+1,B,1,M,T] noise stack
        (x_T then one eps per executed step) for parity runs; `t` = explicit training step ids; `seed` for
        the on-device Philox stream; `want_layer_spans` returns per-step layer-span timings in ret.
        infer=True always runs without a tape (the reference's p_sample is @torch.no_grad()); infer=False builds
        an autograd tape over the differentiable kernels when gradients are enabled."""
        if infer:
            with torch.no_grad():
                return self._forward_infer(txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv,
                                           use_pred_mel2ph, use_pred_pitch, noises, seed, want_layer_spans, n_groups,
                                           persistent)
        return self._forward_train(txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, noises, t, seed)

    def _forward_train(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, noises, t, seed):
        """spec_denoiser.py:168-176: t ~ U{0..steps}, x_t = q_sample(ref) * nonpad, x0 = denoise_fn(x_t, t, cond) * nonpad."""
        F = autograd_ops if torch.is_grad_enabled() else ops
        seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        ret, cond = self.conditioner(txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, False,
                                     dropout_seed=seed)
        tgt_nonpad = ret.pop("tgt_nonpad")
        B, H, T = cond.shape
        M = self.mel_bins
        dev = cond.device
        if t is None:
            t = torch.randint(0, self.num_timesteps + 1, (B,), device=dev).long()
        x_start = ops.btc_to_bct(ref_mels.contiguous())  # [B,M,T]
        eps = noises if noises is not None else ops.randn((B, M, T), dev, seed, 1 << 40)
        ab = torch.stack([self.sqrt_alphas_cumprod[t], self.sqrt_one_minus_alphas_cumprod[t]], -1).contiguous()
        x_t = ops.q_sample(x_start, eps.reshape(B, M, T).contiguous(), ab, nonpad=tgt_nonpad)
        x0 = self.denoise_fn(x_t[:, None], t, cond)[:, 0]
        x0 = F.add_chan_mask(x0.contiguous(), None, tgt_nonpad)
        ret["mel_out_bct"] = x0
        ret["mel_out"] = F.bct_to_btc(x0)
        ret["x_t"] = x_t
        ret["t"] = t
        return ret

    def _forward_infer(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, use_pred_mel2ph,
                       use_pred_pitch, noises, seed, want_layer_spans, n_groups, persistent):
        ret, cond = self.conditioner(txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, True,
                                     use_pred_mel2ph, use_pred_pitch)
        ret.pop("tgt_nonpad")
        ret.pop("pitch_pred_bct", None)
        B, H, T = cond.shape
        M = self.mel_bins
        dev = cond.device
        dn = self.denoise_fn
        steps = self.num_timesteps
        seed = int(seed) if seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        if noises is not None:
            noises = noises.reshape(steps + 1, B, M, T).contiguous()
            x = noises[0].clone()
            eps = noises[1:]
        else:
            x = ops.randn((B, M, T), dev, seed, 0)  # spec_denoiser.py:180
            eps = None
        ids = torch.arange(steps, device=dev)
        coef4 = self.posterior_coef(ids)
        dtab = dn.step_table_all(steps, dev)  # [L*C, steps]; cached while the weights it depends on are unchanged
        # opt-in bf16-operand loop (ops.set_compute_dtype("bf16")): NOT the parity path -- see DESIGN.md section 3.5
        bf16 = None
        if ops.compute_dtype() == "bf16" and dn.use_fused() and dn.encoder_hidden == 192 and T >= 32:
            bf16 = dict(cond=cond.contiguous(), imgs=dn.bf16_layer_images(),
                        b_cond=torch.stack([l.conditioner_projection.bias for l in dn.residual_layers]).contiguous())
        condproj = dn.cond_projections(cond) if bf16 is None else None  # hoisted: independent of the step
        if dn.use_fused():
            def loop():
                return ops.diffusion_loop(
                    x=x, noise=eps, seed=seed, condproj=condproj, dstep=dtab, coef4=coef4, bf16=bf16,
                    w_in=dn._w_in, b_in=dn.input_projection.bias,
                    packs=dn.fused_packs(),
                    w_skip=dn._w_skip, b_skip=dn.skip_projection.bias,
                    w_outp=dn._w_outp, b_outp=dn.output_projection.bias,
                    L=dn.n_layers, steps=steps, dilation_cycle_length=dn.dilation_cycle_length,
                    want_layer_spans=want_layer_spans, n_groups=n_groups, persistent=persistent)
            x_T = x.clone()  # [B, 80, T]: the loop updates x in place
            try:
                spans = loop()
            except ops.SplitRangeError:
                # an activation beyond the fp16 range of the two-piece splitting (never seen with trained weights; the
                # kernel flags it instead of overflowing): repeat the loop from x_T on the three-piece bf16 splitting,
                # which has fp32's range.  Same noise (explicit, or the same Philox stream), so nothing else changes.
                import warnings
                warnings.warn("spec_denoiser: activation outside the fp16 split range; repeating the reverse loop with the "
                              "bf16x3 splitting")
                x.copy_(x_T)
                with ops.split_operand_mode_as(3):
                    spans = loop()
            if spans is not None:
                ret.update(spans)
        else:
            quads = (B * M * T + 3) // 4
            for k, i in enumerate(reversed(range(steps))):  # spec_denoiser.py:181-182
                x0 = dn.denoise(x, condproj, dtab, i, False)
                ops.posterior_step(x0, x, coef4[i:i + 1], eps=None if eps is None else eps[k], out=x, seed=seed,
                                   offset=(k + 1) * quads)
        ret["mel_out"] = ops.bct_to_btc(x)  # x[:, 0].transpose(1, 2)
        ret["cond"] = cond
        return ret


class GaussianDiffusionNormal(GaussianDiffusion):
    """modules/speech_editing/spec_denoiser/spec_denoiser_normal.py:16-184 -- the `wo_masked_predictor` ablation
    (egs/spec_denoiser_wo_masked_predictor.yaml:50).  Same diffusion, same DiffNet; the conditioner is the plain
    FastSpeech of modules/tts/fs.py.

    The reference calls it positionally, `self.fs(txt_tokens, mel2ph, spk_embed, f0, uv, energy, ...)` (:158-159),
    against the signature `forward(txt_tokens, mel2ph, spk_embed, spk_id, f0, uv, ...)` (modules/tts/fs.py:81-82):
    the caller's f0 lands in `spk_id` (unused without use_spk_id), the caller's UV FLAGS land in `f0` and `uv` is None.
    The pitch embedding therefore sees f0_to_coarse(clamp(2**uv, 50, 900)) = bin 1 on every non-padded frame.  A
    drop-in has to produce the same numbers from the same checkpoint, so that binding is kept, not repaired."""
    fs_cls = FastSpeechNormal

    def conditioner(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, infer=False,
                    use_pred_mel2ph=False, use_pred_pitch=False, dropout_seed=0):
        if use_pred_mel2ph or use_pred_pitch:
            raise TypeError("spec_denoiser_normal.GaussianDiffusion.forward has no use_pred_mel2ph / use_pred_pitch")
        F = autograd_ops if torch.is_grad_enabled() else ops
        ret = self.fs(txt_tokens, mel2ph, spk_embed, f0, uv, None, skip_decoder=True, infer=infer,
                      dropout_seed=dropout_seed)
        B, T, M = ref_mels.shape
        tmask = time_mel_masks.reshape(B, T).contiguous()
        masked = ops.mul_one_minus_mask(ref_mels.contiguous(), tmask, M)
        cond = self.mel_encoder.run(ops.btc_to_bct(masked), res=ret.pop("decoder_inp_bct"), mask=ret["tgt_nonpad"])
        ret["decoder_inp"] = F.bct_to_btc(cond)
        return ret, cond
