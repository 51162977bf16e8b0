"""Generate tests/golden/*.npz from the upstream reference (dev container only).

TEST INFRASTRUCTURE.  Imports /root/reference through oracle/ref_import.py,
loads numpy-seeded weights (oracle/weights.py) into the reference modules,
runs them with explicit noise, records inputs/outputs, and cross-checks the
oracle restatement (oracle/oracle.py) against the reference on every case.

    python oracle/make_golden.py            # regenerate everything

Nothing from /root/reference is copied: the fixtures hold only tensors.
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import ref_import, weights as Wt  # noqa: E402

GOLD = Wt.GOLDEN_DIR
torch.set_grad_enabled(False)


@contextlib.contextmanager
def patched_randn(noises):
    """torch.randn pops from `noises` (spec_denoiser.py:180, diffusion_utils.py:65-68)."""
    queue = list(noises)
    real = torch.randn

    def fake(*shape, **kw):
        t = queue.pop(0)
        shp = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        assert tuple(t.shape) == tuple(shp), (t.shape, shp)
        return t.clone()

    torch.randn = fake
    try:
        yield queue
    finally:
        torch.randn = real


VARIANT = "masked"  # "normal": modules/speech_editing/spec_denoiser/spec_denoiser_normal.py (wo_masked_predictor)


def build_ref_model(hp, timesteps, overrides=None):
    hp["timesteps"] = timesteps
    for k, v in (overrides or {}).items():
        hp[k] = v
    if VARIANT == "normal":
        from modules.speech_editing.spec_denoiser import spec_denoiser_normal as SD
    else:
        from modules.speech_editing.spec_denoiser import spec_denoiser as SD
    from modules.speech_editing.spec_denoiser.diffnet import DiffNet
    SD.tqdm = lambda it, **kw: it
    m = SD.GaussianDiffusion(list(range(80)), 80, DiffNet(80), timesteps=timesteps, time_scale=1,
                             loss_type="l1", spec_min=[], spec_max=[])
    m.eval()
    return m


def manifest_of(module):
    return [(k, list(v.shape)) for k, v in module.state_dict().items()]


def load_seeded(module, seed):
    man = [(k, tuple(s)) for k, s in manifest_of(module)]
    W = Wt.seeded_weights(man, seed)
    missing, unexpected = module.load_state_dict(W, strict=False)
    assert not unexpected, unexpected
    assert all(Wt.is_buffer(k) for k in missing), missing
    return W


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def infer_case(hp, name, B, T, T_txt, steps, wseed, iseed, pad_tail=False, overrides=None,
               trace_layers=(), flags=None, keep_steps=None, light_stride=0):
    """light_stride > 0 (the metric-shaped case, T = 800): the fixture keeps mel_out, the integer tensors and every
    `light_stride`-th frame of the kept x0 / x traces -- not decoder_inp / the float pitch tensors (1.6 MB at T = 800)."""
    flags = flags or {}
    model = build_ref_model(hp, steps, overrides)
    W = load_seeded(model, wseed)
    inp = Wt.synthetic_inputs(B, T, T_txt, seed=iseed, pad_tail=pad_tail)
    noises = Wt.synthetic_noises(B, T, steps, seed=iseed + 1)
    rec = {"x0": [], "x": [], "layers": {}}
    hooks = []
    step_ctr = {"n": 0}

    def dn_hook(mod, args, out):
        rec["x0"].append(out.clone())
        step_ctr["n"] += 1

    hooks.append(model.denoise_fn.register_forward_hook(dn_hook))
    # integer index tensors the reference feeds to its embedding tables (fs.py:123-189): captured AT the tables, so the
    # fixtures hold the reference's own indices (dur_embed: masked durations; pitch_embed: 1st call masked-pitch bins,
    # 2nd call the full pitch bins; the `normal` variant has no dur_embed and calls pitch_embed once)
    cap = {"dur": [], "pitch": []}
    if hasattr(model.fs, "dur_embed"):
        hooks.append(model.fs.dur_embed.register_forward_pre_hook(lambda m, a: cap["dur"].append(a[0].clone())))
    if bool(hp["use_pitch_embed"]):
        hooks.append(model.fs.pitch_embed.register_forward_pre_hook(lambda m, a: cap["pitch"].append(a[0].clone())))
    for li in trace_layers:
        def mk(li):
            def h(mod, args, out):
                if step_ctr["n"] == 0:  # first executed step only
                    rec["layers"][li] = (out[0].clone(), out[1].clone())
            return h
        hooks.append(model.denoise_fn.residual_layers[li].register_forward_hook(mk(li)))
    orig_qps = model.q_posterior_sample

    def qps(x_start, x_t, t, repeat_noise=False):
        r = orig_qps(x_start=x_start, x_t=x_t, t=t, repeat_noise=repeat_noise)
        rec["x"].append(r.clone())
        return r

    model.q_posterior_sample = qps
    with patched_randn(noises) as q:
        ret = model(inp["txt_tokens"], inp["time_mel_masks"].clone(), inp["mel2ph"].clone(), inp["spk_embed"],
                    inp["ref_mels"], inp["f0"].clone(), inp["uv"].clone(), infer=True, **flags)
        assert len(q) == 0
    for h in hooks:
        h.remove()
    # ---- oracle cross-check
    dcl = hp["dilation_cycle_length"]
    otrace = []
    use_pitch = bool(hp["use_pitch_embed"])
    oflags = dict(flags) if use_pitch else dict(flags, use_pitch_embed=False)
    if VARIANT != "masked":
        oflags["variant"] = VARIANT
    oret = O.gaussian_diffusion_infer(W, steps, inp, noises, dilation_cycle_length=dcl, trace=otrace, **oflags)
    d_mel = maxdiff(oret["mel_out"], ret["mel_out"])
    d_cond = maxdiff(oret["decoder_inp"], ret["decoder_inp"])
    d_dur = maxdiff(oret["dur"], ret["dur"])
    d_pp = maxdiff(oret["pitch_pred"], ret["pitch_pred"]) if use_pitch else 0.0
    assert use_pitch or "pitch_pred" not in ret
    assert torch.equal(oret["mel2ph"], ret["mel2ph"])
    print("  [%s] oracle vs reference: mel %.2e cond %.2e dur %.2e pitch_pred %.2e" % (name, d_mel, d_cond, d_dur, d_pp))
    assert max(d_mel, d_cond, d_dur, d_pp) < 2e-5, "oracle restatement deviates from the reference"
    for k in range(steps):
        assert maxdiff(otrace[k][0], rec["x0"][k]) < 2e-5 and maxdiff(otrace[k][1], rec["x"][k]) < 2e-5
    # integer intermediates: the reference's own tensors (hooks above); the oracle must agree BIT FOR BIT
    ref_int = {}
    if VARIANT == "masked":
        assert len(cap["dur"]) == 1 and cap["dur"][0].dtype == torch.int64
        ref_int["masked_dur"] = cap["dur"][0]
    if use_pitch:
        assert len(cap["pitch"]) == (2 if VARIANT == "masked" else 1) and cap["pitch"][-1].dtype == torch.int64
        ref_int["pitch"] = cap["pitch"][-1]
        if VARIANT == "masked":
            ref_int["masked_pitch"] = cap["pitch"][0]
    for k, v in ref_int.items():
        assert torch.equal(oret[k], v), "oracle %s differs from the reference's embedding index tensor" % k
    out = dict(
        meta=np.array(json.dumps(dict(B=B, T=T, T_txt=T_txt, steps=steps, wseed=wseed, iseed=iseed,
                                      pad_tail=pad_tail, overrides=overrides or {}, flags=flags,
                                      trace_layers=list(trace_layers), variant=VARIANT, light_stride=light_stride))),
        mel_out=ret["mel_out"], decoder_inp=ret["decoder_inp"], dur=ret["dur"], mel2ph=ret["mel2ph"],
    )
    if light_stride:
        del out["decoder_inp"]
    if VARIANT == "masked":
        out["masked_dur"] = ref_int["masked_dur"]
    if use_pitch:
        out.update(pitch_pred=ret["pitch_pred"], f0_denorm=ret["f0_denorm"], f0_denorm_pred=ret["f0_denorm_pred"],
                   pitch=ref_int["pitch"])
        if light_stride:
            del out["f0_denorm"], out["f0_denorm_pred"]
        if VARIANT == "masked":
            out["masked_pitch"] = ref_int["masked_pitch"]
    ks = range(steps) if keep_steps is None else keep_steps
    for k in ks:
        if light_stride:  # [B, 1, M, T] -> every light_stride-th frame
            out["x0_step%d" % k] = rec["x0"][k][..., ::light_stride]
            out["x_step%d" % k] = rec["x"][k][..., ::light_stride]
            continue
        out["x0_step%d" % k] = rec["x0"][k]
        out["x_step%d" % k] = rec["x"][k]
    for li, (xl, sl) in rec["layers"].items():
        out["layer%d_x" % li] = xl
        out["layer%d_skip" % li] = sl
    npz(name, **out)
    return model


def train_case(hp, name, B, T, T_txt, steps, wseed, iseed):
    model = build_ref_model(hp, steps)
    W = load_seeded(model, wseed)
    inp = Wt.synthetic_inputs(B, T, T_txt, seed=iseed, pad_tail=True)
    rng = np.random.default_rng(iseed + 7)
    t = torch.from_numpy(rng.integers(0, steps + 1, size=(B,), dtype=np.int64))
    eps = torch.from_numpy(rng.standard_normal(size=(B, 1, 80, T), dtype=np.float32))
    real_randint, real_rl = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda x, **k: eps.clone()
    try:
        ret = model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"],
                    inp["ref_mels"], inp["f0"].clone(), inp["uv"].clone(), infer=False)
    finally:
        torch.randint, torch.randn_like = real_randint, real_rl
    oret = O.gaussian_diffusion_train(W, steps, inp, t, eps)
    d = maxdiff(oret["mel_out"], ret["mel_out"])
    print("  [%s] oracle vs reference (train branch): mel %.2e" % (name, d))
    assert d < 2e-5
    npz(name, meta=np.array(json.dumps(dict(B=B, T=T, T_txt=T_txt, steps=steps, wseed=wseed, iseed=iseed,
                                            pad_tail=True))),
        t=t, eps=eps, mel_out=ret["mel_out"], x_t=oret["x_t"], decoder_inp=ret["decoder_inp"])


def train_loss_case(hp, name, B, T, T_txt, steps, wseed, iseed):
    """Reference forward (infer=False, eval mode) + the reference's own loss functions + autograd."""
    from tasks.tts.speech_base import SpeechBaseTask
    from tasks.speech_editing.speech_editing_base import SpeechEditingBaseTask

    class FakeTask:  # the loss methods only touch these attributes of `self`
        l1_loss = SpeechBaseTask.l1_loss
        ssim_loss = SpeechBaseTask.ssim_loss
        add_mel_loss = SpeechBaseTask.add_mel_loss
        add_dur_loss = SpeechEditingBaseTask.add_dur_loss
        add_pitch_loss = SpeechEditingBaseTask.add_pitch_loss

    task = FakeTask()
    task.mel_losses = {"l1": 0.5, "ssim": 0.5}
    task.sil_ph = [1, 2, 3]
    task.token_encoder = type("Enc", (), {"encode": staticmethod(lambda p: [p])})()
    model = build_ref_model(hp, steps)
    W = load_seeded(model, wseed)
    inp = Wt.synthetic_inputs(B, T, T_txt, seed=iseed, pad_tail=True)
    rng = np.random.default_rng(iseed + 7)
    t = torch.from_numpy(rng.integers(0, steps + 1, size=(B,), dtype=np.int64))
    eps = torch.from_numpy(rng.standard_normal(size=(B, 1, 80, T), dtype=np.float32))
    real_randint, real_rl = torch.randint, torch.randn_like
    torch.randint = lambda *a, **k: t.clone()
    torch.randn_like = lambda x, **k: eps.clone()
    try:
        with torch.enable_grad():
            out = model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"],
                        inp["f0"].clone(), inp["uv"].clone(), infer=False)
            tm = inp["time_mel_masks"]
            losses = {}
            task.add_mel_loss(out["mel_out"] * tm, inp["ref_mels"] * tm, losses, postfix="_coarse")
            task.add_dur_loss(out["dur"], inp["mel2ph"], inp["txt_tokens"], losses=losses)
            if hp["use_pitch_embed"]:  # tasks/speech_editing/spec_denoiser.py:55-56
                task.add_pitch_loss(out, {"mel2ph": inp["mel2ph"], "f0": inp["f0"], "uv": inp["uv"]}, losses)
            total = sum(losses.values())
            total.backward()
    finally:
        torch.randint, torch.randn_like = real_randint, real_rl
    grads = {k: p.grad for k, p in model.named_parameters()}
    # ---- oracle cross-check (values and every gradient)
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    with torch.enable_grad():
        olosses, _ = O.training_losses(Wg, steps, inp, t, eps, use_pitch_embed=bool(hp["use_pitch_embed"]),
                                       variant=VARIANT)
        sum(olosses.values()).backward()
    for k in losses:
        assert np.isfinite(float(losses[k])), (k, "pick another input seed")
        assert abs(float(losses[k]) - float(olosses[k])) < 1e-5 * max(1.0, abs(float(losses[k]))), k
    worst = 0.0
    for k, g in grads.items():
        og = Wg[k].grad
        assert (g is None) == (og is None), k
        if g is not None:
            worst = max(worst, float((g - og).abs().max()) / (float(g.abs().max()) + 1e-12))
    print("  [%s] oracle vs reference: losses ok; worst relative grad deviation %.2e" % (name, worst))
    assert worst < 1e-3
    keep = ["denoise_fn.output_projection.bias", "denoise_fn.residual_layers.3.diffusion_projection.bias",
            "denoise_fn.residual_layers.19.dilated_conv.bias", "denoise_fn.mlp.0.bias",
            "fs.dur_predictor.linear.0.weight", "fs.pitch_predictor.linear.bias", "fs.encoder.embed_tokens.weight",
            "fs.spk_embed_proj.weight", "mel_encoder.fc_out.bias", "fs.encoder.res_blocks.0.blocks.0.0.weight",
            "fs.dur_embed.weight", "fs.pitch_embed.weight"]
    keep = [k for k in keep if k in grads]
    names = [k for k, _ in model.named_parameters()]
    norms = np.array([float(grads[k].norm()) if grads[k] is not None else -1.0 for k in names], dtype=np.float64)
    out_np = dict(meta=np.array(json.dumps(dict(B=B, T=T, T_txt=T_txt, steps=steps, wseed=wseed, iseed=iseed,
                                                pad_tail=True, sil_ids=[1, 2, 3], param_names=names,
                                                variant=VARIANT))),
                  t=t, eps=eps, grad_norms=norms, total=total.detach(),
                  **{"loss_" + k: v.detach() for k, v in losses.items()})
    for k in keep:
        g = grads[k]
        if k in ("fs.dur_embed.weight", "fs.pitch_embed.weight"):
            g = g[:64]
        out_np["grad::" + k] = g
    npz(name, **out_np)


def data_feed_case():
    """Host-side data feed: f0 normalisation, editing masks under fixed seeds, and a 3-item
    binarised test set written with the REFERENCE IndexedDatasetBuilder (tests/golden/binary_tiny/)."""
    import random
    from utils.audio.pitch.utils import norm_interp_f0 as ref_nif
    from utils.spec_aug import time_mask as rtm
    from utils.commons.indexed_datasets import IndexedDatasetBuilder as RefBuilder
    rng = np.random.default_rng(5)
    f0 = rng.uniform(80, 400, size=120)
    f0[rng.uniform(size=120) < 0.3] = 0
    f0[:4] = 0
    f0[-3:] = 0
    a, au = ref_nif(f0.copy())
    mel2ph = torch.sort(torch.randint(1, 31, (200,), generator=torch.Generator().manual_seed(3))).values
    spec = torch.zeros(200, 80)
    random.seed(7)
    m1 = rtm.generate_inference_mask(spec, mel2ph, ratio=0.5)
    np.random.seed(11)
    a1 = rtm.generate_alignment_aware_time_mask(spec, mel2ph, ratio=0.8)
    torch.manual_seed(13)
    t1 = rtm.generate_time_mask(spec, ratio=0.3)
    d = os.path.join(GOLD, "binary_tiny")
    os.makedirs(d, exist_ok=True)
    bld = RefBuilder(os.path.join(d, "test"))
    for i, T in enumerate((40, 56, 33)):
        Tt = 8 + i
        m2p = np.sort(rng.integers(1, Tt + 1, size=T)).astype(np.int64)
        f0i = rng.uniform(90, 300, size=T).astype(np.float32)
        f0i[rng.uniform(size=T) < 0.25] = 0
        bld.add_item(dict(item_name="utt%d" % i, txt="hello world %d" % i,
                          ph_token=rng.integers(1, 80, size=Tt).astype(np.int64),
                          mel=np.clip(rng.normal(-3, 1.5, size=(T, 80)), -6, 1.5).astype(np.float32), mel2ph=m2p,
                          f0=f0i, pitch=rng.integers(1, 255, size=T).astype(np.int64),
                          spk_embed=(rng.standard_normal(256) / 16).astype(np.float32), wav_fn="dummy/utt%d.wav" % i))
    bld.finalize()
    np.save(os.path.join(d, "test_lengths.npy"), np.array([40, 56, 33]))
    npz("data_feed", f0=f0, f0_norm=np.asarray(a, dtype=np.float32), uv=np.asarray(au, dtype=np.float32),
        mel2ph=mel2ph, infer_mask_seed7=m1, align_mask_seed11=a1, time_mask_seed13=t1)


def schedule_case(hp):
    out = {}
    for steps in (4, 8, 100):
        m = build_ref_model(hp, steps)
        tab, tab64 = O.diffusion_tables(steps)
        for k in tab:
            ref = getattr(m, k)
            assert torch.equal(ref, tab[k]), (steps, k)
            out["s%d_%s" % (steps, k)] = ref
        out["s%d_betas64" % steps] = tab64["betas"]
    npz("schedule", **out)


def length_regulator_case():
    from modules.commons.nar_tts_modules import LengthRegulator
    rng = np.random.default_rng(99)
    dur = torch.from_numpy(rng.uniform(0, 6, size=(3, 12)).astype(np.float32))
    pad = torch.zeros(3, 12, dtype=torch.bool)
    pad[1, 9:] = True
    pad[2, 5:] = True
    ref = LengthRegulator()(dur, pad)
    mine = O.length_regulator(dur, pad)
    assert torch.equal(ref, mine)
    npz("length_regulator", dur=dur, pad=pad, mel2ph=ref)


def hifigan_case(name, h, B, T, wseed, iseed, manifest=None):
    from modules.vocoder.hifigan.hifigan import HifiGanGenerator
    g = HifiGanGenerator(h).eval()
    man = manifest_of(g)
    W = Wt.seeded_weights([(k, tuple(s)) for k, s in man], wseed)
    g.load_state_dict(W, strict=True)
    if manifest is None:
        with open(os.path.join(GOLD, "manifest_%s.json" % name), "w") as f:
            json.dump(man, f)
    rng = np.random.default_rng(iseed)
    mel = torch.from_numpy(np.clip(rng.normal(-3.0, 1.5, size=(B, 80, T)), -6.0, 1.5).astype(np.float32))
    wav = g(mel)
    mine = O.hifigan_forward(W, h, mel)
    d = maxdiff(wav, mine)
    print("  [%s] oracle vs reference: wav %.2e (|wav| max %.3f)" % (name, d, float(wav.abs().max())))
    assert d < 2e-5
    npz(name, meta=np.array(json.dumps(dict(h=h, B=B, T=T, wseed=wseed, iseed=iseed, manifest=manifest or name))),
        mel=mel, wav=wav)


def hifigan_v23_cases():
    """The V2 (C0 = 128) and V3 (ResBlock2, rates [8, 8, 4], kernels [3, 5, 7], dilations [[1, 2], [2, 6], [3, 12]]) generator shapes of
    the HiFi-GAN paper at T = 200 (51,200 samples: many tiles per stage, every halo crosses tile borders)."""
    hifigan_case("hifigan_v2", Wt.HIFIGAN_V2, B=1, T=200, wseed=24, iseed=205)
    hifigan_case("hifigan_v3", Wt.HIFIGAN_V3, B=1, T=200, wseed=25, iseed=206)


def edit_case(hp, name, seed, steps, wseed, vseed, h, **gen):
    """The reference's SpecDenoiserInfer.forward_model (inference/tts/spec_denoiser.py:63-149) on a synthetic edit
    request: constructor bypassed (it needs dictionaries, a speaker encoder and checkpoints), `input_to_batch`
    replaced by the identity, torch.randn replaced by a recording seeded generator (T_new is only known after the
    duration predictor ran)."""
    from modules.vocoder.hifigan.hifigan import HifiGanGenerator
    SDI = ref_import.import_spec_denoiser_infer()
    model = build_ref_model(hp, steps, dict(residual_layers=20, residual_channels=256, dilation_cycle_length=1))
    W = load_seeded(model, wseed)
    voc = HifiGanGenerator(h).eval()
    Wv = Wt.seeded_weights([(k, tuple(s)) for k, s in manifest_of(voc)], vseed)
    voc.load_state_dict(Wv, strict=True)
    inf = object.__new__(SDI.SpecDenoiserInfer)
    inf.hparams, inf.device, inf.model, inf.vocoder = hp, "cpu", model, voc
    inf.input_to_batch = lambda item: item
    sample = Wt.synthetic_edit_sample(seed, **gen)
    gen_ = torch.Generator().manual_seed(seed + 1)
    rec = []
    real = torch.randn

    def fake(*shape, **kw):
        shp = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        t = real(tuple(shp), generator=gen_)
        rec.append(t.clone())
        return t

    ref_in = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()}
    torch.randn = fake
    try:
        wav_out, wav_gt, mel_out, mel_gt, masked_mel_out, masked_mel_gt = inf.forward_model(ref_in)
    finally:
        torch.randn = real
    assert len(rec) == steps + 1
    o = O.edit_forward_model(W, Wv, h, steps, sample, rec)
    T_new = mel_out.shape[0]
    assert o["edited_mel2ph"].shape[1] == T_new
    d = {k: float(np.abs(o[k].numpy() - v).max()) for k, v in
         (("wav_out", wav_out), ("wav_gt", wav_gt), ("mel_out", mel_out), ("mel_gt", mel_gt),
          ("masked_mel_out", masked_mel_out), ("masked_mel_gt", masked_mel_gt))}
    print("  [%s] T %d -> %d, head %d tail %d; oracle vs reference: %s" % (
        name, sample["mel"].shape[1], T_new, o["head_idx"], o["tail_idx"],
        " ".join("%s %.1e" % kv for kv in d.items())))
    assert max(d.values()) < 2e-5, "oracle restatement deviates from the reference"
    assert np.array_equal(masked_mel_out, o["masked_mel_out"].numpy())  # pure splicing: bit exact
    npz(name, meta=np.array(json.dumps(dict(seed=seed, steps=steps, wseed=wseed, vseed=vseed, h=h, gen=gen))),
        wav_out=wav_out, wav_gt=wav_gt, mel_out=mel_out, masked_mel_out=masked_mel_out, masked_mel_gt=masked_mel_gt,
        noises=torch.stack(rec), edited_mel2ph=o["edited_mel2ph"], pred_mel2ph=o["pred_mel2ph"],
        masked_dur=o["masked_dur"], dur_pred=o["dur_pred"], edited_f0=o["edited_f0"], edited_uv=o["edited_uv"],
        time_mel_masks=o["time_mel_masks"], pitch=o["pitch"], head_tail=np.array([o["head_idx"], o["tail_idx"]]))


def main():
    os.makedirs(GOLD, exist_ok=True)
    hp = ref_import.install(timesteps=4)
    base = dict(residual_layers=20, residual_channels=256, dilation_cycle_length=1)
    schedule_case(hp)
    m = build_ref_model(hp, 4, base)
    with open(os.path.join(GOLD, "manifest_spec_denoiser.json"), "w") as f:
        json.dump(manifest_of(m), f)
    infer_case(hp, "infer_tiny", B=2, T=64, T_txt=16, steps=4, wseed=11, iseed=101, overrides=base,
               trace_layers=(0, 1, 19))
    infer_case(hp, "infer_pad", B=3, T=96, T_txt=24, steps=8, wseed=12, iseed=102, pad_tail=True, overrides=base,
               keep_steps=(0, 7))
    infer_case(hp, "infer_drift100", B=1, T=64, T_txt=16, steps=100, wseed=13, iseed=103, overrides=base,
               keep_steps=(0, 99))
    infer_case(hp, "infer_predpitch", B=2, T=64, T_txt=16, steps=2, wseed=14, iseed=104, overrides=base,
               flags=dict(use_pred_pitch=True), keep_steps=(1,))
    ov = dict(residual_layers=6, residual_channels=256, dilation_cycle_length=4)
    m = build_ref_model(hp, 2, ov)
    with open(os.path.join(GOLD, "manifest_spec_denoiser_dil.json"), "w") as f:
        json.dump(manifest_of(m), f)
    infer_case(hp, "infer_dil", B=1, T=80, T_txt=20, steps=2, wseed=15, iseed=105, overrides=ov,
               trace_layers=(2, 3), keep_steps=(0, 1))
    ov = dict(residual_layers=3, residual_channels=64, dilation_cycle_length=2)
    m = build_ref_model(hp, 2, ov)
    with open(os.path.join(GOLD, "manifest_spec_denoiser_c64.json"), "w") as f:
        json.dump(manifest_of(m), f)
    infer_case(hp, "infer_c64", B=2, T=48, T_txt=12, steps=2, wseed=16, iseed=106, overrides=ov, keep_steps=(1,))
    hp.update(base)
    train_case(hp, "train_tiny", B=2, T=64, T_txt=16, steps=8, wseed=17, iseed=107)
    train_loss_case(hp, "train_losses", B=2, T=64, T_txt=16, steps=8, wseed=18, iseed=108)
    length_regulator_case()
    data_feed_case()
    batch_by_size_case()
    hifigan_case("hifigan_tiny", Wt.HIFIGAN_TINY, B=2, T=24, wseed=21, iseed=201)
    hifigan_case("hifigan_tiny_rb2", Wt.HIFIGAN_TINY_RB2, B=1, T=20, wseed=22, iseed=202)
    hifigan_case("hifigan_v1", Wt.HIFIGAN_V1, B=1, T=12, wseed=23, iseed=203)
    # real-length V1: 56,320 output samples, many tiles per stage, every dilation x kernel halo crosses tile borders
    hifigan_case("hifigan_v1_long", Wt.HIFIGAN_V1, B=1, T=220, wseed=23, iseed=204, manifest="hifigan_v1")
    hifigan_v23_cases()
    ragged_cases(hp)
    full800_case(hp)
    edit_cases(hp)
    nopitch_cases(hp)
    normal_cases(hp)


def ragged_cases(hp):
    """Round 5: a whole-model case whose sizes are multiples of nothing -- T = 77 frames, T_txt = 19 tokens, three utterances with padded tails
    of different lengths, the full 20-layer stack: every other full-model fixture has T in {48, 64, 80, 96}."""
    base = dict(residual_layers=20, residual_channels=256, dilation_cycle_length=1)
    hp.update(base)
    infer_case(hp, "infer_ragged", B=3, T=77, T_txt=19, steps=3, wseed=19, iseed=109, pad_tail=True, overrides=base, keep_steps=(0, 2))
    # and one shorter than every tile, halo and the 16-frame threshold of the MFMA convs: 7 frames, 3 tokens
    infer_case(hp, "infer_short", B=2, T=7, T_txt=3, steps=2, wseed=20, iseed=110, pad_tail=True, overrides=base, keep_steps=(0, 1))
    # training at the ragged sizes: losses + every gradient (weight gradients whose frame rows are not 16-byte aligned, LayerNorm / SSIM tile edges)
    train_loss_case(hp, "train_losses_ragged", B=3, T=77, T_txt=19, steps=8, wseed=24, iseed=112)


def full800_case(hp):
    """Round 6: the shape BASELINE.json's metric is quoted on, run by the reference itself -- T = 800 frames (13 tiles of 64 per utterance, so
    the inter-tile halo hand-off of the stack kernels is exercised over all 100 steps of spec_denoiser.py:178-184), T_txt = 100, 100 steps,
    B = 2 with one padded tail.  The GPU tests also embed the two utterances in a B = 32 batch (the throughput tiles) and run B = 1."""
    base = dict(residual_layers=20, residual_channels=256, dilation_cycle_length=1)
    hp.update(base)
    infer_case(hp, "infer_full800", B=2, T=800, T_txt=100, steps=100, wseed=25, iseed=113, pad_tail=True, overrides=base,
               keep_steps=(0, 50, 99), light_stride=8)


def nopitch_cases(hp):
    """egs/spec_denoiser_libritts.yaml: use_pitch_embed false (:169) -- no pitch_embed / pitch_predictor in the model
    (fs.py:73-78), no pitch block in the conditioner (fs.py:97-99), no uv / f0 losses (tasks/.../spec_denoiser.py:55)."""
    base = dict(residual_layers=20, residual_channels=256, dilation_cycle_length=1, use_pitch_embed=False)
    m = build_ref_model(hp, 4, base)
    with open(os.path.join(GOLD, "manifest_spec_denoiser_nopitch.json"), "w") as f:
        json.dump(manifest_of(m), f)
    infer_case(hp, "infer_nopitch", B=2, T=64, T_txt=16, steps=4, wseed=31, iseed=301, pad_tail=True, overrides=base,
               keep_steps=(0, 3))
    hp.update(base)
    train_loss_case(hp, "train_losses_nopitch", B=2, T=64, T_txt=16, steps=8, wseed=32, iseed=108)
    hp["use_pitch_embed"] = True


def normal_cases(hp):
    """egs/spec_denoiser_wo_masked_predictor.yaml: SpeechDenoiserNormalTask over spec_denoiser_normal.GaussianDiffusion
    (conditioner = modules/tts/fs.py FastSpeech, called with the positional binding of spec_denoiser_normal.py:158)."""
    global VARIANT
    VARIANT = "normal"
    try:
        base = dict(residual_layers=20, residual_channels=256, dilation_cycle_length=1, use_pitch_embed=True)
        m = build_ref_model(hp, 4, base)
        with open(os.path.join(GOLD, "manifest_spec_denoiser_normal.json"), "w") as f:
            json.dump(manifest_of(m), f)
        infer_case(hp, "infer_normal", B=2, T=64, T_txt=16, steps=4, wseed=41, iseed=401, pad_tail=True,
                   overrides=base, keep_steps=(0, 3))
        hp.update(base)
        train_loss_case(hp, "train_losses_normal", B=2, T=64, T_txt=16, steps=8, wseed=43, iseed=109)
    finally:
        VARIANT = "masked"


def build_ref_campnet(hp):
    import yaml
    with open(os.path.join(ref_import.REF_ROOT, "egs", "campnet.yaml")) as f:
        hp.update(yaml.safe_load(f))
    from modules.speech_editing.campnet.campnet import CampNet
    return CampNet(80, 100, hp).eval()


def campnet_case(hp, name, B, T, T_txt, wseed, iseed, pad_tail=True):
    """Reference CampNet forward (egs/campnet.yaml: dropout 0, so train == eval arithmetic) + the reference's own
    mel losses + autograd; oracle cross-check of outputs, losses and every parameter gradient."""
    from tasks.tts.speech_base import SpeechBaseTask

    class FakeTask:
        l1_loss = SpeechBaseTask.l1_loss
        ssim_loss = SpeechBaseTask.ssim_loss
        add_mel_loss = SpeechBaseTask.add_mel_loss

    task = FakeTask()
    task.mel_losses = {"l1": 0.5, "ssim": 0.5}
    model = build_ref_campnet(hp)
    man = manifest_of(model)
    with open(os.path.join(GOLD, "manifest_campnet.json"), "w") as f:
        json.dump(man, f)
    W = load_seeded(model, wseed)
    with torch.no_grad():  # the reference initialises these to exactly 0 / 1; give them real values
        rng = np.random.default_rng(wseed + 1)
        W["mask_emb"] = torch.from_numpy(rng.normal(0, 0.5, size=(1, 1, 80)).astype(np.float32))
        W["decoder_coarse.pos_embed_alpha"] = torch.tensor([0.7], dtype=torch.float32)
        model.load_state_dict(W, strict=False)
    inp = Wt.synthetic_inputs(B, T, T_txt, seed=iseed, pad_tail=pad_tail)
    if pad_tail:
        inp["txt_tokens"][0, -3:] = 0  # padded phonemes too (key padding masks of both attentions)
    txt, mels, tm = inp["txt_tokens"], inp["ref_mels"], inp["time_mel_masks"]
    with torch.enable_grad():
        out = model(txt, mels=mels, time_mel_masks=tm, infer=False, global_step=0)
        losses = {}
        task.add_mel_loss(out["mel_out_coarse"] * tm, mels * tm, losses, postfix="_coarse")
        task.add_mel_loss(out["mel_out_fine"] * tm, mels * tm, losses, postfix="_fine")
        total = sum(losses.values())
        total.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    Wg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in W.items()}
    with torch.enable_grad():
        olosses, oout = O.campnet_losses(Wg, txt, mels, tm)
        sum(olosses.values()).backward()
    d = {k: maxdiff(oout[k], out[k]) for k in ("mel_out_coarse", "mel_out_fine", "attn")}
    for k in losses:
        assert abs(float(losses[k]) - float(olosses[k])) < 1e-5 * max(1.0, abs(float(losses[k]))), k
    worst, n_none = 0.0, 0
    for k, g in grads.items():
        og = Wg[k].grad
        assert (g is None) == (og is None), k
        if g is None:
            n_none += 1
        else:
            worst = max(worst, float((g - og).abs().max()) / (float(g.abs().max()) + 1e-12))
    print("  [%s] oracle vs reference: %s; losses %s; worst rel grad dev %.2e (%d params without grad)" % (
        name, " ".join("%s %.1e" % kv for kv in d.items()),
        " ".join("%s=%.4f" % (k, float(v)) for k, v in losses.items()), worst, n_none))
    assert max(d.values()) < 2e-5 and worst < 1e-3
    names = [k for k, _ in model.named_parameters()]
    norms = np.array([float(grads[k].norm()) if grads[k] is not None else -1.0 for k in names], dtype=np.float64)
    keep = ["mask_emb", "decoder_coarse.pos_embed_alpha", "encoder.embed_tokens.weight",
            "encoder.layers.0.op.self_attn.in_proj_weight", "encoder.layers.2.op.ffn.ffn_2.bias",
            "decoder_coarse.layers.0.op.encoder_attn.in_proj_weight", "decoder_coarse.layers.5.op.self_attn.out_proj.weight",
            "decoder_coarse.layers.3.op.ffn.ffn_1.1.bias", "decoder_coarse.layer_norm.weight", "mel_out_coarse.weight",
            "mel_out_fine.weight", "mel_encoder.fc_out.bias", "decoder_fine.res_blocks.4.blocks.1.4.bias"]
    out_np = dict(meta=np.array(json.dumps(dict(B=B, T=T, T_txt=T_txt, wseed=wseed, iseed=iseed, pad_tail=pad_tail,
                                                param_names=names))),
                  mask_emb=W["mask_emb"], pos_embed_alpha=W["decoder_coarse.pos_embed_alpha"],
                  txt_tokens=txt, mel_out_coarse=out["mel_out_coarse"], mel_out_fine=out["mel_out_fine"],
                  attn=out["attn"], grad_norms=norms, total=total.detach(),
                  **{"loss_" + k: v.detach() for k, v in losses.items()})
    for k in keep:
        g = grads[k]
        out_np["grad::" + k] = g if g.numel() <= 30000 else g.reshape(-1)[:30000]
    npz(name, **out_np)


def batch_by_size_case():
    """utils/commons/dataset_utils.py:65-119 (`batch_by_size`) on random length lists: the reference's own batches."""
    import utils.commons.dataset_utils as RD
    from set_amd import data as D
    rng = np.random.default_rng(77)
    cases = []
    for n, mt, ms, mult in [(50, 3000, 8, 1), (37, 1500, 16, 2), (64, 40000, 16, 8), (23, 900, None, 4),
                            (10, None, 3, 1), (41, 2500, 6, 3)]:
        sizes = rng.integers(80, 800, size=n).tolist()
        order = np.argsort(np.array(sizes), kind="mergesort").tolist() if n % 2 else rng.permutation(n).tolist()
        ref = RD.batch_by_size(np.array(order), lambda i: sizes[i], max_tokens=mt, max_sentences=ms,
                               required_batch_size_multiple=mult)
        ref = [[int(i) for i in b] for b in ref]
        assert D.batch_by_size(order, lambda i: sizes[i], mt, ms, mult) == ref, (n, mt, ms, mult)
        cases.append(dict(sizes=sizes, order=order, max_tokens=mt, max_sentences=ms, mult=mult, batches=ref))
    with open(os.path.join(GOLD, "batch_by_size.json"), "w") as f:
        json.dump(cases, f)
    print("wrote batch_by_size.json", [len(c["batches"]) for c in cases])


def campnet_cases(hp):
    campnet_case(hp, "campnet_tiny", B=2, T=48, T_txt=12, wseed=41, iseed=301)
    campnet_case(hp, "campnet_ragged", B=3, T=77, T_txt=19, wseed=42, iseed=302)


def region_helper_cases():
    """inference/tts/infer_utils.py:29-53 (pure Python): input/output pairs of the reference's own functions."""
    from inference.tts.infer_utils import get_words_region_from_origintxt_region, parse_region_list_from_str
    strs = ["[3,4]", "[7,7][2,3]", "[1,1] [10,12]", "[0,3][4,5]", "", "[12,15][3,4][8,8]"]
    word_lists = [
        (["<BOS>", "this", "is", "|", "a", "test", ",", "okay", "<EOS>"], [[2, 3]]),
        (["<BOS>", "one", "|", "two", "|", "three", "|", "four", "<EOS>"], [[1, 1], [3, 4]]),
        (["hello", "world"], [[2, 2]]),
        (["<BOS>", "a", "|", "b", "!", "c", "<pad>", "d"], [[2, 4]]),
    ]
    out = {"parse": [[s_, parse_region_list_from_str(s_)] for s_ in strs],
           "words_region": [[w, r, get_words_region_from_origintxt_region(w, r)] for w, r in word_lists]}
    with open(os.path.join(GOLD, "edit_regions.json"), "w") as f:
        json.dump(out, f)
    print("wrote edit_regions.json")


def edit_cases(hp):
    region_helper_cases()
    edit_case(hp, "edit_mid", seed=5, steps=4, wseed=31, vseed=21, h=Wt.HIFIGAN_TINY)
    edit_case(hp, "edit_longer", seed=6, steps=3, wseed=32, vseed=21, h=Wt.HIFIGAN_TINY, n_words=8, region=(2, 3),
              n_edited_words=5, max_dur=9)
    edit_case(hp, "edit_end", seed=7, steps=2, wseed=33, vseed=21, h=Wt.HIFIGAN_TINY, n_words=5, region=(5, 5),
              n_edited_words=2)
    edit_case(hp, "edit_start", seed=9, steps=2, wseed=34, vseed=21, h=Wt.HIFIGAN_TINY, n_words=5, region=(1, 2),
              n_edited_words=3)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "edit":  # regenerate only the edit_* cases
        edit_cases(ref_import.install(timesteps=4))
    elif len(sys.argv) > 1 and sys.argv[1] == "nopitch":
        nopitch_cases(ref_import.install(timesteps=4))
    elif len(sys.argv) > 1 and sys.argv[1] == "normal":
        normal_cases(ref_import.install(timesteps=4))
    elif len(sys.argv) > 1 and sys.argv[1] == "campnet":
        campnet_cases(ref_import.install(timesteps=4))
    elif len(sys.argv) > 1 and sys.argv[1] == "bbs":
        ref_import.install(timesteps=4)
        batch_by_size_case()
    elif len(sys.argv) > 1 and sys.argv[1] == "hifigan_v23":
        ref_import.install(timesteps=4)
        hifigan_v23_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == "hifigan_long":
        ref_import.install(timesteps=4)
        hifigan_case("hifigan_v1_long", Wt.HIFIGAN_V1, B=1, T=220, wseed=23, iseed=204, manifest="hifigan_v1")
    else:
        main()
