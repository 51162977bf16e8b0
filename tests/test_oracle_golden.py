"""The oracle (oracle/oracle.py) against the golden vectors generated from the reference itself
(oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle as O
from oracle import weights as Wt

torch.set_grad_enabled(False)
TOL = 2e-5

INFER = [("infer_tiny", "spec_denoiser"), ("infer_pad", "spec_denoiser"), ("infer_ragged", "spec_denoiser"), ("infer_short", "spec_denoiser"), ("infer_predpitch", "spec_denoiser"),
         ("infer_dil", "spec_denoiser_dil"), ("infer_c64", "spec_denoiser_c64"), ("infer_drift100", "spec_denoiser"),
         ("infer_nopitch", "spec_denoiser_nopitch"),  # egs/spec_denoiser_libritts.yaml: use_pitch_embed false
         ("infer_normal", "spec_denoiser_normal")]    # egs/spec_denoiser_wo_masked_predictor.yaml


def _run_infer(case, manifest):
    g = load_golden(case)
    m = g["meta"]
    W = Wt.seeded_weights(Wt.load_manifest(manifest), m["wseed"])
    inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=m["pad_tail"])
    noises = Wt.synthetic_noises(m["B"], m["T"], m["steps"], seed=m["iseed"] + 1)
    dcl = m["overrides"].get("dilation_cycle_length", 1)
    flags = dict(m["flags"])
    if not m["overrides"].get("use_pitch_embed", True):
        flags["use_pitch_embed"] = False
    if m.get("variant", "masked") != "masked":
        flags["variant"] = m["variant"]
    ret = O.gaussian_diffusion_infer(W, m["steps"], inp, noises, dilation_cycle_length=dcl, **flags)
    return g, ret


@pytest.mark.parametrize("case,manifest", INFER)
def test_oracle_infer_matches_reference(case, manifest):
    g, ret = _run_infer(case, manifest)
    assert np.abs(ret["mel_out"].numpy() - g["mel_out"]).max() < TOL
    assert np.abs(ret["decoder_inp"].numpy() - g["decoder_inp"]).max() < TOL
    assert np.abs(ret["dur"].numpy() - g["dur"]).max() < TOL
    assert np.array_equal(ret["mel2ph"].numpy(), g["mel2ph"])
    if "masked_dur" in g:  # the plain FastSpeech of the `normal` variant has no masked predictor inputs
        assert np.array_equal(ret["masked_dur"].numpy(), g["masked_dur"])
    else:
        assert "masked_dur" not in ret and int(ret["pitch"].max()) == 1  # uv flags bound to f0: every frame -> bin 1
    if "pitch_pred" not in g:  # no pitch block: the reference's ret dict has none of the pitch keys either
        assert not ({"pitch_pred", "pitch", "f0_denorm"} & set(ret))
        return
    assert np.abs(ret["pitch_pred"].numpy() - g["pitch_pred"]).max() < TOL
    assert np.array_equal(ret["pitch"].numpy(), g["pitch"])


def test_oracle_full800_100_steps_matches_reference():
    """The metric's own shape run by the reference (round 6): T = 800, T_txt = 100, 100 steps, B = 2 with one padded tail.  The light fixture
    keeps mel_out, the integer tensors and every 8th frame of the x traces of steps 0 / 50 / 99 (spec_denoiser.py:178-184)."""
    g = load_golden("infer_full800")
    m = g["meta"]
    assert (m["B"], m["T"], m["T_txt"], m["steps"], m["light_stride"]) == (2, 800, 100, 100, 8)
    W = Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"])
    inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=True)
    noises = Wt.synthetic_noises(m["B"], m["T"], m["steps"], seed=m["iseed"] + 1)
    trace = []
    ret = O.gaussian_diffusion_infer(W, m["steps"], inp, noises, trace=trace)
    assert np.abs(ret["mel_out"].numpy() - g["mel_out"]).max() < TOL
    for k in ("mel2ph", "masked_dur", "pitch", "masked_pitch"):
        assert np.array_equal(ret[k].numpy(), g[k]), k
    for k in (0, 50, 99):
        assert np.abs(trace[k][0].numpy()[..., ::8] - g["x0_step%d" % k]).max() < TOL
        assert np.abs(trace[k][1].numpy()[..., ::8] - g["x_step%d" % k]).max() < TOL


def test_oracle_train_branch():
    g = load_golden("train_tiny")
    m = g["meta"]
    W = Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"])
    inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=True)
    ret = O.gaussian_diffusion_train(W, m["steps"], inp, torch.from_numpy(g["t"]), torch.from_numpy(g["eps"]))
    assert np.abs(ret["mel_out"].numpy() - g["mel_out"]).max() < TOL


@pytest.mark.parametrize("case,manifest", [("train_losses", "spec_denoiser"), ("train_losses_ragged", "spec_denoiser"),
                                           ("train_losses_nopitch", "spec_denoiser_nopitch"),
                                           ("train_losses_normal", "spec_denoiser_normal")])
def test_oracle_training_losses_and_grads(case, manifest):
    """train_losses.npz = reference model + the reference's own loss functions + autograd (oracle/make_golden.py)."""
    g = load_golden(case)
    m = g["meta"]
    use_pitch = "nopitch" not in case
    assert ("loss_uv" in g) == use_pitch
    W = {k: v.requires_grad_(True) for k, v in Wt.seeded_weights(Wt.load_manifest(manifest), m["wseed"]).items()}
    inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=True)
    with torch.enable_grad():
        losses, _ = O.training_losses(W, m["steps"], inp, torch.from_numpy(g["t"]), torch.from_numpy(g["eps"]),
                                      sil_ids=m["sil_ids"], use_pitch_embed=use_pitch,
                                      variant=m.get("variant", "masked"))
        sum(losses.values()).backward()
    for k, v in losses.items():
        assert abs(float(v) - float(g["loss_" + k])) < 1e-5 * max(1.0, abs(float(v))), k
    norms = dict(zip(m["param_names"], g["grad_norms"]))
    for k, ref in norms.items():
        if ref < 0:
            assert W[k].grad is None, k
        else:
            assert abs(float(W[k].grad.norm()) - ref) < 1e-4 * ref + 1e-9, k
    for key in [k for k in g if k.startswith("grad::")]:
        gr = W[key[6:]].grad
        ref = torch.from_numpy(g[key])
        assert float((gr[:ref.shape[0]] - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-9, key


def test_schedule_known_answers():
    # SURVEY.md 8a rows a1/a2
    b8 = O.vpsde_betas(9)
    assert abs(b8[0] - 0.2269467980000719) < 1e-15 and abs(b8[8] - 0.9849766278643581) < 1e-15
    b100 = O.vpsde_betas(101)
    assert abs(b100[0] - 0.0029414550474994305) < 1e-15 and abs(b100[100] - 0.32570252868784266) < 1e-15
    g = load_golden("schedule")
    for steps in (4, 8, 100):
        tab, _ = O.diffusion_tables(steps)
        for k, v in tab.items():
            assert np.array_equal(v.numpy(), g["s%d_%s" % (steps, k)]), (steps, k)
    tab, t64 = O.diffusion_tables(100)
    assert t64["posterior_mean_coef1"][0] == 1.0 and t64["posterior_mean_coef2"][0] == 0.0
    assert abs(t64["posterior_mean_coef1"][99] - 2.1172884940200542e-05) < 1e-18


def test_length_regulator():
    g = load_golden("length_regulator")
    out = O.length_regulator(torch.from_numpy(g["dur"]), torch.from_numpy(g["pad"]))
    assert np.array_equal(out.numpy(), g["mel2ph"])


@pytest.mark.parametrize("name,h", [("hifigan_tiny", Wt.HIFIGAN_TINY), ("hifigan_tiny_rb2", Wt.HIFIGAN_TINY_RB2),
                                    ("hifigan_v1", Wt.HIFIGAN_V1), ("hifigan_v2", Wt.HIFIGAN_V2), ("hifigan_v3", Wt.HIFIGAN_V3)])
def test_oracle_hifigan(name, h):
    g = load_golden(name)
    W = Wt.seeded_weights(Wt.load_manifest(name), g["meta"]["wseed"])
    wav = O.hifigan_forward(W, h, torch.from_numpy(g["mel"]))
    assert np.abs(wav.numpy() - g["wav"]).max() < TOL


def test_mel_mcd_zero_and_scale():
    rng = np.random.default_rng(0)
    a = rng.normal(size=(50, 80))
    assert O.mel_mcd(a, a) == 0.0
    assert O.mel_mcd(a, a + 1e-4) < 1e-2


def test_pitch_bin_edges_golden():
    """The oracle's f0 -> coarse bin on the reference-generated edge neighbourhoods (every value within 8 ulp of a bin edge)."""
    g = load_golden("pitch_edges")
    f = torch.from_numpy(g["f0_bits"].view(np.float32).copy())
    bins = O.f0_to_coarse(O.denorm_f0(f, None))
    assert torch.equal(bins, torch.from_numpy(g["bins"].astype(np.int64)))
