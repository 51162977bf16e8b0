"""Caller side of inference: SpecDenoiserInfer.forward_model and its helpers.

CPU part: the oracle restatement against the fixtures generated from the reference's own forward_model
(oracle/make_golden.py::edit_case) and the host-side integer planning of the product against the oracle.
GPU part: the product end to end against the same fixtures (integer tensors bit-exact, |dmel| < 1e-4)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, base_hparams, load_golden
from oracle import oracle as O
from oracle import weights as Wt

torch.set_grad_enabled(False)
CASES = ["edit_mid", "edit_longer", "edit_end", "edit_start"]
INT_KEYS = ["masked_dur", "pred_mel2ph", "edited_mel2ph", "pitch"]


def _sample(meta):
    return Wt.synthetic_edit_sample(meta["seed"], **meta["gen"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_edit_forward_model_matches_reference(case):
    g = load_golden(case)
    m = g["meta"]
    W = Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"])
    Wv = Wt.seeded_weights(Wt.load_manifest("hifigan_tiny"), m["vseed"])
    o = O.edit_forward_model(W, Wv, m["h"], m["steps"], _sample(m), list(torch.from_numpy(g["noises"])))
    for k in ("wav_out", "wav_gt", "mel_out", "masked_mel_gt"):
        assert np.abs(o[k].numpy() - g[k]).max() < 2e-5, k
    assert np.array_equal(o["masked_mel_out"].numpy(), g["masked_mel_out"])
    for k in INT_KEYS:
        assert np.array_equal(o[k].numpy(), g[k]), k
    assert [o["head_idx"], o["tail_idx"]] == g["head_tail"].tolist()


def test_region_helpers_match_reference_pairs():
    import set_amd  # noqa: F401
    from set_amd import infer
    with open(os.path.join(GOLDEN, "edit_regions.json")) as f:
        pairs = json.load(f)
    for s, want in pairs["parse"]:
        assert infer.parse_region_list_from_str(s) == want
        assert O.parse_region_list_from_str(s) == want
    for words, regions, want in pairs["words_region"]:
        assert infer.get_words_region_from_origintxt_region(words, regions) == want
        skip = lambda w: (w == "" or not w[0].isalpha()) and w in ("|", "<BOS>", "<pad>")  # noqa: E731
        assert O.words_region_from_text_region(words, regions, skip) == want
    with pytest.raises(AssertionError):
        infer.get_words_region_from_origintxt_region(["a"], [])


@pytest.mark.parametrize("case", CASES)
def test_host_planning_matches_oracle_and_golden(case):
    """plan_masked_dur / plan_splice (numpy, host) are what sizes every device tensor of the edit: bit-exact."""
    import set_amd  # noqa: F401
    from set_amd import infer
    g = load_golden(case)
    s = _sample(g["meta"])
    md = infer.plan_masked_dur(s["ph2word"], s["edited_ph2word"], s["dur"], s["words_region"][0])
    assert np.array_equal(md, g["masked_dur"])
    o_md, o_reg = O.edit_plan_durations(s)
    assert np.array_equal(md, o_md.numpy())
    plan = infer.plan_splice(s["mel2ph"], s["mel2word"], s["edited_ph2word"], g["pred_mel2ph"], s["words_region"][0],
                             s["edited_words_region"][0])
    assert np.array_equal(plan["mel2ph"], g["edited_mel2ph"])
    assert [plan["head_idx"], plan["tail_idx"]] == g["head_tail"].tolist()
    assert np.array_equal(plan["in_region"], o_reg.numpy()[0])
    osp = O.edit_splice(s, torch.from_numpy(g["pred_mel2ph"]))
    assert plan["length_edited"] == osp["length_edited"]


def test_host_planning_error_behaviour():
    import set_amd  # noqa: F401
    from set_amd import infer
    s = Wt.synthetic_edit_sample(5)
    Te = s["edited_ph2word"].shape[1]
    # a predicted alignment that gives the new words no frame: upstream dies in .max() of an empty selection
    pred = torch.ones(1, 20, dtype=torch.int64)
    with pytest.raises(RuntimeError):
        infer.plan_splice(s["mel2ph"], s["mel2word"], s["edited_ph2word"], pred, s["words_region"][0],
                          s["edited_words_region"][0])
    assert Te == s["edited_txt_tokens"].shape[1]


# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _build(dev, m):
    from set_amd.diffnet import DiffNet
    from set_amd.hifigan import HifiGanGenerator
    from set_amd.infer import SpecDenoiserInfer
    from set_amd.spec_denoiser import GaussianDiffusion
    hp = base_hparams(timesteps=m["steps"])
    model = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=m["steps"], time_scale=1,
                              loss_type="l1", spec_min=[], spec_max=[], hp=hp)
    missing, unexpected = model.load_state_dict(Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"]),
                                                strict=False)
    assert not unexpected and all(Wt.is_buffer(k) for k in missing)
    voc = HifiGanGenerator(m["h"])
    voc.load_state_dict(Wt.seeded_weights(Wt.load_manifest("hifigan_tiny"), m["vseed"]), strict=True)
    return SpecDenoiserInfer(hp, device=dev, model=model, vocoder=voc)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_forward_model_matches_reference(dev, case):
    g = load_golden(case)
    m = g["meta"]
    inf = _build(dev, m)
    wav_out, wav_gt, mel_out, mel_gt, masked_mel_out, masked_mel_gt, aux = inf.forward_model(
        _sample(m), noises=torch.from_numpy(g["noises"]), return_aux=True)
    # integer plan + pitch bins: bit exact
    assert np.array_equal(aux["masked_dur"], g["masked_dur"])
    assert np.array_equal(aux["pred_mel2ph"], g["pred_mel2ph"])
    assert np.array_equal(aux["edited_mel2ph"], g["edited_mel2ph"])
    assert np.array_equal(aux["mel2ph_out"], g["edited_mel2ph"])
    assert np.array_equal(aux["pitch"], g["pitch"])
    assert [aux["head_idx"], aux["tail_idx"]] == g["head_tail"].tolist()
    # pure splices: bit exact
    assert np.array_equal(masked_mel_out, g["masked_mel_out"])
    assert np.array_equal(masked_mel_gt, g["masked_mel_gt"])
    assert np.array_equal(aux["edited_f0"], g["edited_f0"]) and np.array_equal(aux["edited_uv"], g["edited_uv"])
    assert np.array_equal(aux["time_mel_masks"], g["time_mel_masks"])
    assert np.array_equal(mel_gt, _sample(m)["mel"][0].numpy())
    # floats
    assert np.abs(aux["dur_pred"] - g["dur_pred"]).max() < 2e-5
    d = np.abs(mel_out - g["mel_out"]).max()
    print("%s: max|dmel| %.2e  max|dwav| %.2e" % (case, d, np.abs(wav_out - g["wav_out"]).max()))
    assert d < 1e-4
    assert np.abs(wav_out - g["wav_out"]).max() < 1e-4 and np.abs(wav_gt - g["wav_gt"]).max() < 1e-4


@pytest.mark.gpu
def test_forward_model_from_item_and_philox_noise(dev):
    """input_to_batch path (numpy item as preprocess_input would hand it over) + on-device noise: deterministic in
    the seed, untouched frames are returned verbatim, output sizes follow the plan."""
    g = load_golden("edit_mid")
    m = g["meta"]
    inf = _build(dev, m)
    s = _sample(m)
    item = {"item_name": "x", "text": "t", "ph": "p", "ph_token": [1] * s["ph2word"].shape[1],
            "edited_ph_token": s["edited_txt_tokens"][0].numpy(), "ph2word": s["ph2word"][0].numpy(),
            "edited_ph2word": s["edited_ph2word"][0].numpy(), "mel2ph": s["mel2ph"][0].numpy(),
            "mel2word": s["mel2word"][0].numpy(), "dur": s["dur"][0].numpy(), "mel": s["mel"][0].numpy(),
            "f0": s["f0"][0].numpy(), "uv": s["uv"][0].numpy(), "spk_embed": s["spk_embed"][0].numpy(),
            "words_region": s["words_region"], "edited_words_region": s["edited_words_region"]}
    a = inf.forward_model(item, seed=3)
    b = inf.forward_model(item, seed=3)
    c = inf.forward_model(item, seed=4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[2], c[2])
    head, tail = g["head_tail"].tolist()
    T_new = g["edited_mel2ph"].shape[1]
    assert a[2].shape == (T_new, 80) and a[0].shape[0] == T_new * 8  # hifigan_tiny hop = 4*2
    assert np.array_equal(a[2][:head], s["mel"][0].numpy()[:head])
    assert np.array_equal(a[2][tail:], a[4][tail:]) and np.isfinite(a[0]).all()
    item.pop("spk_embed")
    with pytest.raises(KeyError):
        inf.forward_model(item)
