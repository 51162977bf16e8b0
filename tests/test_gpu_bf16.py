"""GPU tests of the bf16-operand training path (BASELINE configs[1] "bf16").

Two kinds of check:
* EXACTNESS of the kernels given their contract: operands rounded to bf16 (RNE) and multiplied / accumulated in fp32
  must equal torch's fp32 convolution of the same ROUNDED operands up to fp32 summation order (1e-5 relative) -- this
  pins the MFMA fragment maps, the tap / dilation / halo geometry, the packed weight image and the epilogue;
* TOLERANCE of the arithmetic change itself on the training golden (tests/golden/train_losses.npz, produced by the
  reference): what bf16 operands cost against the fp32 reference, with the bars written below
  (mel-level MCD is the quality measure; the survey measured 0.20 for torch.autocast on this model).
Also here: the deterministic weight gradient (per-slice partials + ordered reduce) is bit-stable run to run.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import base_hparams, load_golden
from oracle import oracle as O
from oracle import weights as Wt

pytestmark = pytest.mark.gpu

# ---- the tolerance of the bf16 path against the fp32 reference golden (train_losses.npz) -------------------------
BF16_LOSS_REL = 2e-2        # every loss term, relative
BF16_MEL_MCD = 0.20         # mel-level MCD between the bf16 and the fp32 `mel_out`
BF16_GRAD_REL_ALL = 8e-2    # all parameter gradients as one vector: ||g_bf16 - g_fp32|| / ||g_fp32||
BF16_GRAD_REL = 0.25        # worst single parameter tensor (small tensors deep in the predictors carry the most noise)
BF16_GRAD_COS = 0.97        # ... and its cosine similarity


@pytest.fixture(scope="module")
def dev(built_lib):
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def bf16():
    from set_amd import ops
    ops.set_compute_dtype("bf16")
    yield
    ops.set_compute_dtype("f32")


def _r(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


CASES = [
    # B, Cin, Cout, K, dil, T, extras
    (2, 256, 512, 3, 1, 300, dict(chan_add=True, res=True)),            # DiffNet dilated conv (the hot shape)
    (2, 256, 512, 3, 4, 130, dict(chan_add=True, res=True)),            # dilation cycle
    (2, 256, 512, 1, 1, 200, dict()),                                    # output projection: KCH = 64 stages
    (3, 192, 512, 1, 1, 97, dict()),                                     # conditioner projection, ragged T
    (2, 512, 256, 3, 1, 131, dict()),                                    # shape of the dilated conv's input gradient
    (2, 192, 384, 5, 1, 70, dict(alpha=5 ** -0.5, act="gelu")),          # encoder conv: 5 taps = 2 tap groups
    (2, 384, 192, 1, 1, 150, dict(res=True, mask=True)),                 # 64-row blocks (Cout = 192)
    (2, 192, 192, 9, 1, 77, dict(act="relu")),                           # 9 taps (CampNet FFN): 3 tap groups
    (1, 256, 80, 1, 1, 260, dict()),                                     # mel output projection: Cout = 80 (padded rows)
    (2, 80, 256, 1, 1, 100, dict(act="relu")),                           # input projection: Cin = 80 -> 96 (3 chunks of 32)
    (1, 256, 256, 1, 1, 64, dict(pro="div", pro_param=math.sqrt(20.0), act="relu")),
    (2, 128, 128, 7, 3, 200, dict(pro="lrelu", pro_param=0.1, res=True)),  # a HiFi-GAN resblock conv shape
]


def _ref_conv(x, w, b, add, res, mask, K, dil, ex):
    xin = x if add is None else x + add[:, :, None]
    if ex.get("pro") == "div":
        xin = xin / ex["pro_param"]
    elif ex.get("pro") == "lrelu":
        xin = F.leaky_relu(xin, ex["pro_param"])
    pad = dil * (K - 1) // 2
    y = F.conv1d(F.pad(_r(xin), (pad, pad)), _r(w), None, dilation=dil)
    y = (y + b[None, :, None]) * ex.get("alpha", 1.0)
    y = {"none": lambda v: v, "relu": F.relu, "gelu": F.gelu}[ex.get("act", "none")](y)
    if res is not None:
        y = y + res
    if mask is not None:
        y = y * mask[:, None, :]
    return y


@pytest.mark.parametrize("case", CASES)
def test_conv1d_bf16_equals_fp32_conv_of_rounded_operands(dev, case):
    from set_amd import ops
    B, Cin, Cout, K, dil, T, ex = case
    g = torch.Generator().manual_seed(Cin + Cout + K + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, T, generator=g) if ex.get("res") else None
    mask = (torch.rand(B, T, generator=g) > 0.3).float() if ex.get("mask") else None
    add = torch.randn(B, Cin, generator=g) if ex.get("chan_add") else None
    want = _ref_conv(x, w, b, add, res, mask, K, dil, ex)
    wd = w.to(dev)
    cw = ops.ConvWeight(lambda: wd, Cout, Cin, K)
    kw = {k: ex[k] for k in ("pro", "pro_param", "act", "alpha") if k in ex}
    got = ops.conv1d(x.to(dev), cw, b.to(dev), dil=dil, pad=dil * (K - 1) // 2,
                     in_chan_add=None if add is None else add.to(dev), res=None if res is None else res.to(dev),
                     mask=None if mask is None else mask.to(dev), impl="bf16", **kw)
    torch.cuda.synchronize()
    assert _rel(got, want) < 1e-5
    # accumulate + out_div epilogue (running mean), on top of a previous output
    prev = torch.randn(B, Cout, T, generator=g)
    out = prev.clone().to(dev)
    ops.conv1d(x.to(dev), cw, b.to(dev), dil=dil, pad=dil * (K - 1) // 2,
               in_chan_add=None if add is None else add.to(dev), res=None if res is None else res.to(dev),
               mask=None if mask is None else mask.to(dev), impl="bf16", out=out, accumulate=True, out_div=3.0, **kw)
    assert _rel(out, (prev + want) / 3.0) < 1e-5


UNIT_CASES = [
    # B, Cin, Cout, K, dil, pad, T, extras: the 16-byte-unit input staging of conv1d_bf16_kernel (T % 4 == 0, Cin % 4 == 0, halo <= 16)
    (2, 192, 768, 9, 1, 4, 800, dict()),                                   # CampNet FFN conv
    (2, 768, 192, 9, 1, 8, 800, dict(act="relu")),                         # ... causal ("LEFT") padding
    (2, 192, 384, 5, 1, 2, 800, dict(alpha=5 ** -0.5, act="gelu")),        # encoder conv: tile starts 2 frames off a quad
    (2, 256, 512, 3, 1, 1, 300, dict(chan_add=True, res=True)),            # DiffNet dilated conv, per-channel add
    (2, 256, 512, 3, 4, 4, 132, dict(chan_add=True)),
    (3, 100, 72, 3, 1, 1, 260, dict(pro="lrelu", pro_param=0.1)),          # partial channel chunk (100 of 128), a 4-frame second tile
    (2, 128, 128, 7, 2, 6, 200, dict(pro="div", pro_param=3.0, mask=True)),
    (1, 36, 200, 5, 1, 2, 4, dict()),                                      # T = one quad
]


@pytest.mark.parametrize("case", UNIT_CASES)
def test_conv1d_bf16_unit_staging_equals_the_per_frame_staging_bit_for_bit(dev, case, monkeypatch):
    """conv1d_bf16_kernel<..., VEC = true> (input in 16-byte units: 4 frames of a channel per load) writes the same bf16 values to the
    same LDS cells as the one-frame-per-load form: torch.equal; and both equal the conv of the rounded operands."""
    from set_amd import ops
    B, Cin, Cout, K, dil, pad, T, ex = case
    g = torch.Generator().manual_seed(Cin + Cout + K + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g) * 0.1
    res = torch.randn(B, Cout, T, generator=g) if ex.get("res") else None
    mask = (torch.rand(B, T, generator=g) > 0.3).float() if ex.get("mask") else None
    add = torch.randn(B, Cin, generator=g) if ex.get("chan_add") else None
    wd = w.to(dev)
    cw = ops.ConvWeight(lambda: wd, Cout, Cin, K)
    kw = {k: ex[k] for k in ("pro", "pro_param", "act", "alpha") if k in ex}
    outs = {}
    for units in ("0", "1"):
        monkeypatch.setenv("SET_AMD_CONV_BF16_UNITS", units)
        outs[units] = ops.conv1d(x.to(dev), cw, b.to(dev), dil=dil, pad=pad, in_chan_add=None if add is None else add.to(dev),
                                 res=None if res is None else res.to(dev), mask=None if mask is None else mask.to(dev), impl="bf16", **kw)
        torch.cuda.synchronize()
    assert torch.equal(outs["0"], outs["1"])
    if pad == dil * (K - 1) // 2:
        assert _rel(outs["1"], _ref_conv(x, w, b, add, res, mask, K, dil, ex)) < 1e-5


@pytest.mark.parametrize("case", CASES[:9])
def test_conv1d_bf16_backward_equals_rounded_operand_gradients(dev, case, bf16):
    """dgrad = conv of the (bf16-rounded) output gradient with the (bf16-rounded) transposed weights; wgrad = products of
    the rounded output gradient and the rounded conv input, summed in fp32."""
    from set_amd import autograd_ops as A, ops
    B, Cin, Cout, K, dil, T, ex = case
    if ex.get("act") == "gelu":
        pytest.skip("split activation: covered by the fp32 backward test; here the conv gradients")
    g = torch.Generator().manual_seed(Cin + Cout + K + T + 1)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K)
    b = torch.randn(Cout, generator=g) * 0.1
    add = torch.randn(B, Cin, generator=g) if ex.get("chan_add") else None
    gy = torch.randn(B, Cout, T, generator=g)
    pad = dil * (K - 1) // 2
    act = ex.get("act", "none")
    mask = (torch.rand(B, T, generator=g) > 0.3).float() if ex.get("mask") else None
    # reference gradients, by hand, on rounded operands
    xin = x if add is None else x + add[:, :, None]
    if ex.get("pro") == "div":
        xin = xin / ex["pro_param"]
    y_pre = F.conv1d(F.pad(_r(xin), (pad, pad)), _r(w), None, dilation=dil) + b[None, :, None]
    gg = gy.clone()
    if mask is not None:
        gg = gg * mask[:, None, :]
    if act == "relu":
        gg = gg * (y_pre > 0).float()
    gr = _r(gg)
    dx_want = F.conv_transpose1d(gr, _r(w), None, dilation=dil)[:, :, pad:pad + T]
    if ex.get("pro") == "div":
        dx_want = dx_want / ex["pro_param"]
    xp = F.pad(_r(xin), (pad, pad))
    dw_want = torch.stack([torch.einsum("bot,bit->oi", gr, xp[:, :, k * dil:k * dil + T]) for k in range(K)], dim=-1)
    d = [t.clone().to(dev).requires_grad_(True) if t is not None else None for t in (x, w, b, add)]
    cw = ops.ConvWeight(lambda: d[1], Cout, Cin, K)
    kw = {k: ex[k] for k in ("pro", "pro_param") if k in ex}
    with torch.enable_grad():
        y = A.conv1d(d[0], cw, d[2], dil=dil, pad=pad, in_chan_add=d[3], act=act,
                     mask=None if mask is None else mask.to(dev), **kw)
        y.backward(gy.to(dev))
    torch.cuda.synchronize()
    assert _rel(d[0].grad, dx_want) < 2e-5
    assert _rel(d[1].grad, dw_want) < 2e-5
    assert _rel(d[2].grad, gg.sum((0, 2))) < 2e-5


TAPS_CASES = [
    # B, Cin, Cout, K, T, prologue, causal: the all-taps weight-gradient kernel (K = 5, 7, 9, dilation 1, "same" or causal padding)
    (2, 192, 768, 9, 800, None, False),           # CampNet FFN conv
    (2, 192, 768, 9, 800, None, True),            # ... of a decoder layer: 8 frames of left padding (transformer FFN, padding='LEFT')
    (3, 100, 200, 9, 333, None, False),           # ragged channels (partial 64-row tiles), ragged T
    (2, 80, 130, 5, 150, ("lrelu", 0.1), False),  # 5 taps, prologue on the conv input
    (2, 80, 130, 5, 150, None, True),
    (2, 64, 64, 7, 65, ("div", 3.0), False),      # 7 taps, a second chunk of one frame
    (2, 64, 96, 7, 130, None, True),
    (1, 256, 1024, 9, 64, None, False),           # exactly one chunk: every halo frame is padding
]


@pytest.mark.parametrize("case", TAPS_CASES)
def test_wgrad_all_taps_kernel_equals_rounded_operand_products(dev, case, monkeypatch):
    """dW[co][ci][k] = sum_{b,t} r(G[b][co][t]) r(P(X[b][ci][t + k - pad])) with zero padding: every tap of the 64 x 64 tile comes
    from shifted views of one staged X row (csrc/bf16.hip: conv1d_wgrad_taps_bf16_kernel); also against the one-tap-per-block kernel
    (same products, another summation order) and bit-stable from run to run."""
    from set_amd import _lib, autograd_ops as A
    B, Cin, Cout, K, T, pro, causal = case
    g = torch.Generator().manual_seed(Cin + Cout + K + T)
    x = torch.randn(B, Cin, T, generator=g)
    gy = torch.randn(B, Cout, T, generator=g)
    pad = K - 1 if causal else (K - 1) // 2
    xin = x
    code, param = 0, 0.0
    if pro is not None:
        code, param = _lib.PRO[pro[0]], pro[1]
        xin = F.leaky_relu(x, param) if pro[0] == "lrelu" else x / param
    xp = F.pad(_r(xin), (pad, K - 1 - pad))
    want = torch.stack([torch.einsum("bot,bit->oi", _r(gy), xp[:, :, k:k + T]) for k in range(K)], dim=-1)
    xd, gd = x.to(dev), gy.to(dev)
    outs = []
    for _ in range(2):
        dw = torch.zeros(Cout, Cin, K, device=dev)
        A.conv_wgrad(gd, xd, None, dw, B, Cin, Cout, K, 1, pad, T, T, code, param, dtype=_lib.DTYPE_BF16)
        outs.append(dw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert _rel(outs[0], want) < 2e-5
    A.conv_wgrad(gd, xd, None, outs[1], B, Cin, Cout, K, 1, pad, T, T, code, param, dtype=_lib.DTYPE_BF16)  # adds on top
    assert _rel(outs[1], 2 * want) < 2e-5


def _q4(t):
    """[B, C, T] -> the channel-quad-interleaved storage [B][C / 4][T][4] the bf16 operands of the weight-gradient kernels use
    (csrc/diffnet_bf16.hip: what the fused layer kernels write); same shape label, other memory order."""
    B, C, T = t.shape
    return t.reshape(B, C // 4, 4, T).transpose(2, 3).contiguous().view(B, C, T)


Q4_WGRAD_CASES = [
    # B, Cin, Cout, K, dil, T, bf16 X too: every loader of a bf16 operand (16-byte units / element-wise, 1 tap / 3 taps in three copies / one copy)
    (2, 256, 512, 1, 1, 800, True),    # output projection: G and X bf16, 16-byte units
    (3, 192, 512, 1, 1, 77, False),    # conditioner projection, ragged T: element-wise G
    (2, 100, 200, 1, 1, 136, True),    # partial row tiles
    (3, 256, 512, 1, 1, 77, True),     # ragged T, both operands element-wise
    (2, 256, 512, 3, 1, 77, False),    # 3 taps, ragged T: three-copy kernel, element-wise G
    (2, 64, 132, 3, 3, 72, False),     # 3 taps, dilation 3: three-copy kernel, G in units
    (2, 256, 512, 3, 2, 264, False),   # one-copy kernel
]


@pytest.mark.parametrize("case", Q4_WGRAD_CASES)
def test_wgrad_reads_bf16_operands_in_the_quad_interleaved_layout(dev, case):
    from set_amd import _lib, autograd_ops as A
    B, Cin, Cout, K, dil, T, x16 = case
    g = torch.Generator().manual_seed(Cin + Cout + K + T)
    x = torch.randn(B, Cin, T, generator=g)
    gy = torch.randn(B, Cout, T, generator=g)
    pad = dil * (K - 1) // 2
    xp = F.pad(_r(x), (pad, pad))
    want = torch.stack([torch.einsum("bot,bit->oi", _r(gy), xp[:, :, k * dil:k * dil + T]) for k in range(K)], dim=-1)
    gd = _q4(gy.to(dev).to(torch.bfloat16))
    xd = _q4(x.to(dev).to(torch.bfloat16)) if x16 else x.to(dev)
    dw = torch.zeros(Cout, Cin, K, device=dev)
    A.conv_wgrad(gd, xd, None, dw, B, Cin, Cout, K, dil, pad, T, T, dtype=_lib.DTYPE_BF16_G16_X16 if x16 else _lib.DTYPE_BF16_G16)
    torch.cuda.synchronize()
    assert _rel(dw, want) < 2e-5


WGRAD3_UNIT_CASES = [
    # B, Cin, Cout, dil, T, per-channel add: the one-copy 3-tap weight-gradient kernel (bf16 output gradient, fp32 conv input in 16-byte units)
    (4, 256, 512, 1, 800, True),    # the DiffNet dilated conv of the training step
    (2, 256, 512, 2, 800, True),
    (2, 256, 512, 4, 800, False),
    (2, 256, 512, 8, 800, True),
    (3, 100, 200, 1, 136, True),    # partial row tiles, a last chunk of 8 frames
    (2, 64, 128, 8, 64, False),     # exactly one chunk: every halo frame is padding
    (5, 72, 132, 2, 72, True),
]


@pytest.mark.parametrize("case", WGRAD3_UNIT_CASES)
def test_wgrad3_one_copy_kernel_equals_the_three_copy_kernel_bit_for_bit(dev, case, monkeypatch):
    """conv1d_wgrad3u_bf16_kernel (one staged X copy, tap windows cut from three aligned groups) against conv1d_wgrad3_bf16_kernel
    (three shifted copies): the same operand values in the same order per accumulator, hence torch.equal; and against the rounded
    operand products."""
    from set_amd import _lib, autograd_ops as A
    B, Cin, Cout, dil, T, with_add = case
    g = torch.Generator().manual_seed(Cin + Cout + dil + T)
    x = torch.randn(B, Cin, T, generator=g)
    gy = torch.randn(B, Cout, T, generator=g)
    add = torch.randn(B, Cin, generator=g) if with_add else None
    xin = x if add is None else x + add[:, :, None]
    xp = F.pad(_r(xin), (dil, dil))
    want = torch.stack([torch.einsum("bot,bit->oi", _r(gy), xp[:, :, k * dil:k * dil + T]) for k in range(3)], dim=-1)
    xd, gd = x.to(dev), _q4(gy.to(dev).to(torch.bfloat16))
    ad = None if add is None else add.to(dev)
    outs = {}
    for units in ("0", "1"):
        monkeypatch.setenv("SET_AMD_WGRAD3_UNITS", units)
        dw = torch.zeros(Cout, Cin, 3, device=dev)
        A.conv_wgrad(gd, xd, ad, dw, B, Cin, Cout, 3, dil, dil, T, T, dtype=_lib.DTYPE_BF16_G16)
        torch.cuda.synchronize()
        outs[units] = dw
    assert torch.equal(outs["0"], outs["1"])
    assert _rel(outs["1"], want) < 2e-5


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_deterministic_wgrad_is_bit_stable_and_matches_the_atomic_kernel(dev, dtype):
    from set_amd import _lib, autograd_ops as A, ops
    B, Cin, Cout, K, dil, T = 8, 256, 512, 3, 1, 800
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(B, Cout, T, generator=g).to(dev)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    add = torch.randn(B, Cin, generator=g).to(dev)
    outs = []
    ops.set_compute_dtype(dtype)
    try:
        for _ in range(3):
            dw = torch.zeros(Cout, Cin, K, device=dev)
            A.conv_wgrad(gy, x, add, dw, B, Cin, Cout, K, dil, dil, T, T)
            outs.append(dw)
        twice = outs[2].clone()
        A.conv_wgrad(gy, x, add, twice, B, Cin, Cout, K, dil, dil, T, T)  # accumulate semantics: adds on top
    finally:
        ops.set_compute_dtype("f32")
    torch.cuda.synchronize()
    assert _rel(twice, 2 * outs[1]) < 1e-6
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.zeros(Cout, Cin, K, device=dev)
    _lib.check(_lib.lib().set_conv1d_wgrad(A._p(gy), A._p(x), A._p(add), A._p(ref), B, Cin, Cout, K, dil, dil, T, T, 0, 0.0,
                                          _lib.IMPL_MFMA, A._stream()), "set_conv1d_wgrad")
    tol = 1e-5 if dtype == "f32" else 1.5e-2  # bf16 operands: 2^-9 relative per product, averaged over 6400 frames
    assert _rel(outs[0], ref) < tol


def _run_training_golden(dev, dtype, monkeypatch, fused=False, fixture="train_losses"):
    from set_amd import hparams as H, ops, tasks
    # fused=False: one differentiable kernel per op for both operand types; True: the fused bf16 layer kernels
    monkeypatch.setenv("SET_AMD_TRAIN_STACK", "1" if fused else "0")
    g = load_golden(fixture)
    m = g["meta"]
    H.hparams.clear()
    H.hparams.update(base_hparams(timesteps=m["steps"]))
    task = tasks.SpeechDenoiserTask(build_vocoder=False)
    task.build_model()
    task.model.load_state_dict(Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"]), strict=False)
    task.model.to(dev).eval()
    inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=True)
    sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], mel2ph=inp["mel2ph"], f0=inp["f0"], uv=inp["uv"],
                  time_mel_masks=inp["time_mel_masks"].squeeze(-1), spk_embed=inp["spk_embed"])
    sample = {k: v.to(dev) for k, v in sample.items()}
    ops.set_compute_dtype(dtype)
    try:
        losses, out = task.run_model(sample, infer=False, t=torch.from_numpy(g["t"]).to(dev),
                                     noises=torch.from_numpy(g["eps"]).to(dev))
        with torch.enable_grad():
            total = sum(losses.values())
        total.backward()
    finally:
        ops.set_compute_dtype("f32")
    torch.cuda.synchronize()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in task.model.named_parameters()}
    return g, {k: float(v) for k, v in losses.items()}, out["mel_out_bct"].detach().transpose(1, 2).cpu(), grads


@pytest.mark.parametrize("fixture", ["train_losses", "train_losses_ragged"])
@pytest.mark.parametrize("layers", ["per_op", "fused_layers"])
def test_training_golden_within_the_bf16_tolerance(dev, monkeypatch, layers, fixture):
    """The reference-generated training golden under bf16 operands: every loss within BF16_LOSS_REL of the reference's
    value, mel_out within BF16_MEL_MCD (mel-level MCD) of the fp32 path's, every parameter gradient within
    BF16_GRAD_REL / BF16_GRAD_COS of the fp32 path's.  `fused_layers`: the DiffNet layers through the fused bf16 forward /
    backward kernels (what a training run uses); `per_op`: one kernel per op."""
    g, l32, mel32, g32 = _run_training_golden(dev, "f32", monkeypatch, fixture=fixture)
    _, l16, mel16, g16 = _run_training_golden(dev, "bf16", monkeypatch, fused=(layers == "fused_layers"), fixture=fixture)
    for k in ("l1_coarse", "ssim_coarse", "pdur", "wdur", "uv", "f0"):
        ref = float(g["loss_" + k])
        assert abs(l32[k] - ref) < 2e-5 * max(1.0, abs(ref))                    # the fp32 leg is the parity path
        assert abs(l16[k] - ref) < BF16_LOSS_REL * max(1.0, abs(ref)), (k, l16[k], ref)
    assert any(l16[k] != l32[k] for k in l16)                                   # bf16 really ran
    mcd = max(O.mel_mcd(mel16[b].numpy(), mel32[b].numpy()) for b in range(mel16.shape[0]))
    assert mcd < BF16_MEL_MCD, mcd
    rows, num, den = [], 0.0, 0.0
    for k, a in g32.items():
        b = g16[k]
        if a is None or float(a.abs().max()) == 0.0:
            assert b is None or float(b.abs().max()) == 0.0, k
            continue
        a, b = a.double().reshape(-1), b.double().reshape(-1)
        rows.append((float((a - b).norm() / a.norm()), float((a @ b) / (a.norm() * b.norm())), k, float(a.norm())))
        num += float((a - b).norm() ** 2)
        den += float(a.norm() ** 2)
    rows.sort(reverse=True)
    rel_all = math.sqrt(num / den)
    worst_rel, worst_cos = rows[0][0], min(r[1] for r in rows)
    print("bf16 vs fp32 training golden: mcd %.4f, all-gradient rel %.3e, worst tensor rel %.3e, worst cos %.6f, losses %s"
          % (mcd, rel_all, worst_rel, worst_cos, {k: "%.5f/%.5f" % (l16[k], l32[k]) for k in l16}))
    for r in rows[:8]:
        print("   rel %.3e cos %.6f |g| %.3e  %s" % r[:2] + (r[3], r[2]) if False else "   rel %.3e cos %.6f |g| %.3e  %s" % (r[0], r[1], r[3], r[2]))
    assert rel_all < BF16_GRAD_REL_ALL and worst_rel < BF16_GRAD_REL and worst_cos > BF16_GRAD_COS, (rel_all, worst_rel, worst_cos)


def test_bf16_training_step_is_deterministic(dev, monkeypatch):
    """Two identical bf16 steps from the same state give bit-identical gradients for every conv / linear weight (the
    weight-gradient GEMMs have no atomics).  Full-step determinism incl. the embedding tables is the fp32 test's job."""
    _, _, _, a = _run_training_golden(dev, "bf16", monkeypatch, fused=True)
    _, _, _, b = _run_training_golden(dev, "bf16", monkeypatch, fused=True)
    n = 0
    for k in a:
        if a[k] is not None and a[k].dim() >= 2 and "embed" not in k and "emb" not in k:
            assert torch.equal(a[k], b[k]), k
            n += 1
    assert n > 100


# ----------------------------------------------------------------------------------------------------------------------
# fused bf16 DiffNet layer kernels (csrc/diffnet_bf16.hip) against a torch emulation with the same rounding points
# ----------------------------------------------------------------------------------------------------------------------
RS2 = 0.70710678118654752440


def _emu_layer_fwd(x, skip, cond, dvec, Wd, bd, Wc, bc, Wo, bo, dil, first):
    xin = _r(x + dvec[:, :, None])
    y = F.conv1d(F.pad(xin, (dil, dil)), _r(Wd), None, dilation=dil) + F.conv1d(_r(cond), _r(Wc)) + (bd + bc)[None, :, None]
    C = x.shape[1]
    z = torch.sigmoid(y[:, :C]) * torch.tanh(y[:, C:])
    o = F.conv1d(_r(z), _r(Wo)) + bo[None, :, None]
    x_out = (x + o[:, :C]) * RS2
    skip_out = o[:, C:] if first else skip + o[:, C:]
    return x_out, skip_out, _r(y), _r(z), xin


def _emu_layer_bwd(dxo, dsk, y16, z16, xin, cond, Wd, Wc, Wo, dil):
    C = dsk.shape[1]
    T = dsk.shape[2]
    d_o = _r(torch.cat([(dxo * RS2) if dxo is not None else torch.zeros_like(dsk), dsk], 1))
    dz = F.conv1d(d_o, _r(Wo).transpose(0, 1).contiguous())
    s, th = torch.sigmoid(y16[:, :C]), torch.tanh(y16[:, C:])
    dy = _r(torch.cat([dz * th * s * (1 - s), dz * s * (1 - th * th)], 1))
    dxd = F.conv_transpose1d(dy, _r(Wd), None, dilation=dil)[:, :, dil:dil + T]
    dx = dxd + (dxo * RS2 if dxo is not None else 0.0)
    dcond = F.conv1d(dy, _r(Wc).transpose(0, 1).contiguous())
    dd = dxd.sum(2)
    xp = F.pad(xin, (dil, dil))
    dWd = torch.stack([torch.einsum("bot,bit->oi", dy, xp[:, :, k * dil:k * dil + T]) for k in range(3)], dim=-1)
    dWc = torch.einsum("bot,bit->oi", dy, _r(cond))[:, :, None]
    dWo = torch.einsum("bot,bit->oi", d_o, z16)[:, :, None]
    return dx, dcond, dd, dWd, dWc, dWo, d_o.sum((0, 2)), dy.sum((0, 2))


@pytest.mark.parametrize("T,cycle", [(300, 1), (97, 2), (800, 1)])
def test_fused_bf16_layers_match_emulation(dev, bf16, T, cycle):
    """2-3 layers through _DiffNetStackBf16Fn (fused forward + fused backward + bf16 weight gradients) against a torch
    emulation that rounds exactly where the kernels round (x+d, cond, z, d_o, dy, weights -> bf16; y / z saved in bf16).
    Tolerances: a bf16 rounding that flips on an fp32 summation-order difference moves one product by 2^-9 relative, so
    the bars are relative to the largest entry of each tensor, not 1e-5."""
    from set_amd import autograd_ops as A
    from set_amd.diffnet import DiffNet
    L = 3 if cycle == 2 else 2
    hp = base_hparams(residual_layers=L, dilation_cycle_length=cycle)
    torch.manual_seed(T)
    dn = DiffNet(80, hp).to(dev)
    B, C, H = 2, 256, 192
    g = torch.Generator().manual_seed(T + 7)
    hx = torch.randn(B, C, T, generator=g)
    cond = torch.randn(B, H, T, generator=g)
    dmat = torch.randn(B, L * C, generator=g) * 0.5
    gsk = torch.randn(B, C, T, generator=g)
    W = []
    for l in dn.residual_layers:
        for p in (l.dilated_conv.weight, l.conditioner_projection.weight, l.output_projection.weight):
            p.data.mul_(0.5)
        for p in (l.dilated_conv.bias, l.conditioner_projection.bias, l.output_projection.bias):
            p.data.normal_(0.0, 0.1)
        W.append([t.detach().cpu().clone() for t in (l.dilated_conv.weight, l.dilated_conv.bias, l.conditioner_projection.weight,
                                                       l.conditioner_projection.bias, l.output_projection.weight,
                                                       l.output_projection.bias)])
    # ---- emulation
    x, skip, saves = hx, None, []
    for l in range(L):
        Wd, bd, Wc, bc, Wo, bo = W[l]
        dil = dn.residual_layers[l].dilation
        xo, skip, y16, z16, xin = _emu_layer_fwd(x, skip, cond, dmat[:, l * C:(l + 1) * C], Wd, bd, Wc, bc, Wo, bo, dil, l == 0)
        saves.append((y16, z16, xin))
        x = xo
    want_skip = skip
    dxo, dcond_w, dd_w, gw = None, torch.zeros_like(cond), [], []
    for l in range(L - 1, -1, -1):
        Wd, bd, Wc, bc, Wo, bo = W[l]
        y16, z16, xin = saves[l]
        dx, dc, dd, dWd, dWc, dWo, dbo, dby = _emu_layer_bwd(dxo, gsk, y16, z16, xin, cond, Wd, Wc, Wo, dn.residual_layers[l].dilation)
        dcond_w += dc
        dd_w.insert(0, dd)
        gw.insert(0, (dWd, dby, dWc, dby, dWo, dbo))
        dxo = dx
    # ---- kernels
    hxd, cd, dmd = (t.clone().to(dev).requires_grad_(True) for t in (hx, cond, dmat))
    with torch.enable_grad():
        got_skip = A.diffnet_stack_train_bf16(dn, hxd, cd, dmd)
        got_skip.backward(gsk.to(dev))
    torch.cuda.synchronize()
    assert _rel(got_skip, want_skip) < 2e-3
    assert _rel(hxd.grad, dxo) < 4e-3
    assert _rel(cd.grad, dcond_w) < 4e-3
    assert _rel(dmd.grad, torch.cat(dd_w, 1)) < 4e-3
    for l, layer in enumerate(dn.residual_layers):
        dWd, dby, dWc, _, dWo, dbo = gw[l]
        assert _rel(layer.dilated_conv.weight.grad, dWd) < 4e-3, l
        assert _rel(layer.conditioner_projection.weight.grad, dWc.reshape(layer.conditioner_projection.weight.shape)) < 4e-3, l
        assert _rel(layer.output_projection.weight.grad, dWo.reshape(layer.output_projection.weight.shape)) < 4e-3, l
        assert _rel(layer.dilated_conv.bias.grad, dby) < 4e-3 and _rel(layer.conditioner_projection.bias.grad, dby) < 4e-3, l
        assert _rel(layer.output_projection.bias.grad, dbo) < 4e-3, l


def test_bf16_inference_loop_tolerance(dev):
    """The opt-in bf16-operand reverse loop (fused bf16 layer kernel per residual layer, conditioner projection inside the
    layer GEMM) on the reference goldens: NOT the parity path (that one is fp32, |dmel| < 1e-4) -- its bar is the quality
    measure of SURVEY.md 8(d), mel-level MCD against the reference mel, and a loose absolute bound."""
    from set_amd import ops
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    worst = {}
    for name in ("infer_tiny", "infer_pad", "infer_drift100"):
        g = load_golden(name)
        m = g["meta"]
        hp = base_hparams(timesteps=m["steps"])
        model = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=m["steps"], time_scale=1, loss_type="l1",
                                  spec_min=[], spec_max=[], hp=hp)
        model.load_state_dict(Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), m["wseed"]), strict=False)
        model.to(dev).eval()
        inp = Wt.synthetic_inputs(m["B"], m["T"], m["T_txt"], seed=m["iseed"], pad_tail=m.get("pad_tail", False))
        noises = torch.stack(Wt.synthetic_noises(m["B"], m["T"], m["steps"], seed=m["iseed"] + 1)).to(dev)
        d = {k: v.to(dev) for k, v in inp.items()}
        ops.set_compute_dtype("bf16")
        try:
            with torch.no_grad():
                ret = model(d["txt_tokens"], d["time_mel_masks"], d["mel2ph"], d["spk_embed"], d["ref_mels"], d["f0"], d["uv"],
                            infer=True, noises=noises)
        finally:
            ops.set_compute_dtype("f32")
        mel = ret["mel_out"].cpu().numpy()
        assert np.isfinite(mel).all()
        mcd = max(O.mel_mcd(mel[b], g["mel_out"][b]) for b in range(mel.shape[0]))
        worst[name] = (mcd, float(np.abs(mel - g["mel_out"]).max()))
        assert torch.equal(ret["mel2ph"].cpu(), torch.from_numpy(g["mel2ph"]))  # integer bookkeeping is untouched
    print("bf16 loop vs reference mel: %s" % {k: "mcd %.4f max|d| %.4f" % v for k, v in worst.items()})
    assert max(v[0] for v in worst.values()) < 0.20  # SURVEY.md 8(d)'s bar for a bf16 path (measured 0.05-0.06)


def test_bf16_inference_loop_quality_at_the_benchmark_size(dev):
    """the quality bar of the bf16-operand loop where it is tight -- B = 32, T = 800, 100 reverse steps, on-device
    Philox noise, the benchmark's random-init weights: mel-level MCD against the fp32 path on the same inputs and noise < 0.20
    (measured 0.179) and max |dmel| < 0.05 (0.026).  The loop runs ten residual layers per launch on 128-frame tiles here."""
    import bench
    from set_amd import _lib, ops
    from set_amd.synthetic import synthetic_inputs
    model = bench.build_model(dev, 100)
    inp = {k: v.to(dev) for k, v in synthetic_inputs(32, 800, 100, seed=1234).items()}

    def run():
        with torch.no_grad():
            return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"], inp["uv"],
                         infer=True, seed=0)["mel_out"].float().cpu().numpy()
    assert int(_lib.lib().set_diffnet_layers_bf16_plan(32, 800, 20, 1)) == 10
    ref = run()
    ops.set_compute_dtype("bf16")
    try:
        got = run()
    finally:
        ops.set_compute_dtype("f32")
    assert np.isfinite(got).all()
    mcd = max(O.mel_mcd(got[b], ref[b]) for b in range(got.shape[0]))
    dmax = float(np.abs(got - ref).max())
    print("bf16 loop vs fp32 path at B=32, T=800, 100 steps: mcd %.4f max|d| %.4f" % (mcd, dmax))
    assert mcd < 0.20 and dmax < 0.05


def test_bf16_loop_utterance_groups_do_not_share_scratch(dev, monkeypatch):
    """with n_groups > 1 the group chains of the bf16 reverse loop run concurrently on auxiliary streams; the 128-frame layer-group
    kernel keeps a block's private skip rows in a scratch buffer indexed by (blockIdx.y, blockIdx.x) of its OWN launch, so every group needs
    its own slice.  Grouping must never change results: 1 / 2 / 3 groups bit-identical at a shape forced onto the 128-frame tile."""
    from set_amd import _lib, ops
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    from set_amd.synthetic import synthetic_inputs
    monkeypatch.setenv("SET_AMD_BF16_FUSE_TILE", "128")
    hp = base_hparams(timesteps=6)
    model = GaussianDiffusion(list(range(80)), 80, DiffNet(80, hp), timesteps=6, time_scale=1, loss_type="l1", spec_min=[], spec_max=[], hp=hp)
    model.load_state_dict(Wt.seeded_weights(Wt.load_manifest("spec_denoiser"), 11), strict=False)
    model.to(dev).eval()
    B, T = 6, 600
    assert int(_lib.lib().set_diffnet_layers_bf16_plan(B // 2, T, 20, 1)) == 10
    inp = {k: v.to(dev) for k, v in synthetic_inputs(B, T, 60, seed=99).items()}
    ops.set_compute_dtype("bf16")
    try:
        outs = []
        for G in (1, 2, 3, 2):
            with torch.no_grad():
                outs.append(model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"], inp["f0"], inp["uv"],
                                  infer=True, seed=3, n_groups=G)["mel_out"].clone())
    finally:
        ops.set_compute_dtype("f32")
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_batch_repack_equals_per_weight_packs(dev):
    """ops.repack_bf16_images (what FlatAdamW.step calls in bf16 mode): every registered bf16 weight image re-rounded from its
    fp32 master weight in ONE launch -- bit-equal to the per-weight pack kernel, for plain, transposed (input-gradient) and
    offset / strided (a slice of a packed projection) weights; afterwards the lazy per-weight check finds the images current."""
    import ctypes as C
    from set_amd import _lib, ops
    g = torch.Generator().manual_seed(7)
    ws = [torch.randn(192, 96, 5, generator=g).to(dev), torch.randn(80, 33, 1, generator=g).to(dev),
          torch.randn(3 * 64, 64, generator=g).to(dev), torch.randn(256, 512, 3, generator=g).to(dev)]
    cws = [ops.ConvWeight(ws[0], 192, 96, 5), ops.ConvWeight(ws[1], 80, 33, 1),
           ops.ConvWeight(ws[2], 128, 64, 1, base=64 * 64),   # rows 64.. of a packed [3 * 64, 64] projection (kv slice)
           ops.ConvWeight(ws[3], 256, 512, 3)]
    cws.append(cws[0].transposed())
    cws.append(cws[3].transposed())
    imgs = [cw.packed_bf16() for cw in cws]          # first use: per-weight packs, registration
    ptrs = [im.data_ptr() for im in imgs]
    for w in ws:
        w.mul_(1.37).add_(0.01)                        # "optimizer step": the master weights move
    ops.bump_weights_epoch()
    n = ops.repack_bf16_images()
    assert n >= len(cws)
    torch.cuda.synchronize()
    for cw, im, ptr in zip(cws, imgs, ptrs):
        assert cw._packed16[0][3] == ops.weights_epoch() and cw._packed16[1].data_ptr() == ptr
        fresh = torch.empty_like(im)
        _lib.check(_lib.lib().set_pack_conv_weight_bf16(C.c_void_p(cw.raw().data_ptr()), C.c_void_p(fresh.data_ptr()), cw.Cout, cw.Cin,
                                                        cw.K, cw.base, cw.sco, cw.sci, cw.stap, None), "pack")
        torch.cuda.synchronize()
        assert torch.equal(im.view(torch.int16), fresh.view(torch.int16))
        assert cw.packed_bf16().data_ptr() == ptr     # current: no re-pack, same storage


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("B,T,L_,dcl,fuse", [(2, 800, 8, 1, 4), (3, 333, 7, 1, 4), (2, 500, 6, 2, 3), (1, 97, 5, 1, 5), (2, 128, 4, 2, 2), (2, 800, 20, 1, 10)])
def test_fused_layer_groups_equal_one_launch_per_layer(dev, monkeypatch, tile, B, T, L_, dcl, fuse):
    """set_diffnet_layers_fwd_bf16 (the tile stays on chip for `fuse` consecutive layers, tile - 2 H stored frames per tile; both tile
    widths: 64 frames with x' and the skip sum in registers, 128 frames with two GEMM-1 passes and the skip rows added at the L2) against
    one set_diffnet_layer_fwd_bf16 launch per layer on the same images: same arithmetic per frame -> x and the skip sum are
    bit-identical, incl. ragged T (partial last tile), a
    group shorter than `fuse` at the end, dilation cycles, a 10-layer group."""
    import ctypes as C
    from set_amd import _lib
    monkeypatch.setenv("SET_AMD_BF16_FUSE_TILE", str(tile))
    Lb = _lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + T + L_)
    n_img = Lb.set_diffnet_layer_bf16_image_size()
    imgs = torch.empty(L_, n_img, dtype=torch.bfloat16, device=dev)
    for l in range(L_):
        wd = (torch.randn(512, 256, 3, generator=g) * 0.03).to(dev)
        wc = (torch.randn(512, 192, generator=g) * 0.05).to(dev)
        wo = (torch.randn(512, 256, generator=g) * 0.05).to(dev)
        _lib.check(Lb.set_pack_diffnet_layer_bf16(wd.data_ptr(), wc.data_ptr(), wo.data_ptr(), imgs[l].data_ptr(), None), "pack")
    x0 = torch.randn(B, 256, T, generator=g).to(dev)
    cond = torch.randn(B, 192, T, generator=g).to(dev)
    dstep = (torch.randn(L_, B, 256, generator=g) * 0.3).to(dev)
    bd, bc, bo = ((torch.randn(L_, 512, generator=g) * 0.1).to(dev) for _ in range(3))

    def per_layer():
        cur, nxt, skip = x0.clone(), torch.empty_like(x0), torch.full_like(x0, float("nan"))
        a = _lib.SetDiffnetLayerBf16Args()
        for l in range(L_):
            a.x_in, a.x_out, a.skip, a.cond = cur.data_ptr(), nxt.data_ptr(), skip.data_ptr(), cond.data_ptr()
            a.dstep, a.img = dstep[l].data_ptr(), imgs[l].data_ptr()
            a.b_dil, a.b_cond, a.b_out = bd[l].data_ptr(), bc[l].data_ptr(), bo[l].data_ptr()
            a.d_bs, a.d_cs, a.B, a.T, a.dil, a.first = 256, 1, B, T, 1 << (l % dcl), int(l == 0)
            _lib.check(Lb.set_diffnet_layer_fwd_bf16(C.byref(a), None), "layer")
            cur, nxt = nxt, cur
        return cur, skip

    def fused():
        cur, nxt, skip = x0.clone(), torch.full_like(x0, float("nan")), torch.full_like(x0, float("nan"))
        n_ws = Lb.set_diffnet_layers_bf16_scratch_floats(B, T, 0, fuse, dcl)
        assert n_ws > 0
        ws = torch.full((n_ws,), float("nan"), device=dev)
        a = _lib.SetDiffnetLayersBf16Args()
        for l in range(0, L_, fuse):
            a.x_in, a.x_out, a.skip, a.cond = cur.data_ptr(), nxt.data_ptr(), skip.data_ptr(), cond.data_ptr()
            a.dstep, a.img = dstep.data_ptr(), imgs[l].data_ptr()
            a.b_dil, a.b_cond, a.b_out = bd[l].data_ptr(), bc[l].data_ptr(), bo[l].data_ptr()
            a.scratch, a.scratch_floats = ws.data_ptr(), n_ws
            a.d_bs, a.d_cs, a.d_ls = 256, 1, B * 256
            a.B, a.T, a.l0, a.nl, a.dilation_cycle_length, a.first = B, T, l, min(fuse, L_ - l), dcl, int(l == 0)
            _lib.check(Lb.set_diffnet_layers_fwd_bf16(C.byref(a), None), "layers")
            cur, nxt = nxt, cur
        return cur, skip

    x_ref, s_ref = per_layer()
    x_f, s_f = fused()
    torch.cuda.synchronize()
    assert torch.isfinite(x_f).all() and torch.isfinite(s_f).all()
    assert torch.equal(x_f, x_ref)
    assert torch.equal(s_f, s_ref)
