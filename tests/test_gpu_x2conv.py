"""The two-piece fp16 conv kernel (SET_IMPL_F16X2, csrc/conv_x2.hip): fp32 operands split into two fp16 values, three fp16
MFMAs per product, fp32 accumulate.  Against the fp32 MFMA kernel (same SetConv1dArgs semantics) and against fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import set_amd  # noqa: F401
    return torch.device("cuda:0")


def _w(dev, Cout, Cin, K, seed, scale=None):
    from set_amd import ops
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(Cout, Cin, K, generator=g) * (scale if scale is not None else (Cin * K) ** -0.5)).to(dev)
    return w, ops.ConvWeight(w, Cout, Cin, K)


CASES = [  # B, Cin, Cout, K, dil, T, extras
    (2, 256, 256, 3, 1, 300, {}),
    (2, 256, 256, 11, 5, 257, dict(pro="lrelu", pro_param=0.1)),
    (1, 128, 128, 7, 3, 1000, dict(pro="lrelu", pro_param=0.1, res=True)),
    (3, 64, 64, 11, 1, 515, dict(pro="lrelu", pro_param=0.1, res=True, accumulate=True, out_div=3.0)),
    (2, 32, 32, 3, 5, 700, dict(pro="lrelu", pro_param=0.1, res=True, accumulate=True)),
    (2, 80, 512, 7, 1, 130, {}),
    (1, 512, 256, 2, -1, 100, dict(pro="lrelu", pro_param=0.1)),        # one polyphase branch: negative dilation, pad 0
    (2, 192, 384, 9, 1, 65, dict(act="gelu", mask=True)),
    (1, 100, 70, 5, 2, 64, dict(act="relu", alpha=0.5)),                # ragged channel counts (padded rows / channels)
]


@pytest.mark.parametrize("case", CASES)
def test_f16x2_conv_matches_fp32_conv_and_fp64(dev, case):
    from set_amd import ops
    import torch.nn.functional as F
    B, Cin, Cout, K, dil, T, ex = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + K)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w, cw = _w(dev, Cout, Cin, K, 5)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    pad = (K - 1) * abs(dil) // 2 if dil > 0 else 0
    T_out = T + 2 * pad - dil * (K - 1) if dil > 0 else T + K - 1
    kw = dict(dil=dil, pad=pad, pro=ex.get("pro", "none"), pro_param=ex.get("pro_param", 0.0), act=ex.get("act", "none"),
              alpha=ex.get("alpha", 1.0))
    if dil < 0:
        kw.update(T_iter=T + K - 1, T_out=T_out)
    res = torch.randn(B, Cout, T_out, generator=g).to(dev) if ex.get("res") else None
    mask = (torch.rand(B, T_out, generator=g) > 0.2).float().to(dev) if ex.get("mask") else None
    prev = torch.randn(B, Cout, T_out, generator=g).to(dev)

    def run(impl):
        out = prev.clone()
        ops.conv1d(x, cw, bias, res=res, mask=mask, out=out, accumulate=bool(ex.get("accumulate")),
                   out_div=ex.get("out_div", 0.0), impl=impl, **kw)
        return out

    ops.conv_x2_range_flag(reset=True)
    y32, y2 = run("mfma"), run("f16x2")
    assert not ops.conv_x2_range_flag()
    scale = float(y32.abs().max())
    assert float((y2 - y32).abs().max()) < 2e-5 * max(1.0, scale), case
    if ex.get("act", "none") in ("none", "relu") and dil > 0:
        xd = x.double()
        if kw["pro"] == "lrelu":
            xd = F.leaky_relu(xd, kw["pro_param"])
        yd = (F.conv1d(xd, w.double(), None, padding=pad, dilation=dil) + bias.double()[None, :, None]) * kw["alpha"]
        if kw["act"] == "relu":
            yd = yd.clamp_min(0)
        if res is not None:
            yd = yd + res.double()
        if ex.get("accumulate"):
            yd = yd + prev.double()
            if ex.get("out_div"):
                yd = yd / ex["out_div"]
        e32, e2 = float((y32.double() - yd).abs().max()), float((y2.double() - yd).abs().max())
        print("%s: max err vs fp64: fp32 MFMA kernel %.3e, f16x2 kernel %.3e" % (case[:6], e32, e2))
        assert e2 < 1.5 * e32 + 1e-7


def test_f16x2_conv_small_weights_and_range_flag(dev):
    """Weights of magnitude 1e-4 (their fp16 residuals would be subnormal without the pack-time power-of-two scale) keep
    fp32-level accuracy; an input beyond the fp16 range raises the sticky flag."""
    from set_amd import ops
    import torch.nn.functional as F
    B, C, K, T = 2, 128, 3, 400
    x = torch.randn(B, C, T, device=dev)
    w, cw = _w(dev, C, C, K, 9, scale=1e-4)
    ops.conv_x2_range_flag(reset=True)
    y2 = ops.conv1d(x, cw, None, pad=1, impl="f16x2")
    y32 = ops.conv1d(x, cw, None, pad=1, impl="mfma")
    yd = F.conv1d(x.double(), w.double(), None, padding=1)
    e32, e2 = float((y32.double() - yd).abs().max()), float((y2.double() - yd).abs().max())
    assert e2 < 1.5 * e32 + 1e-12, (e2, e32)
    assert not ops.conv_x2_range_flag()
    x[1, 5, 77] = 4.0e4
    ops.conv1d(x, cw, None, pad=1, impl="f16x2")
    assert ops.conv_x2_range_flag(reset=True)
    assert not ops.conv_x2_range_flag()


@pytest.mark.parametrize("Cin,Cout,k,u,P,T,B", [(512, 256, 16, 8, 4, 100, 2), (128, 64, 4, 2, 1, 333, 2), (64, 32, 4, 2, 1, 1000, 1),
                                                (256, 128, 8, 4, 2, 65, 3), (96, 48, 7, 3, 2, 200, 2)])
def test_all_phase_transposed_conv_matches_polyphase_fp32_and_fp64(dev, Cin, Cout, k, u, P, T, B):
    """nn.ConvTranspose1d with every output phase in one launch of the two-piece fp16 kernel (rows = (channel, phase), 16-byte
    stores when the stride is a multiple of 4) against the fp32 path (one strided-output conv per phase) and fp64 torch."""
    from set_amd import ops
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Cin + k)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w = (torch.randn(Cin, Cout, k, generator=g) * (Cin * k / u) ** -0.5).to(dev)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    y32 = ops.conv_transpose1d(x, w, bias, Cin, Cout, k, u, P, pro="lrelu", pro_param=0.1, cache={})
    ops.conv_x2_range_flag(reset=True)
    with ops.split_convs():
        y2 = ops.conv_transpose1d(x, w, bias, Cin, Cout, k, u, P, pro="lrelu", pro_param=0.1, cache={})
    assert not ops.conv_x2_range_flag()
    assert y2.shape == y32.shape
    yd = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), bias.double(), stride=u, padding=P)
    e32, e2 = float((y32.double() - yd).abs().max()), float((y2.double() - yd).abs().max())
    print("convT %d->%d k%d s%d: max err vs fp64: fp32 polyphase %.3e, all-phase f16x2 %.3e" % (Cin, Cout, k, u, e32, e2))
    assert float((y2 - y32).abs().max()) < 2e-5 * max(1.0, float(y32.abs().max()))
    assert e2 < 1.5 * e32 + 1e-7


PAIR_CASES = [  # B, C, K, dil, T, accumulate, out_div
    (2, 32, 3, 1, 700, False, 0.0),
    (2, 32, 11, 5, 1000, True, 3.0),      # widest receptive field of the 32-channel stage, MRF mean in the epilogue
    (1, 64, 7, 3, 515, True, 0.0),        # ragged T: last block partly out of range
    (3, 64, 3, 5, 256, False, 0.0),
    (1, 128, 3, 3, 300, True, 3.0),       # 128-row x 128-frame tiles
    (2, 128, 11, 1, 131, False, 0.0),
    (1, 48, 5, 2, 200, False, 0.0),       # channel count that is not a multiple of 32 (padded rows / channels)
    (1, 100, 7, 1, 64, True, 2.0),
    (2, 256, 3, 5, 200, True, 3.0),       # 256 rows x 64-frame tiles (3-tap pairs of the widest stage)
    (1, 160, 5, 1, 97, False, 0.0),
]


@pytest.mark.parametrize("case", PAIR_CASES)
def test_fused_resblock_pair_equals_two_convs(dev, case):
    """set_resblock_pair_x2 (lrelu -> conv(k, d) -> lrelu -> conv(k, 1) -> + x, intermediate in LDS, MRF accumulate /
    divide in the epilogue; hifigan.py:51-58,131-137) against the two set_conv1d(F16X2) launches it replaces -- the same
    products in the same order, the same fp32 intermediate before it is split: BIT-identical -- and against torch in
    fp64 (1e-5 of the output's size, the bar of the split-operand conv tests)."""
    from set_amd import ops
    import torch.nn.functional as F
    B, C, K, dil, T, accumulate, out_div = case
    g = torch.Generator().manual_seed(C * 13 + K * 5 + dil)
    x = torch.randn(B, C, T, generator=g).to(dev)
    w1, cw1 = _w(dev, C, C, K, 7)
    w2, cw2 = _w(dev, C, C, K, 8)
    b1 = (torch.randn(C, generator=g) * 0.1).to(dev)
    b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
    prev = torch.randn(B, C, T, generator=g).to(dev)
    p1, p2 = dil * (K - 1) // 2, (K - 1) // 2
    t = ops.conv1d(x, cw1, b1, dil=dil, pad=p1, pro="lrelu", pro_param=0.1, impl="f16x2")
    want = prev.clone()
    ops.conv1d(t, cw2, b2, dil=1, pad=p2, pro="lrelu", pro_param=0.1, res=x, out=want, accumulate=accumulate,
               out_div=out_div, impl="f16x2")
    got = prev.clone()
    ops.resblock_pair(x, cw1, b1, cw2, b2, dil, slope=0.1, out=got, accumulate=accumulate, out_div=out_div)
    torch.cuda.synchronize()
    assert not ops.conv_x2_range_flag(reset=True)
    assert torch.equal(got, want), float((got - want).abs().max())
    xd = x.double().cpu()
    td = F.conv1d(F.leaky_relu(xd, 0.1), w1.double().cpu(), b1.double().cpu(), dilation=dil, padding=p1)
    yd = F.conv1d(F.leaky_relu(td, 0.1), w2.double().cpu(), b2.double().cpu(), padding=p2) + xd
    if accumulate:
        yd = yd + prev.double().cpu()
    if out_div:
        yd = yd / out_div
    err = float((got.double().cpu() - yd).abs().max())
    assert err < 1e-5 * max(1.0, float(yd.abs().max())), err


def test_fused_resblock_pair_raises_the_range_flag(dev):
    """An activation outside the fp16 range of the splitting -- in x or in the intermediate -- sets the sticky flag that
    HifiGanGenerator.forward checks (it then repeats on the fp32 kernels)."""
    from set_amd import ops
    C, K, T = 32, 3, 128
    w1, cw1 = _w(dev, C, C, K, 7)
    w2, cw2 = _w(dev, C, C, K, 8)
    b = torch.zeros(C, device=dev)
    ops.conv_x2_range_flag(reset=True)
    x = torch.randn(1, C, T, device=dev)
    ops.resblock_pair(x, cw1, b, cw2, b, 1)
    assert not ops.conv_x2_range_flag(reset=True)
    x[0, 3, 50] = 4.0e4
    ops.resblock_pair(x, cw1, b, cw2, b, 1)
    assert ops.conv_x2_range_flag(reset=True)
    big = torch.full((C,), 5.0e4, device=dev)  # the intermediate: conv 1's bias alone leaves the range
    ops.resblock_pair(torch.randn(1, C, T, device=dev), cw1, big, cw2, b, 1)
    assert ops.conv_x2_range_flag(reset=True)
    assert not ops.conv_x2_range_flag(reset=True)


@pytest.mark.parametrize("cfg", [(3, 32, 1, 7, 3, 8192, "lrelu", "tanh"), (2, 17, 2, 3, 1, 4100, "none", "none"),
                                 (1, 64, 1, 9, 4, 4096, "lrelu", "none"), (2, 8, 2, 5, 0, 5000, "none", "relu")])
def test_few_output_channel_conv_matches_torch(dev, cfg):
    """SET_IMPL_FEWOUT (HiFi-GAN conv_post, hifigan.py:123,138-140: 32 -> 1 channels over 13 M samples): the streaming VALU
    kernel against torch fp64 and against the MFMA kernel it replaces on that shape; `auto` must pick it there."""
    from set_amd import ops
    import torch.nn.functional as F
    B, Cin, Cout, K, pad, T, pro, act = cfg
    g = torch.Generator().manual_seed(Cin * 3 + K)
    x = torch.randn(B, Cin, T, generator=g).to(dev)
    w, cw = _w(dev, Cout, Cin, K, 9)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    same = T + 2 * pad - (K - 1) == T
    if not same:
        with pytest.raises(Exception):  # not a same-length convolution: the kernel must refuse, not compute something else
            ops.conv1d(x, cw, bias, pad=pad, pro=pro, pro_param=0.01, act=act, impl="fewout")
        return
    got = ops.conv1d(x, cw, bias, pad=pad, pro=pro, pro_param=0.01, act=act, impl="fewout")
    ref_k = ops.conv1d(x, cw, bias, pad=pad, pro=pro, pro_param=0.01, act=act, impl="mfma")
    torch.cuda.synchronize()
    xd = x.double().cpu()
    xd = F.leaky_relu(xd, 0.01) if pro == "lrelu" else xd
    yd = F.conv1d(xd, w.double().cpu(), bias.double().cpu(), padding=pad)
    yd = torch.tanh(yd) if act == "tanh" else (torch.relu(yd) if act == "relu" else yd)
    assert float((got.double().cpu() - yd).abs().max()) < 2e-5 * max(1.0, float(yd.abs().max()))
    assert float((got - ref_k).abs().max()) < 2e-5 * max(1.0, float(yd.abs().max()))
    picked = ops._pick_impl(None, T, Cout, Cin, K, 1, 1, 0, False, True, pad)
    assert picked == "fewout", picked
