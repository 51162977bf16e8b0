"""HiFi-GAN generator forward on the HIP kernels.

`HifiGanGenerator(h)` takes the same config dict and exposes the same weight-normed parameter names
(`*.weight_g`, `*.weight_v`, `*.bias`) as modules/vocoder/hifigan/hifigan.py:101-124, so `model_gen`
checkpoints load with strict=True.  forward(x[B,80,T]) -> [B,1,T*prod(upsample_rates)]  (:126-142).

Kernel plan: weight norm is folded on the device when the parameters change (set_weight_norm_fold);
every Conv1d is one fused set_conv1d launch (leaky-ReLU prologue, bias + residual epilogue), and each
lrelu -> conv(k, d) -> lrelu -> conv(k, 1) -> +x iteration of a ResBlock1 with <= 128 channels is ONE launch whose
intermediate stays in LDS (set_resblock_pair_x2);
ConvTranspose1d runs as `stride` polyphase stride-1 convolutions writing interleaved outputs; the MRF
sum / mean is folded into the epilogue of each ResBlock's final conv (accumulate + out_div).
"""
import torch
from torch import nn
from torch.nn.utils import weight_norm

from . import ops

LRELU_SLOPE = 0.1


def _padding(k, d=1):
    return int((k * d - d) / 2)


def _init_conv(m, std=0.01):
    m.weight.data.normal_(0.0, std)  # hifigan.py:11-14
    return m


class _WNConv:
    """Folded-weight cache for one weight-normed Conv1d / ConvTranspose1d container."""

    def __init__(self, mod, transposed=False, stride=1):
        self.mod, self.transposed, self.stride = mod, transposed, stride
        self._w, self._key = None, None
        self._cw = None
        self._phases = {}

    def folded(self):
        g, v = self.mod.weight_g.data, self.mod.weight_v.data
        key = (g.data_ptr(), g._version, v.data_ptr(), v._version)
        if self._w is None or key != self._key:
            self._w, self._key = ops.weight_norm_fold(g.contiguous(), v.contiguous()), key
            self._cw, self._phases = None, {}
        return self._w

    def conv_weight(self):
        w = self.folded()
        if self._cw is None:
            cout, cin, k = w.shape
            self._cw = ops.ConvWeight((self, "_w"), cout, cin, k)
        return self._cw


class _ResBlockBase(nn.Module):
    def _wn(self, ch, k, d):
        return _init_conv(nn.Conv1d(ch, ch, k, 1, dilation=d, padding=_padding(k, d)))


class ResBlock1(_ResBlockBase):
    """hifigan.py:27-58: 3 x [lrelu -> conv(k, d) -> lrelu -> conv(k, 1) -> + x]."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.k, self.dil = kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([weight_norm(self._wn(channels, kernel_size, d)) for d in dilation[:3]])
        self.convs2 = nn.ModuleList([weight_norm(self._wn(channels, kernel_size, 1)) for _ in range(3)])
        self._c1 = [_WNConv(m) for m in self.convs1]
        self._c2 = [_WNConv(m) for m in self.convs2]

    def run(self, x, **last):
        """`last` = out / accumulate / out_div of the block's FINAL conv (the MRF sum lives in its epilogue)."""
        k = self.k
        n = len(self._c1)
        for i, (c1, c2, d) in enumerate(zip(self._c1, self._c2, self.dil)):
            if ops.resblock_pair_eligible(x.shape[1], k, d, x.shape[2]):
                # lrelu -> conv(k, d) -> lrelu -> conv(k, 1) -> + x as ONE launch, intermediate in LDS (csrc/resblock_x2.hip);
                # bit-identical to the two launches below on the same (split-operand) arithmetic
                x = ops.resblock_pair(x, c1.conv_weight(), c1.mod.bias.data, c2.conv_weight(), c2.mod.bias.data, d,
                                      slope=LRELU_SLOPE, **(last if i == n - 1 else {}))
                continue
            t = ops.conv1d(x, c1.conv_weight(), c1.mod.bias.data, dil=d, pad=_padding(k, d), pro="lrelu",
                           pro_param=LRELU_SLOPE)
            x = ops.conv1d(t, c2.conv_weight(), c2.mod.bias.data, dil=1, pad=_padding(k, 1), pro="lrelu",
                           pro_param=LRELU_SLOPE, res=x, **(last if i == n - 1 else {}))
        return x


class ResBlock2(_ResBlockBase):
    """hifigan.py:67-84: 2 x [lrelu -> conv(k, d) -> + x]."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3)):
        super().__init__()
        self.k, self.dil = kernel_size, tuple(dilation)
        self.convs = nn.ModuleList([weight_norm(self._wn(channels, kernel_size, d)) for d in dilation[:2]])
        self._c = [_WNConv(m) for m in self.convs]

    def run(self, x, **last):
        k = self.k
        n = len(self._c)
        for i, (c, d) in enumerate(zip(self._c, self.dil)):
            x = ops.conv1d(x, c.conv_weight(), c.mod.bias.data, dil=d, pad=_padding(k, d), pro="lrelu",
                           pro_param=LRELU_SLOPE, res=x, **(last if i == n - 1 else {}))
        return x


class HifiGanGenerator(nn.Module):
    def __init__(self, h, c_out=1):
        super().__init__()
        self.h = h
        self.num_kernels = len(h["resblock_kernel_sizes"])
        self.num_upsamples = len(h["upsample_rates"])
        c0 = h["upsample_initial_channel"]
        self.conv_pre = weight_norm(nn.Conv1d(80, c0, 7, 1, padding=3))
        resblock = ResBlock1 if str(h["resblock"]) == "1" else ResBlock2
        self.ups = nn.ModuleList()
        self._up_cfg = []
        for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
            c_cur = c0 // (2 ** (i + 1))
            self.ups.append(weight_norm(_init_conv(nn.ConvTranspose1d(c_cur * 2, c_cur, k, u, padding=(k - u) // 2))))
            self._up_cfg.append((c_cur * 2, c_cur, k, u, (k - u) // 2))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
                self.resblocks.append(resblock(h, ch, k, d))
        self.conv_post = weight_norm(_init_conv(nn.Conv1d(ch, c_out, 7, 1, padding=3)))
        self._pre = _WNConv(self.conv_pre)
        self._post = _WNConv(self.conv_post)
        self._ups = [_WNConv(m, transposed=True) for m in self.ups]

    @torch.no_grad()
    def forward(self, x, f0=None):
        """The wide convolutions run on the two-piece fp16 kernel (fp32 operands split into two fp16 values, three MFMAs per
        product, fp32 accumulate: fp32-equivalent results, csrc/conv_x2.hip; SET_AMD_VOCODER_SPLIT=0 keeps the fp32 MFMA
        kernels).  Should an activation leave the fp16 range of the splitting (|x| >= 32768; never seen with real
        checkpoints), the kernels raise a flag instead of overflowing and the forward is repeated on the fp32 kernels."""
        import os
        if os.environ.get("SET_AMD_VOCODER_SPLIT", "1") != "0":
            with ops.split_convs():
                y = self._forward(x)
            # one read-back per forward (the flag is sticky and cleared when it is read as set: a stale one from another
            # caller costs a spurious repeat, never a wrong result)
            if not ops.conv_x2_range_flag(reset=True):
                return y
            import warnings
            warnings.warn("HifiGanGenerator: activation outside the fp16 split range; repeating on the fp32 kernels")
        return self._forward(x)

    def _forward(self, x):
        x = x.contiguous()
        x = ops.conv1d(x, self._pre.conv_weight(), self.conv_pre.bias.data, pad=3)
        for i in range(self.num_upsamples):
            cin, cout, k, u, P = self._up_cfg[i]
            up = self._ups[i]
            up.folded()
            x = ops.conv_transpose1d(x, (up, "_w"), self.ups[i].bias.data, cin, cout, k, u, P,
                                     pro="lrelu", pro_param=LRELU_SLOPE, cache=up._phases)
            # MRF (hifigan.py:131-137): xs = rb_0(x); xs += rb_j(x) ...; x = xs / num_kernels.  The running sum lives in
            # the epilogue of each ResBlock's final conv (first block stores, the others accumulate, the last one also
            # divides): same summation order and the same division as the reference, no separate pass over [B,C,T].
            nk = self.num_kernels
            xs = torch.empty_like(x)
            for j in range(nk):
                self.resblocks[i * nk + j].run(x, out=xs, accumulate=j > 0,
                                               out_div=float(nk) if (j == nk - 1 and nk > 1) else 0.0)
            x = xs
        # final leaky_relu uses the default slope 0.01 (hifigan.py:138), then conv_post, tanh
        return ops.conv1d(x, self._post.conv_weight(), self.conv_post.bias.data, pad=3, pro="lrelu", pro_param=0.01,
                          act="tanh")

    def remove_weight_norm(self):
        """hifigan.py:144-151 surface; weight norm is already folded on the device at run time."""
        return None
