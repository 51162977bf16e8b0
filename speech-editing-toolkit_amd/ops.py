"""Tensor-level wrappers over the C ABI (include/set_amd.h).

torch is used for device memory, the current HIP stream and parameter storage
only; every arithmetic op below is a kernel in libset_amd.so.  All tensors must
be fp32 (or int64 for indices), contiguous, on a HIP device.  Nothing here has a
CPU implementation: calling an op without the library / without a GPU raises.
"""
import collections
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import (ACT, PRO, IMPL_BF16, IMPL_MFMA, IMPL_MFMA2, IMPL_NAIVE, SetConv1dArgs, SetDiffnetLayerArgs,
                   SetDiffLoopArgs, SetDiffnetStackArgs, check)

_DEFAULT_IMPL = os.environ.get("SET_AMD_CONV_IMPL", "auto")  # auto | naive | mfma
_WEIGHTS_EPOCH = 0  # bumped by in-place optimizer updates (they do not touch tensor version counters)


_COMPUTE_DTYPE = os.environ.get("SET_AMD_DTYPE", "f32")  # f32 | bf16: MFMA operand type of the conv / wgrad GEMMs


def set_compute_dtype(dtype):
    """'f32' (default; the parity path) or 'bf16': bf16 MFMA operands with fp32 accumulation, fp32 tensors in HBM and
    fp32 master weights (BASELINE configs[1]; what the reference gets from torch.autocast, trainer.py:325).  Applies to
    every conv / linear / weight-gradient GEMM large enough for the bf16 kernels; everything else is unchanged."""
    global _COMPUTE_DTYPE
    if dtype not in ("f32", "bf16"):
        raise ValueError("compute dtype must be 'f32' or 'bf16', got %r" % (dtype,))
    _COMPUTE_DTYPE = dtype


def compute_dtype():
    return _COMPUTE_DTYPE


def bump_weights_epoch():
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


def weights_epoch():
    return _WEIGHTS_EPOCH


import weakref  # noqa: E402

_BF16_IMAGES = weakref.WeakSet()   # ConvWeights that hold a bf16 image (training rows, compute dtype bf16)
_BF16_BATCH = [None, None, 0]      # (signature, device descriptor table, total elements) of the last batch re-pack


def repack_bf16_images():
    """Re-round EVERY live bf16 weight image from its fp32 master weight in ONE launch and mark the images current for the
    present weights epoch.  Called by the optimizer right after its step: the lazy per-weight check in `packed_bf16()` then
    finds nothing to do, instead of launching 100 - 170 tiny pack kernels (and allocating as many tensors) per training step.
    The descriptor table lives on the device and is rebuilt only when the set of images or their storage changes."""
    cws = [cw for cw in _BF16_IMAGES if cw._packed16 is not None]
    if not cws:
        return 0
    ws = [cw.raw() for cw in cws]
    dev = ws[0].device
    keep = [i for i, w in enumerate(ws) if w.device == dev and cws[i]._packed16[1].device == dev]
    cws, ws = [cws[i] for i in keep], [ws[i] for i in keep]
    sig = tuple((id(cw), w.data_ptr(), cw._packed16[1].data_ptr()) for cw, w in zip(cws, ws))
    if _BF16_BATCH[0] != sig:
        arr = (_lib.SetPackBf16Desc * len(cws))()
        start = 0
        for d, cw, w in zip(arr, cws, ws):
            wp = cw._packed16[1]
            d.w, d.wp = w.data_ptr(), wp.data_ptr()
            d.w_base, d.w_sco, d.w_sci, d.w_stap, d.start = cw.base, cw.sco, cw.sci, cw.stap, start
            d.Cout, d.Cin, d.K = cw.Cout, cw.Cin, cw.K
            d.CoutP, d.CinP = (cw.Cout + 127) // 128 * 128, (cw.Cin + 31) // 32 * 32
            assert wp.numel() == d.CoutP * d.K * d.CinP
            start += wp.numel()
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        _BF16_BATCH[0], _BF16_BATCH[1], _BF16_BATCH[2] = sig, raw, start
    check(_lib.lib().set_pack_conv_weights_bf16_batch(C.c_void_p(_BF16_BATCH[1].data_ptr()), len(cws), _BF16_BATCH[2], _stream()),
          "set_pack_conv_weights_bf16_batch")
    for cw, w in zip(cws, ws):
        cw._packed16 = ((w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH), cw._packed16[1])
    return len(cws)


_F32_IMAGES = weakref.WeakSet()    # ConvWeights that hold an fp32 image (set_pack_conv_weight / _v2)
_F32_TABLES = collections.OrderedDict()  # signature -> (device descriptor table, total elements, pinned host copy): the last few batch re-packs


def repack_f32_images():
    """Re-pack every fp32 weight image USED since the last call from its master weight in ONE launch and mark it current for the
    present weights epoch (called by the optimizer right after its step, like repack_bf16_images): the lazy check in `packed()` /
    `packed_v2()` then finds nothing to do instead of launching ~150 pack kernels of ~5 us per fp32 training step (1.1 ms of the
    compute stream's 32 ms at B = 32, T = 800).  An image that was not used stays stale and is re-packed when it is next asked for.
    Same kernel arithmetic (a copy into the image layout): same bits."""
    items = []
    for cw in _F32_IMAGES:
        if cw._used32 and cw._packed is not None:
            items.append((cw, 0, None, cw._packed, 0))
        for slot in cw._used2:
            ent = cw._packed2.get(slot)
            if ent is not None:
                items.append((cw, 1, slot, ent[1], ent[2]))
        cw._used32 = False
        cw._used2 = set()
    if not items:
        return 0
    ws = [it[0].raw() for it in items]
    dev = ws[0].device
    keep = [i for i, w in enumerate(ws) if w.device == dev and items[i][3].device == dev]
    # a deterministic order (the WeakSet's iteration order changes with the set of live images) and a small cache of tables: bucketed
    # ragged batches alternate between a few sets of used images (packed() vs packed_v2() by T) -- each set's table is built and uploaded
    # once, from pinned memory without a host sync (round-5 advisor item)
    order = sorted(range(len(keep)), key=lambda j: (items[keep[j]][3].data_ptr(), items[keep[j]][1], str(items[keep[j]][2])))
    items, ws = [items[keep[j]] for j in order], [ws[keep[j]] for j in order]
    sig = tuple((id(it[0]), it[1], it[2], w.data_ptr(), it[3].data_ptr()) for it, w in zip(items, ws))
    ent = _F32_TABLES.get(sig)
    if ent is not None:
        _F32_TABLES.move_to_end(sig)
    else:
        arr = (_lib.SetPackF32Desc * len(items))()
        start = 0
        for d, (cw, kind, slot, wp, dil), w in zip(arr, items, ws):
            d.w, d.wp = w.data_ptr(), wp.data_ptr()
            d.w_base, d.w_sco, d.w_sci, d.w_stap, d.start = cw.base, cw.sco, cw.sci, cw.stap, start
            d.Cout, d.Cin, d.K = cw.Cout, cw.Cin, cw.K
            n = _lib.lib().set_fill_pack_f32_desc(C.byref(d), kind, int(dil))
            assert n == wp.numel(), (n, wp.numel(), kind)
            start += n
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        if dev.type == "cuda":
            host = host.pin_memory()
        ent = _F32_TABLES[sig] = (host.to(dev, non_blocking=True), start, host)
        while len(_F32_TABLES) > 8:
            _F32_TABLES.popitem(last=False)
    check(_lib.lib().set_pack_conv_weights_f32_batch(C.c_void_p(ent[0].data_ptr()), len(items), ent[1], _stream()),
          "set_pack_conv_weights_f32_batch")
    for (cw, kind, slot, wp, dil), w in zip(items, ws):
        key = (w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH)
        if kind == 0:
            cw._key = key
        else:
            cw._packed2[slot] = (key, wp, dil)
    return len(items)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


import threading  # noqa: E402

_STREAM_TLS = threading.local()  # .leaf: raw handle of the leaf stream while autograd_ops.leaf_work is open IN THIS THREAD (kernels only)


def _stream():
    """The current torch HIP stream of the current device as a raw handle.  (torch.cuda.current_stream() builds a Stream
    object through several Python layers -- ~10 us, once per kernel launch: ~3 ms of a 1,400-launch training step.)"""
    leaf = getattr(_STREAM_TLS, "leaf", None)
    if leaf is not None:
        return leaf
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    """Device address of a tensor as a plain int (ctypes converts it for a `void *` parameter; wrapping it in c_void_p here cost 0.25 us per
    pointer, ~3,000 pointers per training step)."""
    return None if t is None else t.data_ptr()


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the set_amd path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _f(t, name="tensor"):
    _chk(t, torch.float32, name)
    return t


def _fv(t, name="tensor"):
    """fp32 GPU tensor that may be a [B,C,T] view: only the last stride must be 1."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the set_amd path has no CPU fallback" % name)
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1:
        raise ValueError("%s must be an fp32 [B,C,T] tensor/view with unit last stride" % name)
    return t


def _i(t, name="index"):
    _chk(t, torch.int64, name)
    return t


# --------------------------------------------------------------------------
# weights: raw parameter + lazily packed MFMA image, re-packed when the parameter changes
# --------------------------------------------------------------------------
class ConvWeight:
    """Weight operand of set_conv1d.

    `w` is the raw tensor; (base, sco, sci, stap) address W[co][ci][tap] inside it, so the same class serves
    nn.Conv1d [Cout,Cin,K], nn.Linear [Cout,Cin] (K=1) and one polyphase branch of nn.ConvTranspose1d
    [Cin,Cout,k] (base=p, sco=k, sci=Cout*k, stap=u)."""

    def __init__(self, getter, Cout, Cin, K, base=0, sco=None, sci=None, stap=1):
        # `getter`: (owner, "attr.path") -- the weight is looked up on the owner at every use, so a deepcopy / pickle of
        # the model (EMA or teacher copies) follows its own parameters (a closure over the original module would keep
        # computing with the original's weights); a callable or a tensor is accepted too.
        self._getter = getter
        self.Cout, self.Cin, self.K = int(Cout), int(Cin), int(K)
        self.base = int(base)
        self.sco = int(sco if sco is not None else Cin * K)
        self.sci = int(sci if sci is not None else K)
        self.stap = int(stap)
        self._packed = None
        self._key = None
        self._packed2 = {}
        self._packed16 = None
        self._used32, self._used2 = False, set()  # fp32 images asked for since the last repack_f32_images()

    def _resolve(self):
        g = self._getter
        if isinstance(g, tuple):
            obj = g[0]
            for name in g[1].split("."):
                # nn.Module.__getattr__ is the slow path of attribute lookup (three dict probes behind a failed normal lookup, ~0.6 us per
                # level, two levels per weight, several uses per conv): look into the module's own tables first
                d = getattr(obj, "__dict__", None)
                if d is None:
                    obj = getattr(obj, name)
                    continue
                sub = d.get("_modules")
                if sub is not None and name in sub:
                    obj = sub[name]
                    continue
                par = d.get("_parameters")
                if par is not None and name in par and par[name] is not None:
                    obj = par[name]
                    continue
                obj = getattr(obj, name)
            return obj
        return g() if callable(g) else g

    def raw(self):
        return _f(self._resolve(), "weight")

    def transposed(self):
        """W'[ci][co][tap] = W[co][ci][tap]: the weight operand of the input-gradient convolution."""
        if getattr(self, "_tr", None) is None:
            assert self.stap == 1
            self._tr = ConvWeight(self._getter, self.Cin, self.Cout, self.K, base=self.base, sco=self.sci, sci=self.sco,
                                  stap=1)
        return self._tr

    def packed(self):
        w = self.raw()
        key = (w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH)
        if self._packed is None or self._key != key:
            if self._packed is not None and self._packed.device == w.device:
                wp = self._packed
            else:
                n = _lib.lib().set_packed_conv_weight_size(self.Cout, self.Cin, self.K)
                wp = torch.empty(n, dtype=torch.float32, device=w.device)
            check(_lib.lib().set_pack_conv_weight(_p(w), _p(wp), self.Cout, self.Cin, self.K, self.base, self.sco,
                                                  self.sci, self.stap, _stream()), "set_pack_conv_weight")
            self._packed, self._key = wp, key
            _F32_IMAGES.add(self)  # from now on the optimizer re-packs it with the others it used (repack_f32_images)
        self._used32 = True
        return self._packed


    def packed_bf16(self):
        """bf16 image for SET_IMPL_BF16 (re-rounded from the fp32 master weights whenever they change)."""
        w = self.raw()
        key = (w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH)
        if self._packed16 is None or self._packed16[0] != key:
            if self._packed16 is not None and self._packed16[1].device == w.device:
                wp = self._packed16[1]  # re-rounded in place once per optimizer step (stream order keeps earlier readers safe)
            else:
                n = _lib.lib().set_packed_conv_weight_bf16_size(self.Cout, self.Cin, self.K)
                wp = torch.empty(n, dtype=torch.bfloat16, device=w.device)
            check(_lib.lib().set_pack_conv_weight_bf16(_p(w), _p(wp), self.Cout, self.Cin, self.K, self.base, self.sco,
                                                       self.sci, self.stap, _stream()), "set_pack_conv_weight_bf16")
            self._packed16 = (key, wp)
            _BF16_IMAGES.add(self)  # from now on the optimizer re-rounds it with all the others (repack_bf16_images)
        return self._packed16[1]

    def packed_x2(self):
        """Two-piece fp16 image for SET_IMPL_F16X2 (re-split from the fp32 master weights whenever they change); the weights
        are scaled by a power of two so that max |w| lands in [8, 16): one host read-back per weight version."""
        w = self.raw()
        key = (w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH)
        if getattr(self, "_packed_x2", None) is None or self._packed_x2[0] != key:
            import math
            n = _lib.lib().set_packed_conv_weight_x2_size(self.Cout, self.Cin, self.K)
            wp = torch.empty(n, dtype=torch.float16, device=w.device)
            m = float(w.abs().max())
            k = max(-60, min(60, 4 - math.frexp(m)[1])) if m > 0 and math.isfinite(m) else 0
            check(_lib.lib().set_pack_conv_weight_x2(_p(w), _p(wp), self.Cout, self.Cin, self.K, self.base, self.sco, self.sci,
                                                     self.stap, k, _stream()), "set_pack_conv_weight_x2")
            self._packed_x2 = (key, wp)
        return self._packed_x2[1]

    def packed_v2(self, dil):
        """Image for the big-tile kernel (depends on |dil| through the LDS chunking)."""
        w = self.raw()
        key = (w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH)
        halo = (self.K - 1) * abs(dil)
        slot = 128 if (64 + halo) * 256 * 4 > 96 * 1024 else 256
        ent = self._packed2.get(slot)
        if ent is None or ent[0] != key:
            if ent is not None and ent[1].device == w.device:
                wp = ent[1]  # re-packed in place (stream order keeps earlier readers safe)
            else:
                n = _lib.lib().set_packed_conv_weight_v2_size(self.Cout, self.Cin, self.K)
                wp = torch.empty(n, dtype=torch.float32, device=w.device)
            check(_lib.lib().set_pack_conv_weight_v2(_p(w), _p(wp), self.Cout, self.Cin, self.K, int(dil), self.base,
                                                     self.sco, self.sci, self.stap, _stream()), "set_pack_conv_weight_v2")
            ent = (key, wp, int(dil))
            self._packed2[slot] = ent
            _F32_IMAGES.add(self)
        self._used2.add(slot)
        return ent[1]


def f16x2_eligible(T_iter, Cout, Cin, K, dil):
    """Shapes the two-piece fp16 conv kernel takes (and pays for): wide enough, receptive field <= 128 frames."""
    # (K = 1: two k-steps per 32-channel chunk between barriers -- measured slower than the fp32 kernel; not taken)
    return T_iter >= 64 and Cout >= 32 and Cin >= 32 and K >= 2 and Cin * K >= 96 and (K - 1) * abs(dil) <= 128


def conv_x2_range_flag(reset=True):
    """True if an activation left the fp16 range of the F16X2 splitting since the last reset (synchronises the device)."""
    v = C.c_int32(0)
    check(_lib.lib().set_conv_x2_range_flag(C.byref(v), int(bool(reset))), "set_conv_x2_range_flag")
    return bool(v.value)


def bf16_eligible(T_iter, Cout, Cin, K, dil, out_stride=1, out_off=0):
    """Shapes the bf16 kernels take: enough reduction depth and rows for a 64x64 wave tile to pay, stride-1 output."""
    return (T_iter >= 32 and Cout >= 32 and Cin * K >= 64 and out_stride == 1 and out_off == 0
            and (K - 1) * abs(dil) <= 128)


_AUTO_SPLIT = [False]


class split_convs:
    """with split_convs(): ...  -- inside, `auto` convolutions that are wide enough run on the two-piece fp16 kernel
    (SET_IMPL_F16X2: fp32-equivalent results on the 16-bit MFMA pipe).  The caller checks conv_x2_range_flag() afterwards and
    repeats the computation outside the scope if an activation left the fp16 range."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _AUTO_SPLIT[0] = _AUTO_SPLIT[0], self.on

    def __exit__(self, *exc):
        _AUTO_SPLIT[0] = self.prev


def _pick_impl(impl, T_iter, Cout=0, Cin=0, K=1, dil=1, out_stride=1, out_off=0, chan_add=False, plain=False, pad=0):
    impl = impl or _DEFAULT_IMPL
    if impl == "auto":
        if T_iter < 16:
            return "naive"
        # one or two output channels over a long sequence (HiFi-GAN conv_post): a streaming VALU kernel, not a GEMM
        if plain and Cout <= 2 and K <= 9 and dil == 1 and 0 <= pad <= 4 and K - pad <= 5 and out_stride == 1 and \
                T_iter >= 4096 and T_iter % 4 == 0 and _COMPUTE_DTYPE != "bf16":
            return "fewout"
        if _AUTO_SPLIT[0] and not chan_add and f16x2_eligible(T_iter, Cout, Cin, K, dil):
            return "f16x2"
        if _COMPUTE_DTYPE == "bf16" and bf16_eligible(T_iter, Cout, Cin, K, dil, out_stride, out_off):
            return "bf16"
        # big-tile kernel: measured (tools/conv_probe.py, profiles/r01_conv_probe.log) +2 % for the 3-tap convs with a
        # deep reduction (512 -> 256: the input-gradient convs of the DiffNet layers) on short sequences; the small-tile
        # kernel is equal or better everywhere else since its epilogue fetches its operands in batches and its main loop
        # is double-buffered (1x1 convs +3-12 %, 192 -> 192 k5 +10 %, 192 -> 384 k9 +45 %), and it balances better on
        # long sequences (HiFi-GAN)
        if Cout >= 192 and Cin >= 384 and K >= 3 and 32 <= T_iter <= 2048 and (K - 1) * abs(dil) <= 16:
            return "mfma2"
        return "mfma"
    return impl


def conv1d(x, weight, bias=None, *, dil=1, pad=0, pro="none", pro_param=0.0, act="none", act_param=0.0, alpha=1.0,
           res=None, mask=None, in_chan_add=None, out=None, accumulate=False, impl=None,
           T_iter=None, T_out=None, out_stride=1, out_off=0, out_div=0.0):
    """Generic fused conv (see SetConv1dArgs in set_amd.h).  x [B,Cin,T_in] -> out [B,Cout,T_out]."""
    _f(x, "x")
    assert isinstance(weight, ConvWeight)
    B, Cin, T_in = x.shape
    assert Cin == weight.Cin, (Cin, weight.Cin)
    if T_out is None:
        T_out = T_in + 2 * pad - dil * (weight.K - 1) if dil > 0 else T_in
    if T_iter is None:
        T_iter = T_out
    if out is None:
        out = torch.empty(B, weight.Cout, T_out, dtype=torch.float32, device=x.device)
    _fv(out, "out")
    plain = res is None and mask is None and in_chan_add is None and not accumulate and T_out == T_in and T_iter == T_out and \
        out.is_contiguous() and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
    impl = _pick_impl(impl, T_iter, weight.Cout, Cin, weight.K, dil, out_stride, out_off, in_chan_add is not None, plain, pad)
    a = SetConv1dArgs()
    a.inp = x.data_ptr()
    if impl == "mfma2":
        a.w = weight.packed_v2(dil).data_ptr()
    elif impl == "bf16":
        a.w = weight.packed_bf16().data_ptr()
    elif impl == "f16x2":
        a.w = weight.packed_x2().data_ptr()
    else:  # "mfma": packed image; "naive" / "fewout": the raw weight
        a.w = (weight.packed() if impl == "mfma" else weight.raw()).data_ptr()
    a.bias = _f(bias, "bias").data_ptr() if bias is not None else None
    a.res = _fv(res, "res").data_ptr() if res is not None else None
    a.mask = _f(mask, "mask").data_ptr() if mask is not None else None
    a.in_chan_add = _f(in_chan_add, "in_chan_add").data_ptr() if in_chan_add is not None else None
    a.out = out.data_ptr()
    a.in_bs, a.in_cs = Cin * T_in, T_in
    a.out_bs, a.out_cs = out.stride(0), out.stride(1)
    if res is not None:
        a.res_bs, a.res_cs = res.stride(0), res.stride(1)
    a.w_base, a.w_sco, a.w_sci, a.w_stap = weight.base, weight.sco, weight.sci, weight.stap
    a.B, a.Cin, a.Cout, a.K, a.dil, a.pad = B, Cin, weight.Cout, weight.K, dil, pad
    a.T_in, a.T_iter, a.T_out, a.out_stride, a.out_off = T_in, T_iter, T_out, out_stride, out_off
    a.pro, a.act, a.accumulate = PRO[pro], ACT[act], int(bool(accumulate))
    a.impl = {"mfma": IMPL_MFMA, "mfma2": IMPL_MFMA2, "bf16": IMPL_BF16, "f16x2": _lib.IMPL_F16X2,
              "fewout": _lib.IMPL_FEWOUT}.get(impl, IMPL_NAIVE)
    a.pro_param, a.act_param, a.alpha = float(pro_param), float(act_param), float(alpha)
    a.out_div = float(out_div)
    assert not (out_div and not accumulate)
    if CONV_EVENTS is not None and impl in ("bf16", "mfma", "mfma2"):  # measurement hook (bench.py): hipEvents around every MFMA conv launch, keyed by shape
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().set_conv1d(C.byref(a), _stream()), "set_conv1d")
        e1.record()
        CONV_EVENTS.append(((Cin, weight.Cout, weight.K, B, T_iter, impl), e0, e1))
        return out
    check(_lib.lib().set_conv1d(C.byref(a), _stream()), "set_conv1d")
    return out


CONV_EVENTS = None  # bench.py sets a list: ((Cin, Cout, K, B, T, impl), start, end) per bf16 / fp32 MFMA conv launch of the timed steps


def resblock_pair_eligible(C, K, dil, T):
    """Shapes the fused ResBlock-pair kernel takes while the split-operand scope is on (csrc/resblock_x2.hip)."""
    return bool(_AUTO_SPLIT[0]) and _DEFAULT_IMPL == "auto" and os.environ.get("SET_AMD_RESBLOCK_FUSED", "1") != "0" and \
        _lib.lib().set_resblock_pair_x2_supported(int(C), int(K), int(dil), int(T)) == 0


def resblock_pair(x, w1, b1, w2, b2, dil, *, slope=0.1, out=None, accumulate=False, out_div=0.0):
    """out = c2(lrelu(c1(lrelu(x)))) + x (+ out, / out_div): one iteration of HiFi-GAN's ResBlock1 loop (hifigan.py:51-58)
    as ONE launch on the two-piece fp16 operands; the intermediate never leaves LDS.  Bit-identical to the two conv1d
    launches with impl='f16x2'.  w1 / w2: ConvWeight [C, C, K]."""
    _f(x, "x")
    assert isinstance(w1, ConvWeight) and isinstance(w2, ConvWeight)
    B, Cc, T = x.shape
    assert (w1.Cout, w1.Cin, w2.Cout, w2.Cin) == (Cc, Cc, Cc, Cc) and w1.K == w2.K
    if out is None:
        out = torch.empty_like(x)
    _fv(out, "out")
    assert out.data_ptr() != x.data_ptr() and not (out_div and not accumulate)
    a = _lib.SetResblockPairArgs()
    a.x, a.out = x.data_ptr(), out.data_ptr()
    a.w1, a.w2 = w1.packed_x2().data_ptr(), w2.packed_x2().data_ptr()
    a.b1, a.b2 = _f(b1, "b1").data_ptr(), _f(b2, "b2").data_ptr()
    a.x_bs, a.x_cs, a.out_bs, a.out_cs = Cc * T, T, out.stride(0), out.stride(1)
    a.B, a.C, a.K, a.dil, a.T = B, Cc, w1.K, int(dil), T
    a.accumulate, a.slope, a.out_div = int(bool(accumulate)), float(slope), float(out_div)
    check(_lib.lib().set_resblock_pair_x2(C.byref(a), _stream()), "set_resblock_pair_x2")
    return out


def conv_transpose1d(x, w_getter, bias, Cin, Cout, k, stride, padding, *, pro="none", pro_param=0.0, impl=None,
                     cache=None):
    """nn.ConvTranspose1d as `stride` polyphase stride-1 convolutions (hifigan.py:114-115).
    out[co][n] = sum_ci sum_j Wt[ci][co][u*j+p] * in[ci][q-j],  n + P = u*q + p."""
    B, _, T_in = x.shape
    u, P = int(stride), int(padding)
    T_out = (T_in - 1) * u - 2 * P + k
    out = torch.empty(B, Cout, T_out, dtype=torch.float32, device=x.device)
    phases = cache if cache is not None else {}
    if impl is None and _AUTO_SPLIT[0] and Cin >= 32 and u * Cout >= 32 and T_in >= 64 and k >= u:
        # split-operand scope: every phase in one launch on the two-piece fp16 kernel (contiguous 16-byte stores)
        w = _f(ConvWeight(w_getter, Cout, Cin, k)._resolve(), "weight")
        key = (w.data_ptr(), w._version, w.device, _WEIGHTS_EPOCH)
        ent = phases.get("x2")
        if ent is None or ent[0] != key:
            import math
            wp = torch.empty(_lib.lib().set_packed_conv_transpose_x2_size(Cout, Cin, k, u), dtype=torch.float16, device=w.device)
            m = float(w.abs().max())
            ke = max(-60, min(60, 4 - math.frexp(m)[1])) if m > 0 and math.isfinite(m) else 0
            check(_lib.lib().set_pack_conv_transpose_x2(_p(w), _p(wp), Cout, Cin, k, u, ke, _stream()), "set_pack_conv_transpose_x2")
            ent = phases["x2"] = (key, wp)
        _f(x, "x")
        check(_lib.lib().set_conv_transpose1d_x2(_p(x), _p(ent[1]), _p(bias) if bias is not None else None, _p(out), B, Cin, Cout, k,
                                                 u, P, T_in, PRO[pro], float(pro_param), _stream()), "set_conv_transpose1d_x2")
        return out
    for p in range(u):
        J = (k - p + u - 1) // u
        if J <= 0:
            continue
        if p not in phases:
            phases[p] = ConvWeight(w_getter, Cout, Cin, J, base=p, sco=k, sci=Cout * k, stap=u)
        conv1d(x, phases[p], bias, dil=-1, pad=0, pro=pro, pro_param=pro_param, out=out, impl=impl,
               T_iter=T_in + J - 1, T_out=T_out, out_stride=u, out_off=p - P)
    if k < u:  # output samples no tap reaches still carry the bias
        raise NotImplementedError("kernel_size < stride")
    return out


def weight_norm_fold(g, v):
    _f(g), _f(v)
    w = torch.empty_like(v)
    n0 = v.shape[0]
    check(_lib.lib().set_weight_norm_fold(_p(g), _p(v), _p(w), n0, v.numel() // n0, _stream()), "set_weight_norm_fold")
    return w


# --------------------------------------------------------------------------
# conditioner glue
# --------------------------------------------------------------------------
def layernorm_ch(x, gamma, beta, mask=None, eps=1e-5, out=None):
    _f(x), _f(gamma), _f(beta), _f(mask)
    B, Cc, T = x.shape
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().set_layernorm_ch(_p(x), _p(gamma), _p(beta), _p(mask), _p(out), B, Cc, T, float(eps), _stream()),
          "set_layernorm_ch")
    return out


def preln_ffn(x, ln, cw1, b1, cw2, b2, *, dil=1, pad=0, alpha=1.0, act="gelu", act_param=0.0, mask=None, eps=1e-5, T_out=None):
    """Inference form of autograd_ops.preln_ffn: (x + conv1x1(act(alpha conv_k(LN(x))))) (* mask), the activation in the first conv's epilogue."""
    h = layernorm_ch(x, ln[0], ln[1], eps=eps)
    h = conv1d(h, cw1, b1, dil=dil, pad=pad, alpha=alpha, act=act, act_param=act_param, T_out=T_out)
    return conv1d(h, cw2, b2, res=x, mask=mask)


def preln_self_attn(x, ln, attn, key_padding=None, mask=None, eps=1e-5):
    """Inference form of autograd_ops.preln_self_attn."""
    h = layernorm_ch(x, ln[0], ln[1], eps=eps)
    qkv = conv1d(h, attn._w_qkv)
    o, _ = self_attention(qkv, attn.num_heads, key_padding, float("-inf"), attn.scaling)
    return conv1d(o, attn._w_out, res=x, mask=mask)


def preln_cross_attn(x, ln, attn, enc, enc_padding, want_p=True, eps=1e-5):
    """Inference form of autograd_ops.preln_cross_attn: (x + attention, probabilities or None)."""
    h = layernorm_ch(x, ln[0], ln[1], eps=eps)
    q = conv1d(h, attn._w_q)
    kv = conv1d(enc, attn._w_kv)
    o, p = cross_attention(q, kv, attn.num_heads, enc_padding, -1e8, attn.scaling, want_p=want_p)
    return conv1d(o, attn._w_out, res=x), p


_VALIDATE = os.environ.get("SET_AMD_VALIDATE", "0") == "1"


def set_validate(on):
    """Host-side range checks of index tensors before the kernels that would otherwise clamp them (one device->host
    read per call: a debugging aid, off by default; `--validate` / `--debug` runs switch it on)."""
    global _VALIDATE
    _VALIDATE = bool(on)


def embedding_bct(idx, table, scale=1.0, out=None, accumulate=False, padding_idx=None):
    _i(idx), _f(table)
    B, T = idx.shape
    n_rows, Cc = table.shape
    if _VALIDATE and idx.numel() > 0:
        lo, hi = int(idx.min().item()), int(idx.max().item())
        if lo < 0 or hi >= n_rows:  # nn.Embedding raises here; the kernel clamps for memory safety only
            raise IndexError("embedding index out of range: [%d, %d] for a table of %d rows" % (lo, hi, n_rows))
    if out is None:
        assert not accumulate
        out = torch.empty(B, Cc, T, dtype=torch.float32, device=table.device)
    if isinstance(scale, torch.Tensor):  # a scale that lives on the device (a parameter): no host read
        check(_lib.lib().set_embedding_bct_dev_scale(_p(idx), _p(table), _p(_f(out)), B, T, Cc, n_rows, _p(_f(scale, "scale")),
                                                     int(bool(accumulate)), _stream()), "set_embedding_bct_dev_scale")
        return out
    check(_lib.lib().set_embedding_bct(_p(idx), _p(table), _p(_f(out)), B, T, Cc, n_rows, float(scale),
                                       int(bool(accumulate)), _stream()), "set_embedding_bct")
    return out


def abs_sum_mask(x):
    _f(x)
    B, Cc, T = x.shape
    m = torch.empty(B, T, dtype=torch.float32, device=x.device)
    check(_lib.lib().set_abs_sum_mask(_p(x), _p(m), B, Cc, T, _stream()), "set_abs_sum_mask")
    return m


def index_mask(idx):
    _i(idx)
    m = torch.empty(idx.shape, dtype=torch.float32, device=idx.device)
    check(_lib.lib().set_index_mask(_p(idx), _p(m), idx.numel(), _stream()), "set_index_mask")
    return m


def expand_states(enc_bct, mel2ph):
    _f(enc_bct), _i(mel2ph)
    B, Cc, T_txt = enc_bct.shape
    T = mel2ph.shape[1]
    out = torch.empty(B, Cc, T, dtype=torch.float32, device=enc_bct.device)
    check(_lib.lib().set_expand_states(_p(enc_bct), _p(mel2ph), _p(out), B, Cc, T_txt, T, _stream()),
          "set_expand_states")
    return out


def add_chan_mask(x, add=None, mask=None, out=None):
    _f(x), _f(add), _f(mask)
    B, Cc, T = x.shape
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().set_add_chan_mask(_p(x), _p(add), _p(mask), _p(out), B, Cc, T, _stream()), "set_add_chan_mask")
    return out


def masked_dur(mel2ph, tmask_bt, txt_tokens):
    _i(mel2ph), _f(tmask_bt), _i(txt_tokens)
    B, T = mel2ph.shape
    T_txt = txt_tokens.shape[1]
    out = torch.empty(B, T_txt, dtype=torch.int64, device=mel2ph.device)
    check(_lib.lib().set_masked_dur(_p(mel2ph), _p(tmask_bt), _p(txt_tokens), _p(out), B, T, T_txt, _stream()),
          "set_masked_dur")
    return out


def pitch_coarse(f0, uv, tmask=None, mel2ph_pad=None, uv_from_logit=False, want_denorm=True, want_coarse=True):
    _f(f0), _f(uv), _f(tmask), _i(mel2ph_pad)
    n = f0.numel()
    den = torch.empty_like(f0) if want_denorm else None
    co = torch.empty(f0.shape, dtype=torch.int64, device=f0.device) if want_coarse else None
    check(_lib.lib().set_pitch_coarse(_p(f0), _p(uv), _p(tmask), _p(mel2ph_pad), int(bool(uv_from_logit)), _p(den),
                                      _p(co), n, _stream()), "set_pitch_coarse")
    return den, co


def btc_to_bct(x):
    _f(x)
    B, T, Cc = x.shape
    out = torch.empty(B, Cc, T, dtype=torch.float32, device=x.device)
    check(_lib.lib().set_transpose_btc_to_bct(_p(x), _p(out), B, T, Cc, _stream()), "set_transpose_btc_to_bct")
    return out


def bct_to_btc(x):
    _f(x)
    B, Cc, T = x.shape
    out = torch.empty(B, T, Cc, dtype=torch.float32, device=x.device)
    check(_lib.lib().set_transpose_bct_to_btc(_p(x), _p(out), B, Cc, T, _stream()), "set_transpose_bct_to_btc")
    return out


def sum_div(a, b=None, c=None, div=1.0, out=None):
    _f(a), _f(b), _f(c)
    out = torch.empty_like(a) if out is None else out
    check(_lib.lib().set_sum_scale(_p(a), _p(b), _p(c), _p(out), float(div), a.numel(), _stream()), "set_sum_scale")
    return out


def blend_mask(a, b, m, inner):
    """a*(1-m) + b*m with m broadcast over `inner` trailing elements."""
    _f(a), _f(b), _f(m)
    out = torch.empty_like(a)
    check(_lib.lib().set_blend_mask(_p(a), _p(b), _p(m), _p(out), a.numel(), int(inner), _stream()), "set_blend_mask")
    return out


def mul_one_minus_mask(x, m, inner):
    _f(x), _f(m)
    out = torch.empty_like(x)
    check(_lib.lib().set_mul_one_minus_mask(_p(x), _p(m), _p(out), x.numel(), int(inner), _stream()),
          "set_mul_one_minus_mask")
    return out


def length_regulate(dur, txt_tokens):
    """nar_tts_modules.py:42-72.  Host reads max total length (the reference sizes a tensor the same way)."""
    _f(dur), _i(txt_tokens)
    B, T_txt = dur.shape
    total = torch.empty(B, dtype=torch.int64, device=dur.device)
    check(_lib.lib().set_dur_total(_p(dur), _p(txt_tokens), _p(total), B, T_txt, _stream()), "set_dur_total")
    T_out = int(total.max().item())
    out = torch.empty(B, max(T_out, 0), dtype=torch.int64, device=dur.device)
    if T_out > 0:
        check(_lib.lib().set_length_regulate(_p(dur), _p(txt_tokens), _p(out), B, T_txt, T_out, _stream()),
              "set_length_regulate")
    return out


# --------------------------------------------------------------------------
# DiffNet pieces
# --------------------------------------------------------------------------
def sinusoid_embed(t_float, dim):
    """-> [dim][n]"""
    _f(t_float)
    n = t_float.numel()
    out = torch.empty(dim, n, dtype=torch.float32, device=t_float.device)
    check(_lib.lib().set_sinusoid_embed(_p(t_float), _p(out), dim, n, _stream()), "set_sinusoid_embed")
    return out


def gate(y):
    _f(y)
    B, C2, T = y.shape
    z = torch.empty(B, C2 // 2, T, dtype=torch.float32, device=y.device)
    check(_lib.lib().set_gate(_p(y), _p(z), B, C2 // 2, T, _stream()), "set_gate")
    return z


def res_skip(x_in, o, skip, first):
    _f(x_in), _f(o), _f(skip)
    B, Cc, T = x_in.shape
    x_out = torch.empty_like(x_in)
    check(_lib.lib().set_res_skip(_p(x_in), _p(o), _p(x_out), _p(skip), B, Cc, T, int(bool(first)), _stream()),
          "set_res_skip")
    return x_out


def pack_diffnet_layer(w_dil, w_out, w1p=None, w2p=None):
    """Pack one layer; optionally into caller-provided slices of the contiguous [L][...] buffers."""
    _f(w_dil), _f(w_out)
    L = _lib.lib()
    if w1p is None:
        w1p = torch.empty(L.set_diffnet_w1p_size(), dtype=torch.float32, device=w_dil.device)
        w2p = torch.empty(L.set_diffnet_w2p_size(), dtype=torch.float32, device=w_dil.device)
    check(L.set_pack_diffnet_layer(_p(w_dil), _p(w_out), _p(_f(w1p)), _p(_f(w2p)), _stream()), "set_pack_diffnet_layer")
    return w1p, w2p


def pack_diffnet_layer_wino(w_dil, w_out, w1w, w2w):
    """Winograd F(2,3) images of one layer into slices of the contiguous [L][...] buffers."""
    _f(w_dil), _f(w_out), _f(w1w), _f(w2w)
    check(_lib.lib().set_pack_diffnet_layer_wino(_p(w_dil), _p(w_out), _p(w1w), _p(w2w), _stream()),
          "set_pack_diffnet_layer_wino")


def sync_ws_size(B, T):
    return 160 + 2 * B * ((T + 31) // 32)


class SplitRangeError(_lib.SetAmdError):
    """An activation left the range of the fp16 operand splitting (the loop's output is not valid)."""


_SPLIT_MODE_OVERRIDE = [None]


def split_operand_mode():
    """Which splitting the throughput stack kernel uses: 2 = two fp16 pieces (three MFMA products per term), 3 = three bf16
    pieces (six products, fp32 range).  SET_AMD_SPLIT_OPERAND=bf16x3 / f16x2 overrides the default."""
    if _SPLIT_MODE_OVERRIDE[0] is not None:
        return _SPLIT_MODE_OVERRIDE[0]
    v = os.environ.get("SET_AMD_SPLIT_OPERAND", "f16x2").lower()
    if v not in ("f16x2", "bf16x3"):
        raise ValueError("SET_AMD_SPLIT_OPERAND must be f16x2 or bf16x3, not %r" % v)
    return 2 if v == "f16x2" else 3


class split_operand_mode_as:
    """with split_operand_mode_as(3): ...  -- pin the splitting for a block (the fallback of the reverse loop)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev, _SPLIT_MODE_OVERRIDE[0] = _SPLIT_MODE_OVERRIDE[0], self.mode

    def __exit__(self, *exc):
        _SPLIT_MODE_OVERRIDE[0] = self.prev


def stack_variant(B, T, dilation_cycle_length, have_wino=True, have_split=True, x3_mode=None):
    """0 / 1: direct kernel (64- / 32-frame tiles), 2: Winograd kernel, 3: row-split kernel (small batches), 4 / 5: split-operand
    kernel (3 x bf16 / 2 x fp16) -- what set_diffnet_stack would pick.  x3_mode: 0 (no split-operand images), 2, 3, or None =
    split_operand_mode()."""
    m = split_operand_mode() if x3_mode is None else int(x3_mode)
    bits = int(bool(have_wino)) | (2 if have_split else 0) | (4 if m == 3 else 0) | (8 if m == 2 else 0)
    return int(_lib.lib().set_diffnet_stack_variant(int(B), int(T), int(dilation_cycle_length), bits))


def stack_x3_winograd(B, T, dilation_cycle_length, have_wino=True, have_split=True, x3_mode=None):
    """Non-zero when variant 5 (two-piece fp16 split-operand kernel) runs GEMM 1 in its Winograd F(2,3) form for this shape (round 6,
    diffnet_stack_x3v_kernel): the number of 32-frame column blocks per tile -- 3 = 96-frame tiles (shapes with a tile chain for every CU),
    2 = 64-frame tiles; 0 = the direct form."""
    m = split_operand_mode() if x3_mode is None else int(x3_mode)
    bits = int(bool(have_wino)) | (2 if have_split else 0) | (4 if m == 3 else 0) | (8 if m == 2 else 0)
    return int(_lib.lib().set_diffnet_stack_x3_winograd(int(B), int(T), int(dilation_cycle_length), bits))


STACK_VARIANT_NAMES = {0: "diffnet_stack_kernel<2,4,2>", 1: "diffnet_stack_kernel<1,8,2>", 2: "diffnet_stack_wino_kernel",
                       3: "diffnet_stack_split_kernel", 4: "diffnet_stack_x3_kernel<SplitBf16x3>",
                       5: "diffnet_stack_x3_kernel<SplitF16x2>"}


class SplitOperandImages:
    """[L][n] 16-bit images of the split-operand stack kernel + the mode they were packed for."""

    def __init__(self, L, mode, device):
        self.mode = int(mode)
        n = int(_lib.lib().set_diffnet_layer_x3_image_size(self.mode))
        self.data = torch.empty(L, n, dtype=torch.int16, device=device)

    def pack(self, i, w_dil, w_out):
        """Layer i.  fp16 pieces: scale the weights by a power of two so that max |w| lands in [8, 16) (exact; undone on
        the accumulators by the kernel) -- one host read-back per layer and weight version."""
        _f(w_dil), _f(w_out)
        k1 = k2 = 0
        if self.mode == 2:
            import math
            m1, m2 = float(w_dil.abs().max()), float(w_out.abs().max())
            k1 = max(-60, min(60, 3 - math.frexp(m1)[1] + 1)) if m1 > 0 and math.isfinite(m1) else 0
            k2 = max(-60, min(60, 3 - math.frexp(m2)[1] + 1)) if m2 > 0 and math.isfinite(m2) else 0
        check(_lib.lib().set_pack_diffnet_layer_x3(_p(w_dil), _p(w_out), self.data[i].data_ptr(), self.mode, k1, k2, _stream()),
              "set_pack_diffnet_layer_x3")


def split_images(w1p_all, w2p_all):
    """Row-split images of the small-batch stack kernel from the packed layer images: the same values, one 32-row
    block per wave ([L][w][k-step][lane][rb] -> [L][4w + rb][k-step][lane])."""
    L = w1p_all.shape[0]
    return tuple(w.view(L, 4, -1, 64, 4).permute(0, 1, 4, 2, 3).contiguous().view(L, -1) for w in (w1p_all, w2p_all))


def _split_images(a, packs, B, T, dcl, dev):
    """Row-split images + z workspace of the small-batch stack kernel into the args struct (only for shapes the kernel
    is picked for); returns the workspace (the caller keeps it alive; stream-ordered reuse is safe)."""
    x3 = packs[8] if len(packs) >= 9 else None
    if x3 is not None:
        a.wx3_all, a.x3_mode = x3.data.data_ptr(), x3.mode
    if len(packs) < 8 or packs[6] is None or stack_variant(B, T, dcl, x3_mode=x3.mode if x3 is not None else 0) != 3:
        return None
    a.w1s_all, a.w2s_all = packs[6].data_ptr(), packs[7].data_ptr()
    z_ws = torch.empty(B * ((T + 31) // 32) * 256 * 32, dtype=torch.float32, device=dev)
    a.z_ws = z_ws.data_ptr()
    return z_ws


def diffnet_stack(xa, xb, skip, condproj, dstep_ptr, d_bs, d_cs, d_ls, packs, dilation_cycle_length, sync_ws=None,
                  x_all=None, save_y=None, save_z=None, err_flag=None):
    """All L layers in one persistent launch.  condproj [B, L*512, T]; packs = (w1p_all, w2p_all, b_dil_all, b_out_all
    [, w1w_all, w2w_all [, w1s_all, w2s_all [, wx3_all]]]).  Training forward (Winograd kernel only): x_all [L+1,B,256,T] (slab 0 = input) replaces the
    xa/xb ping-pong, save_y [L,B,512,T] / save_z [L,B,256,T] receive what the backward pass needs.
    Returns sync_ws (int32; [1] != 0 means a dependency wait timed out)."""
    _f(xa), _f(xb), _f(skip), _f(condproj)
    B, Cc, T = xa.shape
    assert Cc == 256
    w1p_all, w2p_all, b_dil_all, b_out_all = packs[:4]
    L = b_dil_all.shape[0]
    if sync_ws is None:
        sync_ws = torch.empty(sync_ws_size(B, T), dtype=torch.int32, device=xa.device)
    a = SetDiffnetStackArgs()
    a.xa, a.xb, a.skip = xa.data_ptr(), xb.data_ptr(), skip.data_ptr()
    a.condproj, a.dstep = condproj.data_ptr(), dstep_ptr
    a.w1p_all, a.w2p_all = w1p_all.data_ptr(), w2p_all.data_ptr()
    a.b_dil_all, a.b_out_all = b_dil_all.data_ptr(), b_out_all.data_ptr()
    if len(packs) >= 6 and packs[4] is not None:
        a.w1w_all, a.w2w_all = packs[4].data_ptr(), packs[5].data_ptr()
    if x_all is not None:
        a.x_all, a.save_y, a.save_z = _f(x_all).data_ptr(), _f(save_y).data_ptr(), _f(save_z).data_ptr()
    z_ws = _split_images(a, packs, B, T, dilation_cycle_length, xa.device)  # noqa: F841 (kept alive until the launch is enqueued)
    if err_flag is not None:  # optional sticky int32 error word (1: a dependency wait timed out, 2: out of the fp16 split range)
        a.err_flag = err_flag.data_ptr()
    a.sync_ws = sync_ws.data_ptr()
    a.cp_bs, a.cp_ls = condproj.stride(0), 512 * T
    a.d_bs, a.d_cs, a.d_ls = int(d_bs), int(d_cs), int(d_ls)
    a.B, a.T, a.L, a.dilation_cycle_length = B, T, L, int(dilation_cycle_length)
    check(_lib.lib().set_diffnet_stack(C.byref(a), _stream()), "set_diffnet_stack")
    return sync_ws


def diffnet_layer(x_in, condproj, cp_bs, dstep, d_bs, d_cs, w1p, b_dil, w2p, b_out, x_out, skip, dil, first,
                  dbg_clock=None):
    _f(x_in), _f(x_out), _f(skip)
    B, Cc, T = x_in.shape
    assert Cc == 256
    a = SetDiffnetLayerArgs()
    a.x_in, a.condproj, a.dstep = x_in.data_ptr(), condproj, dstep
    a.w1p, a.b_dil, a.w2p, a.b_out = w1p.data_ptr(), b_dil.data_ptr(), w2p.data_ptr(), b_out.data_ptr()
    a.x_out, a.skip = x_out.data_ptr(), skip.data_ptr()
    a.cp_bs, a.d_bs, a.d_cs = int(cp_bs), int(d_bs), int(d_cs)
    a.B, a.T, a.dil, a.first = B, T, int(dil), int(bool(first))
    a.dbg_clock = dbg_clock.data_ptr() if dbg_clock is not None else None
    check(_lib.lib().set_diffnet_layer(C.byref(a), _stream()), "set_diffnet_layer")


def posterior_step(x0, x_t, coef4, eps=None, out=None, seed=0, offset=0):
    """coef4 [B,4] (or [1,4] shared by the batch)."""
    _f(x0), _f(x_t), _f(coef4), _f(eps)
    B = x0.shape[0]
    per_batch = x0.numel() // B
    out = x_t if out is None else out
    assert coef4.shape[0] in (1, B) and coef4.shape[-1] == 4
    coef_bs = 0 if coef4.shape[0] == 1 else 4
    check(_lib.lib().set_posterior_step(_p(x0), _p(x_t), _p(eps), _p(coef4), coef_bs, _p(out), B, per_batch, int(seed),
                                        int(offset), _stream()), "set_posterior_step")
    return out


def q_sample(x_start, eps, ab2, nonpad=None):
    _f(x_start), _f(eps), _f(ab2), _f(nonpad)
    B, M, T = x_start.shape[0], x_start.shape[-2], x_start.shape[-1]
    out = torch.empty_like(x_start)
    check(_lib.lib().set_q_sample(_p(x_start), _p(eps), _p(ab2), _p(nonpad), _p(out), B, M, T, _stream()),
          "set_q_sample")
    return out


def randn(shape, device, seed=0, offset=0):
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(_lib.lib().set_randn(_p(out), out.numel(), int(seed), int(offset), _stream()), "set_randn")
    return out


def fanout(x, n):
    """Inference backend: n references to the same tensor."""
    return (x,) * n


def grad_scale(x, s):
    """Inference backend: identity (fs.py:144-145 only rescales gradients)."""
    return x


def dropout(x, p, seed, offset=0):
    """Inference backend: identity (eval mode)."""
    return x


def res_skip_fn(x, o, skip_in):
    """Functional form used by the shared layer code: returns (x_out, skip_out); skip_in None on the first layer."""
    first = skip_in is None
    skip = torch.empty_like(x) if first else skip_in
    return res_skip(x, o, skip, first), skip


def selftest_mfma():
    err = C.c_float(-1.0)
    check(_lib.lib().set_selftest_mfma(C.byref(err), _stream()), "set_selftest_mfma")
    return float(err.value)


def diffusion_loop(*, x, noise, seed, condproj, dstep, coef4, w_in, b_in, packs, w_skip, b_skip,
                   w_outp, b_outp, L, steps, dilation_cycle_length, want_layer_spans=False, n_groups=None,
                   persistent=None, bf16=None):
    """Enqueue the whole reverse loop (set_diffusion_loop).  x [B,M,T] is updated in place.
    bf16 = dict(cond=[B,192,T] fp32, imgs=[L, n] bf16 layer images, b_cond=[L,512]): the opt-in bf16-operand loop
    (condproj is then unused and may be None)."""
    _f(x), _f(noise), _f(condproj), _f(dstep), _f(coef4)
    B, M, T = x.shape
    dev = x.device
    ws = [torch.empty(B, 256, T, dtype=torch.float32, device=dev) for _ in range(4)]
    ws_x0pred = torch.empty(B, M, T, dtype=torch.float32, device=dev)
    a = SetDiffLoopArgs()
    a.B, a.T, a.M, a.L, a.steps, a.dilation_cycle_length = B, T, M, L, steps, dilation_cycle_length
    a.x = x.data_ptr()
    a.noise = noise.data_ptr() if noise is not None else None
    a.seed = int(seed)
    a.condproj = condproj.data_ptr() if condproj is not None else None
    a.dstep, a.coef4 = dstep.data_ptr(), coef4.data_ptr()
    if bf16 is not None:
        a.cond, a.img16_all, a.b_cond_all = _f(bf16["cond"]).data_ptr(), bf16["imgs"].data_ptr(), _f(bf16["b_cond"]).data_ptr()
        # workspace of the several-layers-per-launch kernels (128-frame tiles: a block's private copy of its skip rows between the layers
        # of a group), sized for whatever group size the loop picks
        n_ws = max(_lib.lib().set_diffnet_layers_bf16_scratch_floats(B, T, 0, n, dilation_cycle_length) for n in range(2, 17)) \
            if dilation_cycle_length <= 2 else 0
        if n_ws > 0:
            bf16_ws = torch.empty(n_ws, dtype=torch.float32, device=dev)  # noqa: F841 (kept alive until the loop is enqueued: stream-ordered free)
            a.bf16_ws, a.bf16_ws_floats = bf16_ws.data_ptr(), n_ws
    a.w_in_p, a.b_in = w_in.packed().data_ptr(), b_in.data_ptr()
    w1p_all, w2p_all, b_dil_all, b_out_all = packs[:4]
    a.w1p_all, a.w2p_all = w1p_all.data_ptr(), w2p_all.data_ptr()
    a.b_dil_all, a.b_out_all = b_dil_all.data_ptr(), b_out_all.data_ptr()
    if len(packs) >= 6 and packs[4] is not None:
        a.w1w_all, a.w2w_all = packs[4].data_ptr(), packs[5].data_ptr()
    z_ws = _split_images(a, packs, B, T, dilation_cycle_length, dev)  # noqa: F841
    a.persistent = int(default_persistent() if persistent is None else bool(persistent))
    sync_ws = torch.empty(sync_ws_size(B, T), dtype=torch.int32, device=dev)
    a.sync_ws = sync_ws.data_ptr()
    err = torch.zeros(1, dtype=torch.int32, device=dev)  # sticky: a dependency wait of the persistent kernel gave up
    a.err_flag = err.data_ptr()
    bf16_bx2 = bf16 is not None and split_operand_mode() == 2  # the opt-in bf16 loop: its layers are bf16, its step boundary fp32-equivalent
    use_bx2 = ((a.wx3_all and a.x3_mode == 2) or bf16_bx2) and M <= 96 and os.environ.get("SET_AMD_BOUNDARY_X2", "1") != "0"
    if use_bx2:
        # the fused step boundary on the same two-piece fp16 operands
        a.w_skip_x2, a.w_outp_x2, a.w_in_x2 = (w.packed_x2().data_ptr() for w in (w_skip, w_outp, w_in))
    a.w_skip_p, a.b_skip = w_skip.packed().data_ptr(), b_skip.data_ptr()
    a.w_outp_p, a.b_outp = w_outp.packed().data_ptr(), b_outp.data_ptr()
    a.ws_x0, a.ws_x1, a.ws_skip, a.ws_h = (w.data_ptr() for w in ws)
    a.ws_x0pred = ws_x0pred.data_ptr()
    a.n_groups = default_groups(B, T) if n_groups is None else int(n_groups)
    spans, loop_ms = None, None
    if want_layer_spans:
        spans = (C.c_float * steps)()
        a.layer_span_ms = C.cast(spans, C.POINTER(C.c_float))
        loop_ms = C.c_float(0.0)
        a.loop_ms = C.pointer(loop_ms)
    check(_lib.lib().set_diffusion_loop(C.byref(a), _stream()), "set_diffusion_loop")
    # one 4-byte read-back per reverse loop; fail loudly, never return garbage.  The range word can be raised by the persistent stack kernel
    # AND by the split-operand step boundary, which the bf16 loop takes whether or not `persistent` is set
    if a.persistent or use_bx2:
        code = int(err.item())
        if code == 2:
            raise SplitRangeError("set_diffusion_loop: an activation of magnitude >= 32768 is outside the range of the fp16 "
                                  "operand splitting; set SET_AMD_SPLIT_OPERAND=bf16x3 (fp32 range) or SET_AMD_X3=0")
        if code != 0:
            raise _lib.SetAmdError("set_diffusion_loop: a tile dependency wait of the persistent layer-stack kernel timed out")
    # the group chains are joined back into the current stream, so stream-ordered reuse of these buffers is safe
    if spans is None:
        return None
    return {"layer_span_ms": list(spans), "loop_ms": float(loop_ms.value), "n_groups": int(a.n_groups),
            "persistent": int(a.persistent)}


def default_persistent():
    """SET_AMD_PERSISTENT=0/1 overrides; default: persistent layer-stack kernel."""
    return os.environ.get("SET_AMD_PERSISTENT", "1") != "0"


def default_groups(B, T):
    """Utterance groups for the reverse loop (callers may pass groups= themselves).  Measured on MI355X (profiles/): more than
    one group is slower -- kernels from different streams are placed on the same CUs first, so the per-CU imbalance
    of a 416-block launch gets worse, not better -- hence 1.  Grouping never changes results."""
    return 1


# --------------------------------------------------------------------------
# attention building blocks (CampNet rows)
# --------------------------------------------------------------------------
class MatView:
    """A batch of matrices inside a tensor: element (bo, bi, r, c) at  offset + bo*bo_s + bi*bi_s + r*rs + c*cs
    (in floats).  `.t` swaps rows and columns.  Heads of a [B, heads*d, T] activation: MatView.heads(x, heads)."""

    def __init__(self, tensor, n_outer, n_inner, rows, cols, bo_s, bi_s, rs, cs, offset=0):
        self.tensor = _f(tensor)
        self.n_outer, self.n_inner, self.rows, self.cols = int(n_outer), int(n_inner), int(rows), int(cols)
        self.bo_s, self.bi_s, self.rs, self.cs, self.offset = int(bo_s), int(bi_s), int(rs), int(cs), int(offset)

    @property
    def t(self):
        return MatView(self.tensor, self.n_outer, self.n_inner, self.cols, self.rows, self.bo_s, self.bi_s, self.cs,
                       self.rs, self.offset)

    @staticmethod
    def heads(x, heads, chan0=0, width=None):
        """Channels [chan0, chan0+width) of x [B, C, T] (contiguous) split into heads -> per (b, h) the matrix
        [T, d]: rows = frames, cols = head channels.  (q / k / v are channel slices of one packed projection.)"""
        B, Ctot, T = x.shape
        H = Ctot - chan0 if width is None else width
        d = H // heads
        assert x.is_contiguous() and d * heads == H and chan0 + H <= Ctot
        return MatView(x, B, heads, T, d, Ctot * T, d * T, 1, T, offset=chan0 * T)

    @staticmethod
    def scores(s):
        """s [B, heads, Tq, Tk] (contiguous) -> per (b, h) the matrix [Tq, Tk]."""
        B, h, Tq, Tk = s.shape
        assert s.is_contiguous()
        return MatView(s, B, h, Tq, Tk, h * Tq * Tk, Tq * Tk, Tk, 1)

    def ptr(self):
        return self.tensor.data_ptr() + 4 * self.offset


def bmm(a, b, c, alpha=1.0, accumulate=False):
    """c = alpha * a @ b (+ c) on MatViews.  When c is column-major the transposed problem is run so that the
    stores stay coalesced (c^T = b^T a^T)."""
    assert a.cols == b.rows and a.rows == c.rows and b.cols == c.cols
    assert (a.n_outer, a.n_inner) == (b.n_outer, b.n_inner) == (c.n_outer, c.n_inner)
    if c.rs == 1 and c.cs != 1:
        a, b, c = b.t, a.t, c.t
    g = _lib.SetBmmArgs()
    g.A, g.B, g.C = a.ptr(), b.ptr(), c.ptr()
    g.a_bo, g.a_bi, g.a_ms, g.a_ks = a.bo_s, a.bi_s, a.rs, a.cs
    g.b_bo, g.b_bi, g.b_ks, g.b_ns = b.bo_s, b.bi_s, b.rs, b.cs
    g.c_bo, g.c_bi, g.c_ms, g.c_ns = c.bo_s, c.bi_s, c.rs, c.cs
    g.n_outer, g.n_inner, g.M, g.N, g.K = c.n_outer, c.n_inner, c.rows, c.cols, a.cols
    g.alpha, g.accumulate = float(alpha), int(bool(accumulate))
    check(_lib.lib().set_bmm(C.byref(g), _stream()), "set_bmm")
    return c.tensor


def softmax_rows(x, key_padding_mask=None, rows_per_batch=1, fill=float("-inf"), out=None):
    """softmax over the last dim of a contiguous tensor; kpm [B, cols] fp32 (1 = pad), B = rows / rows_per_batch."""
    _f(x), _f(key_padding_mask)
    cols = x.shape[-1]
    rows = x.numel() // cols
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().set_softmax_rows(_p(x), _p(key_padding_mask), _p(out), rows, cols, int(rows_per_batch), float(fill),
                                      _stream()), "set_softmax_rows")
    return out


def softmax_rows_bwd(p, dp):
    _f(p), _f(dp)
    cols = p.shape[-1]
    ds = torch.empty_like(p)
    check(_lib.lib().set_softmax_rows_bwd(_p(p), _p(dp), _p(ds), p.numel() // cols, cols, _stream()), "set_softmax_rows_bwd")
    return ds


def make_positions(tokens=None, x_bct=None):
    """int64 [B,T]: 1,2,3.. over the non-zero tokens (or the non-zero entries of channel 0 of x_bct), 0 elsewhere."""
    if tokens is not None:
        _i(tokens)
        B, T = tokens.shape
        pos = torch.empty(B, T, dtype=torch.int64, device=tokens.device)
        check(_lib.lib().set_make_positions(_p(tokens), None, 0, _p(pos), B, T, _stream()), "set_make_positions")
    else:
        _f(x_bct)
        B, Cc, T = x_bct.shape
        pos = torch.empty(B, T, dtype=torch.int64, device=x_bct.device)
        check(_lib.lib().set_make_positions(None, _p(x_bct), Cc * T, _p(pos), B, T, _stream()), "set_make_positions")
    return pos


def head_mean(p):
    """p [B, heads, ...] -> mean over heads [B, ...]."""
    _f(p)
    B, h = p.shape[:2]
    out = torch.empty((B,) + tuple(p.shape[2:]), dtype=torch.float32, device=p.device)
    check(_lib.lib().set_head_mean(_p(p), _p(out), B, h, out.numel() // B, _stream()), "set_head_mean")
    return out


def mask_fill_chan(x, e, m):
    """x [B,C,T]*(1-m[B,T]) + e[C]*m"""
    _f(x), _f(e), _f(m)
    B, Cc, T = x.shape
    out = torch.empty_like(x)
    check(_lib.lib().set_mask_fill_chan(_p(x), _p(e), _p(m), _p(out), B, Cc, T, _stream()), "set_mask_fill_chan")
    return out


def masked_channel_sum(d, m, out):
    _f(d), _f(m), _f(out)
    B, Cc, T = d.shape
    check(_lib.lib().set_masked_channel_sum(_p(d), _p(m), _p(out), B, Cc, T, _stream()), "set_masked_channel_sum")
    return out


def attention_fused_on(head_dim=None):
    """The fused attention kernels (csrc/attention_fused.hip) are the default for the head sizes they are instantiated for (32, 64,
    96: the shipped 192 / 2 config and its neighbours); any other head size, or SET_AMD_ATTN_FUSED=0, runs the three-launch
    bmm -> softmax -> bmm composition (any head size; also the cross-check of the tests)."""
    if head_dim is not None and int(head_dim) not in (32, 64, 96):
        return False
    return os.environ.get("SET_AMD_ATTN_FUSED", "1") != "0"


def _attn_args(qv, kv, vv, o, lse, heads, key_padding_mask, fill, alpha, p=None):
    d = qv.cols
    assert qv.rs == 1 and kv.rs == 1 and vv.rs == 1 and kv.cols == d and vv.cols == d and kv.rows == vv.rows
    assert qv.bi_s == d * qv.cs and kv.bi_s == d * kv.cs and vv.bi_s == d * vv.cs  # heads = consecutive channel slices
    a = _lib.SetAttnArgs()
    a.q, a.k, a.v = qv.ptr(), kv.ptr(), vv.ptr()
    a.o, a.lse = o.data_ptr(), lse.data_ptr()
    a.p = p.data_ptr() if p is not None else None
    a.kpm = _f(key_padding_mask, "key_padding_mask").data_ptr() if key_padding_mask is not None else None
    a.q_bs, a.k_bs, a.v_bs, a.o_bs = qv.bo_s, kv.bo_s, vv.bo_s, o.stride(0)
    a.q_cs, a.k_cs, a.v_cs, a.o_cs = qv.cs, kv.cs, vv.cs, o.stride(1)
    a.B, a.heads, a.head_dim, a.Tq, a.Tk = qv.n_outer, heads, d, qv.rows, kv.rows
    a.scale, a.fill = float(alpha), float(fill)
    a.bf16 = int(_COMPUTE_DTYPE == "bf16")
    return a


def attention_fused(qv, kv, vv, heads, key_padding_mask=None, fill=float("-inf"), alpha=1.0, want_p=False):
    """o = softmax(alpha q k^T (+ mask)) v per head in ONE launch (scores never in HBM): returns (o [B,H,Tq], lse
    [B,heads,2,Tq] = row max and sum of the softmax, p [B,heads,Tq,Tk] or None).  q / k / v: MatView.heads views."""
    B, Tq, Tk, d = qv.n_outer, qv.rows, kv.rows, qv.cols
    dev = qv.tensor.device
    o = torch.empty(B, heads * d, Tq, dtype=torch.float32, device=dev)
    lse = torch.empty(B, heads, 2, Tq, dtype=torch.float32, device=dev)  # row max m, sum l (log-sum-exp = m + log l)
    p = torch.empty(B, heads, Tq, Tk, dtype=torch.float32, device=dev) if want_p else None
    a = _attn_args(qv, kv, vv, o, lse, heads, key_padding_mask, fill, alpha, p)
    check(_lib.lib().set_attention(C.byref(a), _stream()), "set_attention")
    return o, lse, p


def attention_fused_bwd(qv, kv, vv, o, lse, do, dqv, dkv, dvv, heads, key_padding_mask=None, fill=float("-inf"), alpha=1.0):
    """Gradients of attention_fused into the views dqv / dkv / dvv (two launches, P recomputed from lse)."""
    g = _lib.SetAttnBwdArgs()
    g.fwd = _attn_args(qv, kv, vv, o, lse, heads, key_padding_mask, fill, alpha)
    _f(do, "do")
    assert do.shape == o.shape and do.stride() == o.stride()
    delta = torch.empty(lse.shape[0], lse.shape[1], lse.shape[3], dtype=torch.float32, device=lse.device)
    g.d_o, g.delta = do.data_ptr(), delta.data_ptr()
    g.dq, g.dk, g.dv = dqv.ptr(), dkv.ptr(), dvv.ptr()
    g.dq_bs, g.dk_bs, g.dv_bs = dqv.bo_s, dkv.bo_s, dvv.bo_s
    g.dq_cs, g.dk_cs, g.dv_cs = dqv.cs, dkv.cs, dvv.cs
    check(_lib.lib().set_attention_bwd(C.byref(g), _stream()), "set_attention_bwd")


def attention_views(qv, kv, vv, heads, key_padding_mask=None, fill=float("-inf"), alpha=1.0):
    """o = softmax(alpha * q k^T (+mask)) v per head; q/k/v given as MatView.heads views.  Returns (o [B,H,Tq]
    contiguous, p [B,heads,Tq,Tk])."""
    B, Tq, Tk, d = qv.n_outer, qv.rows, kv.rows, qv.cols
    dev = qv.tensor.device
    s = torch.empty(B, heads, Tq, Tk, dtype=torch.float32, device=dev)
    bmm(qv, kv.t, MatView.scores(s), alpha=alpha)
    p = softmax_rows(s, key_padding_mask, heads * Tq, fill, out=s)
    o = torch.empty(B, heads * d, Tq, dtype=torch.float32, device=dev)
    bmm(MatView.scores(p), vv, MatView.heads(o, heads))
    return o, p


def self_attention(qkv, heads, key_padding_mask=None, fill=float("-inf"), alpha=1.0, want_p=False):
    """qkv [B, 3H, T] = packed in_proj output (transformer.py:421-422) -> (o [B,H,T], p or None)."""
    H = qkv.shape[1] // 3
    views = (MatView.heads(qkv, heads, 0, H), MatView.heads(qkv, heads, H, H), MatView.heads(qkv, heads, 2 * H, H))
    if attention_fused_on(H // heads):
        o, _, p = attention_fused(*views, heads, key_padding_mask, fill, alpha, want_p)
        return o, p
    return attention_views(*views, heads, key_padding_mask, fill, alpha)


def cross_attention(q, kv, heads, key_padding_mask=None, fill=-1e8, alpha=1.0, want_p=True):
    """q [B,H,Tq], kv [B,2H,Tk] (in_proj_k / in_proj_v of the encoder output, transformer.py:433-451)."""
    H = q.shape[1]
    views = (MatView.heads(q, heads), MatView.heads(kv, heads, 0, H), MatView.heads(kv, heads, H, H))
    if attention_fused_on(H // heads):
        o, _, p = attention_fused(*views, heads, key_padding_mask, fill, alpha, want_p)
        return o, p
    return attention_views(*views, heads, key_padding_mask, fill, alpha)


def pos_add(x, alpha, pos, table):
    """x + alpha * table[pos]  (transformer.py:795-796)"""
    return embedding_bct(pos, table, scale=alpha.detach().reshape(1).contiguous(), out=x.clone(), accumulate=True)


def add_masked(a, b, m):
    """a + b * m[b][t] on [B,C,T]"""
    return sum_div(a.contiguous(), add_chan_mask(b, None, m))
