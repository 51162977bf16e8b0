"""Differentiable versions of the ops in ops.py (same names / signatures), for the training path.

torch.autograd is used as the tape only: every forward AND backward below is a kernel of libset_amd.so
(include/set_amd.h, "Training" section).  Ops without gradients (masks, integer bookkeeping) are re-exported
from ops.py unchanged.
"""
import collections
import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import SetAmdError  # noqa: E402
from ._lib import ACT, PRO, IMPL_MFMA, IMPL_NAIVE, check
from .ops import _p, _stream, ConvWeight

# no-gradient ops: identical to the inference path
abs_sum_mask = ops.abs_sum_mask
index_mask = ops.index_mask
masked_dur = ops.masked_dur
pitch_coarse = ops.pitch_coarse
mul_one_minus_mask = ops.mul_one_minus_mask
blend_mask = ops.blend_mask
sinusoid_embed = ops.sinusoid_embed
q_sample = ops.q_sample
randn = ops.randn
length_regulate = ops.length_regulate

L = _lib.lib


class _ZeroArena:
    """Zero-initialised gradient temporaries (split-K wgrad targets, bias sums, scatter-add targets) of one training
    step come out of ONE buffer cleared by ONE memset, instead of ~300 separate small fills per step (each a launch
    of a few microseconds on the critical stream).  Opened by FlatAdamW.zero_grad() and closed by FlatAdamW.step():
    only then is it safe, because the optimizer owns every .grad (views of its flat buffer), so autograd ADDS these
    temporaries into .grad and nothing keeps a reference past the step.  Outside such a step `_gzeros` is torch.zeros.
    Capacity follows the demand of the previous step.  One arena per owner (optimizer)."""

    def __init__(self):
        self.buf, self.off, self.need = None, 0, 0


_ARENAS = {}        # owner id -> _ZeroArena
_ACTIVE = [None]    # the arena of the optimisation step in progress (steps do not nest)

# --------------------------------------------------------------------------------------------------
# Leaf stream: gradient kernels nothing else in the backward pass waits for
# --------------------------------------------------------------------------------------------------
# A weight / bias gradient that goes straight into the flat optimizer's .grad (FlatAdamW.sink) is a LEAF of the step's dependency graph: the
# next reader is the optimizer (or the bucket all-reduce).  On the compute stream these kernels sit in the middle of the critical chain of
# input-gradient kernels: ~3.8 ms of the 11 ms spec_denoiser bf16 step (three grouped layer weight-gradient GEMMs 2.3 ms, ~35 conditioner
# weight gradients, ~70 ordered bias / partial reductions), most of it while the chain itself runs 5-40 us kernels on a few dozen workgroups.
# They are enqueued on a second HIP stream instead, ordered after the compute stream by an event at the point of the call, and the optimizer
# step / bucket launch waits for that stream (leaf_join / leaf_fence).  Results are unchanged bit for bit: same kernels, same operands, each
# target written by exactly one stream.  Tensors handed to a leaf kernel stay referenced until the join (their storage must not be reused by
# the compute stream before the leaf kernel has read it); reduction scratch buffers are per stream.  SET_AMD_LEAF_STREAM=0 switches it off.
# Host cost per fork: one C call (set_stream_order); torch's Event / Stream / record_stream wrappers cost ~28 us per fork (1.1 ms per step).
# Operand lifetime (round 6, advisor item): the operands are no longer all held until the end of backward() -- the fp32 DiffNet stack alone kept
# d_o and dy of 20 layers (2.1 GB at B = 32, T = 800) alive that way, growing with depth.  After every ~SET_AMD_LEAF_MARK_MB (128) MB of newly
# held operands a marker event is recorded on the leaf stream (set_stream_mark); at the next fork the markers that have completed
# (set_stream_mark_done, a non-blocking query) release everything held before them: those kernels have RUN, so whoever gets the storage
# next cannot race them.  Above SET_AMD_LEAF_KEEP_MB (8192; 0 = no accounting) MB of held operand STORAGE (views of one saved buffer count
# once) the compute stream is made to wait for the leaf stream (an early leaf_join) -- the bound, sized for a 288 GB part; not the normal
# path at the benchmark shapes: every early join gives up overlap (a first accounting that counted every view -- 18 GB "held" in the fp32
# step -- joined early and cost it 0.9 ms: 31.9 against 30.9 ms, profiles/r06_leaf_accounting_ab.log).  Measured (profiles/r06_leaf_operands.log): the lowest-priority leaf stream runs late, so markers rarely
# complete inside a backward pass -- it is the cap that bounds the memory; with a 64 MB cap the step joins 26 times and stays bit-identical.
_LEAF = {}  # device index -> {"stream" (torch object, kept alive), "raw" (handle), "dirty", "keep" (operands of queued leaf kernels), ...}
_LEAF_STATS = {"max_keep_bytes": 0, "released_by_marker": 0, "early_joins": 0}


def _mb_env(name, default):
    try:
        return max(0, int(os.environ.get(name, default))) << 20
    except ValueError:
        return int(default) << 20


def _storages(obj, out):
    """(storage address, storage bytes) of every tensor in obj (tuples / lists nested): operands are often VIEWS of one saved buffer
    (every layer's slice of the [L, B, 512, T] tensors of the fused stack) -- counting numel per view made the fp32 step look like
    18 GB of held operands and join early for nothing (profiles/r06_leaf_accounting_ab.log)."""
    if isinstance(obj, torch.Tensor):
        s_ = obj.untyped_storage()
        out.append((s_.data_ptr(), s_.nbytes()))
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _storages(o, out)


def _leaf_hold(st, obj):
    keys = None
    if st["cap_bytes"]:  # SET_AMD_LEAF_KEEP_MB=0: no accounting (operands held until the join, as in round 5)
        keys, stor = [], st["stor"]
        _storages(obj, keys)
        for k, n in keys:
            ent = stor.get(k)
            if ent is None:
                stor[k] = [1, n]
                st["bytes"] += n
                st["unmarked"] += n
            else:
                ent[0] += 1
    st["keep"].append((obj, keys))
    st["count"] += 1
    if st["bytes"] > _LEAF_STATS["max_keep_bytes"]:
        _LEAF_STATS["max_keep_bytes"] = st["bytes"]


def _leaf_drop(st, keys):
    if keys:
        stor = st["stor"]
        for k, n in keys:
            ent = stor.get(k)
            if ent is not None:
                ent[0] -= 1
                if ent[0] <= 0:
                    del stor[k]
                    st["bytes"] -= n


def _leaf_release_all(st):
    st["keep"].clear()
    st["marks"].clear()
    st["stor"].clear()
    st["base"], st["bytes"], st["unmarked"] = st["count"], 0, 0


def _leaf_poll(st):
    """Release the operands whose leaf kernels have finished (markers that completed); non-blocking."""
    marks = st["marks"]
    while marks and L().set_stream_mark_done(marks[0][0]) == 1:
        _, upto = marks.popleft()
        keep = st["keep"]
        while st["base"] < upto:
            _, keys = keep.popleft()
            _leaf_drop(st, keys)
            st["base"] += 1
            _LEAF_STATS["released_by_marker"] += 1


def _leaf_mark(st):
    """After the kernels of a fork have been enqueued: a marker once enough new operands are held and a slot of the device's ring is free."""
    if st["unmarked"] < st["mark_bytes"] or len(st["marks"]) >= 8:
        return
    busy = {m[0] for m in st["marks"]}
    slot = next(s for s in range(8 * (st["idx"] % 8), 8 * (st["idx"] % 8) + 8) if s not in busy)
    check(L().set_stream_mark(st["raw"], slot, st["idx"]), "set_stream_mark")
    st["marks"].append((slot, st["count"]))
    st["unmarked"] = 0


def leaf_stats(reset=False):
    """{"max_keep_bytes", "released_by_marker", "early_joins"} since the last reset (tests / probes)."""
    out = dict(_LEAF_STATS)
    if reset:
        for k in _LEAF_STATS:
            _LEAF_STATS[k] = 0
    return out


def leaf_enabled():
    return os.environ.get("SET_AMD_LEAF_STREAM", "1") != "0"


def _leaf_state(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _LEAF.get(idx)
    if st is None:
        # lowest priority: its chip-filling GEMMs yield workgroup slots to the compute stream's short kernels (a default-priority stream
        # measured the same step time, profiles/r05_training_ab.log; one form kept)
        raw = C.c_void_p()
        with torch.cuda.device(idx):
            check(L().set_stream_create_low_priority(C.byref(raw)), "set_stream_create_low_priority")
        stream = torch.cuda.ExternalStream(raw.value, device=torch.device("cuda", idx))
        st = _LEAF[idx] = {"stream": stream, "raw": raw, "dirty": False, "keep": collections.deque(), "idx": idx,
                           "marks": collections.deque(), "stor": {}, "count": 0, "base": 0, "bytes": 0, "unmarked": 0,
                           "mark_bytes": _mb_env("SET_AMD_LEAF_MARK_MB", 128), "cap_bytes": _mb_env("SET_AMD_LEAF_KEEP_MB", 8192)}
    return st


def _order(st, fork):
    """fork: the leaf stream waits for the current stream; else (join) the current stream waits for the leaf stream: one C call on a cached
    event (torch's Event / Stream.wait_event / record_stream wrappers cost ~28 us per fork)."""
    if fork:
        check(L().set_stream_order(_stream(), st["raw"], 2 * st["idx"]), "set_stream_order")
    else:
        check(L().set_stream_order(st["raw"], _stream(), 2 * st["idx"] + 1), "set_stream_order")


class leaf_work:
    """with leaf_work(device, use, *tensors) as on_leaf: kernels launched inside (ops._stream()) go to the device's leaf stream, ordered after
    everything enqueued on the current stream so far, when `use` holds.  `tensors`: the operands; they are kept referenced until the
    compute stream has been made to wait for the leaf stream (leaf_join), so the allocator cannot hand their storage to a compute-stream
    kernel that might run before the leaf kernel has read it.  Nothing may be ALLOCATED inside (torch's current stream is unchanged)."""

    def __init__(self, dev, use, *tensors):
        self.on = bool(use) and dev.type == "cuda" and leaf_enabled() and getattr(ops._STREAM_TLS, "leaf", None) is None
        self.dev, self.tensors = dev, tensors

    def __enter__(self):
        if not self.on:
            return False
        st = _leaf_state(self.dev)
        if st["marks"]:
            _leaf_poll(st)
        if st["cap_bytes"] and st["bytes"] > st["cap_bytes"]:  # the bound: the compute stream waits for the leaf stream, everything held so far is released
            _order(st, False)
            _leaf_release_all(st)
            _LEAF_STATS["early_joins"] += 1
        _order(st, True)
        _leaf_hold(st, self.tensors)
        ops._STREAM_TLS.leaf = st["raw"]
        self._st = st
        if not st["dirty"]:
            st["dirty"] = True
            try:  # the stream that ran backward() waits for the leaf stream when the pass is over: callers may read .grad right away
                torch.autograd.Variable._execution_engine.queue_callback(leaf_join)
            except RuntimeError:
                pass  # not inside a backward pass (a test calling a backward function by hand): FlatAdamW.step() / abort_step() join
        return True

    def __exit__(self, *exc):
        if self.on:
            ops._STREAM_TLS.leaf = None
            if exc[0] is None:
                _leaf_mark(self._st)
        return False


def leaf_join():
    """The current stream waits for everything enqueued on the leaf streams (before the optimizer reads the gradients); the operands
    held for the leaf kernels are released (whatever reuses their storage is ordered after this point)."""
    for idx, st in _LEAF.items():
        if st["dirty"]:
            with torch.cuda.device(idx):
                _order(st, False)
            st["dirty"] = False
            _leaf_release_all(st)


def leaf_fence(dev):
    """For a consumer that must see BOTH streams' work without stalling the compute stream (a bucket all-reduce launched from an autograd
    hook): returns the leaf stream (torch object) after making it wait for the current stream, or None when no leaf work is pending."""
    st = _LEAF.get(dev.index if dev.index is not None else torch.cuda.current_device()) if dev.type == "cuda" else None
    if st is None or not st["dirty"]:
        return None
    _order(st, True)
    return st["stream"]


def zero_arena_begin(device, min_floats=0, owner=None):
    a = _ARENAS.setdefault(id(owner), _ZeroArena())
    want = max(int(min_floats), int(a.need * 1.05) + 4096)
    if a.buf is None or a.buf.device != device or a.buf.numel() < want:
        if a.buf is not None and a.buf.device == device:
            want = max(want, int(a.buf.numel() * 1.5))
        a.buf = torch.empty(want, dtype=torch.float32, device=device)
    a.buf.zero_()
    a.off, a.need = 0, 0
    _ACTIVE[0] = a


def zero_arena_end():
    _ACTIVE[0] = None


def _gzeros(shape, device):
    a = _ACTIVE[0]
    shape = tuple(int(d) for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
    n = 1
    for d in shape:
        n *= d
    if a is not None and a.buf.device == device:
        step = (n + 63) // 64 * 64  # 256-byte aligned slices
        a.need += step
        if a.off + step <= a.buf.numel():
            t = a.buf[a.off:a.off + n].view(shape)
            a.off += step
            return t
    return torch.zeros(shape, dtype=torch.float32, device=device)


def _zeros_like(t):
    return _gzeros(t.shape, t.device)


def grad_sink(param):
    """(target, owner): the optimizer-owned .grad view to accumulate a parameter gradient into, or (None, None) when the
    gradient has to travel through autograd (no flat optimizer, first step, parameter used more than once)."""
    owner = getattr(param, "_flat_owner", None) if param is not None else None
    if owner is None:
        return None, None
    t = owner.sink(param)
    return (t, owner) if t is not None else (None, None)


def _wgrad_impl(T):
    return IMPL_MFMA if T >= 16 else IMPL_NAIVE


_DET_SCRATCH = {}  # (device, stream) -> per-block / per-slice partial results of the ordered reductions (reused: stream-ordered)


def _stream_key(device):
    """(device, raw handle of the current stream): scratch buffers are reused in stream order, so every stream has its own."""
    return (device, _stream().value)


def _leaf_keep(buf):
    """A scratch buffer replaced while the leaf stream is the launch stream: kernels queued there may still use it -- held until the join."""
    if getattr(ops._STREAM_TLS, "leaf", None) is not None:
        for st in _LEAF.values():
            if st["dirty"]:
                _leaf_hold(st, buf)


def _det_scratch(device, n_floats):
    key = _stream_key(device)
    buf = _DET_SCRATCH.get(key)
    if buf is None or buf.numel() < n_floats:
        if buf is not None:
            _leaf_keep(buf)
        buf = torch.empty(int(n_floats * 1.25) + 4096, dtype=torch.float32, device=device)
        _DET_SCRATCH[key] = buf
    return buf


def _wg_scratch(device, need):
    key = _stream_key(device)
    buf = _WG_SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        if buf is not None:
            _leaf_keep(buf)
        buf = torch.empty(int(need * 1.25) + 1024, dtype=torch.float32, device=device)
        _WG_SCRATCH[key] = buf
    return buf


def channel_sum_(x, out, B, Cc, T):
    """out[c] += sum_{b,t} x[b][c][t], per-slice partials combined in slice order (deterministic)."""
    check(L().set_channel_sum_det(_p(x), _p(out), B, Cc, T, _p(_det_scratch(x.device, 2048 + Cc)), _stream()), "set_channel_sum_det")


_WG_SCRATCH = {}  # (device, stream) -> slice-partial buffer of the deterministic weight-gradient path (reused: stream-ordered)
DETERMINISTIC_WGRAD = True  # False: the round-1 split-K kernel with fp32 atomics (order-dependent bits)


def conv_wgrad(g, x, chan_add, dw, B, Cin, Cout, K, dil, pad, T, T_in, pro=0, pro_param=0.0, dw_ptr=None, dtype=None):
    """dW[co][ci][tap] += sum_{b,t} G[b][co][t] * P(X[b][ci][t + tap*dil - pad] + chan_add[b][ci]).  MFMA shapes go
    through set_conv1d_wgrad_det (per-slice partial sums reduced in slice order: bit-stable, no atomics) with bf16 or
    fp32 operands according to ops.compute_dtype(); tiny T uses the one-thread-per-weight kernel."""
    ptr = C.c_void_p(dw.data_ptr() if dw_ptr is None else dw_ptr)
    if T < 16 or not DETERMINISTIC_WGRAD:
        check(L().set_conv1d_wgrad(_p(g), _p(x), _p(chan_add), ptr, B, Cin, Cout, K, dil, pad, T, T_in, pro,
                                   float(pro_param), _wgrad_impl(T), _stream()), "set_conv1d_wgrad")
        return
    if dtype is not None:
        dt = dtype  # bf16 operands already in HBM (fused layer kernels)
    else:
        dt = _lib.DTYPE_BF16 if (ops.compute_dtype() == "bf16" and Cout >= 32 and Cin >= 32) else _lib.DTYPE_F32
    buf = _wg_scratch(g.device, L().set_conv1d_wgrad_scratch_floats(B, Cin, Cout, K, T, dt))
    check(L().set_conv1d_wgrad_det(_p(g), _p(x), _p(chan_add), ptr, B, Cin, Cout, K, dil, pad, T, T_in, pro,
                                   float(pro_param), dt, _p(buf), buf.numel(), _stream()), "set_conv1d_wgrad_det")


def conv_wgrad_grouped(g, x, chan_add, dw_ptr, groups, g_gs, x_gs, add_gs, dw_gs, B, Cin, Cout, K, dil, pad, T, T_in, dtype):
    """`groups` equal bf16 weight-gradient GEMMs (the same conv of every residual layer) in one launch + one ordered reduce;
    group q reads g + q g_gs, x + q x_gs, chan_add + q add_gs and adds into dw_ptr + q dw_gs (element strides)."""
    buf = _wg_scratch(g.device, L().set_conv1d_wgrad_grouped_scratch_floats(groups, B, Cin, Cout, K, T))
    check(L().set_conv1d_wgrad_det_grouped(_p(g), _p(x), _p(chan_add), C.c_void_p(dw_ptr), groups, g_gs, x_gs, add_gs, dw_gs, B, Cin,
                                           Cout, K, dil, pad, T, T_in, dtype, _p(buf), buf.numel(), _stream()),
          "set_conv1d_wgrad_det_grouped")


def _grouped_targets(params, dev):
    """Where the gradients of one parameter kind of all layers go: (base pointer, element stride between layers, what to hand to
    autograd per layer).  In place when the flat optimizer owns every one of them at a uniform stride (its layout is: each layer's
    parameters contiguous, in the same order); otherwise one zeroed [L, ...] temporary whose slices autograd accumulates."""
    sinks = [grad_sink(p_)[0] for p_ in params]
    n = params[0].numel()
    if all(t is not None for t in sinks):
        ptrs = [t.data_ptr() for t in sinks]
        step = (ptrs[1] - ptrs[0]) if len(ptrs) > 1 else 4 * n
        if step % 4 == 0 and step >= 4 * n and all(ptrs[i] == ptrs[0] + i * step for i in range(len(ptrs))):
            return ptrs[0], step // 4, [None] * len(params)
        # owned, but not at a uniform stride (never seen with FlatAdamW): through autograd like unowned parameters
    tmp = _gzeros(len(params) * n, dev).view(len(params), *params[0].shape)
    return tmp.data_ptr(), n, [tmp[l] for l in range(len(params))]


# --------------------------------------------------------------------------------------------------
class _Conv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, chan_add, res, cw, dil, pad, pro, pro_param, act, alpha, mask, impl, t_out=None):
        x = x.contiguous()
        y = ops.conv1d(x, cw, bias, dil=dil, pad=pad, pro=pro, pro_param=pro_param, act=act, alpha=alpha, res=res,
                       mask=mask, in_chan_add=chan_add, impl=impl, T_out=t_out)
        ctx.cw, ctx.cfg = cw, (dil, pad, pro, pro_param, act, alpha, impl)
        ctx.has = (bias is not None, chan_add is not None, res is not None)
        ctx.wparam, ctx.bparam = weight, bias  # the Parameter objects (gradient sinks of the flat optimizer)
        ctx.save_for_backward(x, chan_add, mask, y if act == "relu" else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, chan_add, mask, y = ctx.saved_tensors
        dil, pad, pro, pro_param, act, alpha, impl = ctx.cfg
        cw = ctx.cw
        has_bias, has_add, has_res = ctx.has
        dy = dy.contiguous()
        B, Cout, T = dy.shape
        Cin, T_in = x.shape[1], x.shape[2]
        if act == "none" and mask is None and alpha == 1.0:
            g = dy  # identity epilogue: no kernel, no copy (every DiffNet layer conv; 14 us x ~700 launches per step)
        else:
            g = torch.empty_like(dy)
            check(L().set_conv_epilogue_bwd(_p(dy), _p(y), _p(mask), _p(g), B, Cout, T, ACT[act], float(alpha), _stream()),
                  "set_conv_epilogue_bwd")
        dres = None
        if has_res and ctx.needs_input_grad[4]:
            if act == "none" and alpha == 1.0:
                dres = g
            else:
                dres = torch.empty_like(dy)
                check(L().set_conv_epilogue_bwd(_p(dy), None, _p(mask), _p(dres), B, Cout, T, 0, 1.0, _stream()),
                      "set_conv_epilogue_bwd")
        dx = dadd = None
        if ctx.needs_input_grad[0] or (has_add and ctx.needs_input_grad[3]):
            # input gradient = convolution of G with W'[ci][co][tap] = W[co][ci][tap], taps at s + pad - tap*dil
            dx = ops.conv1d(g, cw.transposed(), None, dil=-dil, pad=-pad, alpha=(1.0 / pro_param) if pro == "div" else 1.0,
                            T_iter=T_in, T_out=T_in, impl=impl)
            if has_add and ctx.needs_input_grad[3]:
                dadd = torch.empty(B, Cin, dtype=torch.float32, device=dy.device)
                check(L().set_row_sum(_p(dx), _p(dadd), B * Cin, T_in, 1.0, _stream()), "set_row_sum")
        dw = db = None
        want_w, want_b = ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        w_sink = b_sink = None
        if want_w:
            w = cw.raw()
            # plain [Cout,Cin,K] rows, possibly a row slice of a larger parameter (packed q/k/v projections)
            assert cw.stap == 1 and cw.sci == cw.K and cw.sco == cw.Cin * cw.K, "plain conv layout"
            whole = cw.base == 0 and w.numel() == Cout * Cin * cw.K
            w_sink = grad_sink(ctx.wparam)[0] if whole else None
            dw = w_sink if w_sink is not None else _gzeros(w.shape, dy.device)
        if want_b:
            b_sink = grad_sink(ctx.bparam)[0] if ctx.bparam.numel() == Cout else None
            db = b_sink if b_sink is not None else _gzeros(Cout, dy.device)
        # gradients that go straight into .grad are leaves of this backward pass (the optimizer is their next reader): leaf stream
        # (g may be the engine's own gradient buffer dy, or be handed on as the residual's gradient: the autograd engine accumulates a later
        # arrival into such a buffer IN PLACE only while it holds the last reference to it (input_buffer.cpp: can_accumulate_inplace), and
        # leaf_work keeps a reference to its operands until the streams are joined -- so nothing rewrites g under the leaf kernels)
        on_leaf = (not want_w or w_sink is not None) and (not want_b or b_sink is not None) and (want_w or want_b)
        with leaf_work(dy.device, on_leaf, g, x, chan_add):
            if want_w:
                conv_wgrad(g, x, chan_add, dw, B, Cin, Cout, cw.K, dil, pad, T, T_in, PRO[pro], pro_param,
                           dw_ptr=dw.data_ptr() + 4 * cw.base)
            if want_b:
                channel_sum_(g, db, B, Cout, T)
        if w_sink is not None:
            dw = None  # written in place; autograd still fires the parameter's post-accumulate hook (bucket launch)
        if b_sink is not None:
            db = None
        return (dx if ctx.needs_input_grad[0] else None, dw, db, dadd, dres, None, None, None, None, None, None, None,
                None, None, None)


_SPLIT_ACTS = ("gelu", "mish", "softplus", "tanh")


def conv1d(x, weight, bias=None, *, dil=1, pad=0, pro="none", pro_param=0.0, act="none", act_param=0.0, alpha=1.0,
           res=None, mask=None, in_chan_add=None, out=None, accumulate=False, impl=None, T_iter=None, T_out=None,
           out_stride=1, out_off=0):
    """Differentiable set_conv1d (stride-1, plain [Cout,Cin,K] weights).  Activations other than ReLU run as a
    separate kernel so their pre-activation is available to the backward."""
    assert isinstance(weight, ConvWeight) and out is None and not accumulate and out_stride == 1 and out_off == 0
    assert pro in ("none", "div") and T_iter is None
    if act in _SPLIT_ACTS:
        assert res is None
        z = _Conv1dFn.apply(x, weight.raw(), bias, in_chan_add, None, weight, dil, pad, pro, pro_param, "none", alpha,
                            None, impl, T_out)
        y = activation(z, act, act_param)
        return add_chan_mask(y, None, mask) if mask is not None else y
    return _Conv1dFn.apply(x, weight.raw(), bias, in_chan_add, res, weight, dil, pad, pro, pro_param, act, alpha, mask,
                           impl, T_out)


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, act, p):
        z = z.contiguous()
        y = torch.empty_like(z)
        check(L().set_act_fwd(_p(z), _p(y), z.numel(), ACT[act], float(p), _stream()), "set_act_fwd")
        ctx.save_for_backward(z)
        ctx.cfg = (act, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (z,) = ctx.saved_tensors
        act, p = ctx.cfg
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        check(L().set_act_bwd(_p(z), _p(dy), _p(dz), z.numel(), ACT[act], float(p), _stream()), "set_act_bwd")
        return dz, None, None


def activation(z, act, p=0.0):
    return _ActFn.apply(z, act, p)


class _LayerNormChFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mask, eps):
        x = x.contiguous()
        y = ops.layernorm_ch(x, gamma, beta, mask, eps)
        ctx.save_for_backward(x, gamma, mask)
        ctx.eps = eps
        ctx.gparam, ctx.bparam = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mask = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cc, T = x.shape
        dx = torch.empty_like(x)
        sg, _ = grad_sink(ctx.gparam)
        sb, _ = grad_sink(ctx.bparam)
        dg = sg if sg is not None else _zeros_like(gamma)
        db = sb if sb is not None else _zeros_like(gamma)
        part = _det_scratch(x.device, L().set_layernorm_ch_bwd_scratch(B, Cc, T))  # per-block partial rows (reused: stream-ordered)
        check(L().set_layernorm_ch_bwd(_p(x), _p(gamma), _p(mask), _p(dy), _p(dx), _p(dg), _p(db), _p(part), B, Cc, T,
                                       float(ctx.eps), _stream()), "set_layernorm_ch_bwd")
        return dx, (None if sg is not None else dg), (None if sb is not None else db), None, None


def layernorm_ch(x, gamma, beta, mask=None, eps=1e-5, out=None):
    return _LayerNormChFn.apply(x, gamma, beta, mask, eps)


class _PreLnFfnFn(torch.autograd.Function):
    """One tape node for the pre-LayerNorm feed-forward sub-block both models are built from (modules/commons/conv.py:24-65 ResidualBlock
    unit; modules/speech_editing/commons/transformer.py:76-113 + :619-652 TransformerFFNLayer inside Enc/DecSALayer):

        y = (x + W2 act(alpha (W1 (*) LN(x)) + alpha b1 ...) + b2) (* mask)      -- set_conv1d semantics: alpha scales conv + bias

    Forward = the four launches of the per-op tape (LN, conv k, activation on the saved pre-activation, 1x1 conv with the residual and the
    mask in its epilogue); backward = the same kernels in hand order.  What the single node saves: four autograd nodes per sub-block in
    each direction (17 sub-blocks per CampNet step, 8 per spec_denoiser step: the steps are bound by the host's enqueue time), the
    alpha-scaling launch (folded into the activation backward: set_act_bwd_scaled, same two roundings) and the `.contiguous()` /
    fan-out bookkeeping between them.  Gradients are bit-identical to the per-op tape's (tests: all-gradients goldens)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, cw1, cw2, dil, pad, alpha, act, act_param, eps, mask, t_out):
        x = x.contiguous()
        h = ops.layernorm_ch(x, gamma, beta, None, eps)
        z = ops.conv1d(h, cw1, b1, dil=dil, pad=pad, alpha=alpha, T_out=t_out)
        f = torch.empty_like(z)
        check(L().set_act_fwd(_p(z), _p(f), z.numel(), ACT[act], float(act_param), _stream()), "set_act_fwd")
        y = ops.conv1d(f, cw2, b2, res=x, mask=mask)
        ctx.save_for_backward(x, gamma, mask, h, z, f)
        ctx.cws, ctx.cfg = (cw1, cw2), (dil, pad, alpha, act, act_param, eps)
        ctx.params = (gamma, beta, w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mask, h, z, f = ctx.saved_tensors
        cw1, cw2 = ctx.cws
        dil, pad, alpha, act, act_param, eps = ctx.cfg
        p_gamma, p_beta, p_w1, p_b1, p_w2, p_b2 = ctx.params
        dy = dy.contiguous()
        B, Cc, T = dy.shape
        Cmid, T1, T_in = z.shape[1], z.shape[2], h.shape[2]
        dev = dy.device

        def tgt(param, cw=None):  # (accumulation target, gradient to hand to autograd or None when written in place)
            whole = cw is None or (cw.base == 0 and param.numel() == cw.Cout * cw.Cin * cw.K)
            sink = grad_sink(param)[0] if whole else None
            if sink is not None:
                return sink, None
            tmp = _gzeros(param.shape, dev)
            return tmp, tmp

        # ---- second conv (1x1, + residual, * mask): G2 = dy * mask is also the residual's gradient
        if mask is None:
            g2 = dy
        else:
            g2 = torch.empty_like(dy)
            check(L().set_conv_epilogue_bwd(_p(dy), None, _p(mask), _p(g2), B, Cc, T, 0, 1.0, _stream()), "set_conv_epilogue_bwd")
        df = ops.conv1d(g2, cw2.transposed(), None, dil=-1, pad=0, T_iter=T1, T_out=T1)
        (t_w2, r_w2), (t_b2, r_b2) = tgt(p_w2, cw2), tgt(p_b2)
        with leaf_work(dev, r_w2 is None and r_b2 is None, g2, f):  # (g2 may be dy itself: kept referenced, see _Conv1dFn.backward)
            conv_wgrad(g2, f, None, t_w2, B, Cmid, Cc, 1, 1, 0, T, T1, dw_ptr=t_w2.data_ptr() + 4 * cw2.base)
            channel_sum_(g2, t_b2, B, Cc, T)
        # ---- activation backward with the first conv's alpha folded in: G1 = gradient of the raw conv + bias
        g1 = torch.empty_like(z)
        check(L().set_act_bwd_scaled(_p(z), _p(df), _p(g1), z.numel(), ACT[act], float(act_param), float(alpha), _stream()), "set_act_bwd_scaled")
        dh = ops.conv1d(g1, cw1.transposed(), None, dil=-dil, pad=-pad, T_iter=T_in, T_out=T_in)
        (t_w1, r_w1), (t_b1, r_b1) = tgt(p_w1, cw1), tgt(p_b1)
        with leaf_work(dev, r_w1 is None and r_b1 is None, g1, h):
            conv_wgrad(g1, h, None, t_w1, B, Cc, Cmid, cw1.K, dil, pad, T1, T_in, dw_ptr=t_w1.data_ptr() + 4 * cw1.base)
            channel_sum_(g1, t_b1, B, Cmid, T1)
        # ---- LayerNorm backward, then the residual branch joins (same order as the fan-out node of the per-op tape: LN branch + residual)
        dxl = torch.empty_like(x)
        (t_g, r_g), (t_bt, r_bt) = tgt(p_gamma), tgt(p_beta)
        part = _det_scratch(dev, L().set_layernorm_ch_bwd_scratch(B, Cc, T_in))
        check(L().set_layernorm_ch_bwd_add(_p(x), _p(gamma), None, _p(dh), _p(g2), _p(dxl), _p(t_g), _p(t_bt), _p(part), B, Cc, T_in, float(eps),
                                           _stream()), "set_layernorm_ch_bwd_add")  # dx = LN gradient + residual gradient, one launch
        return (dxl, r_g, r_bt, r_w1, r_b1, r_w2, r_b2) + (None,) * 10


def preln_ffn(x, ln, cw1, b1, cw2, b2, *, dil=1, pad=0, alpha=1.0, act="gelu", act_param=0.0, mask=None, eps=1e-5, T_out=None):
    """(x + conv1x1(act(alpha conv_k(LN(x))))) (* mask) as one tape node; ln = (gamma, beta).  SET_AMD_FUSED_NODES=0: the per-op tape
    (fan-out, LayerNorm, conv, activation, conv: five nodes) -- the cross-check of tests/test_gpu_training.py and the A/B of the bench."""
    if not _fused_nodes():
        x_ln, x_res = fanout(x, 2)
        h = layernorm_ch(x_ln, ln[0], ln[1], eps=eps)
        h = conv1d(h, cw1, b1, dil=dil, pad=pad, alpha=alpha, act=act, act_param=act_param, T_out=T_out)
        return conv1d(h, cw2, b2, res=x_res, mask=mask)
    return _PreLnFfnFn.apply(x, ln[0], ln[1], cw1.raw(), b1, cw2.raw(), b2, cw1, cw2, dil, pad, alpha, act, act_param, eps, mask, T_out)


class _EmbeddingFn(torch.autograd.Function):
    """out = (base or 0) + scale * table[idx]  written channel-major."""

    @staticmethod
    def forward(ctx, idx, table, base, scale, padding_idx):
        ctx.padding_idx = -1 if padding_idx is None else int(padding_idx)
        if base is None:
            out = ops.embedding_bct(idx, table, scale=scale)
        else:
            out = ops.embedding_bct(idx, table, scale=scale, out=base.detach().clone(), accumulate=True)
        ctx.save_for_backward(idx)
        ctx.cfg = (tuple(table.shape), scale, base is not None)
        ctx.tparam = table
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        (n_rows, Cc), scale, has_base = ctx.cfg
        dout = dout.contiguous()
        B, T = idx.shape
        sink, _ = grad_sink(ctx.tparam)  # (a table that is a parameter used once: its gradient goes straight into .grad, on the leaf stream)
        dtab = sink if sink is not None else _gzeros((n_rows, Cc), dout.device)
        # ordered scatter (no atomics): gradient rows transposed to [B][T][C], per-(utterance, segment) partial tables
        doutT = ops.bct_to_btc(dout)
        S = L().set_scatter_rows_segments(T)
        with leaf_work(dout.device, sink is not None, dout, doutT, idx):
            check(L().set_scatter_rows_det(_p(idx), _p(doutT), _p(dtab), B, T, Cc, n_rows, float(scale), ctx.padding_idx, 0,
                                           _p(_det_scratch(dout.device, B * S * n_rows * Cc)), _stream()), "set_scatter_rows_det")
        return None, (None if sink is not None else dtab), (dout if has_base else None), None, None


def embedding_bct(idx, table, scale=1.0, out=None, accumulate=False, padding_idx=None):
    return _EmbeddingFn.apply(idx, table, out if accumulate else None, scale, padding_idx)


class _ExpandStatesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, mel2ph):
        enc = enc.contiguous()
        ctx.save_for_backward(mel2ph)
        ctx.shape = tuple(enc.shape)
        return ops.expand_states(enc, mel2ph)

    @staticmethod
    def backward(ctx, dout):
        (mel2ph,) = ctx.saved_tensors
        B, Cc, T_txt = ctx.shape
        dout = dout.contiguous()
        T = mel2ph.shape[1]
        doutT = ops.bct_to_btc(dout)
        dencT = _gzeros((B, T_txt, Cc), dout.device)
        S = L().set_scatter_rows_segments(T)
        check(L().set_scatter_rows_det(_p(mel2ph), _p(doutT), _p(dencT), B, T, Cc, T_txt, 1.0, -1, 1,
                                       _p(_det_scratch(dout.device, B * S * T_txt * Cc)), _stream()), "set_scatter_rows_det")
        return ops.btc_to_bct(dencT), None


def expand_states(enc_bct, mel2ph):
    return _ExpandStatesFn.apply(enc_bct, mel2ph)


class _AddChanMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, add, mask):
        x = x.contiguous()
        ctx.save_for_backward(mask)
        ctx.has_add = add is not None
        return ops.add_chan_mask(x, add, mask)

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        dout = dout.contiguous()
        B, Cc, T = dout.shape
        dx = ops.add_chan_mask(dout, None, mask) if mask is not None else dout
        dadd = None
        if ctx.has_add and ctx.needs_input_grad[1]:
            dadd = torch.empty(B, Cc, dtype=torch.float32, device=dout.device)
            check(L().set_row_sum(_p(dx), _p(dadd), B * Cc, T, 1.0, _stream()), "set_row_sum")
        return dx, dadd, None


def add_chan_mask(x, add=None, mask=None, out=None):
    return _AddChanMaskFn.apply(x, add, mask)


class _TransposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, to_bct):
        ctx.to_bct = to_bct
        x = x.contiguous()
        return ops.btc_to_bct(x) if to_bct else ops.bct_to_btc(x)

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        return (ops.bct_to_btc(d) if ctx.to_bct else ops.btc_to_bct(d)), None


def btc_to_bct(x):
    return _TransposeFn.apply(x, True)


def bct_to_btc(x):
    return _TransposeFn.apply(x, False)


class _GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        y = y.contiguous()
        ctx.save_for_backward(y)
        return ops.gate(y)

    @staticmethod
    def backward(ctx, dz):
        (y,) = ctx.saved_tensors
        dz = dz.contiguous()
        B, C2, T = y.shape
        dy = torch.empty_like(y)
        check(L().set_gate_bwd(_p(y), _p(dz), _p(dy), B, C2 // 2, T, _stream()), "set_gate_bwd")
        return dy


def gate(y):
    return _GateFn.apply(y)


class _ResSkipFn(torch.autograd.Function):
    """(x_out, skip_out) = ((x + o[:, :C]) / sqrt2, skip_in + o[:, C:])   (functional form of set_res_skip)."""

    @staticmethod
    def forward(ctx, x, o, skip_in):
        x, o = x.contiguous(), o.contiguous()
        first = skip_in is None
        skip = torch.empty_like(x) if first else skip_in.detach().clone()
        x_out = ops.res_skip(x, o, skip, first)
        ctx.first = first
        return x_out, skip

    @staticmethod
    def backward(ctx, dx_out, dskip):
        dx_out, dskip = dx_out.contiguous(), dskip.contiguous()
        B, Cc, T = dx_out.shape
        dx = torch.empty_like(dx_out)
        d_o = torch.empty(B, 2 * Cc, T, dtype=torch.float32, device=dx_out.device)
        check(L().set_res_skip_bwd(_p(dx_out), _p(dskip), _p(dx), _p(d_o), B, Cc, T, _stream()), "set_res_skip_bwd")
        return dx, d_o, (None if ctx.first else dskip)


def res_skip_fn(x, o, skip_in):
    return _ResSkipFn.apply(x, o, skip_in)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(L().set_dropout(_p(x), _p(y), x.numel(), float(p), int(seed), int(offset), _stream()), "set_dropout")
        ctx.cfg = (p, seed, offset)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(L().set_dropout(_p(dy), _p(dx), dy.numel(), float(p), int(seed), int(offset), _stream()), "set_dropout")
        return dx, None, None, None


def dropout(x, p, seed, offset=0):
    return _DropoutFn.apply(x, p, seed, offset) if p > 0 else x


class _FanoutFn(torch.autograd.Function):
    """n uses of one tensor: forward returns n aliases, backward sums the n gradients with set_sum_scale (so the
    fan-in accumulation is our kernel, not the autograd engine's elementwise add)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g.contiguous() for g in grads if g is not None]
        acc = gs[0]
        k = 1
        while k < len(gs):
            acc = ops.sum_div(acc, gs[k], gs[k + 1] if k + 1 < len(gs) else None, 1.0)
            k += 2
        return acc, None


def fanout(x, n):
    return _FanoutFn.apply(x, n) if n > 1 else (x,)


class _GradScaleFn(torch.autograd.Function):
    """x.detach() + s * (x - x.detach())  (fs.py:144-145,167-169): identity forward, gradient scaled by s."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        out = torch.empty_like(d)
        check(L().set_scale_bcast(_p(d), None, _p(out), d.numel(), 1, None, float(ctx.s), _stream()), "set_scale_bcast")
        return out, None


def grad_scale(x, s):
    return _GradScaleFn.apply(x, s) if s != 1 else x


# --------------------------------------------------------------------------------------------------
# DiffNet layer stack: fused forward (persistent Winograd kernel), hand-ordered backward
# --------------------------------------------------------------------------------------------------
class _DiffNetStackFn(torch.autograd.Function):
    """All L residual layers (diffnet.py:60-81,121-127) as ONE forward launch that also stores every layer's input x_l,
    pre-gate y_l and gated z_l; the backward walks the layers in reverse with the same kernels the per-op tape uses
    (res-skip / gate backward, conv dgrad = set_conv1d on the transposed weights, wgrad, channel / row sums).
    Inputs: hx [B,256,T], cond [B,H,T], dmat [B, L*256] (diffusion projections), then per layer
    (W_cond, b_cond, W_dil, b_dil, W_out, b_out).  Output: the skip sum [B,256,T] (the last layer's x is unused)."""

    @staticmethod
    def forward(ctx, dn, hx, cond, dmat, *params):
        L_, C_ = dn.n_layers, dn.C
        hx, cond, dmat = hx.contiguous(), cond.contiguous(), dmat.contiguous()
        B, _, T = hx.shape
        dev = hx.device
        layers = list(dn.residual_layers)
        condproj = torch.empty(B, L_ * 2 * C_, T, dtype=torch.float32, device=dev)
        for l, layer in enumerate(layers):
            ops.conv1d(cond, layer._w_cond, layer.conditioner_projection.bias, out=condproj[:, l * 2 * C_:(l + 1) * 2 * C_])
        x_all = torch.empty(L_ + 1, B, C_, T, dtype=torch.float32, device=dev)
        x_all[0].copy_(hx)
        y_all = torch.empty(L_, B, 2 * C_, T, dtype=torch.float32, device=dev)
        z_all = torch.empty(L_, B, C_, T, dtype=torch.float32, device=dev)
        skip = torch.empty(B, C_, T, dtype=torch.float32, device=dev)
        ws = ops.diffnet_stack(x_all[0], x_all[1], skip, condproj, dmat.data_ptr(), L_ * C_, 1, C_, dn.fused_packs(inference=False),
                               dn.dilation_cycle_length, x_all=x_all, save_y=y_all, save_z=z_all)
        ctx.dn, ctx.ws = dn, ws
        ctx.save_for_backward(cond, dmat, x_all, y_all, z_all)
        return skip

    @staticmethod
    def backward(ctx, dskip):
        dn = ctx.dn
        if int(ctx.ws[1]) != 0:  # (the forward's abort word; this read-back is the first host sync of the step)
            raise SetAmdError("set_diffnet_stack: a tile dependency wait of the training forward timed out")
        cond, dmat, x_all, y_all, z_all = ctx.saved_tensors
        L_, C_ = dn.n_layers, dn.C
        B, H, T = cond.shape
        dev = cond.device
        layers = list(dn.residual_layers)
        dskip = dskip.contiguous()
        need_cond = ctx.needs_input_grad[2]
        dcond = _zeros_like(cond) if need_cond else None
        dd = torch.empty(B, L_ * C_, dtype=torch.float32, device=dev)
        dx = _gzeros((B, C_, T), dev)  # the last layer's x output feeds nothing
        grads = []
        impl_w = _wgrad_impl(T)
        for l in range(L_ - 1, -1, -1):
            layer = layers[l]
            dil = layer.dilation
            dxr = torch.empty_like(dx)
            d_o = torch.empty(B, 2 * C_, T, dtype=torch.float32, device=dev)
            check(L().set_res_skip_bwd(_p(dx), _p(dskip), _p(dxr), _p(d_o), B, C_, T, _stream()), "set_res_skip_bwd")
            # output_projection (1x1, 256 -> 512)
            def tgt(param):  # the optimizer's .grad view when it owns the parameter, else a zeroed temporary
                sk, _ = grad_sink(param)
                return (sk, True) if sk is not None else (_zeros_like(param), False)

            dw_out, d1 = tgt(layer.output_projection.weight)
            db_out, d2 = tgt(layer.output_projection.bias)
            dw_cond, d3 = tgt(layer.conditioner_projection.weight)
            db, d4 = tgt(layer.conditioner_projection.bias)
            db2, d5 = tgt(layer.dilated_conv.bias)
            dw_dil, d6 = tgt(layer.dilated_conv.weight)
            dz = ops.conv1d(d_o, layer._w_out.transposed(), None, dil=-1, pad=0, T_iter=T, T_out=T)
            # gate
            dy = torch.empty(B, 2 * C_, T, dtype=torch.float32, device=dev)
            check(L().set_gate_bwd(_p(y_all[l]), _p(dz), _p(dy), B, C_, T, _stream()), "set_gate_bwd")
            # conditioner_projection (1x1, H -> 512): y = ... + W_cond cond + b_cond
            if need_cond:
                ops.conv1d(dy, layer._w_cond.transposed(), None, dil=-1, pad=0, T_iter=T, T_out=T, out=dcond,
                           accumulate=True)
            # dilated conv (k=3) on x_l + d_l
            dl = dmat[:, l * C_:(l + 1) * C_].contiguous()
            dxd = ops.conv1d(dy, layer._w_dil.transposed(), None, dil=-dil, pad=-dil, T_iter=T, T_out=T)
            ddl = torch.empty(B, C_, dtype=torch.float32, device=dev)
            check(L().set_row_sum(_p(dxd), _p(ddl), B * C_, T, 1.0, _stream()), "set_row_sum")
            dd[:, l * C_:(l + 1) * C_] = ddl
            dx = ops.sum_div(dxd, dxr)
            # the layer's six parameter gradients: leaves (leaf stream when they go straight into .grad)
            with leaf_work(dev, d1 and d2 and d3 and d4 and d5 and d6, d_o, dy, z_all, cond, x_all, dl):
                conv_wgrad(d_o, z_all[l], None, dw_out, B, C_, 2 * C_, 1, 1, 0, T, T)
                channel_sum_(d_o, db_out, B, 2 * C_, T)
                conv_wgrad(dy, cond, None, dw_cond, B, H, 2 * C_, 1, 1, 0, T, T)
                channel_sum_(dy, db, B, 2 * C_, T)  # = db_cond = db_dil
                channel_sum_(dy, db2, B, 2 * C_, T)
                conv_wgrad(dy, x_all[l], dl, dw_dil, B, C_, 2 * C_, 3, dil, dil, T, T)
            grads.append((None if d3 else dw_cond, None if d4 else db, None if d6 else dw_dil, None if d5 else db2,
                          None if d1 else dw_out, None if d2 else db_out))
        grads.reverse()
        flat = [g for tup in grads for g in tup]
        return (None, dx, dcond, dd, *flat)


class _SumLossesFn(torch.autograd.Function):
    """total = sum of the 0-d loss terms as ONE stack + ONE reduction (Python's sum() is an add launch per term, starting from 0, and an
    add / mul node per term on the way back); every term's gradient is the incoming one."""

    @staticmethod
    def forward(ctx, *vals):
        return torch.stack([v.reshape(()) for v in vals]).sum()

    @staticmethod
    def backward(ctx, g):
        return tuple(g for _ in ctx.needs_input_grad)


def sum_losses(vals):
    return _SumLossesFn.apply(*vals)


def _bf16_layer_pointers(dn, layers, imgs):
    """[(image, b_dil, b_cond, b_out, dilation)] of the residual layers as integers, cached on the DiffNet while the tensors stay where they are
    (parameters are views of the optimizer's flat buffer; the images are rebuilt in place)."""
    key = (imgs[0].data_ptr(), imgs[-1].data_ptr(), layers[0].dilated_conv.bias.data_ptr(), layers[-1].output_projection.bias.data_ptr(), len(layers))
    ent = getattr(dn, "_bf16_ptrs", None)
    if ent is None or ent[0] != key:
        ent = (key, [(imgs[l].data_ptr(), ly.dilated_conv.bias.data_ptr(), ly.conditioner_projection.bias.data_ptr(),
                      ly.output_projection.bias.data_ptr(), int(ly.dilation)) for l, ly in enumerate(layers)])
        dn._bf16_ptrs = ent
    return ent[1]


SWEEP_EVENTS = None  # bench.py sets a list: (start, end, launches) hipEvent pairs around the layer-backward sweep of every step


class _DiffNetStackBf16Fn(torch.autograd.Function):
    """All L residual layers with bf16 MFMA operands: one fused forward launch per layer that also saves the pre-gate y
    and the gated z in bf16, one fused backward launch per layer (d_o -> dz -> gate derivative -> dx, dcond, bias / step
    partial sums) and three weight-gradient GEMMs per layer on the saved bf16 operands (csrc/diffnet_bf16.hip).
    Same inputs / outputs as _DiffNetStackFn."""

    @staticmethod
    def forward(ctx, dn, hx, cond, dmat, *params):
        L_, C_ = dn.n_layers, dn.C
        hx, cond, dmat = hx.contiguous(), cond.contiguous(), dmat.contiguous()
        B, _, T = hx.shape
        dev = hx.device
        layers = list(dn.residual_layers)
        imgs = dn.bf16_layer_images()
        x_all = torch.empty(L_ + 1, B, C_, T, dtype=torch.float32, device=dev)
        x_all[0].copy_(hx)
        y16 = torch.empty(L_, B, 2 * C_, T, dtype=torch.bfloat16, device=dev)
        z16 = torch.empty(L_, B, C_, T, dtype=torch.bfloat16, device=dev)
        skip = torch.empty(B, C_, T, dtype=torch.float32, device=dev)
        a = _lib.SetDiffnetLayerBf16Args()
        a.skip, a.cond = skip.data_ptr(), cond.data_ptr()
        a.d_bs, a.d_cs = dmat.stride(0), 1
        a.B, a.T = B, T
        # per-layer addresses as integers (a tensor index or an nn.Module attribute per operand and layer cost ~0.4 ms of host time per step
        # in this loop and the backward sweep: the step is bound by the host's enqueue time)
        x0, sx = x_all.data_ptr(), 4 * B * C_ * T
        y0, z0, d0 = y16.data_ptr(), z16.data_ptr(), dmat.data_ptr()
        per = _bf16_layer_pointers(dn, layers, imgs)
        fn, st, ref = L().set_diffnet_layer_fwd_bf16, _stream(), C.byref(a)
        for l in range(L_):
            a.x_in, a.x_out = x0 + l * sx, x0 + (l + 1) * sx
            a.dstep = d0 + 4 * l * C_
            a.img, a.b_dil, a.b_cond, a.b_out, a.dil = per[l]
            a.y16, a.z16 = y0 + l * sx, z0 + l * (sx // 2)  # bf16 [B][2C][T] and [B][C][T]
            a.first = int(l == 0)
            check(fn(ref, st), "set_diffnet_layer_fwd_bf16")
        ctx.dn, ctx.imgs = dn, imgs
        ctx.save_for_backward(cond, dmat, x_all, y16, z16)
        return skip

    @staticmethod
    def backward(ctx, dskip):
        dn, imgs = ctx.dn, ctx.imgs
        cond, dmat, x_all, y16, z16 = ctx.saved_tensors
        L_, C_ = dn.n_layers, dn.C
        B, H, T = cond.shape
        dev = cond.device
        layers = list(dn.residual_layers)
        dskip = dskip.contiguous()
        dcond = torch.empty_like(cond)
        dd = torch.empty(B, L_ * C_, dtype=torch.float32, device=dev)
        # Weight gradients of ALL layers in three grouped launches after the sweep (dy / d_o of every layer are kept: 2 x L x 26 MB
        # at B = 32, T = 800) when the layers share one dilation; else three GEMMs per layer inside the sweep.
        grouped = len({layer.dilation for layer in layers}) == 1
        # the grouped form keeps dy / d_o of ALL layers alive (2 x L x B x 2C x T bf16: ~1 GB at L = 20, B = 32, T = 800, next to x_all,
        # y16, z16); beyond a budget (SET_AMD_GROUPED_WGRAD_MB, default 4096 MB -- 1.4 % of the 288 GB) it falls back to the per-layer form
        if grouped and 2 * L_ * B * 2 * C_ * T * 2 > float(os.environ.get("SET_AMD_GROUPED_WGRAD_MB", "4096")) * 2 ** 20:
            grouped = False
        n_slab = L_ if grouped else 1
        dy16_all = torch.empty(n_slab, B, 2 * C_, T, dtype=torch.bfloat16, device=dev)
        do16_all = torch.empty(n_slab, B, 2 * C_, T, dtype=torch.bfloat16, device=dev)
        dx = [torch.empty(B, C_, T, dtype=torch.float32, device=dev) for _ in range(2)]
        G16, GX16 = _lib.DTYPE_BF16_G16, _lib.DTYPE_BF16_G16_X16
        a = _lib.SetDiffnetLayerBf16BwdArgs()
        a.dskip, a.dcond = dskip.data_ptr(), dcond.data_ptr()
        a.B, a.T = B, T
        grads = []
        cur = None  # gradient w.r.t. the current layer's x_out (None for the last layer: its x_out feeds nothing)
        # grouped: the per-tile partial sums (bias / step-offset gradients) of ALL layers are kept and reduced by ONE launch after the sweep
        # (L x ~2 MB at B = 32, T = 800) instead of one 5 us launch per layer in the middle of the chain of layer kernels
        if grouped:
            tiles_g = L().set_diffnet_layer_bwd_bf16_tiles(T, layers[0].dilation)
            pdbo_all = torch.empty(L_, B * tiles_g, 2 * C_, dtype=torch.float32, device=dev)
            pdby_all = torch.empty(L_, B * tiles_g, 2 * C_, dtype=torch.float32, device=dev)
            pdd_all = torch.empty(L_, B * tiles_g, C_, dtype=torch.float32, device=dev)
        if SWEEP_EVENTS is not None:  # measurement hook (bench.py): hipEvents around the L launches of diffnet_layer_bwd_bf16_kernel
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if grouped:  # the sweep as 20 launches on integer addresses (see forward); everything else of this form happens after the loop
            per = _bf16_layer_pointers(dn, layers, imgs)
            dy0, do0, y0, s16 = dy16_all.data_ptr(), do16_all.data_ptr(), y16.data_ptr(), 2 * B * 2 * C_ * T
            pb0, py0, pd0, sp = pdbo_all.data_ptr(), pdby_all.data_ptr(), pdd_all.data_ptr(), 4 * B * tiles_g * C_
            dxp = (dx[0].data_ptr(), dx[1].data_ptr())
            fn, st, ref = L().set_diffnet_layer_bwd_bf16, _stream(), C.byref(a)
            a.dil = layers[0].dilation
            curp = None
            for l in range(L_ - 1, -1, -1):
                a.dy16, a.do16, a.y16 = dy0 + l * s16, do0 + l * s16, y0 + l * s16
                a.dx_out, a.img, a.dx = curp, per[l][0], dxp[l & 1]
                a.part_dbo, a.part_dby, a.part_dd = pb0 + l * 2 * sp, py0 + l * 2 * sp, pd0 + l * sp
                a.dcond_first = int(l == L_ - 1)
                check(fn(ref, st), "set_diffnet_layer_bwd_bf16")
                grads.append([None] * 6)
                curp = dxp[l & 1]
            cur = dx[0]
        for l in (range(L_ - 1, -1, -1) if not grouped else ()):
            layer = layers[l]
            dil = layer.dilation
            tiles = L().set_diffnet_layer_bwd_bf16_tiles(T, dil)
            if grouped:
                pdbo_c, pdby_c, pdd_c = pdbo_all[l], pdby_all[l], pdd_all[l]
            else:
                pdbo_c = torch.empty(B * tiles, 2 * C_, dtype=torch.float32, device=dev)
                pdby_c = torch.empty(B * tiles, 2 * C_, dtype=torch.float32, device=dev)
                pdd_c = torch.empty(B * tiles, C_, dtype=torch.float32, device=dev)
            out = dx[l & 1]
            dy16, do16 = dy16_all[l if grouped else 0], do16_all[l if grouped else 0]
            a.dy16, a.do16 = dy16.data_ptr(), do16.data_ptr()
            a.dx_out = None if cur is None else cur.data_ptr()
            a.y16, a.img, a.dx = y16[l].data_ptr(), imgs[l].data_ptr(), out.data_ptr()
            a.part_dbo, a.part_dby, a.part_dd = pdbo_c.data_ptr(), pdby_c.data_ptr(), pdd_c.data_ptr()
            a.dil, a.dcond_first = dil, int(l == L_ - 1)
            check(L().set_diffnet_layer_bwd_bf16(C.byref(a), _stream()), "set_diffnet_layer_bwd_bf16")
            # bias / step-offset gradients: ordered sums of the per-tile partials, one launch per layer (straight into
            # .grad when the flat optimizer owns it, else into a zeroed temporary that autograd accumulates)
            def btgt(param):  # (accumulation target, gradient to hand to autograd or None when written in place)
                sink, _ = grad_sink(param)
                if sink is not None:
                    return sink, None
                tmp = _gzeros(2 * C_, dev)
                return tmp, tmp

            def wg(param, *args, **kw):
                sink, owner = grad_sink(param)
                tgt = sink if sink is not None else _zeros_like(param)
                conv_wgrad(args[0], args[1], args[2], tgt, *args[3:], **kw)  # (compute stream: the next layer's launch overwrites dy16 / do16)
                return None if sink is not None else tgt

            if grouped:
                dw_out = dw_cond = dw_dil = db_out = db_dil = db_cond = None  # filled in below
            else:
                (t_out, db_out), (t_dil, db_dil), (t_cond, db_cond) = (btgt(layer.output_projection.bias),
                                                                       btgt(layer.dilated_conv.bias),
                                                                       btgt(layer.conditioner_projection.bias))
                check(L().set_diffnet_layer_bwd_reduce(_p(pdbo_c), _p(pdby_c), _p(pdd_c), B, tiles, _p(t_out), _p(t_dil), _p(t_cond),
                                                       dd.data_ptr() + 4 * l * C_, dd.stride(0), _stream()), "set_diffnet_layer_bwd_reduce")
                dw_out = wg(layer.output_projection.weight, do16, z16[l], None, B, C_, 2 * C_, 1, 1, 0, T, T, dtype=GX16)
                dw_cond = wg(layer.conditioner_projection.weight, dy16, cond, None, B, H, 2 * C_, 1, 1, 0, T, T, dtype=G16)
                dl = dmat[:, l * C_:(l + 1) * C_].contiguous()
                dw_dil = wg(layer.dilated_conv.weight, dy16, x_all[l], dl, B, C_, 2 * C_, 3, dil, dil, T, T, dtype=G16)
            grads.append([dw_cond, db_cond, dw_dil, db_dil, dw_out, db_out])
            cur = out
        if SWEEP_EVENTS is not None:
            ev1.record()
            SWEEP_EVENTS.append((ev0, ev1, L_ if grouped else 4 * L_))  # (grouped: nothing but the L layer launches lies between the events)
        grads.reverse()
        if grouped:
            dil = layers[0].dilation
            n_act = B * 2 * C_ * T
            pb_o, sb_o, rb_o = _grouped_targets([ly.output_projection.bias for ly in layers], dev)
            pb_d, sb_d, rb_d = _grouped_targets([ly.dilated_conv.bias for ly in layers], dev)
            pb_c, sb_c, rb_c = _grouped_targets([ly.conditioner_projection.bias for ly in layers], dev)
            check(L().set_diffnet_layers_bwd_reduce(_p(pdbo_all), _p(pdby_all), _p(pdd_all), B, tiles_g, L_, C.c_void_p(pb_o), sb_o,
                                                    C.c_void_p(pb_d), sb_d, C.c_void_p(pb_c), sb_c, _p(dd), dd.stride(0), C_, _stream()),
                  "set_diffnet_layers_bwd_reduce")
            for l in range(L_):
                grads[l][1], grads[l][3], grads[l][5] = rb_c[l], rb_d[l], rb_o[l]
            p_out, s_out, r_out = _grouped_targets([ly.output_projection.weight for ly in layers], dev)
            p_c, s_c, r_c = _grouped_targets([ly.conditioner_projection.weight for ly in layers], dev)
            p_d, s_d, r_d = _grouped_targets([ly.dilated_conv.weight for ly in layers], dev)
            dl_all = dmat.view(B, L_, C_).transpose(0, 1).contiguous()  # [L][B][C] step offsets (the conv's input is x + d)
            # written straight into .grad (every r_* entry None): leaves of the backward pass -- 2.3 ms of chip-filling GEMMs that run on the
            # leaf stream under the conditioner's backward (hundreds of 5-40 us kernels on a few dozen workgroups each)
            in_place = all(r is None for r in r_out + r_c + r_d)
            with leaf_work(dev, in_place, do16_all, dy16_all, z16, cond, x_all, dl_all):
                conv_wgrad_grouped(do16_all, z16, None, p_out, L_, n_act, B * C_ * T, 0, s_out, B, C_, 2 * C_, 1, 1, 0, T, T, GX16)
                conv_wgrad_grouped(dy16_all, cond, None, p_c, L_, n_act, 0, 0, s_c, B, H, 2 * C_, 1, 1, 0, T, T, G16)
                conv_wgrad_grouped(dy16_all, x_all, dl_all, p_d, L_, n_act, B * C_ * T, B * C_, s_d, B, C_, 2 * C_, 3, dil, dil, T, T, G16)
            for l in range(L_):
                grads[l][0], grads[l][2], grads[l][4] = r_c[l], r_d[l], r_out[l]
        flat = [g for tup in grads for g in tup]
        need_cond = ctx.needs_input_grad[2]
        return (None, cur, dcond if need_cond else None, dd, *flat)


def _uniform_stride(tensors):
    """Element stride between the (equal-shaped, contiguous fp32) tensors when they sit at one stride in memory, else None."""
    ptrs = [t.data_ptr() for t in tensors]
    n = tensors[0].numel()
    step = (ptrs[1] - ptrs[0]) if len(ptrs) > 1 else 4 * n
    if step % 4 or step < 4 * n or any(ptrs[i] != ptrs[0] + i * step for i in range(len(ptrs))):
        return None
    if any((not t.is_contiguous()) or t.dtype != torch.float32 for t in tensors):
        return None
    return step // 4


class _StepProjFn(torch.autograd.Function):
    """dmat[n][l*C + co] = diffusion_projection_l(h)[co] for every residual layer l in one launch (diffnet.py:66,72); backward: one
    launch for the input gradient partials (+ their ordered sum) and one for all weight / bias gradients (csrc/train.hip)."""

    @staticmethod
    def forward(ctx, h, w_ls, b_ls, *params):
        L_ = len(params) // 2
        ws, bs = params[:L_], params[L_:]
        Cc, N = h.shape[1], h.shape[2]
        h = h.contiguous()
        out = torch.empty(N, L_ * Cc, dtype=torch.float32, device=h.device)
        check(L().set_step_proj_fwd(_p(h), _p(ws[0]), w_ls, _p(bs[0]), b_ls, _p(out), L_, Cc, N, _stream()), "set_step_proj_fwd")
        ctx.ws, ctx.bs, ctx.w_ls = ws, bs, w_ls
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        ws, bs = ctx.ws, ctx.bs
        L_, Cc, N = len(ws), h.shape[1], h.shape[2]
        g = g.contiguous()
        dev = h.device
        p_w, s_w, r_w = _grouped_targets(list(ws), dev)
        p_b, s_b, r_b = _grouped_targets(list(bs), dev)
        dh = torch.empty_like(h)
        scratch = _det_scratch(dev, L().set_step_proj_bwd_scratch_floats(L_, Cc, N))
        check(L().set_step_proj_bwd_dh(_p(g), _p(ws[0]), ctx.w_ls, _p(dh), _p(scratch), L_, Cc, N, _stream()), "set_step_proj_bwd_dh")
        sinks = all(r is None for r in r_w) and all(r is None for r in r_b)  # every dW / db goes straight into the flat gradient buffer
        with leaf_work(dev, sinks, h, g):
            check(L().set_step_proj_bwd_dw(_p(h), _p(g), C.c_void_p(p_w), s_w, C.c_void_p(p_b), s_b, L_, Cc, N, _stream()),
                  "set_step_proj_bwd_dw")
        return (dh, None, None, *r_w, *r_b)


def step_projections(dn, h):
    """[n, L*C] step offsets of all residual layers from the step embedding h [1, C, n]; None when the layers' parameters are not laid
    out at one stride (no flat optimizer and separately allocated tensors) or the shape is outside the kernel."""
    ws = [layer.diffusion_projection.weight for layer in dn.residual_layers]
    bs = [layer.diffusion_projection.bias for layer in dn.residual_layers]
    Cc, N = h.shape[1], h.shape[2]
    if h.shape[0] != 1 or Cc % 64 or N > 64 or Cc * (32 if N <= 32 else 64) > 16384:
        return None
    w_ls, b_ls = _uniform_stride(ws), _uniform_stride(bs)
    if w_ls is None or b_ls is None:
        return None
    return _StepProjFn.apply(h, w_ls, b_ls, *ws, *bs)


def diffnet_stack_train_bf16(dn, hx, cond, dmat):
    params = []
    for layer in dn.residual_layers:
        params += [layer.conditioner_projection.weight, layer.conditioner_projection.bias, layer.dilated_conv.weight,
                   layer.dilated_conv.bias, layer.output_projection.weight, layer.output_projection.bias]
    return _DiffNetStackBf16Fn.apply(dn, hx, cond, dmat, *params)


def diffnet_stack_train(dn, hx, cond, dmat):
    params = []
    for layer in dn.residual_layers:
        params += [layer.conditioner_projection.weight, layer.conditioner_projection.bias, layer.dilated_conv.weight,
                   layer.dilated_conv.bias, layer.output_projection.weight, layer.output_projection.bias]
    return _DiffNetStackFn.apply(dn, hx, cond, dmat, *params)


# --------------------------------------------------------------------------------------------------
# attention + the small CampNet ops
# --------------------------------------------------------------------------------------------------
def _attn_bwd(qv, kv, vv, p, do, dqv, dkv, dvv, heads, alpha):
    """Gradients of o = softmax(alpha q k^T) v into the views dqv / dkv / dvv (four strided batched GEMMs + one
    softmax backward)."""
    MV = ops.MatView
    dov = MV.heads(do, heads)
    dp = torch.empty_like(p)
    ops.bmm(dov, vv.t, MV.scores(dp))            # dP = dO V^T
    ops.bmm(MV.scores(p).t, dov, dvv)            # dV = P^T dO
    ds = ops.softmax_rows_bwd(p, dp)
    ops.bmm(MV.scores(ds), kv, dqv, alpha=alpha)   # dQ = alpha dS K
    ops.bmm(MV.scores(ds).t, qv, dkv, alpha=alpha)  # dK = alpha dS^T Q


class _SelfAttnFn(torch.autograd.Function):
    """qkv [B, 3H, T] (packed in_proj output) -> o [B, H, T] (+ p when asked).  Default: the fused kernels -- the scores
    never reach HBM and the backward recomputes P from the saved log-sum-exp (csrc/attention_fused.hip); SET_AMD_ATTN_FUSED=0:
    bmm -> softmax -> bmm with the probabilities saved."""

    @staticmethod
    def forward(ctx, qkv, heads, kpm, fill, alpha, want_p):
        qkv = qkv.contiguous()
        H = qkv.shape[1] // 3
        MV = ops.MatView
        ctx.cfg = (heads, alpha, kpm, fill)
        if ops.attention_fused_on(H // heads):
            o, lse, p = ops.attention_fused(MV.heads(qkv, heads, 0, H), MV.heads(qkv, heads, H, H), MV.heads(qkv, heads, 2 * H, H),
                                            heads, kpm, fill, alpha, want_p)
            ctx.fused = True
            ctx.save_for_backward(qkv, o, lse)
        else:
            o, p = ops.attention_views(MV.heads(qkv, heads, 0, H), MV.heads(qkv, heads, H, H), MV.heads(qkv, heads, 2 * H, H),
                                       heads, kpm, fill, alpha)
            ctx.fused = False
            ctx.save_for_backward(qkv, p)
        if p is None:
            p = qkv.new_empty(0)
        ctx.mark_non_differentiable(p)
        return o, p

    @staticmethod
    def backward(ctx, do, _dp):
        heads, alpha, kpm, fill = ctx.cfg
        qkv = ctx.saved_tensors[0]
        H = qkv.shape[1] // 3
        MV = ops.MatView
        d = torch.empty_like(qkv)
        views = (MV.heads(qkv, heads, 0, H), MV.heads(qkv, heads, H, H), MV.heads(qkv, heads, 2 * H, H))
        dviews = (MV.heads(d, heads, 0, H), MV.heads(d, heads, H, H), MV.heads(d, heads, 2 * H, H))
        if ctx.fused:
            _, o, lse = ctx.saved_tensors
            ops.attention_fused_bwd(*views, o, lse, do.contiguous(), *dviews, heads, kpm, fill, alpha)
        else:
            _attn_bwd(*views, ctx.saved_tensors[1], do.contiguous(), *dviews, heads, alpha)
        return d, None, None, None, None, None


class _CrossAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, heads, kpm, fill, alpha, want_p):
        q, kv = q.contiguous(), kv.contiguous()
        H = q.shape[1]
        MV = ops.MatView
        ctx.cfg = (heads, alpha, kpm, fill)
        if ops.attention_fused_on(H // heads):
            o, lse, p = ops.attention_fused(MV.heads(q, heads), MV.heads(kv, heads, 0, H), MV.heads(kv, heads, H, H), heads, kpm,
                                            fill, alpha, want_p)
            ctx.fused = True
            ctx.save_for_backward(q, kv, o, lse)
        else:
            o, p = ops.attention_views(MV.heads(q, heads), MV.heads(kv, heads, 0, H), MV.heads(kv, heads, H, H), heads, kpm,
                                       fill, alpha)
            ctx.fused = False
            ctx.save_for_backward(q, kv, p)
        if p is None:
            p = q.new_empty(0)
        ctx.mark_non_differentiable(p)
        return o, p

    @staticmethod
    def backward(ctx, do, _dp):
        heads, alpha, kpm, fill = ctx.cfg
        q, kv = ctx.saved_tensors[:2]
        H = q.shape[1]
        MV = ops.MatView
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        views = (MV.heads(q, heads), MV.heads(kv, heads, 0, H), MV.heads(kv, heads, H, H))
        dviews = (MV.heads(dq, heads), MV.heads(dkv, heads, 0, H), MV.heads(dkv, heads, H, H))
        if ctx.fused:
            o, lse = ctx.saved_tensors[2:]
            ops.attention_fused_bwd(*views, o, lse, do.contiguous(), *dviews, heads, kpm, fill, alpha)
        else:
            _attn_bwd(*views, ctx.saved_tensors[2], do.contiguous(), *dviews, heads, alpha)
        return dq, dkv, None, None, None, None, None


def self_attention(qkv, heads, key_padding_mask=None, fill=float("-inf"), alpha=1.0, want_p=False):
    """(o, p): p is an empty tensor unless want_p (the fused kernels do not materialise the probabilities)."""
    return _SelfAttnFn.apply(qkv, heads, key_padding_mask, fill, alpha, want_p)


def cross_attention(q, kv, heads, key_padding_mask=None, fill=-1e8, alpha=1.0, want_p=True):
    return _CrossAttnFn.apply(q, kv, heads, key_padding_mask, fill, alpha, want_p)


def _tape_tgt(param, dev, whole=True):
    """(accumulation target, gradient to hand to autograd or None when the kernel writes the optimizer's .grad view in place)."""
    sink = grad_sink(param)[0] if whole else None
    if sink is not None:
        return sink, None
    tmp = _gzeros(param.shape, dev)
    return tmp, tmp


def _preln_tail(x, gamma, p_gamma, p_beta, dh, g2, eps):
    """LayerNorm backward of a pre-LN sub-block, then the residual branch joins (LN branch + residual, the per-op tape's order)."""
    B, Cc, T = x.shape
    dxl = torch.empty_like(x)
    (t_g, r_g), (t_b, r_b) = _tape_tgt(p_gamma, x.device), _tape_tgt(p_beta, x.device)
    part = _det_scratch(x.device, L().set_layernorm_ch_bwd_scratch(B, Cc, T))
    check(L().set_layernorm_ch_bwd_add(_p(x), _p(gamma), None, _p(dh), _p(g2), _p(dxl), _p(t_g), _p(t_b), _p(part), B, Cc, T, float(eps),
                                       _stream()), "set_layernorm_ch_bwd_add")  # dx = LN gradient + residual gradient, one launch
    return dxl, r_g, r_b


class _PreLnSelfAttnFn(torch.autograd.Function):
    """One tape node for x + out_proj(self_attention(in_proj(LN(x)))) (* mask): the self-attention sub-block of Enc/DecSALayer
    (modules/speech_editing/commons/transformer.py:138-189,421-422,619-652; bias-free packed projections).  Same kernels as the per-op tape
    (LN, packed 1x1 in_proj, fused attention, 1x1 out_proj with the residual and the mask in its epilogue); fused attention kernels only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w_in, w_out, cw_qkv, cw_out, heads, kpm, fill, alpha, mask, eps):
        x = x.contiguous()
        H = x.shape[1]
        MV = ops.MatView
        h = ops.layernorm_ch(x, gamma, beta, None, eps)
        qkv = ops.conv1d(h, cw_qkv)
        o, lse, _ = ops.attention_fused(MV.heads(qkv, heads, 0, H), MV.heads(qkv, heads, H, H), MV.heads(qkv, heads, 2 * H, H), heads, kpm,
                                        fill, alpha, False)
        y = ops.conv1d(o, cw_out, None, res=x, mask=mask)
        ctx.save_for_backward(x, gamma, mask, kpm, h, qkv, o, lse)
        ctx.cfg, ctx.cws, ctx.params = (heads, fill, alpha, eps), (cw_qkv, cw_out), (gamma, beta, w_in, w_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mask, kpm, h, qkv, o, lse = ctx.saved_tensors
        heads, fill, alpha, eps = ctx.cfg
        cw_qkv, cw_out = ctx.cws
        p_gamma, p_beta, p_win, p_wout = ctx.params
        dy = dy.contiguous()
        B, H, T = dy.shape
        dev, MV = dy.device, ops.MatView
        if mask is None:
            g2 = dy
        else:
            g2 = torch.empty_like(dy)
            check(L().set_conv_epilogue_bwd(_p(dy), None, _p(mask), _p(g2), B, H, T, 0, 1.0, _stream()), "set_conv_epilogue_bwd")
        do = ops.conv1d(g2, cw_out.transposed(), None, dil=-1, pad=0, T_iter=T, T_out=T)
        t_wo, r_wo = _tape_tgt(p_wout, dev)
        with leaf_work(dev, r_wo is None, g2, o):
            conv_wgrad(g2, o, None, t_wo, B, H, H, 1, 1, 0, T, T)
        dqkv = torch.empty_like(qkv)
        ops.attention_fused_bwd(MV.heads(qkv, heads, 0, H), MV.heads(qkv, heads, H, H), MV.heads(qkv, heads, 2 * H, H), o, lse, do,
                                MV.heads(dqkv, heads, 0, H), MV.heads(dqkv, heads, H, H), MV.heads(dqkv, heads, 2 * H, H), heads, kpm, fill, alpha)
        dh = ops.conv1d(dqkv, cw_qkv.transposed(), None, dil=-1, pad=0, T_iter=T, T_out=T)
        t_wi, r_wi = _tape_tgt(p_win, dev)
        with leaf_work(dev, r_wi is None, dqkv, h):
            conv_wgrad(dqkv, h, None, t_wi, B, H, 3 * H, 1, 1, 0, T, T)
        dx, r_g, r_b = _preln_tail(x, gamma, p_gamma, p_beta, dh, g2, eps)
        return (dx, r_g, r_b, r_wi, r_wo) + (None,) * 8


class _PreLnCrossAttnFn(torch.autograd.Function):
    """One tape node for x + out_proj(attention(q = in_proj_q(LN(x)), k, v = in_proj_kv(enc))): the encoder-decoder attention sub-block of
    DecSALayer (transformer.py:283-410,531-609); returns (y, probabilities or an empty tensor).  The q rows and the k / v rows of the packed
    in_proj weight get ONE gradient (the per-op tape returns two full-size temporaries that autograd adds); fused attention kernels only."""

    @staticmethod
    def forward(ctx, x, enc, gamma, beta, w_in, w_out, cw_q, cw_kv, cw_out, heads, kpm, fill, alpha, want_p, eps):
        x, enc = x.contiguous(), enc.contiguous()
        H = x.shape[1]
        MV = ops.MatView
        h = ops.layernorm_ch(x, gamma, beta, None, eps)
        q = ops.conv1d(h, cw_q)
        kv = ops.conv1d(enc, cw_kv)
        o, lse, p = ops.attention_fused(MV.heads(q, heads), MV.heads(kv, heads, 0, H), MV.heads(kv, heads, H, H), heads, kpm, fill, alpha, want_p)
        y = ops.conv1d(o, cw_out, None, res=x)
        ctx.save_for_backward(x, enc, gamma, kpm, h, q, kv, o, lse)
        ctx.cfg, ctx.cws, ctx.params = (heads, fill, alpha, eps), (cw_q, cw_kv, cw_out), (gamma, beta, w_in, w_out)
        if p is None:
            p = x.new_empty(0)
        ctx.mark_non_differentiable(p)
        return y, p

    @staticmethod
    def backward(ctx, dy, _dp):
        x, enc, gamma, kpm, h, q, kv, o, lse = ctx.saved_tensors
        heads, fill, alpha, eps = ctx.cfg
        cw_q, cw_kv, cw_out = ctx.cws
        p_gamma, p_beta, p_win, p_wout = ctx.params
        g2 = dy.contiguous()
        B, H, T = g2.shape
        Tk = enc.shape[2]
        dev, MV = g2.device, ops.MatView
        do = ops.conv1d(g2, cw_out.transposed(), None, dil=-1, pad=0, T_iter=T, T_out=T)
        t_wo, r_wo = _tape_tgt(p_wout, dev)
        with leaf_work(dev, r_wo is None, g2, o):
            conv_wgrad(g2, o, None, t_wo, B, H, H, 1, 1, 0, T, T)
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        ops.attention_fused_bwd(MV.heads(q, heads), MV.heads(kv, heads, 0, H), MV.heads(kv, heads, H, H), o, lse, do,
                                MV.heads(dq, heads), MV.heads(dkv, heads, 0, H), MV.heads(dkv, heads, H, H), heads, kpm, fill, alpha)
        dh = ops.conv1d(dq, cw_q.transposed(), None, dil=-1, pad=0, T_iter=T, T_out=T)
        denc = ops.conv1d(dkv, cw_kv.transposed(), None, dil=-1, pad=0, T_iter=Tk, T_out=Tk) if ctx.needs_input_grad[1] else None
        # rows [0, H) of the packed weight from the queries, rows [H, 3H) from the keys / values: together the whole parameter, once
        t_wi, r_wi = _tape_tgt(p_win, dev)
        with leaf_work(dev, r_wi is None, dq, dkv, h, enc):
            conv_wgrad(dq, h, None, t_wi, B, H, H, 1, 1, 0, T, T, dw_ptr=t_wi.data_ptr() + 4 * cw_q.base)
            conv_wgrad(dkv, enc, None, t_wi, B, H, 2 * H, 1, 1, 0, Tk, Tk, dw_ptr=t_wi.data_ptr() + 4 * cw_kv.base)
        dx, r_g, r_b = _preln_tail(x, gamma, p_gamma, p_beta, dh, g2, eps)
        return (dx, denc, r_g, r_b, r_wi, r_wo) + (None,) * 9


def _fused_nodes():
    return os.environ.get("SET_AMD_FUSED_NODES", "1") != "0"


def preln_self_attn(x, ln, attn, key_padding=None, mask=None, eps=1e-5):
    """x + self-attention(LayerNorm(x)) (* mask); attn: the module holding the packed projections (campnet.MultiheadAttention)."""
    if _fused_nodes() and ops.attention_fused_on(attn.embed_dim // attn.num_heads):
        return _PreLnSelfAttnFn.apply(x, ln[0], ln[1], attn._w_qkv.raw(), attn._w_out.raw(), attn._w_qkv, attn._w_out, attn.num_heads,
                                      key_padding, float("-inf"), attn.scaling, mask, eps)
    x_ln, x_res = fanout(x, 2)
    h = layernorm_ch(x_ln, ln[0], ln[1], eps=eps)
    qkv = conv1d(h, attn._w_qkv)
    o, _ = self_attention(qkv, attn.num_heads, key_padding, float("-inf"), attn.scaling)
    return conv1d(o, attn._w_out, res=x_res, mask=mask)


def preln_cross_attn(x, ln, attn, enc, enc_padding, want_p=True, eps=1e-5):
    """(x + encoder-decoder attention(LayerNorm(x), enc), probabilities [B, heads, T, T_txt] or an empty tensor)."""
    if _fused_nodes() and ops.attention_fused_on(attn.embed_dim // attn.num_heads):
        return _PreLnCrossAttnFn.apply(x, enc, ln[0], ln[1], attn._w_q.raw(), attn._w_out.raw(), attn._w_q, attn._w_kv, attn._w_out,
                                       attn.num_heads, enc_padding, -1e8, attn.scaling, want_p, eps)
    x_ln, x_res = fanout(x, 2)
    h = layernorm_ch(x_ln, ln[0], ln[1], eps=eps)
    q = conv1d(h, attn._w_q)
    kv = conv1d(enc, attn._w_kv)
    o, p = cross_attention(q, kv, attn.num_heads, enc_padding, -1e8, attn.scaling, want_p=want_p)
    return conv1d(o, attn._w_out, res=x_res), p


class _PosAddFn(torch.autograd.Function):
    """x + alpha * table[pos]   (transformer.py:795-796; alpha is the learnable pos_embed_alpha)."""

    @staticmethod
    def forward(ctx, x, alpha, pos, table):
        out = ops.embedding_bct(pos, table, scale=alpha.detach().reshape(1).contiguous(), out=x.clone(), accumulate=True)
        ctx.save_for_backward(pos, table)
        return out

    @staticmethod
    def backward(ctx, d):
        pos, table = ctx.saved_tensors
        dalpha = None
        if ctx.needs_input_grad[1]:
            pe = ops.embedding_bct(pos, table)
            prod = ops.blend_mask(torch.zeros_like(pe), pe, d.contiguous(), 1)  # pe * d elementwise
            dalpha = _sum(prod, arena=False).reshape(1)
        return d, dalpha, None, None


def pos_add(x, alpha, pos, table):
    return _PosAddFn.apply(x, alpha, pos, table)


class _MaskFillChanFn(torch.autograd.Function):
    """x*(1-m) + e[c]*m on [B,C,T]; x carries no gradient here (it is the input mel), e = mask_emb does."""

    @staticmethod
    def forward(ctx, x, e, m):
        ctx.save_for_backward(m)
        ctx.C = x.shape[1]
        return ops.mask_fill_chan(x, e.reshape(-1).contiguous(), m)

    @staticmethod
    def backward(ctx, d):
        (m,) = ctx.saved_tensors
        de = _gzeros(ctx.C, d.device)
        ops.masked_channel_sum(d.contiguous(), m, de)
        return None, de.reshape(1, 1, -1), None


def mask_fill_chan(x, e, m):
    return _MaskFillChanFn.apply(x, e, m)


class _AddMaskedFn(torch.autograd.Function):
    """a + b * m[b][t]  on [B,C,T]   (campnet.py:62,68: `mels*(1-mask) + coarse*mask`, `mel_coarse + fine*mask`)."""

    @staticmethod
    def forward(ctx, a, b, m):
        ctx.save_for_backward(m)
        return ops.sum_div(a.contiguous(), ops.add_chan_mask(b, None, m))

    @staticmethod
    def backward(ctx, d):
        (m,) = ctx.saved_tensors
        d = d.contiguous()
        return (d if ctx.needs_input_grad[0] else None), ops.add_chan_mask(d, None, m), None


def add_masked(a, b, m):
    return _AddMaskedFn.apply(a, b, m)


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
def frame_weights(target_btm):
    """weights_nonzero_speech(target)[..., 0]  -> [B*T]"""
    B, T, M = target_btm.shape
    w = torch.empty(B * T, dtype=torch.float32, device=target_btm.device)
    check(L().set_frame_weight(_p(target_btm.contiguous()), _p(w), B * T, M, _stream()), "set_frame_weight")
    return w


def _sum(x, w=None, inner=1, arena=True):
    # arena: the accumulator is a slice of the step's zero arena inside an optimisation step (no fill launch) -- for values consumed within the
    # step; a result handed to autograd as a gradient (which may keep it as .grad) owns its storage instead
    out = _gzeros(1, x.device) if arena else torch.zeros(1, dtype=torch.float32, device=x.device)
    check(L().set_weighted_sum_det(_p(x), _p(w), _p(out), x.numel(), inner, _p(_det_scratch(x.device, 1024)), _stream()),
          "set_weighted_sum_det")
    return out


class _MaskedL1Fn(torch.autograd.Function):
    """(|pred - target| * w).sum() / w_full.sum()   (speech_base.py:223-229; w broadcast over the mel axis)."""

    @staticmethod
    def forward(ctx, pred, target, w):
        pred, target = pred.contiguous(), target.contiguous()
        M = pred.shape[-1]
        absd = torch.empty_like(pred)
        check(L().set_l1_elem(_p(pred), _p(target), _p(absd), None, pred.numel(), _stream()), "set_l1_elem")
        num = _sum(absd, w, M)
        den = _sum(w) * float(M)
        ctx.save_for_backward(pred, target, w, den)
        return (num / den).reshape(())

    @staticmethod
    def backward(ctx, gout):
        pred, target, w, den = ctx.saved_tensors
        M = pred.shape[-1]
        sgn = torch.empty_like(pred)
        check(L().set_l1_elem(_p(pred), _p(target), None, _p(sgn), pred.numel(), _stream()), "set_l1_elem")
        scale = (gout.reshape(1) / den).contiguous()
        d = torch.empty_like(pred)
        check(L().set_scale_bcast(_p(sgn), _p(w), _p(d), pred.numel(), M, _p(scale), 1.0, _stream()), "set_scale_bcast")
        return d, None, None


def masked_l1(pred, target, w):
    return _MaskedL1Fn.apply(pred, target, w)


class _SSIMLossFn(torch.autograd.Function):
    """((1 - ssim_map(pred+bias, target+bias)) * w).sum() / w_full.sum()   (speech_base.py:247-257, ssim.py:24-44)."""

    @staticmethod
    def forward(ctx, pred, target, w, bias):
        B, T, M = pred.shape
        dev = pred.device
        img1, img2 = pred.contiguous(), target.contiguous()
        maps = [torch.empty(B, T, M, dtype=torch.float32, device=dev) for _ in range(5)]
        check(L().set_ssim_filter(_p(img1), _p(img2), float(bias), *[_p(m) for m in maps], B, T, M, _stream()), "set_ssim_filter")
        one_minus = torch.empty_like(pred)
        d = [torch.empty_like(pred) for _ in range(3)]
        check(L().set_ssim_map(*[_p(m) for m in maps], _p(one_minus), _p(d[0]), _p(d[1]), _p(d[2]), pred.numel(), _stream()),
              "set_ssim_map")
        num = _sum(one_minus, w, M)
        den = _sum(w) * float(M)
        ctx.save_for_backward(img1, img2, w, den, *d)
        ctx.bias = bias
        return (num / den).reshape(())

    @staticmethod
    def backward(ctx, gout):
        img1, img2, w, den, d0, d1, d2 = ctx.saved_tensors
        B, T, M = img1.shape
        scale = (-gout.reshape(1) / den).contiguous()  # d loss / d ssim = -w / den
        g = []
        for dk in (d0, d1, d2):
            t = torch.empty_like(dk)
            check(L().set_scale_bcast(_p(dk), _p(w), _p(t), dk.numel(), M, _p(scale), 1.0, _stream()), "set_scale_bcast")
            g.append(t)
        dimg = torch.empty_like(img1)
        check(L().set_ssim_bwd(_p(img1), _p(img2), float(ctx.bias), _p(g[0]), _p(g[1]), _p(g[2]), _p(dimg), B, T, M, _stream()),
              "set_ssim_bwd")
        return dimg, None, None, None


def ssim_loss(pred, target, w, bias=6.0):
    return _SSIMLossFn.apply(pred, target, w, bias)


class _DurLossFn(torch.autograd.Function):
    """returns (pdur, wdur) already multiplied by their lambdas (speech_editing_base.py:58-90)."""

    @staticmethod
    def forward(ctx, dur_pred, mel2ph, txt, word_id, n_words, lam_p, lam_w):
        dur_pred = dur_pred.contiguous()
        B, T_txt = dur_pred.shape
        T = mel2ph.shape[1]
        sums = _gzeros(4, dur_pred.device)
        check(L().set_dur_loss_sums_det(_p(dur_pred), _p(mel2ph), _p(txt), _p(word_id), _p(sums), B, T, T_txt, n_words,
                                        _p(_det_scratch(dur_pred.device, 4 * B)), _stream()), "set_dur_loss_sums_det")
        ctx.save_for_backward(dur_pred, mel2ph, txt, word_id, sums)
        ctx.cfg = (n_words, lam_p, lam_w)
        pd = sums[0] / sums[1] * lam_p
        wd = sums[2] / sums[3] * lam_w
        return pd, wd

    @staticmethod
    def backward(ctx, gp, gw):
        dur_pred, mel2ph, txt, word_id, sums = ctx.saved_tensors
        n_words, lam_p, lam_w = ctx.cfg
        B, T_txt = dur_pred.shape
        # both outputs normally receive the same upstream gradient (1.0 from the loss sum); handle them separately
        d = torch.empty_like(dur_pred)
        out = None
        for lam_a, lam_b, gg in ((lam_p, 0.0, gp), (0.0, lam_w, gw)):
            check(L().set_dur_loss(_p(dur_pred), _p(mel2ph), _p(txt), _p(word_id), None, _p(sums), _p(d), B, mel2ph.shape[1],
                                   T_txt, n_words, float(lam_a), float(lam_b), 1.0, _stream()), "set_dur_loss")
            term = torch.empty_like(d)
            check(L().set_scale_bcast(_p(d), None, _p(term), d.numel(), 1, _p(gg.reshape(1).contiguous()), 1.0, _stream()),
                  "set_scale_bcast")
            out = term if out is None else ops.sum_div(out, term, None, 1.0)
        return out, None, None, None, None, None, None


def dur_losses(dur_pred, mel2ph, txt, word_id, n_words, lam_p, lam_w):
    return _DurLossFn.apply(dur_pred, mel2ph, txt, word_id, n_words, lam_p, lam_w)


class _PitchLossFn(torch.autograd.Function):
    """returns (uv_loss, f0_loss) times their lambdas; pp is channel-major [B,2,T] (speech_editing_base.py:92-108)."""

    @staticmethod
    def forward(ctx, pp, f0, uv, mel2ph, lam_uv, lam_f0):
        pp = pp.contiguous()
        B, _, T = pp.shape
        sums = _gzeros(4, pp.device)
        check(L().set_pitch_loss_sums_det(_p(pp), _p(f0), _p(uv), _p(mel2ph), _p(sums), B, T,
                                          _p(_det_scratch(pp.device, 4 * ((B * T + 255) // 256))), _stream()),
              "set_pitch_loss_sums_det")
        ctx.save_for_backward(pp, f0, uv, mel2ph, sums)
        ctx.cfg = (lam_uv, lam_f0)
        return sums[0] / sums[1] * lam_uv, sums[2] / sums[3] * lam_f0

    @staticmethod
    def backward(ctx, g_uv, g_f0):
        pp, f0, uv, mel2ph, sums = ctx.saved_tensors
        lam_uv, lam_f0 = ctx.cfg
        B, _, T = pp.shape
        d = torch.empty_like(pp)
        check(L().set_pitch_loss(_p(pp), _p(f0), _p(uv), _p(mel2ph), None, _p(sums), _p(d), B, T, float(lam_uv),
                                 float(lam_f0), 1.0, _stream()), "set_pitch_loss")
        # rows carry independent upstream gradients: row 0 (f0) * g_f0, row 1 (uv) * g_uv
        gsc = torch.stack([g_f0.reshape(()), g_uv.reshape(())]).reshape(1, 2, 1).expand(B, 2, 1).contiguous().reshape(-1)
        out = torch.empty_like(d)
        check(L().set_scale_bcast(_p(d), _p(gsc), _p(out), d.numel(), T, None, 1.0, _stream()), "set_scale_bcast")
        return out, None, None, None, None, None


def pitch_losses(pp_bct, f0, uv, mel2ph, lam_uv, lam_f0):
    return _PitchLossFn.apply(pp_bct, f0, uv, mel2ph, lam_uv, lam_f0)


# --------------------------------------------------------------------------------------------------
# optimizer
# --------------------------------------------------------------------------------------------------
def grad_sumsq(flat_grad):
    out = torch.zeros(1, dtype=torch.float32, device=flat_grad.device)
    check(L().set_sumsq_det(_p(flat_grad), _p(out), flat_grad.numel(), _p(_det_scratch(flat_grad.device, 2048)), _stream()),
          "set_sumsq_det")
    return out


def adamw_step(flat_p, flat_g, m, v, lr, beta1, beta2, eps, weight_decay, step, sumsq=None, max_norm=0.0,
               grad_scale=1.0):
    check(L().set_adamw(_p(flat_p), _p(flat_g), _p(m), _p(v), flat_p.numel(), float(lr), float(beta1), float(beta2),
                        float(eps), float(weight_decay), int(step), _p(sumsq), float(max_norm), float(grad_scale),
                        _stream()), "set_adamw")
