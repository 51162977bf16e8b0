"""DiffNet denoiser (`DIFF_DECODERS['wavenet']`) on the HIP kernels.

Same constructor contract, `forward(spec, diffusion_step, cond)` signature and state_dict keys as the
reference's modules/speech_editing/spec_denoiser/diffnet.py:84-132, so reference checkpoints load unchanged.
nn.Conv1d / nn.Linear modules are used as parameter containers only; no torch arithmetic runs in forward.
"""
import math
import os

import torch
from torch import nn

from . import autograd_ops, ops
from .hparams import hparams as _global_hparams


def _kaiming_conv1d(cin, cout, k, **kw):
    layer = nn.Conv1d(cin, cout, k, **kw)  # diffnet.py:49-52
    nn.init.kaiming_normal_(layer.weight)
    return layer


class ResidualBlock(nn.Module):
    """Parameter container for one residual layer (diffnet.py:60-66)."""

    def __init__(self, encoder_hidden, residual_channels, dilation):
        super().__init__()
        C = residual_channels
        self.dilation = dilation
        self.dilated_conv = _kaiming_conv1d(C, 2 * C, 3, padding=dilation, dilation=dilation)
        self.diffusion_projection = nn.Linear(C, C)
        self.conditioner_projection = _kaiming_conv1d(encoder_hidden, 2 * C, 1)
        self.output_projection = _kaiming_conv1d(C, 2 * C, 1)
        H = encoder_hidden
        self._w_dil = ops.ConvWeight((self, "dilated_conv.weight"), 2 * C, C, 3)
        self._w_dproj = ops.ConvWeight((self, "diffusion_projection.weight"), C, C, 1)
        self._w_cond = ops.ConvWeight((self, "conditioner_projection.weight"), 2 * C, H, 1)
        self._w_out = ops.ConvWeight((self, "output_projection.weight"), 2 * C, C, 1)
        self._fused = None
        self._fused_key = None

    def fused_weights(self):
        wd, wo = self.dilated_conv.weight, self.output_projection.weight
        key = (wd.data_ptr(), wd._version, wo.data_ptr(), wo._version, ops.weights_epoch())
        if self._fused is None or key != self._fused_key:
            self._fused = ops.pack_diffnet_layer(wd, wo)
            self._fused_key = key
        return self._fused


class DiffNet(nn.Module):
    FUSED_CHANNELS = 256  # residual_channels the fused layer kernel is specialised for
    FUSED_MAX_DIL = 8

    def __init__(self, in_dims=80, hp=None):
        super().__init__()
        hp = hp if hp is not None else _global_hparams
        self.in_dims = in_dims
        self.encoder_hidden = hp["hidden_size"]
        self.n_layers = hp["residual_layers"]
        self.C = C = hp["residual_channels"]
        self.dilation_cycle_length = hp["dilation_cycle_length"]
        self.input_projection = _kaiming_conv1d(in_dims, C, 1)
        self.mlp = nn.Sequential(nn.Linear(C, C * 4), nn.Identity(), nn.Linear(C * 4, C))  # [1] is Mish (no params)
        self.residual_layers = nn.ModuleList([
            ResidualBlock(self.encoder_hidden, C, 2 ** (i % self.dilation_cycle_length))
            for i in range(self.n_layers)])
        self.skip_projection = _kaiming_conv1d(C, C, 1)
        self.output_projection = _kaiming_conv1d(C, in_dims, 1)
        nn.init.zeros_(self.output_projection.weight)
        self._w_in = ops.ConvWeight((self, "input_projection.weight"), C, in_dims, 1)
        self._w_mlp0 = ops.ConvWeight((self, "mlp.0.weight"), 4 * C, C, 1)
        self._w_mlp2 = ops.ConvWeight((self, "mlp.2.weight"), C, 4 * C, 1)
        self._w_skip = ops.ConvWeight((self, "skip_projection.weight"), C, C, 1)
        self._w_outp = ops.ConvWeight((self, "output_projection.weight"), in_dims, C, 1)
        self.impl = "auto"  # auto | fused | unfused  (unfused = generic kernels; device-side cross-check)
        self._packs, self._packs_key = None, None

    # ---- helpers -------------------------------------------------------------------------------------
    def can_fuse(self):
        return self.C == self.FUSED_CHANNELS and 2 ** (self.dilation_cycle_length - 1) <= self.FUSED_MAX_DIL

    def use_fused(self):
        if self.impl == "unfused":
            return False
        if self.impl == "fused" and not self.can_fuse():
            raise RuntimeError("fused DiffNet layer kernel needs residual_channels == 256 and dilation <= 8")
        return self.can_fuse()

    def fused_packs(self, inference=True):
        """Contiguous [L][...] packed weights + biases for the fused layer / persistent stack kernels (re-packed when any
        layer parameter changes): (w1p, w2p, b_dil, b_out, w1w, w2w [Winograd], w1s, w2s [row-split], wx3 [split-operand]).
        inference=False (the training forward, whose weights change every step): only what the Winograd training kernel
        reads -- the last three are None (no extra 40 MB permutation and no per-layer scale read-backs per step)."""
        layers = list(self.residual_layers)
        key = tuple((p.data_ptr(), p._version) for l in layers for p in
                    (l.dilated_conv.weight, l.output_projection.weight, l.dilated_conv.bias, l.output_projection.bias))
        key = key + (ops.weights_epoch(),)
        dev = layers[0].dilated_conv.weight.device
        L = len(layers)
        if self._packs is None or key != self._packs_key:
            w1 = torch.empty(L, 512 * 768, dtype=torch.float32, device=dev)
            w2 = torch.empty(L, 512 * 256, dtype=torch.float32, device=dev)
            wino = self.dilation_cycle_length <= 4  # Winograd F(2,3) images for the persistent stack kernel (d <= 8)
            w1w = torch.empty(L, 512 * 256 * 4, dtype=torch.float32, device=dev) if wino else None
            w2w = torch.empty(L, 512 * 256, dtype=torch.float32, device=dev) if wino else None
            from .autograd_ops import _uniform_stride
            wds, wos = [l.dilated_conv.weight.detach() for l in layers], [l.output_projection.weight.detach() for l in layers]
            sd, so = _uniform_stride(wds), _uniform_stride(wos)
            if sd is not None and so is not None and wds[0].is_cuda:  # (the flat optimizer's layout) every layer in one launch per image family
                from . import _lib
                _lib.check(_lib.lib().set_pack_diffnet_layers(ops._p(wds[0]), ops._p(wos[0]), sd, so, ops._p(w1), ops._p(w2), ops._p(w1w),
                                                              ops._p(w2w), L, ops._stream()), "set_pack_diffnet_layers")
            else:
                for i, l in enumerate(layers):
                    ops.pack_diffnet_layer(l.dilated_conv.weight, l.output_projection.weight, w1[i], w2[i])
                    if wino:
                        ops.pack_diffnet_layer_wino(wds[i], wos[i], w1w[i], w2w[i])
            bd = torch.stack([l.dilated_conv.bias for l in layers]).contiguous()
            bo = torch.stack([l.output_projection.bias for l in layers]).contiguous()
            self._packs, self._packs_key = (w1, w2, bd, bo, w1w, w2w), key
            self._packs_extra_key = None
        if not inference:
            return self._packs + (None, None, None)
        ekey = key + (ops.split_operand_mode(),)
        if getattr(self, "_packs_extra_key", None) != ekey:
            w1s, w2s = ops.split_images(self._packs[0], self._packs[1])  # small-batch (row-split) stack kernel
            wx3 = None
            if self.dilation_cycle_length <= 4:  # split-operand images of the throughput kernel (csrc/diffnet_x3.hip)
                wx3 = ops.SplitOperandImages(L, ops.split_operand_mode(), dev)
                for i, l in enumerate(layers):
                    wx3.pack(i, l.dilated_conv.weight.detach(), l.output_projection.weight.detach())
            self._packs_extra, self._packs_extra_key = (w1s, w2s, wx3), ekey
        return self._packs + self._packs_extra

    def bf16_layer_images(self):
        """Per-layer packed bf16 weight images of the fused training kernels (forward GEMM 1 / 2 and the three transposed
        images of the backward), re-rounded from the fp32 master weights whenever they change."""
        layers = list(self.residual_layers)
        # the 60 weight tensors as a flat list, looked up once (module attribute chains cost ~1 us each: this key is computed every step)
        ps = getattr(self, "_img16_params", None)
        if ps is None or len(ps) != 3 * len(layers) or ps[0] is not layers[0].dilated_conv.weight or ps[-1] is not layers[-1].output_projection.weight:
            ps = self._img16_params = [p for l in layers for p in (l.dilated_conv.weight, l.conditioner_projection.weight, l.output_projection.weight)]
            self._img16_strides = None
        key = tuple((p.data_ptr(), p._version) for p in ps)
        key = key + (ops.weights_epoch(),)
        if getattr(self, "_img16", None) is None or self._img16_key != key:
            from . import _lib
            n = _lib.lib().set_diffnet_layer_bf16_image_size()
            dev = layers[0].dilated_conv.weight.device
            img = torch.empty(len(layers), n, dtype=torch.bfloat16, device=dev)
            from .autograd_ops import _uniform_stride
            # the layers' strides are a property of where the tensors live: recomputed when any address changes
            addr = tuple(k[0] for k in key[:-1])
            if self._img16_strides is None or self._img16_strides[0] != addr:
                self._img16_strides = (addr, [_uniform_stride([p.detach() for p in ps[j::3]]) for j in range(3)])
            strides = self._img16_strides[1]
            if all(st is not None for st in strides):  # (the flat optimizer's layout: every layer's tensors at one stride) one launch
                l0 = layers[0]
                _lib.check(_lib.lib().set_pack_diffnet_layers_bf16(
                    ops._p(l0.dilated_conv.weight.detach()), ops._p(l0.conditioner_projection.weight.detach()),
                    ops._p(l0.output_projection.weight.detach()), strides[0], strides[1], strides[2], ops._p(img), len(layers),
                    ops._stream()), "set_pack_diffnet_layers_bf16")
            else:
                for i, l in enumerate(layers):
                    _lib.check(_lib.lib().set_pack_diffnet_layer_bf16(
                        ops._p(l.dilated_conv.weight.detach()), ops._p(l.conditioner_projection.weight.detach()),
                        ops._p(l.output_projection.weight.detach()), ops._p(img[i]), ops._stream()), "set_pack_diffnet_layer_bf16")
            self._img16, self._img16_key = img, key
        return self._img16

    def step_table(self, t_values):
        """d[l][c][n] = diffusion_projection_l(mlp(sinusoid(t_n)))[c]  ->  tensor [L*C, n]
        (diffnet.py:121-122 and :69).  t_values: float tensor [n]; depends on t only, so the reverse
        loop computes it once for all steps."""
        C, L = self.C, self.n_layers
        n = t_values.numel()
        emb = ops.sinusoid_embed(t_values, C).view(1, C, n)
        h = ops.conv1d(emb, self._w_mlp0, self.mlp[0].bias, act="mish")
        h = ops.conv1d(h, self._w_mlp2, self.mlp[2].bias)
        out = torch.empty(1, L * C, n, dtype=torch.float32, device=h.device)
        for l, layer in enumerate(self.residual_layers):
            ops.conv1d(h, layer._w_dproj, layer.diffusion_projection.bias, out=out[:, l * C:(l + 1) * C, :])
        return out.view(L * C, n)

    def step_table_all(self, steps, dev):
        """step_table(arange(steps)) for the reverse loop, kept across loops for as long as the tensors it is computed from are unchanged: it
        depends on the step MLP and the layers' diffusion projections only (22 launches of work on 100 columns -- 0.8 ms of a 150 ms loop at
        B = 32, T = 800).  Keyed by storage, version counter and the optimizer's weights epoch of every one of those tensors."""
        ps = [self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias]
        for layer in self.residual_layers:
            ps += [layer.diffusion_projection.weight, layer.diffusion_projection.bias]
        key = (int(steps), str(dev), ops.weights_epoch(), torch.is_grad_enabled()) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_dtab_key", None) != key or torch.is_grad_enabled():
            self._dtab = self.step_table(torch.arange(int(steps), device=dev).to(torch.float32))
            self._dtab_key = key
        return self._dtab

    def cond_projections(self, cond):
        """conditioner_projection_l(cond) for every layer -> [B, L*2C, T].  It does not depend on the
        diffusion step, so the reverse loop hoists it (the reference recomputes it every step, diffnet.py:70)."""
        B, H, T = cond.shape
        C, L = self.C, self.n_layers
        out = torch.empty(B, L * 2 * C, T, dtype=torch.float32, device=cond.device)
        if not torch.is_grad_enabled():
            # one launch on the layers' weights stacked along the output channels (bit-identical to the per-layer launches: same kernel, same
            # k order per output; 1.17 against 1.4 - 1.55 ms at B = 32, T = 800, profiles/r06_condproj_probe.log); the stacked copy follows the
            # layers' tensors by storage, version counter and the optimizer's weights epoch
            ps = [t for layer in self.residual_layers for t in (layer.conditioner_projection.weight, layer.conditioner_projection.bias)]
            key = (str(cond.device), ops.weights_epoch()) + tuple((p.data_ptr(), p._version) for p in ps)
            if getattr(self, "_wc_all_key", None) != key:
                self._wc_all = torch.cat([p.detach() for p in ps[0::2]], 0).contiguous()
                self._bc_all = torch.cat([p.detach() for p in ps[1::2]], 0).contiguous()
                self._wc_all_cw = ops.ConvWeight((self, "_wc_all"), L * 2 * C, H, 1)
                self._wc_all_key = key
            ops.conv1d(cond, self._wc_all_cw, self._bc_all, out=out)
            return out
        for l, layer in enumerate(self.residual_layers):
            ops.conv1d(cond, layer._w_cond, layer.conditioner_projection.bias,
                       out=out[:, l * 2 * C:(l + 1) * 2 * C, :])
        return out

    def denoise(self, x, condproj, dtab, col, batch_cols):
        """One DiffNet pass.  x [B,M,T]; condproj [B,L*2C,T]; dtab [L*C,n];
        column of batch b is `col + b` if batch_cols else `col`.  Returns x0 [B,M,T]."""
        B, M, T = x.shape
        C, L = self.C, self.n_layers
        n = dtab.shape[1]
        fused = self.use_fused()
        h = ops.conv1d(x, self._w_in, self.input_projection.bias, act="relu")
        skip = torch.empty(B, C, T, dtype=torch.float32, device=x.device)
        nxt = torch.empty_like(h)
        for l, layer in enumerate(self.residual_layers):
            cp = condproj[:, l * 2 * C:(l + 1) * 2 * C, :]
            if fused:
                w1p, w2p = layer.fused_weights()
                dptr = dtab.data_ptr() + 4 * (l * C * n + col)
                ops.diffnet_layer(h, cp.data_ptr(), condproj.stride(0), dptr, 1 if batch_cols else 0, n,
                                  w1p, layer.dilated_conv.bias, w2p, layer.output_projection.bias,
                                  nxt, skip, layer.dilation, l == 0)
                h, nxt = nxt, h
            else:
                d = dtab.view(L, C, n)[l]  # [C, n]
                d = d[:, col:col + B].t().contiguous() if batch_cols else d[:, col].reshape(1, C).expand(B, C).contiguous()
                y = ops.conv1d(h, layer._w_dil, layer.dilated_conv.bias, dil=layer.dilation,
                               pad=layer.dilation, in_chan_add=d, res=cp)
                z = ops.gate(y)
                o = ops.conv1d(z, layer._w_out, layer.output_projection.bias)
                h = ops.res_skip(h, o, skip, l == 0)
        hs = ops.conv1d(skip, self._w_skip, self.skip_projection.bias, pro="div", pro_param=math.sqrt(L),
                        act="relu")
        return ops.conv1d(hs, self._w_outp, self.output_projection.bias)

    def forward_train(self, spec, diffusion_step, cond):
        """DiffNet.forward with an autograd tape (training): generic differentiable kernels per op
        (conv1d dgrad/wgrad on fp32 MFMA, gate / res-skip backward); same math as `forward`."""
        A = autograd_ops
        C, L = self.C, self.n_layers
        x = spec[:, 0].contiguous()
        n = x.shape[0]
        emb = ops.sinusoid_embed(diffusion_step.to(torch.float32).contiguous(), C).view(1, C, n)
        h = A.conv1d(emb, self._w_mlp0, self.mlp[0].bias, act="mish")
        h = A.conv1d(h, self._w_mlp2, self.mlp[2].bias)
        hx = A.conv1d(x, self._w_in, self.input_projection.bias, act="relu")
        cond = cond.contiguous()
        skip = None
        # (the persistent Winograd stack kernel is an fp32-operand kernel: with bf16 operands the layers run op by op)
        use_stack = (self.can_fuse() and self.impl != "unfused" and os.environ.get("SET_AMD_TRAIN_STACK", "1") != "0"
                     and ops.compute_dtype() == "f32"
                     and ops.stack_variant(x.shape[0], x.shape[2], self.dilation_cycle_length, have_split=False, x3_mode=0) == 2)
        use_bf16_layers = (self.can_fuse() and self.impl != "unfused" and ops.compute_dtype() == "bf16"
                           and self.encoder_hidden == 192 and os.environ.get("SET_AMD_TRAIN_STACK", "1") != "0"
                           and x.shape[2] >= 32)
        dmat = None
        if use_bf16_layers or use_stack:
            # [n, L*C] step offsets of every layer: one launch when the layers' parameters sit at one stride (flat optimizer), else
            # one 1x1 conv per layer over the fanned-out embedding
            dmat = A.step_projections(self, h) if os.environ.get("SET_AMD_STEP_PROJ", "1") != "0" else None
            if dmat is None:
                hs_ = A.fanout(h, L)
                dmat = torch.cat([A.conv1d(hs_[li], layer._w_dproj, layer.diffusion_projection.bias)[0].t()
                                  for li, layer in enumerate(self.residual_layers)], dim=1)
        if use_bf16_layers:
            # bf16 operands: one fused forward and one fused backward launch per layer (csrc/diffnet_bf16.hip)
            skip = A.diffnet_stack_train_bf16(self, hx, cond, dmat)
        elif use_stack:
            # fused forward: one persistent Winograd launch for all L layers (+ saved x/y/z), hand-ordered backward
            skip = A.diffnet_stack_train(self, hx, cond, dmat)
        else:
            hs_ = A.fanout(h, L)         # every layer's diffusion_projection reads the step embedding
            conds = A.fanout(cond, L)    # ... and its conditioner_projection reads cond
            for li, layer in enumerate(self.residual_layers):
                h, cond = hs_[li], conds[li]
                hx, hx_res = A.fanout(hx, 2)  # dilated conv input + residual path
                d = A.conv1d(h, layer._w_dproj, layer.diffusion_projection.bias)  # [1, C, n]
                d_bc = d[0].t().contiguous()  # [n, C]: per-utterance channel offsets (layout change only)
                cp = A.conv1d(cond, layer._w_cond, layer.conditioner_projection.bias)
                y = A.conv1d(hx, layer._w_dil, layer.dilated_conv.bias, dil=layer.dilation, pad=layer.dilation,
                             in_chan_add=d_bc, res=cp)
                z = A.gate(y)
                o = A.conv1d(z, layer._w_out, layer.output_projection.bias)
                hx, skip = A.res_skip_fn(hx_res, o, skip)
        hs = A.conv1d(skip, self._w_skip, self.skip_projection.bias, pro="div", pro_param=math.sqrt(L), act="relu")
        return A.conv1d(hs, self._w_outp, self.output_projection.bias)[:, None, :, :]

    # ---- reference signature ---------------------------------------------------------------------------
    def forward(self, spec, diffusion_step, cond):
        """spec [B,1,M,T] fp32, diffusion_step int64 [B], cond [B,H,T] -> [B,1,M,T]  (diffnet.py:110-132)."""
        if torch.is_grad_enabled():
            return self.forward_train(spec, diffusion_step, cond)
        x = spec[:, 0].contiguous()
        dtab = self.step_table(diffusion_step.to(torch.float32).contiguous())
        condproj = self.cond_projections(cond.contiguous())
        x0 = self.denoise(x, condproj, dtab, 0, True)
        return x0[:, None, :, :]
