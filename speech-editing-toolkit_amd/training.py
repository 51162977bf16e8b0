"""Optimizer / data-parallel glue for the training rows of the path.

* parameters and gradients live in two flat fp32 buffers (each nn.Parameter / .grad is a view), so the optimizer is
  ONE fused kernel (`set_adamw`: clip_grad_norm_ + AdamW, torch semantics) and the data-parallel gradient exchange is
  a few large RCCL all-reduces over xGMI, launched from autograd hooks so they overlap with the rest of backward
  (`parallel.GradBucketer`), instead of one per tensor (the reference relies on torch DDP's 25 MB
  buckets, utils/commons/trainer.py:475-479; fs.decoder / fs.mel_out never receive gradients there either);
* learning rate = WarmupSchedule (utils/nn/schedulers.py:42-57): lr * min(step / warmup, 1), floored at 1e-7.
"""
import os

import torch

from . import autograd_ops as A
from . import ops, parallel


class FlatAdamW:
    def __init__(self, model, lr=2e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, clip_grad_norm=1.0,
                 warmup_updates=8000, bucket_mb=25):
        named = list(model.named_parameters())
        self.params = [p for _, p in named]          # model order (= the order of torch.optim.AdamW's state_dict)
        self.param_names = [n for n, _ in named]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        n_pad = (n + 255) // 256 * 256
        self.flat_p = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        # layout: the parameters a loss can reach first (model order), then the ones the model declares unreachable
        # (`unused_parameter_prefixes`): the gradient exchange covers the first part only
        unused_pre = tuple(getattr(model, "unused_parameter_prefixes", ()) or ())
        if os.environ.get("SET_AMD_EXCHANGE_UNUSED", "0") == "1":
            unused_pre = ()
        self.is_unused = [bool(unused_pre) and name.startswith(unused_pre) for name in self.param_names]
        self.layout = [i for i, u in enumerate(self.is_unused) if not u] + [i for i, u in enumerate(self.is_unused) if u]
        self.offs = [0] * len(self.params)
        off = 0
        for i in self.layout:
            p = self.params[i]
            k = p.numel()
            self.offs[i] = off
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            off += k
        self.n_exchanged = sum(self.params[i].numel() for i in self.layout if not self.is_unused[i])
        self.n = n
        # Direct gradient sinks: the weight / bias gradient kernels accumulate straight into the flat buffer instead of
        # returning a temporary that the autograd engine then ADDS into .grad (one ATen elementwise launch and one zeroed
        # temporary per parameter: ~250 launches, ~1 ms per step at B=32, T=800).  The first step runs through autograd
        # and counts how often each parameter receives a gradient; afterwards a parameter that receives exactly one
        # (every conv / linear here) is written directly; autograd still fires its post-accumulate hook when the backward
        # function returns (None gradient), which launches the bucket.  Parameters used several times per step keep the
        # autograd path (the hook then fires once, after the last use).
        self.direct = True
        self.in_step, self._learned, self._uses = False, False, {}
        for p in self.params:
            p._flat_owner = self
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.lr0, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.clip, self.warmup = clip_grad_norm, warmup_updates
        self.bucket = int(bucket_mb * (1 << 20) // 4)
        # gradient exchange overlapped with backward: buckets launch from autograd hooks (no-op when world == 1)
        # (buckets follow the order gradients appear in: cut from the end of the buffer, never across a top-level module)
        used = [i for i in self.layout if not self.is_unused[i]]
        self.bucketer = parallel.GradBucketer([self.params[i] for i in used], self.flat_g[:self.n_exchanged], self.bucket,
                                              force=os.environ.get("SET_AMD_FORCE_BUCKETER", "0") == "1",
                                              groups=[self.param_names[i].split(".", 1)[0] for i in used])
        self._unused_checked = self.n_exchanged == self.n
        self.num_updates = 0
        ops.bump_weights_epoch()

    def sink(self, param):
        """The .grad view a backward kernel may accumulate into directly, or None (gradient goes through autograd)."""
        if not (self.direct and self.in_step) or param.grad is None:
            return None
        if not self._learned:
            self._uses[id(param)] = self._uses.get(id(param), 0) + 1
            return None
        if self._uses.get(id(param), 0) != 1:
            return None
        return param.grad

    def lr_at(self, num_updates):
        warm = min(num_updates / self.warmup, 1.0) if self.warmup > 0 else 1.0
        return max(self.lr0 * warm, 1e-7)

    def zero_grad(self, accumulate=False):
        """Start an optimizer step.  accumulate=True: more than one backward will run before step() (the reference's
        `accumulate_grad_batches`, utils/commons/trainer.py:331-340); the gradient exchange then waits for step()."""
        self.flat_g.zero_()
        A.zero_arena_begin(self.flat_g.device, self.n, owner=self)  # this step's zero-initialised gradient temporaries
        self.bucketer.reset(defer=accumulate)
        self.in_step = True
        for p in self.params:  # autograd may have re-pointed .grad; restore the views
            if p.grad is None or p.grad.data_ptr() < self.flat_g.data_ptr() or \
                    p.grad.data_ptr() >= self.flat_g.data_ptr() + 4 * self.flat_g.numel():
                raise RuntimeError("a .grad left the flat gradient buffer")

    def abort_step(self):
        """A step that failed between zero_grad() and step(): close the zero arena and leave the "in step" state, so that a
        later backward outside a zero_grad()/step() pair goes through autograd instead of writing into the flat buffer."""
        A.zero_arena_end()
        A.leaf_join()
        self.in_step = False

    def step(self):
        """gradient all-reduce (SUM, few large buckets, launched from autograd hooks while backward was still running;
        whatever is left goes now) -> clip_grad_norm_(max_norm) + AdamW on the mean gradient, then the warm-up
        schedule (base_task.py:129-137)."""
        A.zero_arena_end()
        A.leaf_join()  # (normally done already, by the callback at the end of backward())
        self.in_step = False
        self._learned = True
        if not self._unused_checked:  # once: a parameter declared unreachable that did receive a gradient would silently diverge across ranks
            self._unused_checked = True
            if bool(torch.count_nonzero(self.flat_g[self.n_exchanged:self.n]).item()):
                bad = [self.param_names[i] for i in self.layout if self.is_unused[i] and bool(torch.count_nonzero(self.params[i].grad).item())]
                raise RuntimeError("parameters declared unused received gradients: %s" % bad[:5])
        world = self.bucketer.finish()
        sumsq = A.grad_sumsq(self.flat_g) if self.clip > 0 else None
        lr = self.lr_at(self.num_updates)
        self.num_updates += 1
        A.adamw_step(self.flat_p, self.flat_g, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                     self.num_updates, sumsq, self.clip, 1.0 / world)
        ops.bump_weights_epoch()
        if ops.compute_dtype() == "bf16":
            ops.repack_bf16_images()  # every bf16 weight image in one launch (they are all stale now)
        ops.repack_f32_images()  # and every fp32 image the step used
        return lr, sumsq

    # ---- checkpoint interchange with torch.optim.AdamW (the reference's optimizer, tasks/tts/speech_base.py:163-170):
    #      `optimizer_states[0]` of a reference checkpoint (utils/commons/trainer.py:459-471) loads here and vice versa
    def _offsets(self):
        """(offset in the flat buffers, numel, shape) per parameter in MODEL order (the flat layout puts unreachable parameters last)."""
        for i, p in enumerate(self.params):
            yield self.offs[i], p.numel(), p.shape

    def state_dict(self):
        state = {}
        if self.num_updates > 0:
            for i, (off, n, shape) in enumerate(self._offsets()):
                state[i] = {"step": torch.tensor(float(self.num_updates)),
                            "exp_avg": self.m[off:off + n].view(shape).clone(),
                            "exp_avg_sq": self.v[off:off + n].view(shape).clone()}
        group = {"lr": self.lr_at(self.num_updates), "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.wd, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "initial_lr": self.lr0,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW's format; parameters without saved state (never reached by a gradient in the
        saved run) start from zero moments.  The step counter is shared (torch keeps one per parameter)."""
        groups = sd["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(self.params):
            raise ValueError("optimizer parameters do not match: %d saved vs %d here" % (len(order), len(self.params)))
        offs = list(self._offsets())
        for pos, key in enumerate(order):  # validate everything BEFORE touching m / v: a mismatch must not leave them half-copied
            st = sd["state"].get(key)
            if st is not None and tuple(st["exp_avg"].shape) != tuple(offs[pos][2]):
                raise ValueError("optimizer state %d has shape %s, parameter has %s"
                                 % (key, tuple(st["exp_avg"].shape), tuple(offs[pos][2])))
        self.m.zero_()
        self.v.zero_()
        steps = 0
        for pos, key in enumerate(order):
            st = sd["state"].get(key)
            if st is None:
                continue
            off, n, shape = offs[pos]
            self.m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps = max(steps, int(float(st["step"])))
        self.num_updates = steps
        g0 = groups[0]
        self.betas, self.eps, self.wd = tuple(g0["betas"]), g0["eps"], g0["weight_decay"]
        self.lr0 = g0.get("initial_lr", self.lr0)
