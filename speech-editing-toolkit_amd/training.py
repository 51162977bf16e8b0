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

    def step(self, graph_hyper=None):
        """gradient all-reduce (SUM, few large buckets, launched from autograd hooks while backward was still running;
        whatever is left goes now) -> clip_grad_norm_(max_norm) + AdamW on the mean gradient, then the warm-up
        schedule (base_task.py:129-137).  graph_hyper (GraphedTrainStep, while a step is being captured): a device tensor
        [lr, bc1, bc2] the update reads instead of host values; the update counter is then advanced by the replaying caller."""
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
        if graph_hyper is not None:
            lr = None
            A.adamw_step_dev(self.flat_p, self.flat_g, self.m, self.v, graph_hyper, self.betas[0], self.betas[1], self.eps, self.wd,
                             sumsq, self.clip, 1.0 / world)
        else:
            lr = self.lr_at(self.num_updates)
            self.num_updates += 1
            A.adamw_step(self.flat_p, self.flat_g, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                         self.num_updates, sumsq, self.clip, 1.0 / world)
        ops.bump_weights_epoch()
        if ops.compute_dtype() == "bf16":
            ops.repack_bf16_images()  # every bf16 weight image in one launch (they are all stale now)
        return lr, sumsq

    def next_hyper(self):
        """Host side of one update for a replayed graph: advances the counter, returns (lr, bc1, bc2) of that update."""
        lr = self.lr_at(self.num_updates)
        self.num_updates += 1
        bc1, bc2 = A.adamw_hyper(self.betas[0], self.betas[1], self.num_updates)
        return lr, bc1, bc2

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


class GraphedTrainStep:
    """One optimisation step (zero_grad -> forward + losses -> backward -> clip + AdamW -> bf16 weight images) as ONE captured HIP
    graph per batch shape, replayed for every later update of that shape.

    Why: a training step of this path is ~500 (spec_denoiser) / ~1,250 (CampNet) launches of 5 - 50 us; enqueueing them from Python
    took 8.6 / 12.5 ms of a 12.1 / 16.1 ms step (round 3): the GPU waited for the host.  A replay costs one launch.
    What makes the replay the SAME step as the eager one (tests/test_gpu_training.py compares them bit for bit):
      * the batch is copied into the static input tensors the graph was captured on;
      * every Philox kernel adds a device word to its seed argument (set_rng_seed_delta): the replay of step k stores
        seed_k - seed_captured there, so noise / dropout are the eager step's; explicit diffusion steps `t` are a static tensor;
      * the optimizer's lr and bias corrections come from a device tensor (set_adamw_dev) filled before each replay; the update
        counter advances on the host as in the eager step;
      * nothing in a step synchronises with the host or depends on host state that changes between steps (the first
        `eager_steps` steps of a shape run eagerly: weight-image caches, the zero arena and the direct-gradient sinks reach
        their steady state there).
    Opt-in (SET_AMD_GRAPH_STEP=1), because it does NOT make these steps faster: measured (profiles/r04_graph.log) the eager steps were
    never waiting for the host -- Python enqueues a step in 8.6 / 12.4 ms WHILE the GPU executes the previous one for 12.1 / 16.1 ms of
    pure kernel time (rocprof: 670 kernels, 12.8 ms busy per step) -- and a replay adds ~0.5 ms (the batch copies, and a graph cannot
    overlap its own previous launch): spec_denoiser bf16 12.10 ms eager / 12.62 ms graphed, CampNet 16.12 / 16.89, host enqueue
    8.56 -> 0.27 ms and 12.36 -> 1.50 ms.  It is the right tool when the host is the bottleneck (small batches, a busy host).
    Never graphed: more than one rank (the bucketed all-reduce runs from autograd hooks), gradient accumulation, more than
    `max_graphs` distinct shapes (a graph holds its activations: ~2 GB at B=32, T=800)."""

    def __init__(self, task, optimizer, eager_steps=3, max_graphs=4):
        self.task, self.opt = task, optimizer
        self.eager_steps, self.max_graphs = max(2, int(eager_steps)), int(max_graphs)
        self.entries = {}
        self.hyper = self.seed_delta = self.stream = None
        self.replays = 0

    def __del__(self):
        for _ in range(getattr(self, "_n_captured", 0)):
            A.graph_released()

    def usable(self):
        dev = self.opt.flat_p.device
        return (dev.type == "cuda" and os.environ.get("SET_AMD_GRAPH_STEP", "0") == "1" and not self.opt.bucketer.enabled)

    @staticmethod
    def _key(sample, t):
        return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(sample.items()) if isinstance(v, torch.Tensor)) + (t is not None,)

    def eager_mode(self):
        """Call before any eager work that draws Philox numbers once a graph exists (an eager step of another shape, a validation
        pass): the seed-delta word of the last replay is process-wide and would shift those seeds."""
        if self.seed_delta is not None:
            self.seed_delta.zero_()

    def _eager(self, sample, seed, t):
        from .tasks import _optimisation_step
        self.eager_mode()
        kw = {"seed": seed}
        if t is not None:
            kw["t"] = t
        return _optimisation_step(self.task, sample, self.opt, **kw)

    def __call__(self, sample, seed, t=None):
        if not self.usable():
            return self._eager(sample, seed, t)
        key = self._key(sample, t)
        e = self.entries.setdefault(key, {"count": 0, "graph": None})
        if e["graph"] is None:
            n_graphs = sum(1 for v in self.entries.values() if v["graph"] is not None)
            if e["count"] < self.eager_steps or n_graphs >= self.max_graphs:
                e["count"] += 1
                return self._eager(sample, seed, t)
            self._capture(e, sample, seed, t)
        return self._replay(e, sample, seed, t)

    def _capture(self, e, sample, seed, t):
        from . import _lib
        from .tasks import _optimisation_step
        dev = self.opt.flat_p.device
        if self.hyper is None:
            self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
            self.seed_delta = torch.zeros(1, dtype=torch.int64, device=dev)
            _lib.check(_lib.lib().set_rng_seed_delta(self.seed_delta.data_ptr()), "set_rng_seed_delta")
        e["static"] = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sample.items()}
        e["t"] = t.clone() if t is not None else None
        e["seed"] = int(seed)
        self.seed_delta.zero_()
        kw = {"seed": e["seed"]}
        if t is not None:
            kw["t"] = e["t"]
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        del A.CAPTURED_ABORT_WORDS[:]
        A.graph_captured()  # from here on replaced arena / scratch buffers are parked, not freed (the graph holds their addresses)
        self._n_captured = getattr(self, "_n_captured", 0) + 1
        with torch.cuda.graph(g):
            total, parts, _ = _optimisation_step(self.task, e["static"], self.opt, _graph_hyper=self.hyper, **kw)
            names = sorted(parts)
            e["out"] = torch.stack([total.reshape(())] + [parts[n].reshape(()).to(total.dtype) for n in names])
        e["names"], e["graph"] = names, g
        e["abort_words"] = list(A.CAPTURED_ABORT_WORDS)  # polled every 64 replays (one host read each)

    def _replay(self, e, sample, seed, t):
        # The replay runs on a stream of its own, tied to the caller's stream by events on both sides.  (Launched into the legacy
        # default stream -- torch's current stream unless the caller changed it -- the graph ran concurrently with default-stream work
        # enqueued right after it: an eager step of another replica shared the zero arena and the seed-delta word with the still
        # running replay, gradients came out as garbage.  Event dependencies do not rely on the null stream's implicit ordering.)
        cur = torch.cuda.current_stream()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for k, v in sample.items():
                if isinstance(v, torch.Tensor):
                    e["static"][k].copy_(v, non_blocking=True)
                    v.record_stream(self.stream)
            if t is not None:
                e["t"].copy_(t, non_blocking=True)
                t.record_stream(self.stream)
            delta = (int(seed) - e["seed"]) % (1 << 64)
            self.seed_delta.fill_(delta - (1 << 64) if delta >= (1 << 63) else delta)
            lr, bc1, bc2 = self.opt.next_hyper()
            self.hyper[0:1].fill_(lr)
            self.hyper[1:2].fill_(bc1)
            self.hyper[2:3].fill_(bc2)
            e["graph"].replay()
            self.replays += 1
            # the replay updated the parameters: host-side packed-weight caches (packed() / packed_x2(), keyed on the weights epoch) built by
            # eager work between replays are stale now -- FlatAdamW.step's own bump ran only once, at capture time
            ops.bump_weights_epoch()
            out = e["out"].clone()  # the graph's output tensors are overwritten by the next replay
        cur.wait_stream(self.stream)
        out.record_stream(cur)
        if e.get("abort_words") and self.replays % 64 == 0:
            for ws in e["abort_words"]:
                if int(ws[1]) != 0:
                    raise A.SetAmdError("set_diffnet_stack: a tile dependency wait of a replayed training forward timed out")
        return out[0], {n: out[1 + i] for i, n in enumerate(e["names"])}, lr
