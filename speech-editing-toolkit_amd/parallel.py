"""Data-parallel helpers: one process per GPU, utterances sharded across ranks, no data-path collective
for inference.  Mirrors the reference's batch striding `x[rank::num_replicas]`
(tasks/tts/speech_base.py:128-131) and its env-var rendezvous on 127.0.0.1 (utils/commons/trainer.py:481-485);
backend 'nccl' is RCCL on ROCm, 'gloo' is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_batch(sample, rank, world):
    """Rank r takes utterances r, r+world, ... of every batched tensor (first dim = batch)."""
    if world == 1:
        return sample
    out = {}
    for k, v in sample.items():
        out[k] = v[rank::world].contiguous() if isinstance(v, torch.Tensor) and v.dim() > 0 else v
    return out


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (timing)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def _forced():
    """SET_AMD_FORCE_BUCKETER=1 keeps the collective machinery on at world size 1, so that a 1-GPU box still drives every
    broadcast / all-reduce through RCCL (tests/test_gpu_dist.py)."""
    return os.environ.get("SET_AMD_FORCE_BUCKETER", "0") == "1"


def _collectives_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def broadcast_module_(module, src=0, flat_params=None):
    """DDP construction semantics (utils/commons/trainer.py:475-479: torch's DistributedDataParallel broadcasts the
    rank-0 parameters and buffers when it wraps the task, so replicas agree even when every rank initialised with its
    own random state).  With a flat parameter buffer (training.FlatAdamW) that is ONE collective for all parameters;
    buffers (the diffusion schedule tables) follow one by one.  Returns the number of bytes sent from `src`."""
    if not _collectives_on():
        return 0
    sent = 0
    with torch.no_grad():
        if flat_params is not None:
            dist.broadcast(flat_params, src=src)
            sent += flat_params.numel() * flat_params.element_size()
        else:
            for p in module.parameters():
                dist.broadcast(p.data, src=src)
                sent += p.numel() * p.element_size()
        for b in module.buffers():
            if b.numel() > 0:
                dist.broadcast(b, src=src)
                sent += b.numel() * b.element_size()
    return sent


def configure_ddp(module, optimizer=None, src=0):
    """What `Trainer.run_single_process` does for a replica (utils/commons/trainer.py:166-170, 402): wait until every
    rank has built / restored its model, make the replicas identical to rank 0's, wait again.  `optimizer` (FlatAdamW)
    supplies the flat parameter buffer and also gets its Adam moments and step counter broadcast so a resumed run agrees
    on every rank."""
    if not _collectives_on():
        return 0
    dist.barrier()
    sent = broadcast_module_(module, src, flat_params=getattr(optimizer, "flat_p", None))
    if optimizer is not None and hasattr(optimizer, "m"):
        dist.broadcast(optimizer.m, src=src)
        dist.broadcast(optimizer.v, src=src)
        n = torch.tensor([optimizer.num_updates], dtype=torch.int64, device=optimizer.m.device)
        dist.broadcast(n, src=src)
        optimizer.num_updates = int(n.item())
        from . import ops
        ops.bump_weights_epoch()  # packed weight images were built from the pre-broadcast values
    dist.barrier()
    return sent


def bucketed_all_reduce_sum_(flat, bucket_elems):
    """In-place SUM all-reduce of a flat buffer in a few large buckets (RCCL ring/direct over xGMI is per-link bound,
    so few large messages beat many small ones).  Returns the world size (the caller folds 1/world into its next
    kernel).  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1
    n = flat.numel()
    works = []
    for s in range(0, n, bucket_elems):
        works.append(dist.all_reduce(flat[s:s + bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    return dist.get_world_size()


class GradBucketer:
    """Overlap the data-parallel gradient exchange with backward ("bucketed and overlapped").

    The gradients live in ONE flat buffer in parameter order (training.FlatAdamW).  Backward produces them roughly in
    reverse order WITHIN a top-level module, and module by module in reverse data-flow order: for the spec_denoiser the
    parameter order is denoise_fn.* (57.7 MB), fs.*, mel_encoder.* while backward reaches denoise_fn FIRST (back to
    front) and the conditioner LAST.  So the buffer is cut into contiguous buckets from the END, a bucket is closed when
    it holds `bucket_elems` gradients -- and also at a top-level module boundary (`groups`), so that the tail of
    denoise_fn (the first gradients to exist) never waits in one bucket with the conditioner's (the last).  Every parameter
    gets a post-accumulate-grad hook: when the last gradient of a bucket has been accumulated, that bucket's SUM
    all-reduce is launched asynchronously (RCCL runs it on its own stream behind an event on the compute stream, i.e.
    under the rest of backward).  `finish()` launches whatever never fired (parameters without gradients: fs.decoder /
    fs.mel_out stay zero) and waits.  A few ~25 MB buckets: xGMI rings are per-link bound, small messages waste them,
    one 64 MB bucket that ends in the conditioner leaves most of the exchange exposed after backward."""

    def __init__(self, params, flat_g, bucket_elems, force=False, groups=None):
        self.flat_g = flat_g
        # force: keep the machinery on at world size 1 (a single-rank RCCL group on a 1-GPU box still runs every
        # all-reduce through the library: tests/test_gpu_dist.py)
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.buckets = []      # [start, end) element ranges, last bucket of the buffer first
        self.launch_log = []   # bucket ids in launch order, "hook" / "finish" (diagnostics + tests)
        self.event_log = []    # ("arrive", parameter index) / ("launch", bucket id) in program order (tests)
        self._pending, self._works, self._handles = [], {}, []
        self.defer, self.bytes_reduced, self.exposed_s, self._owner_of = False, 0, 0.0, {}
        self.bucket_of = []    # bucket id of every parameter
        if not self.enabled:
            return
        self.buckets, self.bucket_of = self.cut_buckets([p.numel() for p in params], bucket_elems, groups, flat_g.numel())
        owner = self.bucket_of
        self._members = [0] * len(self.buckets)
        for i, p in enumerate(params):
            if p.requires_grad:
                self._members[owner[i]] += 1
                self._owner_of[id(p)] = owner[i]
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(owner[i], i)))
        self.reset()

    @staticmethod
    def cut_buckets(numels, bucket_elems, groups=None, flat_numel=None):
        """(buckets, bucket id per parameter): contiguous [start, end) ranges cut from the END of the flat buffer; closed
        at `bucket_elems` elements or at a change of `groups[i]` once the bucket holds bucket_elems / 8 (a tiny module --
        the 0.36 MB mel_encoder -- rides with its neighbour rather than paying a collective's latency for itself)."""
        offs, off = [], 0
        for n in numels:
            offs.append((off, off + n))
            off += n
        buckets, owner = [], [0] * len(numels)
        end, count, cur = off, 0, 0
        for i in range(len(numels) - 1, -1, -1):
            owner[i] = cur
            count += numels[i]
            boundary = groups is not None and i > 0 and groups[i - 1] != groups[i] and count >= max(1, bucket_elems // 8)
            if count >= bucket_elems or boundary or i == 0:
                buckets.append((offs[i][0], end))
                end, count, cur = offs[i][0], 0, cur + 1
        if buckets and flat_numel is not None and buckets[0][1] < flat_numel:  # padding tail of the flat buffer rides with bucket 0
            buckets[0] = (buckets[0][0], flat_numel)
        return buckets, owner

    def _make_hook(self, b, index=-1):
        # NB torch fires a post-accumulate-grad hook for every leaf the backward graph reaches, also when the backward
        # function returned None for it -- which is what lets the gradient kernels write .grad directly (training.
        # FlatAdamW.sink) and still have their bucket launched from here, right after their kernel was enqueued.
        def hook(_param):
            self.event_log.append(("arrive", index))
            self._arrive(b)
        return hook

    def _arrive(self, b):
        if self.defer or not self.enabled:
            return
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b, "hook")
        elif self._pending[b] < 0:
            # a second backward into the same optimizer step: the bucket has already been SUM-reduced, local
            # gradients added on top of it would never be exchanged and the replicas would silently diverge
            raise RuntimeError("GradBucketer: a gradient arrived after its bucket was all-reduced (more than one "
                               "backward per optimizer step); use reset(defer=True) for gradient accumulation")

    def param_ready(self, param):
        """Manual arrival, for a gradient producer outside autograd (autograd itself fires the hook of every leaf it
        reaches, also for gradients a kernel wrote straight into the flat buffer)."""
        if self.enabled:
            b = self._owner_of.get(id(param))
            if b is not None:
                self._arrive(b)

    def _launch(self, b, why):
        s, e = self.buckets[b]
        leaf = None
        if self.flat_g.is_cuda:
            # gradient kernels of this bucket may still be queued on the leaf stream (autograd_ops.leaf_work): the collective is enqueued
            # behind BOTH streams -- from the leaf stream, after it has been made to wait for the compute stream -- without stalling backward
            from . import autograd_ops as A
            leaf = A.leaf_fence(self.flat_g.device)
        if leaf is not None:
            with torch.cuda.stream(leaf):
                self._works[b] = dist.all_reduce(self.flat_g[s:e], op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._works[b] = dist.all_reduce(self.flat_g[s:e], op=dist.ReduceOp.SUM, async_op=True)
        self.launch_log.append((b, why))
        self.event_log.append(("launch", b))
        self.bytes_reduced += 4 * (e - s)

    def reset(self, defer=False):
        """Call before every backward of a new optimizer step.  defer=True (gradient accumulation: several backwards per
        step, the reference's `accumulate_grad_batches`): nothing is launched from the hooks, finish() reduces every
        bucket once after the last backward."""
        if self.enabled:
            self._pending = list(self._members)
            self._works = {}
            self.launch_log = []
            self.event_log = []
            self.defer = bool(defer)
            self.bytes_reduced = 0

    def finish(self):
        """All buckets reduced (SUM) when this returns; returns the world size (1 when not distributed).
        `exposed_s` = host time spent waiting here (communication that backward did not hide)."""
        if not self.enabled:
            return 1
        import time
        t0 = time.perf_counter()
        for b in range(len(self.buckets)):
            if b not in self._works:
                self._launch(b, "finish")
        for w in self._works.values():
            w.wait()
        self.exposed_s = time.perf_counter() - t0
        return dist.get_world_size()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
