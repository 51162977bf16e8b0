"""Optimizer / data-parallel glue for the training rows of the path.

* parameters and gradients live in two flat fp32 buffers (each nn.Parameter / .grad is a view), so the optimizer is
  ONE fused kernel (`set_adamw`: clip_grad_norm_ + AdamW, torch semantics) and the data-parallel gradient exchange is
  a few large RCCL all-reduces over xGMI, launched from autograd hooks so they overlap with the rest of backward
  (`parallel.GradBucketer`), instead of one per tensor (the reference relies on torch DDP's 25 MB
  buckets, utils/commons/trainer.py:475-479; fs.decoder / fs.mel_out never receive gradients there either);
* learning rate = WarmupSchedule (utils/nn/schedulers.py:42-57): lr * min(step / warmup, 1), floored at 1e-7.
"""
import torch

from . import autograd_ops as A
from . import ops, parallel


class FlatAdamW:
    def __init__(self, model, lr=2e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0, clip_grad_norm=1.0,
                 warmup_updates=8000, bucket_mb=64):
        self.params = [p for p in model.parameters()]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        n_pad = (n + 255) // 256 * 256
        self.flat_p = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n_pad, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            off += k
        self.n = n
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.lr0, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.clip, self.warmup = clip_grad_norm, warmup_updates
        self.bucket = int(bucket_mb * (1 << 20) // 4)
        # gradient exchange overlapped with backward: buckets launch from autograd hooks (no-op when world == 1)
        self.bucketer = parallel.GradBucketer(self.params, self.flat_g, self.bucket)
        self.num_updates = 0
        ops.bump_weights_epoch()

    def lr_at(self, num_updates):
        warm = min(num_updates / self.warmup, 1.0) if self.warmup > 0 else 1.0
        return max(self.lr0 * warm, 1e-7)

    def zero_grad(self):
        self.flat_g.zero_()
        self.bucketer.reset()
        for p in self.params:  # autograd may have re-pointed .grad; restore the views
            if p.grad is None or p.grad.data_ptr() < self.flat_g.data_ptr() or \
                    p.grad.data_ptr() >= self.flat_g.data_ptr() + 4 * self.flat_g.numel():
                raise RuntimeError("a .grad left the flat gradient buffer")

    def step(self):
        """gradient all-reduce (SUM, few large buckets, launched from autograd hooks while backward was still running;
        whatever is left goes now) -> clip_grad_norm_(max_norm) + AdamW on the mean gradient, then the warm-up
        schedule (base_task.py:129-137)."""
        world = self.bucketer.finish()
        sumsq = A.grad_sumsq(self.flat_g) if self.clip > 0 else None
        lr = self.lr_at(self.num_updates)
        self.num_updates += 1
        A.adamw_step(self.flat_p, self.flat_g, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                     self.num_updates, sumsq, self.clip, 1.0 / world)
        ops.bump_weights_epoch()
        return lr, sumsq
