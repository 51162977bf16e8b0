"""Trainer counterpart for the path's training rows (utils/commons/trainer.py:116-123,140-200,
256-379,431-485 in the reference): build the task, restore the newest checkpoint, make the data-parallel replicas
identical, then loop `run_training_batch` with a validation + checkpoint every `val_check_interval` updates.

What is deliberately different from the reference (and why):
* one process per GPU is started by the launcher (`torchrun`, or `bench.py`'s self-launch); `fit()` joins the process
  group from the environment instead of calling `mp.spawn` itself -- RCCL wants one process per device either way;
* every random stream of a step is a function of (hparams['seed'], global_step): the diffusion step ids `t`, the Philox
  seed of noise / dropout, and the numpy / python / torch generators the dataset's mask generators draw from.  A run
  resumed from `model_ckpt_steps_N.ckpt` therefore continues with exactly the batches and random numbers the
  uninterrupted run would have used (the reference reseeds and restarts its endless batch list from the beginning);
* `global_step` counts OPTIMIZER UPDATES; with `accumulate_grad_batches` = a > 1 an update consumes a batches (the
  reference counts batches and steps the optimizer on every a-th, scheduling on global_step // a): `max_updates`,
  `val_check_interval`, the warm-up and the step number in checkpoint names therefore mean a times more data here than in
  a reference run with the same yaml; a checkpoint without the `global_step_unit` key resumed with a > 1 takes its unit from
  hparams['resume_global_step_unit'] = 'updates' | 'batches', else from the restored optimizer's own step count, else is refused;
* the optimizer is the fused flat AdamW (training.FlatAdamW) whose state_dict is torch.optim.AdamW's, so checkpoints
  interchange (`optimizer_states[0]`); clip + schedule are inside its step (base_task.py:129-137).
No tensorboard, no progress bars, no code snapshots: logging is a dict per `log_interval` updates on rank 0.
"""
import os
import random
import threading
import time

import numpy as np
import torch

from . import ckpt_utils, parallel
from .hparams import hparams


def move_to_device(batch, device):
    return {k: (v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def step_seed(seed, global_step):
    """Philox seed / generator seed of one update: a fixed mix of the run seed and the update index."""
    return (int(seed) * 1000003 + int(global_step) * 7919 + 12345) % (2 ** 31 - 1)


# One lock for every piece of code that reseeds or draws from the process-global numpy / python / torch generators on behalf of the
# data feed: BatchLoader.fetch (prefetch worker thread AND main thread: validation batches) and the batch-list construction of
# tasks._loader.  Without it the worker's fetch(k + 1) and a validation pass on the main thread reseed the same globals while the
# other one is drawing, and both batches depend on thread timing.
RNG_LOCK = threading.RLock()


class BatchLoader:
    """`dataset[i]` + `collater` over a list of index batches, addressed by update index.  The dataset's mask
    generators draw from the global numpy / python / torch generators (utils/spec_aug/time_mask.py:6-93), so these are
    seeded per batch position: what batch k contains does not depend on how many batches were fetched before it.
    `fetch` holds RNG_LOCK from the reseed to the last draw and puts the three global generator states back afterwards:
    it is a pure function of k whatever other thread fetches at the same time, and it leaves no trace in the global streams."""

    def __init__(self, dataset, batches, seed):
        self.dataset, self.batches, self.seed = dataset, batches, int(seed)

    def __len__(self):
        return len(self.batches)

    def fetch(self, k):
        idx = self.batches[k % len(self.batches)]
        s = step_seed(self.seed, k)
        with RNG_LOCK:
            saved = (np.random.get_state(), random.getstate(), torch.get_rng_state())
            try:
                np.random.seed(s)
                random.seed(s)
                torch.default_generator.manual_seed(s)  # the CPU generator only: torch.manual_seed would also reseed every CUDA generator
                return self.dataset.collater([self.dataset[i] for i in idx])
            finally:
                np.random.set_state(saved[0])
                random.setstate(saved[1])
                torch.set_rng_state(saved[2])


class Trainer:
    def __init__(self, work_dir, val_check_interval=2000, max_updates=160000, num_sanity_val_steps=0,
                 accumulate_grad_batches=1, num_ckpt_keep=3, seed=1234, log_interval=100, device=None, **_unused):
        self.work_dir = work_dir
        self.val_check_interval = int(val_check_interval)
        self.max_updates = int(max_updates)
        self.num_sanity_val_steps = int(num_sanity_val_steps)
        self.accumulate_grad_batches = int(accumulate_grad_batches)
        self.num_ckpt_keep = int(num_ckpt_keep)
        self.seed = int(seed)
        self.log_interval = int(log_interval)
        self.device = device
        self.global_step, self.current_epoch = 0, 0
        self.best_val_results = None
        self.first_epoch = True
        self.testing = False
        self.task = self.optimizer = None
        self.rank, self.world = 0, 1
        self.history = []  # (global_step, total loss, losses) of every update of this process (tests, logging)
        self._prefetch = None

    # ---- entry points (trainer.py:112-137) -----------------------------------------------------------------------
    def test(self, task_cls):
        self.testing = True
        return self.fit(task_cls)

    def fit(self, task_cls):
        self.rank, self.world, local_rank = parallel.init_from_env()
        if self.device is None:
            self.device = torch.device("cuda", local_rank)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        task = task_cls() if isinstance(task_cls, type) else task_cls
        task.trainer = self
        self.task = task
        self.run_single_process(task)
        return 1

    def run_single_process(self, task):
        """trainer.py:140-200: build, restore, optimizer, DDP set-up, then test or train."""
        if self.testing:
            task.test()
            return
        model = task.build_model()
        model.to(self.device).train()
        self.optimizer = task.configure_optimizers()
        self.global_step, self.current_epoch, meta = ckpt_utils.restore_ckpt(self.work_dir, model, self.optimizer, return_meta=True)
        if self.global_step > 0:
            self.best_val_results = meta.get("checkpoint_callback_best")
            if self.accumulate_grad_batches > 1 and meta.get("global_step_unit") != "updates":
                # a checkpoint without the key is either the reference trainer's (global_step counts BATCHES) or one this trainer wrote
                # before the key existed (it already counted updates).  hparams['resume_global_step_unit'] = 'updates' | 'batches' settles
                # it; unset, the restored optimizer's own step count decides (updates: == global_step, batches: == global_step // a); if
                # neither fits the run is refused -- a silent wrong guess mis-schedules warm-up, lr, validation and max_updates by a
                unit = hparams.get("resume_global_step_unit")
                if unit not in ("updates", "batches"):
                    n_opt = int(getattr(self.optimizer, "num_updates", -1))
                    a = self.accumulate_grad_batches
                    # the reference steps its optimizer exactly floor(N / a) times for N batches (utils/commons/trainer.py:366): only that
                    # count means "batches"; when BOTH readings fit (tiny global_step) the run is refused like any other ambiguous one
                    fits_updates = n_opt > 0 and n_opt == self.global_step
                    fits_batches = n_opt > 0 and n_opt == self.global_step // a
                    if fits_updates and not fits_batches:
                        unit = "updates"
                    elif fits_batches and not fits_updates:
                        unit = "batches"
                    else:
                        raise RuntimeError(
                            "resuming a checkpoint without 'global_step_unit' under accumulate_grad_batches=%d: cannot tell whether its "
                            "global_step=%d counts optimizer updates or batches (optimizer step count %d); set "
                            "resume_global_step_unit=updates|batches (a reference-written checkpoint counts batches)"
                            % (a, self.global_step, n_opt))
                    if self.rank == 0:
                        print("| resume: checkpoint has no 'global_step_unit'; optimizer step count %d => global_step counts %s"
                              % (n_opt, unit), flush=True)
                if unit == "batches":
                    self.global_step //= self.accumulate_grad_batches
        task.global_step = self.global_step
        # barrier, rank-0 parameter / buffer / optimizer-state broadcast, barrier (trainer.py:166-170,402,475-479)
        parallel.configure_ddp(model, self.optimizer)
        self.train()

    # ---- training loop (trainer.py:256-304) -------------------------------------------------------------------------
    def train(self):
        task = self.task
        if self.num_sanity_val_steps > 0:
            self.evaluate(max_batches=self.num_sanity_val_steps)
        loader = task.train_dataloader()
        t_log = time.perf_counter()
        while self.global_step <= self.max_updates:
            if self.global_step % self.val_check_interval == 0 and not self.first_epoch:
                self.run_evaluation()
            self.run_training_batch(loader)
            self.first_epoch = False
            self.global_step += 1
            task.global_step = self.global_step
            if self.rank == 0 and self.log_interval > 0 and self.global_step % self.log_interval == 0:
                gs, total, parts = self.history[-1]
                total, parts = float(total), {k: float(v) for k, v in parts.items()}
                dt = time.perf_counter() - t_log
                t_log = time.perf_counter()
                print("| step %d loss %.4f %s (%.1f ms/step)" % (self.global_step, total, " ".join(
                    "%s=%.4f" % kv for kv in sorted(parts.items())), 1e3 * dt / self.log_interval), flush=True)
        if self.rank == 0:
            print("| Training end..")

    def _next_batch(self, loader, k):
        """Batch k of the training list.  With hparams['ds_workers'] > 0 (the reference's DataLoader-worker knob,
        utils/commons/dataset_utils.py:213-215) batch k + 1 is assembled (dataset[i], mask generators, collate, pinned host
        buffers) on a background thread while update k runs on the GPU; `fetch(k)` is a pure function of k (it reseeds, draws
        and restores the global generators under RNG_LOCK), so the prefetched batch is the one a synchronous fetch would have
        produced whatever the main thread does meanwhile -- validation passes fetch through the same lock, and
        `run_evaluation` waits for the batch in flight first (`_drain_prefetch`): a resumed run stays bit-identical."""
        if int(hparams.get("ds_workers", 0) or 0) <= 0:
            return loader.fetch(k)
        if self._prefetch is None or self._prefetch[0] is not loader:
            from concurrent.futures import ThreadPoolExecutor
            self._prefetch = (loader, ThreadPoolExecutor(max_workers=1), {})
        _, pool, pending = self._prefetch
        pin = self.device is not None and self.device.type == "cuda"

        def work(kk):
            b = loader.fetch(kk)
            return {n: (v.pin_memory() if pin and isinstance(v, torch.Tensor) else v) for n, v in b.items()}
        fut = pending.pop(k, None)
        for stale in [j for j in pending if j != k + 1]:
            pending.pop(stale).cancel()
        batch = fut.result() if fut is not None else work(k)
        if k + 1 not in pending:
            pending[k + 1] = pool.submit(work, k + 1)
        return batch

    def _drain_prefetch(self):
        """Wait for the batch the worker is assembling (its result stays queued: it is the batch of its index either way)."""
        if self._prefetch is not None:
            for fut in list(self._prefetch[2].values()):
                try:
                    fut.result()
                except Exception:
                    pass  # surfaces when the batch is consumed

    def run_training_batch(self, loader):
        """trainer.py:306-379 for one optimizer: `accumulate_grad_batches` forward/backward passes, then clip + AdamW
        + schedule.  Gradient exchange: launched from autograd hooks during the (single) backward, or after the last
        backward when accumulating."""
        task, opt = self.task, self.optimizer
        acc = self.accumulate_grad_batches
        opt.zero_grad(accumulate=acc > 1)
        total, parts = None, {}
        try:
            for micro in range(acc):
                k = self.global_step * acc + micro
                batch = move_to_device(self._next_batch(loader, k), self.device)
                # model-side randomness (diffusion steps, noise, dropout) also depends on the rank: data-parallel replicas
                # must not draw the same t / eps / dropout stream for their i-th sample (the batch LIST stays rank-free)
                seed = (step_seed(self.seed, k) + 104729 * self.rank) % (2 ** 31 - 1)
                B = batch["txt_tokens"].shape[0]
                t = torch.from_numpy(np.random.default_rng([self.seed, k] + ([self.rank] if self.rank else [])).integers(
                    0, int(task.model.num_timesteps) + 1, size=(B,), dtype=np.int64)).to(self.device) \
                    if hasattr(task.model, "num_timesteps") else None
                loss, log = task._training_step(batch, k, seed=seed, t=t)
                with torch.enable_grad():
                    (loss / acc if acc > 1 else loss).backward()
                total = loss.detach() if total is None else total + loss.detach()
                parts = log  # device scalars: converting here would add a host sync per loss per update
        except BaseException:
            opt.abort_step()
            raise
        opt.step()
        self.history.append((self.global_step, total / acc, parts))
        if len(self.history) > 1024:
            del self.history[:512]

    # ---- validation + checkpoints (trainer.py:205-254, 431-470) ------------------------------------------------------
    def run_evaluation(self):
        self._drain_prefetch()
        res = self.evaluate(max_batches=hparams.get("eval_max_batches", -1))
        if self.rank == 0:
            self.save_checkpoint(logs=res)
        parallel.barrier()
        return res

    def evaluate(self, max_batches=None):
        task = self.task
        loader = task.val_dataloader()
        if loader is None or len(loader) == 0:
            return None
        if max_batches is None or max_batches < 0:
            max_batches = len(loader)
        task.model.eval()
        tot, n = {}, 0
        with torch.no_grad():
            for k in range(min(max_batches, len(loader))):
                batch = move_to_device(loader.fetch(k), self.device)
                out = task.validation_step(batch, k)
                ns = int(batch.get("nsamples", 1))
                for kk, v in out["losses"].items():
                    tot[kk] = tot.get(kk, 0.0) + float(v) * ns
                tot["total_loss"] = tot.get("total_loss", 0.0) + float(out["total_loss"]) * ns
                n += ns
        task.model.train()
        res = {k: round(v / max(n, 1), 4) for k, v in tot.items()}
        if self.rank == 0:
            print("| Validation results@%d: %s" % (self.global_step, res), flush=True)
        return {"val_loss": res.get("total_loss"), "losses": res}

    def save_checkpoint(self, logs=None):
        cur = None if logs is None else logs.get("val_loss")
        if cur is not None and (self.best_val_results is None or cur < self.best_val_results):
            self.best_val_results = cur
        return ckpt_utils.save_ckpt(self.work_dir, self.task.model, self.optimizer, global_step=self.global_step,
                                    epoch=self.current_epoch, best=self.best_val_results,
                                    num_ckpt_keep=self.num_ckpt_keep)
