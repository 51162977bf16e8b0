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
