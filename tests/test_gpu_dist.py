"""GPU tests of the data-parallel training path (SURVEY.md 8 row e / a21; reference: utils/commons/trainer.py:116-137,
166-170,402,475-485 and torch DDP's gradient averaging).

* two ranks, real kernels: rank-divergent initial weights become rank 0's after `configure_ddp`, and the bucketed
  SUM all-reduce / world of the two shards' gradients equals the single-process gradient of the concatenated batch.
  On a box with >= 2 GPUs this runs over RCCL ('nccl', one GPU per rank); on the 1-GPU box both ranks share cuda:0 and
  the collective is gloo on device tensors (RCCL refuses two ranks on one device) -- same host logic, same kernels.
* a single-rank RCCL group on one GPU drives the same code through the library itself (broadcast, bucketed
  all-reduce from autograd hooks, barrier).
* `bench.py --gpus N` refuses to report an N-GPU number from fewer devices.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _paired_batch(B, T, T_txt):
    """Global batch whose shards r::2 share txt / mel2ph / masks / uv (so every masked-mean loss has the same
    denominators on both shards and mean-of-shard-gradients == gradient of the concatenated batch) but differ in the
    mels, f0 values and speaker embeddings."""
    from set_amd.synthetic import synthetic_inputs
    a = synthetic_inputs(B // 2, T, T_txt, seed=31, pad_tail=True)
    b = synthetic_inputs(B // 2, T, T_txt, seed=32, pad_tail=True)
    out = {}
    for k in a:
        rows = []
        for i in range(B // 2):
            rows += [a[k][i], (a if k in ("txt_tokens", "mel2ph", "time_mel_masks", "uv") else b)[k][i]]
        out[k] = torch.stack(rows)
    out["txt_tokens"][:, 3], out["txt_tokens"][:, 9] = 2, 1  # silence tokens: every utterance has words (finite wdur)
    pad = out["mel2ph"] == 0  # keep the padded frames of the second copy exactly zero too
    out["ref_mels"][pad] = 0.0
    out["f0"][pad] = 0.0
    return out


def _worker(rank, world, port, backend, force, q, model="denoiser"):
    try:
        _worker_body(rank, world, port, backend, force, q, model)
    except BaseException:
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc()))
        raise


def _worker_body(rank, world, port, backend, force, q, model="denoiser"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        os.environ["SET_AMD_FORCE_BUCKETER"] = "1"
    import torch.distributed as dist
    from conftest import base_hparams
    import set_amd  # noqa: F401
    from set_amd import hparams as H, parallel, tasks
    from set_amd.training import FlatAdamW
    dev_id = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if world > 1 or force:
        os.environ["LOCAL_RANK"] = str(dev_id)
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=rank, world_size=world)
    H.hparams.clear()
    torch.manual_seed(100 + rank)  # rank-divergent initial weights ON PURPOSE
    if model == "campnet":  # BASELINE configs[4]: the masked-mel transformer under the same data-parallel machinery
        import yaml
        with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "campnet.yaml")) as f:
            H.hparams.update(yaml.safe_load(f))
        H.hparams.update(binary_data_dir="", vocoder_ckpt="")
        task = tasks.CampNetTask(80, 100, build_vocoder=False)
        task.build_model()
        late_prefix = "encoder."      # the text encoder is the LAST module backward reaches
    else:
        H.hparams.update(base_hparams(timesteps=4, residual_layers=3))
        task = tasks.SpeechDenoiserTask(build_vocoder=False)
        task.build_model()
        torch.nn.init.normal_(task.model.denoise_fn.output_projection.weight, std=1.0 / 16.0)
        late_prefix = "fs."           # ... the conditioner here (parameter order: denoise_fn, fs, mel_encoder)
    task.model.to(dev).eval()  # eval: no predictor dropout (its Philox counters depend on the batch layout)
    opt = FlatAdamW(task.model, lr=2e-4, clip_grad_norm=1.0, warmup_updates=8000, bucket_mb=4)
    own = float(opt.flat_p.double().sum())
    sent = parallel.configure_ddp(task.model, opt)
    chk = opt.flat_p.double().sum().reshape(1)
    if world > 1:
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        replicas_equal = all(float(c) == float(allc[0]) for c in allc)
    else:
        replicas_equal = True

    B, T, T_txt = 4, 64, 16
    full = _paired_batch(B, T, T_txt)
    g = torch.Generator().manual_seed(5)
    t_full = torch.tensor([3, 3, 1, 1])
    eps_full = torch.randn(B, 80, T, generator=g)

    def sample_of(d):
        d = {k: v.to(dev) for k, v in d.items()}
        return dict(txt_tokens=d["txt_tokens"], mels=d["ref_mels"], mel2ph=d["mel2ph"], f0=d["f0"], uv=d["uv"],
                    time_mel_masks=d["time_mel_masks"].squeeze(-1).contiguous(), spk_embed=d["spk_embed"])

    def grads_of(sample, t, eps):
        opt.zero_grad()
        kw = dict(t=t.to(dev), noises=eps.to(dev).contiguous()) if model != "campnet" else {}
        losses, _ = task.run_model(sample, infer=False, **kw)
        with torch.enable_grad():
            total = sum(losses.values())
        total.backward()
        from set_amd import autograd_ops as A
        A.zero_arena_end()
        w = opt.bucketer.finish()
        return opt.flat_g.clone() / w, float(total), w

    sh = parallel.shard_batch(full, rank, world)
    g_first, _, _ = grads_of(sample_of(sh), t_full[rank::world], eps_full[rank::world])
    opt._learned = True  # (normally set by the first optimizer.step(): from now on the gradient kernels write .grad directly)
    g_dist, loss_local, w = grads_of(sample_of(sh), t_full[rank::world], eps_full[rank::world])
    direct_equal = bool(torch.equal(g_first, g_dist))  # autograd-accumulated vs directly written gradients: same bits
    n_direct = sum(1 for v in opt._uses.values() if v == 1)
    log = list(opt.bucketer.launch_log)
    # bucket order follows gradient arrival: the first all-reduce is in flight before the first gradient of the module
    # backward reaches last exists (a bucket that ends in the conditioner would leave the exchange exposed)
    ev = list(opt.bucketer.event_log)
    first_launch = next((i for i, e in enumerate(ev) if e[0] == "launch"), None)
    first_late = next((i for i, e in enumerate(ev) if e[0] == "arrive" and opt.param_names[e[1]].startswith(late_prefix)), None)
    launch_before_late = first_launch is not None and first_late is not None and first_launch < first_late
    spans = len({opt.param_names[i].split(".", 1)[0] for i, b in enumerate(opt.bucketer.bucket_of) if b == 0})
    n_buckets = len(opt.bucketer.buckets)
    reduced = opt.bucketer.bytes_reduced
    opt.bucketer.enabled = False  # single-process reference on the concatenated batch: no collective
    g_full, loss_full, _ = grads_of(sample_of(full), t_full, eps_full)
    rel = float((g_dist - g_full).abs().max() / g_full.abs().max())
    q.put(dict(rank=rank, own=own, after=float(chk), sent=sent, replicas_equal=replicas_equal, w=w, rel=rel,
               loss_local=loss_local, loss_full=loss_full, log=log, n_buckets=n_buckets, reduced=reduced, n=opt.n, n_exchanged=opt.n_exchanged,
               direct_equal=direct_equal, n_direct=n_direct, launch_before_late=launch_before_late, bucket0_modules=spans,
               backend=dist.get_backend() if dist.is_initialized() else None))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _run(world, backend, force=False, model="denoiser"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, force, q, model)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=240) for _ in procs), key=lambda d: d["rank"])
    except Exception:
        for p in procs:  # a worker died (its traceback is on stderr): do not leave the other rank waiting in a collective
            if p.is_alive():
                p.kill()
        raise
    for r in res:
        assert "error" not in r, r["error"]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _check_two_ranks(res):
    r0, r1 = res
    assert r0["own"] != r1["own"]                          # the ranks really initialised differently
    assert r0["replicas_equal"] and r1["replicas_equal"]
    assert r0["after"] == r1["after"] == r0["own"]         # ... and both now hold rank 0's parameters
    assert r0["sent"] >= 4 * r0["n"]
    for r in res:
        assert r["direct_equal"] and r["n_direct"] > 50  # second pass: kernels accumulate into .grad, buckets fire by hand
        assert r["w"] == 2
        # mean of the two shards' gradients == gradient of the concatenated batch (fp32 summation order differs)
        assert r["rel"] < 2e-5, r["rel"]
        assert r["n_buckets"] >= 3 and sorted(b for b, _ in r["log"]) == list(range(r["n_buckets"]))
        assert sum(1 for _, why in r["log"] if why == "hook") >= 2   # launched during backward
        assert r["launch_before_late"], "no all-reduce was in flight before backward reached its last module"
        assert r["reduced"] == 4 * r["n_exchanged"] and r["n_exchanged"] <= r["n"]  # spec_denoiser: fs.decoder / fs.mel_out (never reached by a loss) are not sent
    assert r0["log"] == r1["log"]                          # same launch order on both ranks (no deadlock by construction)
    assert abs(0.5 * (r0["loss_local"] + r1["loss_local"]) - r0["loss_full"]) < 1e-4 * abs(r0["loss_full"])


# (the RCCL tests come first: on a box with >= 2 GPUs they are the evidence the others cannot give)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_rccl_campnet_gradients_equal_concatenated_batch():
    res = _run(2, "nccl", model="campnet")
    _check_two_ranks(res)
    assert res[0]["backend"] == "nccl"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL refuses two ranks on one device)")
def test_two_ranks_rccl_gradients_equal_concatenated_batch():
    res = _run(2, "nccl")
    _check_two_ranks(res)
    assert res[0]["backend"] == "nccl"


def test_two_ranks_share_one_gpu_gradients_equal_concatenated_batch():
    """Runs on the 1-GPU box: both ranks on cuda:0, gloo moves the device tensors."""
    _check_two_ranks(_run(2, "gloo"))


def test_two_ranks_share_one_gpu_campnet_gradients_equal_concatenated_batch():
    """BASELINE configs[4] (CampNet under DDP): the same check on the masked-mel transformer -- rank-divergent weights
    become rank 0's, the mean of the two shards' gradients equals the gradient of the concatenated batch, buckets launch
    from the hooks in the same order on both ranks and the first one before backward reaches the text encoder."""
    _check_two_ranks(_run(2, "gloo", model="campnet"))


def test_single_rank_rccl_group_runs_the_collectives():
    """One rank, backend 'nccl' (= RCCL): the parameter broadcast, the bucketed all-reduces launched from autograd hooks
    and the barriers all go through the library; with one rank they must be the identity."""
    (r,) = _run(1, "nccl", force=True)
    assert r["backend"] == "nccl" and r["w"] == 1
    assert r["sent"] >= 4 * r["n"] and r["after"] == r["own"]
    assert r["n_buckets"] >= 3 and sorted(b for b, _ in r["log"]) == list(range(r["n_buckets"]))
    assert r["reduced"] == 4 * r["n_exchanged"] and r["n_exchanged"] < r["n"]
    assert r["rel"] < 1e-5  # same kernels, same batch, all-reduce over one rank = identity
    assert r["direct_equal"] and r["n_direct"] > 50


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
