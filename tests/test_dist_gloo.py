"""world_size-2 gloo test of the multi-GPU path: utterance sharding (no data-path collective) and the
barrier + max-over-ranks timing reduction bench.py uses.  Runs on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import set_amd  # noqa: F401
    from set_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sample = {"mels": torch.arange(6 * 4 * 2, dtype=torch.float32).reshape(6, 4, 2), "mel2ph": torch.arange(24).reshape(6, 4),
              "nsamples": 6}
    sh = parallel.shard_batch(sample, r, w)
    parallel.barrier()
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    w_ret = parallel.bucketed_all_reduce_sum_(flat, 4)  # 3 buckets: 4 + 4 + 2
    assert w_ret == world and torch.equal(flat, torch.arange(10, dtype=torch.float32) * 3)
    tmax = parallel.max_over_ranks(1.0 + rank)
    tot = parallel.sum_over_ranks(float(sh["mels"].shape[0]))
    q.put((rank, sh["mels"][:, 0, 0].tolist(), sh["mel2ph"].shape[0], sh["nsamples"], tmax, tot))
    dist.destroy_process_group()


def test_shard_and_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, n0, ns0, t0, tot0), (r1, m1, n1, ns1, t1, tot1) = res
    assert m0 == [0.0, 16.0, 32.0] and m1 == [8.0, 24.0, 40.0]  # utterances 0,2,4 / 1,3,5
    assert n0 == n1 == 3 and ns0 == 6
    assert t0 == t1 == 2.0 and tot0 == tot1 == 6.0


def _toy(seed=0):
    torch.manual_seed(seed)
    layers = [torch.nn.Linear(8, 8) for _ in range(5)]
    unused = torch.nn.Linear(8, 3)  # never part of the graph (like fs.decoder / fs.mel_out in the reference)
    params = [p for l in layers for p in l.parameters()] + list(unused.parameters())
    n = sum(p.numel() for p in params)
    flat_g = torch.zeros((n + 255) // 256 * 256)
    off = 0
    for p in params:
        p.grad = flat_g[off:off + p.numel()].view(p.shape)
        off += p.numel()

    def loss_of(x):
        h = x
        for l in layers:
            h = torch.tanh(l(h))
        return (h * h).sum()

    return params, flat_g, loss_of


def _bucket_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import set_amd  # noqa: F401
    from set_amd import parallel
    parallel.init_from_env(backend="gloo")
    params, flat_g, loss_of = _toy()
    bk = parallel.GradBucketer(params, flat_g, bucket_elems=100)  # 72-element layers -> several buckets
    xs = [torch.randn(4, 8, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    logs = []
    for it in range(2):  # second iteration: reset() re-arms the hooks
        flat_g.zero_()
        bk.reset()
        loss_of(xs[rank]).backward()
        n_hook = sum(1 for _, why in bk.launch_log if why == "hook")
        assert bk.finish() == world
        logs.append((list(bk.launch_log), n_hook))
    got = flat_g.clone()
    # expected: sum over ranks of the local gradients, computed without any hook machinery
    want = torch.zeros_like(flat_g)
    for r in range(world):
        p2, g2, loss2 = _toy()
        loss2(xs[r]).backward()
        want += g2
    q.put((rank, float((got - want).abs().max()), logs, [b for b in bk.buckets], all(p.grad.data_ptr() >= flat_g.data_ptr() for p in params)))
    dist.destroy_process_group()


def test_grad_bucketer_overlaps_and_matches_sum_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, logs, buckets, in_flat in res:
        assert err < 1e-6 and in_flat
        assert len(buckets) >= 3 and buckets[0][1] % 256 == 0          # several buckets, padding rides with the last range
        assert sorted(b for b in buckets) == sorted(buckets) and all(s < e for s, e in buckets)
        for log, n_hook in logs:
            assert n_hook >= 2                                          # launched from autograd hooks, i.e. during backward
            assert sorted(b for b, _ in log) == list(range(len(buckets)))  # every bucket exactly once
            assert any(why == "finish" for _, why in log)               # the bucket holding the unused parameters
    assert res[0][2] == res[1][2]                                       # same launch order on both ranks


def test_single_process_is_identity():
    import set_amd  # noqa: F401
    from set_amd import parallel
    s = {"a": torch.ones(3, 2)}
    assert parallel.shard_batch(s, 0, 1) is s
    assert parallel.max_over_ranks(3.5) == 3.5
    f = torch.ones(5)
    assert parallel.bucketed_all_reduce_sum_(f, 2) == 1 and torch.equal(f, torch.ones(5))
    bk = parallel.GradBucketer([torch.nn.Parameter(torch.ones(5))], f, 2)
    assert not bk.enabled and bk.finish() == 1


def _ddp_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import set_amd  # noqa: F401
    from set_amd import parallel
    from set_amd.training import FlatAdamW
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)  # rank-divergent initial weights ON PURPOSE
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    model.register_buffer("table", torch.randn(7))
    opt = FlatAdamW(model, lr=1e-3, warmup_updates=10, bucket_mb=1e-4)
    opt.m.fill_(float(rank + 1))
    opt.num_updates = 5 * (rank + 1)
    before = opt.flat_p.clone()
    sent = parallel.configure_ddp(model, opt)
    # every replica now holds rank 0's parameters, buffers, Adam moments and step counter
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref_table = torch.randn(7)
    same = all(torch.equal(a, b) for a, b in zip(model.parameters(), ref.parameters())) and torch.equal(model.table, ref_table)
    views_ok = all(p.data_ptr() >= opt.flat_p.data_ptr() and p.data_ptr() < opt.flat_p.data_ptr() + 4 * opt.flat_p.numel()
                   for p in model.parameters())
    # gradient accumulation guard: a second backward after the buckets were reduced must raise, defer=True must not
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(rank))
    opt.zero_grad()
    model(x).pow(2).sum().backward()
    raised = False
    try:
        model(x).pow(2).sum().backward()
    except RuntimeError as e:
        raised = "more than one backward" in str(e)
    opt.bucketer.finish()
    opt.zero_grad(accumulate=True)
    model(x).pow(2).sum().backward()
    model(x).pow(2).sum().backward()
    n_hook = sum(1 for _, why in opt.bucketer.launch_log if why == "hook")
    world_ret = opt.bucketer.finish()
    g_sum = opt.flat_g.clone()
    # expected: sum over ranks of 2 x the local gradient
    want = torch.zeros_like(g_sum)
    for r in range(world):
        m2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        m2.load_state_dict(ref.state_dict())
        xr = torch.randn(4, 6, generator=torch.Generator().manual_seed(r))
        (2 * m2(xr).pow(2).sum()).backward()
        want[:opt.n] += torch.cat([p.grad.reshape(-1) for p in m2.parameters()])
    q.put((rank, same, views_ok, sent, bool((before != opt.flat_p).any()), float(opt.m[0]), opt.num_updates, raised, n_hook,
           world_ret, float((g_sum - want).abs().max()), opt.bucketer.bytes_reduced))
    dist.destroy_process_group()


def test_configure_ddp_broadcasts_rank0_state_and_accumulation_is_guarded_world2():
    """Reference: DistributedDataParallel(task) + dist.barrier() (utils/commons/trainer.py:166-170,402,475-479) make
    rank-divergent replicas identical to rank 0; ADVICE r1: a second backward per step must not double-reduce."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, views_ok, sent, changed, m0, n_upd, raised, n_hook, world_ret, err, reduced in res:
        assert same and views_ok
        assert sent == 4 * (256 + 7)            # one flat parameter buffer (padded to 256) + the buffer
        assert changed == (rank == 1)           # rank 1's own init was overwritten, rank 0's was not
        assert m0 == 1.0 and n_upd == 5         # optimizer state follows rank 0 too
        assert raised                           # second backward without accumulate=True is refused
        assert n_hook == 0 and world_ret == 2   # accumulate=True: nothing launched from hooks, finish() reduces
        assert err < 1e-5
        assert reduced == 4 * (6 * 5 + 5 + 5 * 3 + 3)   # the parameters themselves: the padding of the flat buffer is not sent


def test_bucket_cut_follows_gradient_arrival_order_of_the_spec_denoiser():
    """parameter order is denoise_fn.* (57.7 MB) -> fs.* -> mel_encoder.* while backward reaches denoise_fn
    first (back to front) and the conditioner last.  Buckets are contiguous ranges cut from the END of the flat buffer;
    cutting by size alone (64 MB) made bucket 0 = conditioner + the back half of DiffNet, which can only launch at the very
    end of backward.  With the module-boundary rule and ~25 MB buckets: no bucket mixes denoise_fn with the conditioner, the
    buckets that hold only denoise_fn tensors come in back-to-front order, the tiny mel_encoder rides with fs, and the whole
    buffer is covered exactly once."""
    import json
    import os
    import numpy as np
    from conftest import GOLDEN
    from oracle import weights as Wt
    from set_amd.parallel import GradBucketer
    man = [(k, int(np.prod(shape)) if shape else 1) for k, shape in json.load(open(os.path.join(GOLDEN, "manifest_spec_denoiser.json")))
           if not Wt.is_buffer(k)]
    names, numels = [k for k, _ in man], [n for _, n in man]
    groups = [k.split(".", 1)[0] for k in names]
    assert groups[0] == "denoise_fn" and groups[-1] == "mel_encoder" and "fs" in groups
    total = sum(numels)
    for mb in (25, 64):
        buckets, owner = GradBucketer.cut_buckets(numels, mb * (1 << 20) // 4, groups, total + 7)
        # exact cover, last bucket of the buffer first, padding tail in bucket 0
        assert buckets[0][1] == total + 7 and buckets[-1][0] == 0
        assert all(buckets[i][0] == buckets[i + 1][1] for i in range(len(buckets) - 1))
        mods = [sorted({groups[i] for i in range(len(names)) if owner[i] == b}) for b in range(len(buckets))]
        assert all(not ("denoise_fn" in m and len(m) > 1) for m in mods), mods
        assert mods[0] == ["fs", "mel_encoder"]            # 0.36 MB does not get a collective of its own
        dn = [b for b, m in enumerate(mods) if m == ["denoise_fn"]]
        assert dn == list(range(dn[0], len(buckets)))      # contiguous, and they end the list: front of the buffer last
        if mb == 25:
            assert len(dn) == 3 and len(buckets) == 5, (len(dn), len(buckets))
            sizes = [4 * (e - s) / 2 ** 20 for s, e in buckets]
            assert all(sz <= 2 * 25 for sz in sizes)
    # without groups: the old behaviour (one bucket spanning both) is still what a caller without names gets
    b_old, own_old = GradBucketer.cut_buckets(numels, 64 * (1 << 20) // 4, None, total)
    assert len({groups[i] for i in range(len(names)) if own_old[i] == 0}) == 3


def test_bench_train_mode_launches_its_own_ranks_end_to_end_on_a_stubbed_device():
    """`python bench.py --gpus 2 --mode train` from a bare shell -- the driver's multi-GPU command minus the launcher --
    must start its own two ranks under torch.distributed.run on 127.0.0.1, make the rank-divergent replicas identical (rank-0
    broadcast), shard the batch, exchange the gradients in buckets launched from autograd hooks, time with barrier + max over
    ranks and print ONE JSON line on rank 0.  No GPU here: SET_AMD_BENCH_STUB_DEVICE=cpu swaps the task for a toy torch module and
    RCCL for gloo; everything between `main()` and the JSON line is the code the real run executes
    (reference: mp.spawn + DDP, utils/commons/trainer.py:116-137,166-170,475-485)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SET_AMD_BENCH_STUB_DEVICE"] = "cpu"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["device"] == "cpu-stub" and d["metric"].startswith("STUB") and d["roofline"] is None
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo" and d["steps"] == 3 and d["warmup"] == 1
    assert len(d["per_rank_ms_per_step"]) == 2 and d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) - 1e-9   # max over ranks
    assert d["scaling"] == "weak" and d["config"]["B_per_gpu"] == 8
    assert d["replicas_identical_after_steps"] is True
    # the toy module: 16*64+64 + 64*64+64 + 64*8+8 reachable parameters, 8*8+8 behind `unused_parameter_prefixes`
    n_used, n_dead = 16 * 64 + 64 + 64 * 64 + 64 + 64 * 8 + 8, 8 * 8 + 8
    assert d["grad_elems"] == n_used + n_dead and d["grad_elems_exchanged"] == n_used
    assert d["allreduce_bytes_per_step"] == 4 * n_used           # every step exchanges the reachable gradients exactly once
    assert d["param_broadcast_bytes"] == 4 * ((n_used + n_dead + 255) // 256 * 256)
    assert "in 3 buckets" in d["config"]["sharding"]
    # and the refusals: a launcher that started a different number of ranks, and no GPU without the test hook
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train"],
                        env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "launcher started 1 rank" in (r2.stderr + r2.stdout)
    if not torch.cuda.is_available():
        env3 = {k: v for k, v in env.items() if k != "SET_AMD_BENCH_STUB_DEVICE"}
        r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "train"], env=env3,
                            capture_output=True, text=True, timeout=300)
        assert r3.returncode != 0 and "needs an MI355X" in (r3.stderr + r3.stdout)


def test_bench_infer_mode_launches_its_own_ranks_end_to_end_on_a_stubbed_device():
    """`python bench.py --gpus 2` (default --mode infer) is the command the driver's scaling run issues; its world > 1
    branches -- self-launch under torch.distributed.run on 127.0.0.1, shard_batch r::N of ONE global batch of 32 x N utterances, barrier,
    max over ranks on the rank's device, the all-reduce / all-gather of dist_facts, one JSON line on rank 0 -- had never executed anywhere.
    SET_AMD_BENCH_STUB_DEVICE=cpu swaps the model for a toy torch-CPU callable with the same call signature and return keys, RCCL for
    gloo; everything else between main() and the printed line is the code the real run executes (no collective on the data path:
    SURVEY.md 8(e); the reference shards the same way, tasks/tts/speech_base.py:128-131)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SET_AMD_BENCH_STUB_DEVICE"] = "cpu"
    lines_by_n = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
        lines_by_n[n] = d = json.loads(lines[0])
        assert d["device"] == "cpu-stub" and d["metric"].startswith("STUB")
        assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["steps"] == 3 and d["warmup"] == 1
        assert d["dist_backend"] == ("gloo" if n > 1 else None)
        assert len(d["per_rank_ms_per_step"]) == n and d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) - 1e-9   # max over ranks
        assert d["scaling"] == "weak" and d["config"]["B_per_gpu"] == 32 and d["higher_is_better"] is True
        # whole-job aggregate: all N x 32 utterances x 800 frames x steps over the slowest rank's time
        assert abs(d["value"] - n * 32 * 800 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
        for k in ("roofline", "unit", "vs_baseline", "dtype", "data"):
            assert k in d
        assert "cpu_baseline" not in d and "train_bf16" not in d      # 1-GPU extras never run on the stub
    # rank 0 of the 2-rank run holds utterances 0, 2, 4, ... of a 64-utterance batch: not the 1-rank run's 32
    assert lines_by_n[1]["stub_shard_checksum"] != lines_by_n[2]["stub_shard_checksum"]
    # the refusal: a launcher that started a different number of ranks
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "launcher started 1 rank" in (r2.stderr + r2.stdout)
    if not torch.cuda.is_available():
        env3 = {k: v for k, v in env.items() if k != "SET_AMD_BENCH_STUB_DEVICE"}
        r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env3, capture_output=True, text=True, timeout=300)
        assert r3.returncode != 0 and "needs an MI355X" in (r3.stderr + r3.stdout)
