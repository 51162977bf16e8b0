"""Headline benchmark: diffusion mel-frames/s of one full GaussianDiffusion.forward(infer=True)
(conditioner once + 100 x (DiffNet + posterior step)) at B=32 per GPU, T=800, fp32.

  python bench.py --gpus N --steps K --warmup W [--mode infer|train] [--dtype f32|bf16]

With N > 1 and no WORLD_SIZE in the environment the script re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
(one rank per GPU over RCCL; the reference spawns its ranks the same way, utils/commons/trainer.py:116-137,481-485)
and fails loudly when fewer than N devices are visible.  Under an external launcher it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* as usual.

One "step" = one pass of the hot path over one synthetic batch.
  --mode infer (default, BASELINE.json's metric): utterances shard across ranks with NO data-path collective (weak
      scaling, B=32 per rank).
  --mode train (BASELINE configs[1]/[2]): forward + losses + backward + bucketed RCCL gradient all-reduce overlapped
      with backward + clip + AdamW, after the rank-0 parameter broadcast and the two barriers of the reference's DDP
      set-up; reports samples/s, all-reduce bytes per step and the exposed (not hidden by backward) exchange time.
Timing = barrier + synchronize on both sides, MAX over ranks.  Rank 0 prints ONE JSON line with the contract fields plus
`roofline` (dominant kernel = the persistent DiffNet layer-stack kernel, one launch = all 20 residual layers of one
denoise step, measured with hipEvents on the launch stream inside the timed region) and, at N=1, `cpu_baseline` (the CPU
oracle = a port of the reference's torch-CPU path, timed on bounded samples: SURVEY.md 8(d)'s protocol).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml  # noqa: E402

B_PER_GPU, T, T_TXT, DIFF_STEPS, M, L, C, H = 32, 800, 100, 100, 80, 20, 256, 192
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0
# algorithmic work of the residual layers of ONE denoise step, per frame and layer:
#   k3 dilated conv 512x768 + output projection 512x256 = 524,288 MAC per frame (conditioner projection hoisted)
FLOP_PER_FRAME_LAYER = 2 * (512 * 768 + 512 * 256)
# algorithmic HBM bytes per frame per layer (fp32): x in 1024 + condproj 2048 + x out 1024 + skip rmw 2048
BYTES_PER_FRAME_LAYER = 1024 + 2048 + 1024 + 2048
# one training step, algorithmic FLOPs per frame (~3 x forward of one DiffNet pass + conditioner)
TRAIN_FLOP_PER_FRAME = 3 * (25_116_672 + 2_100_000)


def load_hparams():
    with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser.yaml")) as f:
        return yaml.safe_load(f)


def build_model(dev, steps):
    import set_amd  # noqa: F401
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    hp = load_hparams()
    hp["timesteps"] = steps
    torch.manual_seed(1234)
    model = GaussianDiffusion(list(range(80)), M, DiffNet(M, hp), timesteps=steps, time_scale=1, loss_type="l1",
                              spec_min=[], spec_max=[], hp=hp)
    # random-init weights of the architecture; the reference zero-inits output_projection.weight (diffnet.py:108),
    # give it real values so x0 depends on the network
    torch.nn.init.normal_(model.denoise_fn.output_projection.weight, std=1.0 / 16.0)
    return model.to(dev).eval()


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY.md 8(d) "CPU baseline timing"): the oracle on the same synthetic workload, bounded samples
# ------------------------------------------------------------------------------------------------------------------
def _cpu_steps(O, W, tab, cond, n_timed, seed=0, gpu_model=None):
    """1 warm-up + n_timed DiffNet+posterior steps on the rows of `cond`; returns (mean seconds per step, parity)."""
    B = cond.shape[0]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, M, T, generator=g)
    times, parity = [], None
    for k in range(1 + n_timed):
        tt = torch.full((B,), DIFF_STEPS - 1 - k, dtype=torch.long)
        eps = torch.randn(B, 1, M, T, generator=g)
        t0 = time.perf_counter()
        x0 = O.diffnet_forward(W, x, tt, cond)
        if parity is None and gpu_model is not None:
            # the checker's by-product: the same DiffNet pass on the GPU path at the full benchmark size
            # (this is iteration 0, the warm-up, which is not part of the timed mean)
            dev = next(gpu_model.parameters()).device
            with torch.no_grad():
                x0_gpu = gpu_model.denoise_fn(x.to(dev), tt.to(dev), cond.to(dev).contiguous())
            parity = float((x0_gpu.cpu().double() - x0.double()).abs().max())
        x = O.q_posterior_sample(tab, x0, x, tt, eps)
        times.append(time.perf_counter() - t0)
    return sum(times[1:]) / n_timed, parity


def host_cores():
    """What this process may really use: the affinity set (cpuset) and, when the container runs under a CFS quota, the quota in cores --
    os.cpu_count() sees neither (round-5 review: 256 "cpus" whose 32-process leg ran 5 x slower than one 16-thread process)."""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            a, b = f.read().split()
            quota = None if a == "max" else float(a) / float(b)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                a, b = float(f.read()), float(g.read())
                quota = None if a <= 0 else a / b
        except (OSError, ValueError):
            pass
    usable = len(aff) if quota is None else max(1, min(len(aff), int(quota)))
    return {"os_cpu_count": os.cpu_count() or 1, "affinity": aff, "cgroup_quota_cores": quota, "usable": usable}


def _cpu_mp_worker(rank, W, cond, threads, barrier, q, cores):
    """One of P concurrent CPU processes (all usable host cores busy), pinned to its own disjoint core range: 1 warm-up + 2 timed steps on
    its own 4 utterances."""
    if cores:
        try:
            os.sched_setaffinity(0, cores)  # threads the OpenMP pool creates later inherit the mask
        except OSError:
            pass
    torch.set_num_threads(threads)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass
    torch.set_grad_enabled(False)
    from oracle import oracle as O
    tab, _ = O.diffusion_tables(DIFF_STEPS)
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(cond.shape[0], 1, M, T, generator=g)
    times = []
    for k in range(3):
        if k == 1:
            barrier.wait()
        tt = torch.full((cond.shape[0],), DIFF_STEPS - 1 - k, dtype=torch.long)
        eps = torch.randn(cond.shape[0], 1, M, T, generator=g)
        t0 = time.perf_counter()
        x0 = O.diffnet_forward(W, x, tt, cond)
        x = O.q_posterior_sample(tab, x0, x, tt, eps)
        times.append(time.perf_counter() - t0)
    q.put((rank, times[1:]))


def cpu_all_cores(W, cond, threads, hc):
    """Throughput with EVERY usable host core busy: P = usable // threads processes, `threads` threads each, each process pinned to its own
    core range of the affinity set (OMP_NUM_THREADS set before the spawn), 4 utterances per process -- what a CPU deployment of the
    reference would do for throughput; one PyTorch process does not scale past a few tens of threads on this workload."""
    import torch.multiprocessing as mp
    usable = hc["usable"]
    P = max(1, min(usable // threads, 32))
    pin = hc["cgroup_quota_cores"] is None or len(hc["affinity"]) <= usable  # under a CFS quota the cores are shared: do not pin
    ranges = [hc["affinity"][r * threads:(r + 1) * threads] if pin else None for r in range(P)]
    ctx = mp.get_context("spawn")
    q, barrier = ctx.Queue(), ctx.Barrier(P)
    Wd = {k: v for k, v in W.items() if k.startswith("denoise_fn.")}
    for v in Wd.values():
        v.share_memory_()
    c4 = cond[:4].contiguous().share_memory_()
    old = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ["OMP_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = str(threads)
    try:
        procs = [ctx.Process(target=_cpu_mp_worker, args=(r, Wd, c4, threads, barrier, q, ranges[r])) for r in range(P)]
        for p in procs:
            p.start()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    res, deadline = [], time.time() + 300
    import queue as _queue
    while len(res) < P:
        try:
            res.append(q.get(timeout=2))
        except _queue.Empty:
            if time.time() > deadline or any(p.exitcode not in (None, 0) for p in procs):
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise RuntimeError("a CPU baseline worker died or timed out")
    for p in procs:
        p.join(timeout=60)
    step = max(sum(t) / len(t) for _, t in res)  # the slowest process bounds the aggregate
    return {"value": P * 4 * T / (DIFF_STEPS * step), "unit": "mel-frames/s", "cores": P * threads, "processes": P,
            "threads_per_process": threads, "pinned": bool(pin),
            "sample": "%d processes x %d threads%s, 4 utterances each, 2 timed DiffNet+posterior steps after a common "
                      "barrier, scaled to %d steps; slowest process s/step=%.3f"
                      % (P, threads, " pinned to disjoint core ranges" if pin else "", DIFF_STEPS, step)}


def cpu_baseline(model, inp, protocol="full"):
    """`value` = the CPU's best: conditioner once + timed DiffNet+posterior steps at B=32, T=800 scaled to 100 steps,
    at the best thread count of a sweep over {8, 16, 32, 64, all} (the sweep itself runs on the first 8 utterances).
    Also reported: the single-thread figure (the reference pins OMP_NUM_THREADS=1, tasks/run.py:3), one COMPLETE
    100-step run at B=4 as a cross-check of the scaling, and HiFi-GAN V1 at B=4."""
    from oracle import oracle as O
    W = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cpu_in = {k: v.cpu() for k, v in inp.items()}
    tab, _ = O.diffusion_tables(DIFF_STEPS)
    hc = host_cores()
    ncpu = hc["usable"]  # affinity set / cgroup quota, not os.cpu_count()
    torch.set_num_threads(min(ncpu, 32))
    t0 = time.perf_counter()
    ret, cond = O.conditioner(W, cpu_in["txt_tokens"], cpu_in["time_mel_masks"], cpu_in["mel2ph"],
                              cpu_in["spk_embed"], cpu_in["ref_mels"], cpu_in["f0"], cpu_in["uv"])
    t_cond = time.perf_counter() - t0
    B = cond.shape[0]
    sweep = {}
    settings = sorted({n for n in (8, 16, 32, 64, ncpu) if n <= ncpu}) if protocol != "quick" else [min(ncpu, 32)]
    for n in settings:
        torch.set_num_threads(n)
        s, _ = _cpu_steps(O, W, tab, cond[:8].contiguous(), 2)
        sweep[n] = 8 * T / (DIFF_STEPS * s)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t_step, parity = _cpu_steps(O, W, tab, cond, 3, gpu_model=model)
    total = t_cond + DIFF_STEPS * t_step
    out = {"value": B * T / total, "unit": "mel-frames/s", "cores": best, "host_cpus": hc["os_cpu_count"],
           "host_cpus_affinity": len(hc["affinity"]), "host_cgroup_quota_cores": hc["cgroup_quota_cores"], "host_cpus_usable": ncpu,
           "kind": "port",
           "parity_max_abs_dx0_one_pass": parity,
           "thread_sweep_frames_per_s_B8": {str(k): v for k, v in sweep.items()},
           "sample": "conditioner once + 3 timed DiffNet+posterior steps (after 1 warm-up) at B=%d,T=%d with %d threads "
                     "(best of a sweep over %s on 8 utterances), scaled to %d steps; s/step=%.3f, conditioner s=%.3f"
                     % (B, T, best, settings, DIFF_STEPS, t_step, t_cond)}
    if protocol == "full":
        # single thread, as the reference runs (OMP_NUM_THREADS=1): 2 utterances, 1 warm-up + 2 timed steps
        torch.set_num_threads(1)
        s1, _ = _cpu_steps(O, W, tab, cond[:2].contiguous(), 2)
        out["single_thread"] = {"value": 2 * T / (DIFF_STEPS * s1), "unit": "mel-frames/s", "cores": 1,
                                "sample": "2 timed steps at B=2,T=%d scaled to %d steps; s/step=%.3f" % (T, DIFF_STEPS, s1)}
        # one complete 100-step reverse loop at B=4 (cross-check of the "scale 3 steps to 100" shortcut)
        torch.set_num_threads(best)
        c4 = cond[:4].contiguous()
        g = torch.Generator().manual_seed(7)
        x = torch.randn(4, 1, M, T, generator=g)
        t0 = time.perf_counter()
        for i in reversed(range(DIFF_STEPS)):
            tt = torch.full((4,), i, dtype=torch.long)
            x0 = O.diffnet_forward(W, x, tt, c4)
            x = O.q_posterior_sample(tab, x0, x, tt, torch.randn(4, 1, M, T, generator=g))
        t100 = time.perf_counter() - t0
        out["full_100_steps_B4"] = {"value": 4 * T / t100, "unit": "mel-frames/s", "cores": best, "seconds": t100}
        # HiFi-GAN V1 generator forward at B=4, T=800 (the vocoder half of BASELINE configs[3])
        from oracle import weights as Wt
        Wg = Wt.seeded_weights(Wt.load_manifest("hifigan_v1"), 21)
        mel = cpu_in["ref_mels"][:4].transpose(1, 2).contiguous()
        O.hifigan_forward(Wg, Wt.HIFIGAN_V1, mel[:1, :, :64])  # warm-up
        t0 = time.perf_counter()
        O.hifigan_forward(Wg, Wt.HIFIGAN_V1, mel)
        th = time.perf_counter() - t0
        out["hifigan_v1_B4"] = {"value": 4 * T / th, "unit": "mel-frames/s", "cores": best, "seconds": th}
        # every host core busy: P processes x `best` threads
        try:
            out["all_cores_multiprocess"] = cpu_all_cores(W, cond, min(best, 8), hc)
        except Exception as e:  # a reported baseline must not take the benchmark down
            out["all_cores_multiprocess"] = {"error": repr(e)}
        # `value` is the CPU's best over everything measured above
        cands = [("one process, B=32 (scaled from 3 steps)", out["value"], best),
                 ("one process, complete 100 steps at B=4", out["full_100_steps_B4"]["value"], best)]
        if "value" in out["all_cores_multiprocess"]:
            cands.append(("all cores, %d processes" % out["all_cores_multiprocess"]["processes"],
                          out["all_cores_multiprocess"]["value"], out["all_cores_multiprocess"]["cores"]))
        name, val, cores = max(cands, key=lambda c: c[1])
        am = out["all_cores_multiprocess"]
        if "value" in am and am["value"] < max(c[1] for c in cands[:2]):
            # every usable core busy must not lose to one process; when it does, say what the box looked like
            am["note"] = ("aggregate below the one-process figure: %d processes x %d threads on %d usable cores (affinity %d, cgroup quota %s, "
                          "os.cpu_count %d) -- the cores are shared or throttled; `value` stays the best measured figure"
                          % (am["processes"], am["threads_per_process"], ncpu, len(hc["affinity"]), hc["cgroup_quota_cores"], hc["os_cpu_count"]))
        out["one_process_B32"] = {"value": out["value"], "cores": best}
        out["value"], out["cores"], out["best_of"] = val, cores, name
    torch.set_num_threads(min(ncpu, 64))
    return out


# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def stub_device():
    """Test hook (tests/test_dist_gloo.py): SET_AMD_BENCH_STUB_DEVICE=cpu runs either mode with NO kernels -- a toy torch-CPU
    module stands in for the task / the model, gloo for RCCL -- so that the self-launch, rendezvous, rank-0 broadcast, sharding,
    hook-launched bucket all-reduce (train), barrier / max-over-ranks timing, the all-gather of per-rank times and the rank-0 JSON
    line can be driven end to end on a box without GPUs.  The line it prints says so ("device": "cpu-stub", "value" is not a
    measurement of this framework)."""
    return os.environ.get("SET_AMD_BENCH_STUB_DEVICE", "") == "cpu"


class _StubInferModel:
    """Stands in for GaussianDiffusion when SET_AMD_BENCH_STUB_DEVICE=cpu (`--mode infer`): same call signature and the same keys in
    the returned dict as the real forward(infer=True, want_layer_spans=True); the "mel" is a function of the rank's OWN shard so that
    the test can see the utterance sharding."""

    class _Dn:
        dilation_cycle_length = 1

    denoise_fn = _Dn()

    def __call__(self, txt_tokens, time_mel_masks, mel2ph, spk_embed, ref_mels, f0, uv, infer=True, seed=0, want_layer_spans=False):
        t0 = time.perf_counter()
        mel = torch.tanh(ref_mels * (1.0 - time_mel_masks)) + 1e-3 * float(seed % 7)
        time.sleep(0.002 * ref_mels.shape[0] / B_PER_GPU)
        ms = 1e3 * (time.perf_counter() - t0)
        ret = {"mel_out": mel}
        if want_layer_spans:
            ret.update(layer_span_ms=[ms / DIFF_STEPS] * DIFF_STEPS, loop_ms=ms, n_groups=1, persistent=1)
        return ret


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize()


def maybe_self_launch(args):
    """`python bench.py --gpus N` from a bare shell: start N ranks (one per GPU) and hand over to them."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    n_dev = args.gpus if stub_device() else torch.cuda.device_count()
    if n_dev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- refusing to report a %d-GPU number from fewer "
                         "devices" % (args.gpus, n_dev, args.gpus))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def dist_facts(dev, world, elapsed):
    """Proof that the ranks really talk: world size as the process group sees it, an all-reduce that every rank must
    join (sum of rank+1), and every rank's own elapsed time."""
    import torch.distributed as dist
    me = device_identity(dev)
    if world == 1:
        return {"rccl_ranks": 1, "backend": None, "per_rank_s": [elapsed], "devices": [me]}
    one = torch.tensor([float(dist.get_rank() + 1)], device=dev)
    dist.all_reduce(one)
    assert int(one.item()) == world * (world + 1) // 2, "all-reduce did not reach every rank"
    per = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(per, torch.tensor([elapsed], dtype=torch.float64, device=dev))
    # which physical device every rank ran on (a first multi-GPU run must show N DIFFERENT devices, not N ranks on one)
    devs = [None] * world
    dist.all_gather_object(devs, me)
    ids = [d.get("uuid") or d.get("pci") for d in devs]
    return {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "per_rank_s": [float(p.item()) for p in per],
            "devices": devs, "distinct_devices": len(set(ids)) if all(ids) else None}


def overlap_frac(bucketer):
    if not getattr(bucketer, "enabled", False) or not bucketer.buckets:
        return None
    tot = sum(e - s0 for s0, e in bucketer.buckets)
    hook = sum(bucketer.buckets[b][1] - bucketer.buckets[b][0] for b, how in getattr(bucketer, "launch_log", []) if how == "hook")
    return hook / tot if tot else None


def device_identity(dev):
    """{"rank", "local_device", "name", "uuid", "pci", "hip_visible_devices"} of the device this rank runs on."""
    out = {"rank": int(os.environ.get("RANK", 0)), "local_device": str(dev), "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES")}
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        out["name"] = pr.name
        u = getattr(pr, "uuid", None)
        out["uuid"] = str(u) if u is not None else None
        out["pci"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
        out["n_cu"] = pr.multi_processor_count
    return out


def run_infer(args, rank, world, dev):
    from set_amd import _lib, parallel
    from set_amd.synthetic import synthetic_inputs
    torch.set_grad_enabled(False)
    stub = stub_device()
    model = _StubInferModel() if stub else build_model(dev, DIFF_STEPS)
    # global batch = 32 per rank; every rank takes its utterances r::world of the same synthetic batch
    full = synthetic_inputs(B_PER_GPU * world, T, T_TXT, seed=1234)
    inp = {k: v.to(dev) for k, v in parallel.shard_batch(full, rank, world).items()}

    def step(seed, spans=False):
        return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"],
                     inp["f0"], inp["uv"], infer=True, seed=seed, want_layer_spans=spans)

    for w in range(args.warmup):
        step(1000 + w)
    _sync(dev)
    parallel.barrier()
    spans, loop_ms, groups = [], [], 1
    t0 = time.perf_counter()
    for k in range(args.steps):
        ret = step(k, spans=True)
        spans.extend(ret["layer_span_ms"])
        loop_ms.append(ret["loop_ms"])
        groups = ret["n_groups"]
    _sync(dev)
    elapsed = time.perf_counter() - t0
    parallel.barrier()
    t_max = parallel.max_over_ranks(elapsed, device=dev if world > 1 else "cpu")
    assert torch.isfinite(ret["mel_out"]).all()
    assert ret["mel_out"].shape[0] == B_PER_GPU, "this rank did not get %d utterances of the global batch" % B_PER_GPU
    facts = dist_facts(dev, world, elapsed)
    # every rank's shard is r::world of ONE global batch (speech_base.py:128-131): the first utterance ids add up accordingly
    first_ids = parallel.sum_over_ranks(float(rank), device=dev if world > 1 else "cpu")
    assert first_ids == world * (world - 1) / 2

    frames = B_PER_GPU * world * T * args.steps
    value = frames / t_max
    # Dominant kernel.  Default path: ONE persistent layer-stack launch per denoise step runs all L residual layers
    # (task queue over (layer, tile)); SET_AMD_PERSISTENT=0: L diffnet_layer_kernel launches per step.
    #   launch_ms = mean duration of one launch, from hipEvent pairs on the launch stream over the timed region
    #   achieved  = algorithmic FLOPs of one launch / launch_ms  (x concurrent launches if utterance groups > 1)
    persistent = bool(ret.get("persistent", 0))
    from set_amd import ops
    variant = ops.stack_variant(B_PER_GPU // groups, T, model.denoise_fn.dilation_cycle_length)
    stack_kernel = ops.STACK_VARIANT_NAMES[variant]
    # MFMA work actually issued per algorithmic FLOP, and the pipe it is issued on:
    #   Winograd kernel (fp32 pipe): 3/4 (k=3 conv as F(2,3): 512x512 instead of 512x768 MACs per frame, plus the 1x1 conv)
    #   split-operand kernels (16-bit pipe): every fp32 product = 6 bf16 x bf16 (three pieces) or 3 fp16 x fp16 (two pieces)
    #   MFMA products with fp32 accumulation -- fp32-equivalent results on the 2.5 PFLOP/s pipe
    executed_ratio, pipe_peak, pipe = 1.0, PEAK_F32_MFMA_TFLOPS, "v_mfma_f32_32x32x2_f32"
    if persistent and variant == 2:
        executed_ratio = (512 * 512 + 512 * 256) / (512 * 768 + 512 * 256)
    elif persistent and variant == 4:
        executed_ratio, pipe_peak, pipe = 6.0, PEAK_BF16_MFMA_TFLOPS, "v_mfma_f32_32x32x16_bf16"
    elif persistent and variant == 5:
        executed_ratio, pipe_peak, pipe = 3.0, PEAK_BF16_MFMA_TFLOPS, "v_mfma_f32_32x32x16_f16"
    x3w = int(ops.stack_x3_winograd(B_PER_GPU // groups, T, model.denoise_fn.dilation_cycle_length)) if persistent and variant == 5 else 0
    x3w_name = "diffnet_stack_x3v_kernel<%d>" % x3w if x3w else None  # x3w = column blocks (32 frames) per tile: 2 or 3
    if x3w:  # round 6: GEMM 1 of the two-piece fp16 kernel in its Winograd F(2,3) form: 3/4 of the layer's MFMAs are issued
        stack_kernel = "diffnet_stack_x3v_kernel<%d> (SplitF16x2, Winograd F(2,3) form of GEMM 1, %d-frame tiles on v_mfma_f32_16x16x32_f16)" % (x3w, 32 * x3w)
        executed_ratio = 3.0 * (512 * 512 + 512 * 256) / (512 * 768 + 512 * 256)
        pipe = "v_mfma_f32_16x16x32_f16 (GEMM 1) + v_mfma_f32_32x32x16_f16 (GEMM 2)"
    split_operands = persistent and variant in (4, 5)
    layers_per_launch = L if persistent else 1
    launch_ms = sum(spans) / len(spans) / (L // layers_per_launch)
    flop_per_launch = FLOP_PER_FRAME_LAYER * (B_PER_GPU / groups) * T * layers_per_launch
    bytes_per_launch = BYTES_PER_FRAME_LAYER * (B_PER_GPU / groups) * T * layers_per_launch
    ach_tflops = groups * flop_per_launch / (launch_ms * 1e-3) / 1e12
    ach_wall = FLOP_PER_FRAME_LAYER * B_PER_GPU * T * L * DIFF_STEPS * len(loop_ms) / (sum(loop_ms) * 1e-3) / 1e12
    ach_gbs = groups * bytes_per_launch / (launch_ms * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel from the PMC passes (tools/sessions/gpu_pmc_r04.sh: separate --pmc runs, 2 x FETCH_SIZE +
    # WRITE_SIZE per MI355X_MICROARCH.md).  Counters cannot be collected inside this process, so the figure is the one measured
    # on the SAME kernel sources: the file carries their sha256 and the figure is withheld (null) when the sources changed since.
    traffic, traffic_note, pmc = None, None, None
    tfile = os.path.join(ROOT, "profiles", "r06_pmc_x3.json")
    if not os.path.exists(tfile):
        tfile = os.path.join(ROOT, "profiles", "r05_pmc_x3.json")
    tname = os.path.relpath(tfile, ROOT)
    if os.path.exists(tfile) and persistent and variant in (4, 5):
        import hashlib
        with open(tfile) as f:
            tj = json.load(f)
        h = hashlib.sha256()
        for src in tj.get("kernel_sources", []):
            with open(os.path.join(ROOT, src), "rb") as f:
                h.update(f.read())
        ent = tj.get(x3w_name if x3w else "diffnet_stack_x3_kernel<%s>" % ("SplitF16x2" if variant == 5 else "SplitBf16x3"))
        if ent is not None and h.hexdigest() == tj.get("kernel_source_sha256"):
            traffic, pmc = ent["traffic_bytes"], ent
            traffic_note = "PMC passes of this kernel build (%s, source sha256 matches)" % tname
        else:
            traffic_note = "withheld: the kernel sources changed since the PMC passes in %s" % tname
    extras = rank == 0 and world == 1 and not stub  # the comparison lines and sub-benchmarks of the 1-GPU default run
    out = {
        "metric": "diffusion mel-frames/s (100-step p_sample, B=32/GPU, T=800)" if not stub else
                  "STUB (toy torch-CPU model over gloo: exercises bench.py's launch / sharding / timing path, measures nothing of this framework)",
        **({"device": "cpu-stub", "stub_shard_checksum": float(ret["mel_out"].double().sum())} if stub else {}),
        "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 (operands carried as %s, fp32 accumulate: fp32-equivalent, see roofline.note)" % (
            "2 fp16 pieces, 3 MFMA products per term" if variant == 5 else "3 bf16 pieces, 6 MFMA products per term")
                  if split_operands else "f32"),
        "data": "synthetic",
        "config": {"workload": "FluentSpeech spec_denoiser 100-step p_sample inference (conditioner + 100 x "
                               "(DiffNet + posterior)), synthetic 80-mel T=800 batches, B=32 per GPU (shapes of "
                               "BASELINE configs[1]); on-device Philox noise",
                   "B_per_gpu": B_PER_GPU, "T": T, "T_txt": T_TXT, "denoise_steps": DIFF_STEPS,
                   "sharding": "utterances r::N, no collective"},
        "rccl_ranks": facts["rccl_ranks"], "dist_backend": facts["backend"],
        "per_rank_ms_per_step": [1e3 * s / args.steps for s in facts["per_rank_s"]],
        "devices": facts.get("devices"), "distinct_devices": facts.get("distinct_devices"),
        "collective_bytes_per_step": 0,  # inference shards by utterance: no data-path collective
        # achieved = MFMA FLOPs the dominant kernel issues per launch / its mean launch duration, against the dense peak of
        # the pipe it issues them on; algorithmic_* = the fp32 FLOPs of the layer math (conditioner
        # projection hoisted) over the same time, for comparison with rounds that ran on the fp32 pipe
        "roofline": {"kernel": stack_kernel if persistent else "diffnet_layer_kernel", "bound": "mfma",
                     "achieved": executed_ratio * ach_tflops, "peak": pipe_peak, "unit": "TFLOP/s",
                     "frac": executed_ratio * ach_tflops / pipe_peak, "traffic": traffic, "launch_ms": launch_ms,
                     # the same time priced with the ALGORITHMIC fp32 FLOPs of the layer math against the pipe the products run on
                     # (`frac` counts the 3 / 6 emulation products per fp32 product as achieved work)
                     "frac_algorithmic": ach_tflops / pipe_peak,
                     "mfma_instruction": pipe, "executed_mfma_flop_ratio": executed_ratio,
                     "flop_per_launch": flop_per_launch, "layers_per_launch": layers_per_launch,
                     "algorithmic_fp32_TFLOPs": ach_tflops,
                     "algorithmic_vs_fp32_mfma_peak": ach_tflops / PEAK_F32_MFMA_TFLOPS,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "concurrent_launches": groups,
                     "achieved_wall_lower_bound": executed_ratio * ach_wall,
                     "frac_wall_lower_bound": executed_ratio * ach_wall / pipe_peak,
                     "hbm_algorithmic_GBps": ach_gbs, "hbm_frac": ach_gbs / PEAK_HBM_GBS,
                     # the same launch priced with the bytes the PMC passes counted (L2 misses: HBM + MALL), for what the memory system actually moves
                     "hbm_measured_GBps": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None,
                     # what the bare matrix instructions of this kernel's mix sustain at the package power cap on the same chip (registers only, no
                     # operand streams; tools/hw/mfma_ceiling_w.hip, profiles/r06_ceiling_w.log: 16-wide 2,140 / 32-wide 1,875 TFLOP/s): GEMM 1 is 2/3
                     # of the executed FLOPs (16-wide), GEMM 2 1/3 (32-wide) in the Winograd kernel
                     "power_capped_mfma_ceiling_TFLOPs": (1.0 / (2.0 / 3.0 / 2140.0 + 1.0 / 3.0 / 1875.0)) if x3w else (1875.0 if variant == 5 else None),
                     "frac_of_power_capped_mfma_ceiling": (executed_ratio * ach_tflops / (1.0 / (2.0 / 3.0 / 2140.0 + 1.0 / 3.0 / 1875.0))) if x3w else
                                                          (executed_ratio * ach_tflops / 1875.0 if variant == 5 else None),
                     "traffic_note": traffic_note,
                     "traffic_over_algorithmic_bytes": traffic / bytes_per_launch if traffic else None,
                     "pmc_mfma_busy_frac_of_simd_cycles": pmc["mfma_busy_frac_of_simd_cycles"] if pmc else None,
                     "pmc_sclk_GHz": pmc["cycles_per_launch"] / (launch_ms * 1e6) if pmc and pmc.get("cycles_per_launch") else None,
                     "limit": "power: hwmon telemetry over 8 s loops of this kernel reads 1.37-1.40 kW of the 1.4 kW cap at 1.86-2.04 GHz "
                              "(2.4 GHz max; profiles/r04_power.log); a bare v_mfma_f32_32x32x16_f16 loop on the same box reaches "
                              "1,836 TFLOP/s at 1.29 kW, 1,356 TFLOP/s with this kernel's weight-fragment stream from L2 "
                              "(profiles/r04_mfma_ceiling.log); removing the kernel's ring stalls (round 4) lowered the clock from "
                              "2.00 to 1.86 GHz and the time by 1.4 %; round 6 (profiles/r06_*_ab.log, power and clock per variant): the whole "
                              "loop as one launch -2.3 %, XCD-aware task claiming -2.7 %, next-task prefetch -1.1 % -- only fewer matrix "
                              "instructions and fewer operand bytes moved it: the Winograd form of GEMM 1 (3/4 of the MFMAs) +2.8 % at 20-30 W less, then 96-frame "
                              "tiles on the 16-wide instruction (2/3 of the weight-fragment stream per frame, two accumulator sets): 1.43 against 1.61 ms "
                              "per launch on the same box (profiles/r06_x3v_ab.log; operand-stream ceilings of the three GEMM loops in "
                              "profiles/r06_ceiling_w.log: 1,139 / 1,434 / 1,373 TFLOP/s executed for 64- / 96- / 128-frame tiles)",
                     "note": (("fp32-equivalent results from 16-bit MFMAs.  f16x2: every fp32 operand is carried as TWO fp16 pieces "
                               "(11 + 11 = 22 significand bits, 2 fewer than fp32's 24) and a product is a0 b0 + a0 b1 + a1 b0 -- "
                               "the a1 b1 term (~2^-22 |ab|) is dropped; fp32 accumulation.  " if variant == 5 else
                               "fp32-equivalent results from 16-bit MFMAs.  bf16x3: every fp32 operand is the sum of THREE bf16 "
                               "pieces (8 + 8 + 8 = 24 significand bits) and a product is the six cross terms down to 2^-16; fp32 "
                               "accumulation.  ") +
                              ("GEMM 1 (the k = 3 dilated conv) runs in its Winograd F(2,3) form over output pairs on the same two-piece "
                                 "operands (tests ::test_x3v_*: error against fp64 within 2 x the fp32 chain's, the T = 800 x 100-step reference "
                                 "golden at 1.7e-6).  " if x3w else "") +
                              "Asserted: error of a layer stack against fp64 <= 1.5 x the native fp32-MFMA chain's "
                              "(tests/test_gpu_parity.py::test_x3_stack_matches_fp32_stack_and_fp64), the reference goldens incl. "
                              "the 100-step drift case at |dmel| ~1.2e-6 through this kernel (::test_full_inference_*split_operand*). "
                              "native_fp32_loop = the same loop on the fp32 MFMA pipe (SET_AMD_X3=0); bf16x3_operand_loop = the same "
                              "loop on the exact 24-bit splitting")
                     if split_operands else None},
    }
    if extras and not args.no_quality:
        out.update(quality_vs_oracle(model))
    if extras and split_operands and not args.no_native_fp32:
        out["native_fp32_loop"] = native_fp32_line(model, inp, args, step(0)["mel_out"])
    if extras and split_operands and variant == 5 and not args.no_bf16x3_loop:
        out["bf16x3_operand_loop"] = bf16x3_line(model, inp, args, step(0)["mel_out"])
    if extras and not args.no_bf16_loop:
        out["bf16_operand_loop"] = bf16_loop_line(model, inp, args, ret_f32_seed0=step(0)["mel_out"])
    if extras and args.cpu_baseline != "off":
        out["cpu_baseline"] = cpu_baseline(model, inp, args.cpu_baseline)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if extras and not args.no_secondary:
        # the other BASELINE configs, each in a process of its own (fresh hparams / allocator; this process idles meanwhile)
        del model
        torch.cuda.empty_cache()
        out["e2e_b64_vocoder"] = e2e_line()
        out["train_bf16"] = train_line("spec_denoiser", "bf16")
        out["campnet_train_bf16"] = train_line("campnet", "bf16")
        out["train_f32"] = train_line("spec_denoiser", "f32")  # the reference's default precision (egs/spec_denoiser.yaml:4), same shape as configs[1]
    return out


def quality_vs_oracle(model):
    """BASELINE's "MCD vs ref" on the headline path, over the FULL loop (round 6): the metric's own configuration -- B = 32, T = 800, all 100
    reverse steps of the shipped kernels with explicit noise -- and four of its utterances (rows 0, 9, 20, 31) through the CPU oracle (the port
    of the reference's torch path, pinned on the reference-generated goldens in tests/, the T = 800 x 100-step one included) on the same weights,
    inputs and noise: mel-level MCD (utils/eval/mcd.py:89-95 restated in oracle.mel_mcd) and max |dmel| (spec_denoiser.py:178-184)."""
    from oracle import oracle as O
    from set_amd.synthetic import synthetic_inputs
    dev = next(model.parameters()).device
    rows = [0, 9, 20, 31]
    inp = synthetic_inputs(B_PER_GPU, T, T_TXT, seed=4321)
    g = torch.Generator().manual_seed(99)
    nz_rows = [torch.randn(len(rows), 1, M, T, generator=g) for _ in range(DIFF_STEPS + 1)]
    gd = torch.Generator(device=dev).manual_seed(98)
    noises = torch.randn(DIFF_STEPS + 1, B_PER_GPU, 1, M, T, device=dev, generator=gd)
    noises[:, rows] = torch.stack(nz_rows).to(dev)
    d = {k: v.to(dev) for k, v in inp.items()}
    with torch.no_grad():
        ret = model(d["txt_tokens"], d["time_mel_masks"], d["mel2ph"], d["spk_embed"], d["ref_mels"], d["f0"], d["uv"], infer=True,
                    noises=noises)
    del noises
    W = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    sub = {k: v[rows].contiguous() for k, v in inp.items()}
    t0 = time.perf_counter()
    oret = O.gaussian_diffusion_infer(W, DIFF_STEPS, sub, nz_rows)
    dt = time.perf_counter() - t0
    a, b = ret["mel_out"][rows].float().cpu().numpy(), oret["mel_out"].float().numpy()
    return {"mcd_vs_oracle": max(O.mel_mcd(a[i], b[i]) for i in range(len(rows))), "max_abs_dmel_vs_oracle": float(abs(a - b).max()),
            "quality_sample": "B=%d x T=%d x %d explicit-noise reverse steps on the shipped kernels (the metric's configuration); rows %s "
                              "against the CPU oracle's complete %d-step loop (%.1f s of CPU)" % (B_PER_GPU, T, DIFF_STEPS, rows, DIFF_STEPS, dt),
            "integers_equal_vs_oracle": bool(torch.equal(ret["pitch"][rows].cpu(), oret["pitch"]) and
                                             torch.equal(ret["masked_dur"][rows].cpu(), oret["masked_dur"]))}


def _sub_bench(argv, timeout=420):
    """Run this script in a process of its own and parse its JSON line."""
    cmd = [sys.executable, os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
    return json.loads(lines[-1])


def train_line(model, dtype):
    """BASELINE configs[1] (spec_denoiser, B=32, T=800, bf16) / configs[4] per GPU (CampNet, B=16, T=800): `bench.py --mode train` in a
    sub-process."""
    d = _sub_bench(["--mode", "train", "--model", model, "--dtype", dtype, "--steps", "20", "--warmup", "5"])
    if "error" in d:
        return d
    out = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
           "host_enqueue_ms_per_step": d["host_enqueue_ms_per_step"], "loss": d["loss"],
           "allreduce_bytes_per_step_at_N_ranks": 4 * d.get("grad_elems_exchanged", 0),
           "roofline": d["roofline"], "whole_step": d.get("whole_step"), "launches_per_step": d.get("launches_per_step")}
    return out


def e2e_line():
    """BASELINE configs[3]: 100-step batched inference + HiFi-GAN V1 vocoder at B = 64, T = 800 on one GPU (tools/e2e_bench.py)."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py")]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}
    return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "diffusion_ms": d["diffusion_ms"], "vocoder_ms": d["vocoder_ms"],
            "wav_shape": d["wav_shape"], "finite": d["finite"], "dtype": "f32 (diffusion: split-operand kernel; vocoder: split-operand convs)",
            "vocoder_config": "assumed HiFi-GAN V1 (the reference's config.yaml is an external download)"}


def native_fp32_line(model, inp, args, mel_split):
    """The same 100-step loop with the residual layers on the fp32 MFMA pipe (the Winograd persistent kernel, round 1's
    headline path; SET_AMD_X3=0) and the difference between the two outputs on the same inputs and Philox noise."""
    os.environ["SET_AMD_X3"] = "0"
    try:
        def step(seed, spans=False):
            return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"],
                         inp["f0"], inp["uv"], infer=True, seed=seed, want_layer_spans=spans)
        step(1000)
        torch.cuda.synchronize()
        spans = []
        t0 = time.perf_counter()
        for k in range(args.steps):
            spans.extend(step(k, spans=True)["layer_span_ms"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        mel32 = step(0)["mel_out"]
    finally:
        del os.environ["SET_AMD_X3"]
    launch_ms = sum(spans) / len(spans)
    tfl = FLOP_PER_FRAME_LAYER * B_PER_GPU * T * L / (launch_ms * 1e-3) / 1e12
    return {"value": B_PER_GPU * T / dt, "unit": "mel-frames/s", "ms_per_step": 1e3 * dt, "dtype": "f32",
            "kernel": "diffnet_stack_wino_kernel", "launch_ms": launch_ms, "algorithmic_fp32_TFLOPs": tfl,
            "frac_of_fp32_mfma_peak": tfl / PEAK_F32_MFMA_TFLOPS,
            "max_abs_dmel_split_vs_native": float((mel_split.float() - mel32.float()).abs().max())}


def bf16x3_line(model, inp, args, mel_f16x2):
    """The same 100-step loop with every fp32 operand carried as THREE bf16 pieces (8 + 8 + 8 = all 24 significand bits, six
    MFMA products per term, fp32 range; SET_AMD_SPLIT_OPERAND=bf16x3) -- the exact splitting, twice the matrix-pipe work of
    the headline's two-piece fp16 one -- and the difference between the two outputs on the same inputs and Philox noise."""
    os.environ["SET_AMD_SPLIT_OPERAND"] = "bf16x3"
    try:
        def step(seed, spans=False):
            return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"],
                         inp["f0"], inp["uv"], infer=True, seed=seed, want_layer_spans=spans)
        step(1000)
        torch.cuda.synchronize()
        spans = []
        t0 = time.perf_counter()
        for k in range(args.steps):
            spans.extend(step(k, spans=True)["layer_span_ms"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        mel3 = step(0)["mel_out"]
    finally:
        del os.environ["SET_AMD_SPLIT_OPERAND"]
    launch_ms = sum(spans) / len(spans)
    tfl = FLOP_PER_FRAME_LAYER * B_PER_GPU * T * L / (launch_ms * 1e-3) / 1e12
    return {"value": B_PER_GPU * T / dt, "unit": "mel-frames/s", "ms_per_step": 1e3 * dt,
            "dtype": "f32 (operands carried as 3 bf16 pieces, 6 MFMA products per term, fp32 accumulate)",
            "kernel": "diffnet_stack_x3_kernel<SplitBf16x3>", "launch_ms": launch_ms, "algorithmic_fp32_TFLOPs": tfl,
            "executed_bf16_mfma_TFLOPs": 6.0 * tfl, "frac_of_bf16_mfma_peak": 6.0 * tfl / PEAK_BF16_MFMA_TFLOPS,
            "max_abs_dmel_bf16x3_vs_f16x2": float((mel3.float() - mel_f16x2.float()).abs().max())}


def bf16_loop_line(model, inp, args, ret_f32_seed0):
    """NOT the headline and NOT the parity path: the same 100-step loop with bf16 MFMA operands in the residual layers
    (one fused launch per layer, conditioner projection inside the layer GEMM, fp32 x / skip / cond in HBM, fp32
    accumulation; csrc/diffnet_bf16.hip).  Reported because BASELINE's north_star asks for the HBM-roofline fraction of
    the DenoiseNet inner loop, which fp32 MFMA caps at 7.7 %: here the layers stream
    4,864 B per frame and layer (x in 1024 + cond 768 + x out 1024 + skip r/w 2048).  Quality: mel-level MCD and max |dmel|
    against the fp32 path on the same inputs and the same Philox noise."""
    from oracle import oracle as O
    from set_amd import ops
    ops.set_compute_dtype("bf16")
    try:
        def step(seed, spans=False):
            return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"],
                         inp["f0"], inp["uv"], infer=True, seed=seed, want_layer_spans=spans)
        step(1000)
        torch.cuda.synchronize()
        spans = []
        t0 = time.perf_counter()
        for k in range(args.steps):
            ret = step(k, spans=True)
            spans.extend(ret["layer_span_ms"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        mel16 = step(0)["mel_out"]
    finally:
        ops.set_compute_dtype("f32")
    span_ms = sum(spans) / len(spans)  # the L layer launches of one denoise step
    bytes_step = 4864.0 * B_PER_GPU * T * L
    a, b = mel16.float().cpu().numpy(), ret_f32_seed0.float().cpu().numpy()
    mcd = max(O.mel_mcd(a[i], b[i]) for i in range(a.shape[0]))
    gbs = bytes_step / (span_ms * 1e-3) / 1e9
    tfl = (FLOP_PER_FRAME_LAYER + 2 * 512 * 192) * B_PER_GPU * T * L / (span_ms * 1e-3) / 1e12
    from set_amd import _lib
    fuse = int(_lib.lib().set_diffnet_layers_bf16_plan(B_PER_GPU, T, L, 1))
    # HBM bytes per group launch from the PMC passes of THESE kernel sources (tools/sessions/gpu_pmc_bf16_layers.sh; sha256-checked like the headline's)
    traffic, traffic_note, pmc = None, "no PMC file", None
    tfile = os.path.join(ROOT, "profiles", "r06_pmc_bf16_layers.json")
    if not os.path.exists(tfile):
        tfile = os.path.join(ROOT, "profiles", "r05_pmc_bf16_layers.json")
    if os.path.exists(tfile):
        import hashlib
        with open(tfile) as f:
            tj = json.load(f)
        h = hashlib.sha256()
        for src in tj.get("kernel_sources", []):
            with open(os.path.join(ROOT, src), "rb") as f:
                h.update(f.read())
        ent = tj.get("layers_per_launch_%d" % fuse)
        if ent is not None and h.hexdigest() == tj.get("kernel_source_sha256"):
            pmc = ent
            traffic = ent.get("hbm_bytes_per_launch")
            traffic_note = "PMC passes of this kernel build (%s, source sha256 matches); per launch of %d layers" % (os.path.relpath(tfile, ROOT), fuse)
        else:
            traffic_note = "withheld: the kernel sources changed since the PMC passes in %s" % os.path.relpath(tfile, ROOT)
    return {"note": "opt-in bf16 MFMA operands in the residual layers; not the parity path, not the headline",
            "value": B_PER_GPU * T / dt, "unit": "mel-frames/s", "ms_per_step": 1e3 * dt, "dtype": "bf16 operands, f32 accumulate",
            "mcd_vs_f32_path": mcd, "max_abs_dmel_vs_f32_path": float(abs(a - b).max()),
            "roofline": {"kernel": "diffnet_layers_t128_bf16_kernel (%d residual layers per launch, %d launches per denoise step)" % (fuse, (L + fuse - 1) // fuse)
                         if fuse >= 10 else "diffnet_layers_reg_bf16_kernel (%d residual layers per launch)" % fuse,
                         "bound": "hbm", "achieved": gbs,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "layers_span_ms": span_ms, "algorithmic_bytes_per_step": bytes_step, "layers_per_launch": fuse,
                         "mfma_TFLOPs": tfl, "mfma_frac_of_bf16_peak": tfl / PEAK_BF16_MFMA_TFLOPS,
                         "pmc_mfma_busy_frac_of_simd_cycles": pmc.get("mfma_busy_frac_of_simd_cycles") if pmc else None,
                         "pmc_sclk_GHz": pmc.get("sclk_GHz") if pmc else None,
                         "limit": "power (hwmon: 1.36-1.37 kW of the 1.4 kW cap at 1.9-2.0 GHz over a sustained loop of this kernel, "
                                  "profiles/r04_power.log): time follows the energy of a layer -- without weight-fragment, LDS-fragment and "
                                  "skip traffic the same kernel runs 28.8 instead of 40.7 us per layer (profiles/r04_t128_exp.log)"}}


# CampNet (BASELINE configs[4]), algorithmic FLOPs of one training step per mel frame at T = 800, T_txt = 100: forward MACs =
# mel-side weights (decoder_coarse 10.63 M + decoder_fine 4.54 M + mel encoder / output layers 0.13 M: every weight once per
# frame) + text encoder 5.89 M per TOKEN (x 100 / 800) + attention (6 x (2 T + 2 T_txt) x 192 per frame); x 2 FLOP x 3 (fwd + bwd)
CAMPNET_B_PER_GPU = 16
CAMPNET_TRAIN_FLOP_PER_FRAME = 3 * 2 * (15.30e6 + 5.89e6 * T_TXT / T + 6 * (2 * T + 2 * T_TXT) * 192)


class _StubTask:
    """Stands in for the task when SET_AMD_BENCH_STUB_DEVICE=cpu: two top-level modules (so that the bucketer has a module boundary
    to respect) and one the loss never reaches (so that the exchanged prefix is shorter than the buffer)."""

    def __init__(self):
        class Net(torch.nn.Module):
            unused_parameter_prefixes = ("dead.",)

            def __init__(self):
                super().__init__()
                self.front = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64))
                self.back = torch.nn.Sequential(torch.nn.Tanh(), torch.nn.Linear(64, 8))
                self.dead = torch.nn.Linear(8, 8)

            def forward(self, x):
                return self.back(self.front(x))
        self.model = Net()


def _stub_step_fn(task, opt):
    """zero_grad -> backward (bucket all-reduces launch from the autograd hooks) -> finish -> SGD on the mean gradient, in torch."""
    class Step:
        def __call__(self, sample, seed):
            opt.zero_grad()
            loss = (task.model(sample["x"]) - sample["y"]).pow(2).mean()
            loss.backward()
            world = opt.bucketer.finish()
            opt.in_step = False
            lr = opt.lr_at(opt.num_updates)
            opt.num_updates += 1
            with torch.no_grad():
                opt.flat_p.add_(opt.flat_g, alpha=-lr / world)
            return loss.detach(), {"mse": loss.detach()}, lr
    return Step()


def _train_setup(args, rank, world, dev):
    """(task, optimizer, this rank's sample, step function, utterances per GPU) of --mode train."""
    from set_amd import hparams as HP, ops, parallel, tasks
    from set_amd.synthetic import synthetic_inputs
    from set_amd.training import FlatAdamW
    if stub_device():
        torch.manual_seed(1234 + rank)
        task = _StubTask()
        opt = FlatAdamW(task.model, lr=1e-2, warmup_updates=0, bucket_mb=4e-3)
        g = torch.Generator().manual_seed(7)
        full = {"x": torch.randn(8 * world, 16, generator=g), "y": torch.randn(8 * world, 8, generator=g)}
        return task, opt, parallel.shard_batch(full, rank, world), _stub_step_fn(task, opt), 8
    campnet = args.model == "campnet"
    bpg = CAMPNET_B_PER_GPU if campnet else B_PER_GPU
    HP.hparams.clear()
    if campnet:
        with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "campnet.yaml")) as f:
            HP.hparams.update(yaml.safe_load(f))
        HP.hparams.update(binary_data_dir="", vocoder_ckpt="")
    else:
        HP.hparams.update(load_hparams())
    if args.dtype == "bf16":
        ops.set_compute_dtype("bf16")
    # every rank seeds DIFFERENTLY on purpose: the replicas must agree because of the rank-0 broadcast, not by luck
    torch.manual_seed(1234 + rank)
    if campnet:
        task = tasks.CampNetTask(80, 100, build_vocoder=False)
        task.build_model()
        with torch.no_grad():
            task.model.mask_emb.normal_(0, 0.5)
    else:
        task = tasks.SpeechDenoiserTask(build_vocoder=False)
        task.build_model()
        torch.nn.init.normal_(task.model.denoise_fn.output_projection.weight, std=1.0 / 16.0)
    task.model.to(dev).train()
    opt = FlatAdamW(task.model, lr=HP.hparams["lr"], betas=(0.9, 0.98), weight_decay=0.0, clip_grad_norm=1.0,
                    warmup_updates=8000)
    full = synthetic_inputs(bpg * world, T, T_TXT, seed=1234, pad_tail=True)
    inp = {k: v.to(dev) for k, v in parallel.shard_batch(full, rank, world).items()}
    if campnet:
        sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], spk_embed=inp["spk_embed"],
                      time_mel_masks=inp["time_mel_masks"].squeeze(-1).contiguous())
    else:
        sample = dict(txt_tokens=inp["txt_tokens"], mels=inp["ref_mels"], mel2ph=inp["mel2ph"], f0=inp["f0"], uv=inp["uv"],
                      time_mel_masks=inp["time_mel_masks"].squeeze(-1).contiguous(), spk_embed=inp["spk_embed"])
    return task, opt, sample, _EagerStep(task, opt), bpg


class _EagerStep:
    """One optimisation step of the task per call: zero_grad -> forward + losses -> backward -> (bucket all-reduce) -> clip + AdamW."""

    def __init__(self, task, opt):
        self.task, self.opt = task, opt

    def __call__(self, sample, seed):
        return self.task.training_step(sample, self.opt, seed=seed)


TRAIN_PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_train.json")
if not os.path.exists(TRAIN_PMC_FILE):
    TRAIN_PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_train.json")
TRAIN_KERNEL_SOURCES = ["speech-editing-toolkit_amd/csrc/diffnet_bf16.hip", "speech-editing-toolkit_amd/csrc/bf16.hip",
                        "speech-editing-toolkit_amd/csrc/common.h", "speech-editing-toolkit_amd/csrc/rows_sum.h",
                  "speech-editing-toolkit_amd/csrc/train.hip", "speech-editing-toolkit_amd/autograd_ops.py"]


def _profiled(key, model, dtype):
    """A figure that cannot be measured inside this process (rocprofv3 PMC / kernel-trace passes of `bench.py --mode train`,
    tools/sessions/gpu_r6_final.sh -> profiles/r06_pmc_train.json), quoted only while the sha256 of the kernel sources it was taken on matches."""
    if not os.path.exists(TRAIN_PMC_FILE):
        return None
    import hashlib
    with open(TRAIN_PMC_FILE) as f:
        tj = json.load(f)
    h = hashlib.sha256()
    for src in tj.get("kernel_sources", []):
        with open(os.path.join(ROOT, src), "rb") as f:
            h.update(f.read())
    if h.hexdigest() != tj.get("kernel_source_sha256"):
        return None
    return tj.get("%s_%s" % (model, dtype), {}).get(key)


def _train_dominant_kernel(args, step_fn, sample, dev, bpg):
    """The kernel with the largest share of a training step's GPU time, timed with hipEvents on its launch stream over 5 extra steps.
    spec_denoiser bf16: diffnet_layer_bwd_bf16_kernel (20 launches per step, HBM-bound: per frame and layer it reads dx' 1024 + dskip 1024
    + y (bf16) 1024 + dcond 768 and writes dx 1024 + dy (bf16) 1024 + d_o (bf16) 1024 + dcond 768 = 7,680 B, csrc/diffnet_bf16.hip).
    CampNet (and spec_denoiser fp32): the bf16 conv shape with the largest total time (MFMA-bound: 2 B T Cin Cout K FLOP per launch)."""
    from set_amd import autograd_ops as A, ops
    campnet = args.model == "campnet"
    A.SWEEP_EVENTS, ops.CONV_EVENTS = [], []
    try:
        for k in range(5):
            step_fn(sample, seed=5000 + k)
        _sync(dev)
        sweeps, convs = list(A.SWEEP_EVENTS), list(ops.CONV_EVENTS)
    finally:
        A.SWEEP_EVENTS, ops.CONV_EVENTS = None, None
    if sweeps and not campnet:
        per_launch_ms = sum(e0.elapsed_time(e1) / n for e0, e1, n in sweeps) / len(sweeps)
        bytes_launch = 7680.0 * bpg * T
        gbs = bytes_launch / (per_launch_ms * 1e-3) / 1e9
        ent = _profiled("diffnet_layer_bwd_bf16_kernel", "spec_denoiser", args.dtype) or {}
        return {"kernel": "diffnet_layer_bwd_bf16_kernel", "launches_per_step": L, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": ent.get("traffic_bytes"), "launch_ms": per_launch_ms,
                "algorithmic_bytes_per_launch": bytes_launch, "share_of_step_gpu_time": L * per_launch_ms,
                "traffic_note": "PMC passes of this kernel build (%s, source sha256 matches)" % os.path.relpath(TRAIN_PMC_FILE, ROOT) if ent else
                                "no PMC figure for these kernel sources"}
    if not convs:
        return None
    by = {}
    for key, e0, e1 in convs:
        d = by.setdefault(key, [0.0, 0])
        d[0] += e0.elapsed_time(e1)
        d[1] += 1
    key, (ms, n) = max(by.items(), key=lambda kv: kv[1][0])
    Cin, Cout, K, Bc, Tc, impl = key
    fl = 2.0 * Bc * Tc * Cin * Cout * K
    by_l = 4.0 * Bc * Tc * (Cin + Cout)  # fp32 activations in and out (weights are L2-resident)
    tfl = fl / (ms / n * 1e-3) / 1e12
    gbs = by_l / (ms / n * 1e-3) / 1e9
    if impl == "bf16":
        # (csrc/bf16.hip: 1x1 convs with Cin > 32 and Cout > 64 run in conv1x1_oneshot_bf16_kernel<Cin rounded up>, the rest in conv1d_bf16_kernel)
        family, peak = ("conv1x1_oneshot_bf16_kernel" if (K == 1 and Cin > 32 and Cout > 64) else "conv1d_bf16_kernel"), PEAK_BF16_MFMA_TFLOPS
    else:  # fp32 operands on v_mfma_f32_32x32x2_f32 (csrc/conv1d.hip): the 64-row kernel or the big-tile one
        family, peak = ("conv1d_mfma_v2_kernel" if impl == "mfma2" else "conv1d_mfma_kernel"), PEAK_F32_MFMA_TFLOPS
    name = "%s %d->%d k=%d (B=%d, T=%d)" % (family, Cin, Cout, K, Bc, Tc)
    ent = _profiled(family, "campnet" if campnet else "spec_denoiser", args.dtype) or {}
    # the bound follows the shape's arithmetic intensity against the ridge of the pipe (bf16: 2.5 PFLOP/s / 8 TB/s = 312 FLOP/B; fp32: 19.7)
    hbm_bound = fl / by_l < peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    return {"kernel": name, "launches_per_step": n / 5.0, "bound": "hbm" if hbm_bound else "mfma",
            "achieved": gbs if hbm_bound else tfl, "peak": PEAK_HBM_GBS if hbm_bound else peak,
            "unit": "GB/s" if hbm_bound else "TFLOP/s", "frac": gbs / PEAK_HBM_GBS if hbm_bound else tfl / peak,
            "traffic": ent.get("traffic_bytes"), "launch_ms": ms / n, "flop_per_launch": fl, "algorithmic_bytes_per_launch": by_l,
            "mfma_TFLOPs": tfl, "hbm_GBps": gbs, "share_of_step_gpu_time": ms / 5.0,
            "traffic_note": "PMC passes, mean over ALL launches of this kernel family in a step, i.e. over its shapes (%s)" % os.path.relpath(TRAIN_PMC_FILE, ROOT) if ent else
                            "no PMC figure for these kernel sources"}


def run_train(args, rank, world, dev):
    """--model spec_denoiser: BASELINE configs[1] (1 GPU) / configs[2] (8 GPUs, B=256 global); --model campnet: configs[4]
    (CampNet masked-mel transformer, B=16 per GPU = egs/campnet.yaml max_sentences).  One optimisation step of the task
    under the reference's DDP set-up (barrier, rank-0 broadcast, barrier; bucketed gradient all-reduce from autograd hooks)."""
    from set_amd import parallel
    campnet = args.model == "campnet"
    task, opt, sample, step_fn, bpg = _train_setup(args, rank, world, dev)
    bcast_bytes = parallel.configure_ddp(task.model, opt)  # barrier, rank-0 broadcast, barrier
    if world > 1:  # replicas identical now?
        chk = opt.flat_p.double().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        assert float(lo) == float(hi), "parameter broadcast left the replicas different"
    for w in range(args.warmup):
        step_fn(sample, seed=100 + w)
    _sync(dev)
    parallel.barrier()
    exposed, reduced, enqueue = 0.0, 0, 0.0
    t0 = time.perf_counter()
    for k in range(args.steps):
        t1 = time.perf_counter()
        total, parts, lr = step_fn(sample, seed=k)
        enqueue += time.perf_counter() - t1  # host time to ENQUEUE the step (no synchronisation inside)
        exposed += opt.bucketer.exposed_s
        reduced = opt.bucketer.bytes_reduced
    _sync(dev)
    elapsed = time.perf_counter() - t0
    parallel.barrier()
    t_max = parallel.max_over_ranks(elapsed, device=dev if world > 1 else "cpu")
    facts = dist_facts(dev, world, elapsed)
    assert torch.isfinite(total).all()
    same_after = True
    if world > 1:  # every rank applied the same all-reduced gradient to the same weights: the replicas must still be bit-identical
        chk = opt.flat_p.double().sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        same_after = float(lo) == float(hi)
        assert same_after, "the replicas diverged during the timed steps"
    dominant = None if stub_device() else _train_dominant_kernel(args, step_fn, sample, dev, bpg)
    n_samples = bpg * world * args.steps
    flop = (CAMPNET_TRAIN_FLOP_PER_FRAME if campnet else TRAIN_FLOP_PER_FRAME) * bpg * T * args.steps  # per rank
    peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    ach = flop / t_max / 1e12
    if campnet:
        metric = "CampNet training samples/s (B=16/GPU, T=800)"
        workload = ("CampNet masked-mel transformer training step (text encoder + coarse transformer decoder with self / "
                    "encoder-decoder attention + fine conv decoder, coarse + fine l1/ssim losses, backward, gradient all-reduce, "
                    "clip + AdamW), synthetic 80-mel T=800 batches, B=16 per GPU (BASELINE configs[4]: egs/campnet.yaml)")
    else:
        metric = "spec_denoiser training samples/s (B=32/GPU, T=800)"
        workload = ("FluentSpeech spec_denoiser training step (conditioner + one DiffNet pass + l1/ssim/dur/"
                    "pitch losses + backward + gradient all-reduce + clip + AdamW), synthetic 80-mel T=800 "
                    "batches, B=32 per GPU (BASELINE configs[1]; configs[2] at 8 GPUs = 256 global)")
    if stub_device():
        metric = "STUB (toy torch-CPU module over gloo: exercises bench.py's launch / exchange / timing path, measures nothing of this framework)"
        workload = "stub"
    return {
        "metric": metric, "value": n_samples / t_max, "unit": "samples/s", **({"device": "cpu-stub"} if stub_device() else {}),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_max / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": workload,
                   "B_per_gpu": bpg, "T": T, "T_txt": T_TXT, "sharding": "utterances r::N, gradient all-reduce (SUM, "
                   "1/N folded into AdamW) in %d buckets" % max(1, len(opt.bucketer.buckets))},
        "frames_per_s": None if stub_device() else n_samples * T / t_max, "host_enqueue_ms_per_step": 1e3 * enqueue / args.steps,
        "rccl_ranks": facts["rccl_ranks"], "dist_backend": facts["backend"], "replicas_identical_after_steps": same_after,
        "per_rank_ms_per_step": [1e3 * s / args.steps for s in facts["per_rank_s"]],
        "devices": facts.get("devices"), "distinct_devices": facts.get("distinct_devices"),
        "allreduce_bytes_per_step": reduced, "allreduce_exposed_ms_per_step": 1e3 * exposed / args.steps,
        # of the bytes exchanged in the last step: the share whose all-reduce was LAUNCHED from a gradient hook, i.e. under the rest of
        # backward (the remainder was launched by finish() after backward); with exposed_ms this says how much of the exchange was hidden
        "allreduce_launched_under_backward_frac": overlap_frac(opt.bucketer),
        "param_broadcast_bytes": bcast_bytes, "grad_elems": opt.n, "grad_elems_exchanged": opt.n_exchanged,
        "loss": float(total), "lr": lr, "losses": {k: float(v) for k, v in parts.items()},
        # roofline = the step's dominant kernel (largest share of the GPU time; hipEvents on its launch stream over extra steps right
        # behind the timed ones -- inside them, the ~100 event records per step would cost the step itself 0.4 ms of host time);
        # whole_step = the algorithmic FLOPs of the step over the step time
        "roofline": dominant,
        "whole_step": None if stub_device() else {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                                  "flop_per_step": flop / args.steps},
        "launches_per_step": None if stub_device() else _profiled("launches_per_step", "campnet" if campnet else "spec_denoiser", args.dtype),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("infer", "train"), default="infer")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32", help="train mode only: MFMA operand type")
    ap.add_argument("--cpu-baseline", choices=("full", "quick", "off"), default="full")
    ap.add_argument("--no-cpu-baseline", action="store_const", const="off", dest="cpu_baseline")
    ap.add_argument("--no-native-fp32", action="store_true", help="skip the fp32-MFMA-pipe comparison run of the same loop")
    ap.add_argument("--no-bf16-loop", action="store_true", help="skip the extra (non-headline) bf16-operand loop line")
    ap.add_argument("--no-bf16x3-loop", action="store_true", help="skip the exact three-piece splitting of the same loop")
    ap.add_argument("--no-quality", action="store_true", help="skip the MCD / max |dmel| of the shipped path against the CPU oracle")
    ap.add_argument("--no-secondary", action="store_true", help="skip the sub-benchmarks of BASELINE configs[1], [3], [4] (sub-processes)")
    ap.add_argument("--model", choices=("spec_denoiser", "campnet"), default="spec_denoiser",
                    help="train mode only: campnet = BASELINE configs[4] (B=16/GPU, T=800)")
    args = ap.parse_args()

    stub = stub_device()
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the set_amd hot path has no CPU fallback")
    maybe_self_launch(args)

    import set_amd  # noqa: F401
    from set_amd import _lib, parallel
    _lib.build()
    rank, world, local_rank = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if stub:
        dev = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    out = (run_train if args.mode == "train" else run_infer)(args, rank, world, dev)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
