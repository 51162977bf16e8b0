"""Headline benchmark: diffusion mel-frames/s of one full GaussianDiffusion.forward(infer=True)
(conditioner once + 100 x (DiffNet + posterior step)) at B=32 per GPU, T=800, fp32.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic batch.  Utterances shard across ranks with NO
data-path collective (weak scaling: B=32 per rank); timing = barrier + synchronize on both sides, MAX over
ranks.  Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel =
diffnet_layer_kernel, measured with hipEvents on the launch stream inside the timed region) and, at N=1,
`cpu_baseline` (the CPU oracle = a port of the reference's torch-CPU path, timed on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import yaml  # noqa: E402

B_PER_GPU, T, T_TXT, DIFF_STEPS, M, L, C, H = 32, 800, 100, 100, 80, 20, 256, 192
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
# algorithmic work of ONE diffnet_layer_kernel launch (DESIGN.md section "roofline"):
#   k3 dilated conv 512x768 + output projection 512x256 = 524,288 MAC per frame (conditioner projection hoisted)
FLOP_PER_FRAME_LAYER = 2 * (512 * 768 + 512 * 256)
# algorithmic HBM bytes per frame per layer launch (fp32): x in 1024 + condproj 2048 + x out 1024 + skip rmw 2048
BYTES_PER_FRAME_LAYER = 1024 + 2048 + 1024 + 2048


def build_model(dev, steps):
    import set_amd  # noqa: F401
    from set_amd.diffnet import DiffNet
    from set_amd.spec_denoiser import GaussianDiffusion
    with open(os.path.join(ROOT, "speech-editing-toolkit_amd", "egs", "spec_denoiser.yaml")) as f:
        hp = yaml.safe_load(f)
    hp["timesteps"] = steps
    torch.manual_seed(1234)
    model = GaussianDiffusion(list(range(80)), M, DiffNet(M, hp), timesteps=steps, time_scale=1, loss_type="l1",
                              spec_min=[], spec_max=[], hp=hp)
    # random-init weights of the architecture; the reference zero-inits output_projection.weight (diffnet.py:108),
    # give it real values so x0 depends on the network
    torch.nn.init.normal_(model.denoise_fn.output_projection.weight, std=1.0 / 16.0)
    return model.to(dev).eval()


def cpu_baseline(model, inp, n_timed=3):
    """The oracle (port of the reference's torch CPU path) on the SAME workload, bounded sample:
    conditioner once + (1 warm-up + n_timed) DiffNet+posterior steps at B=32,T=800, scaled to 100 steps."""
    from oracle import oracle as O
    W = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cpu_in = {k: v.cpu() for k, v in inp.items()}
    tab, _ = O.diffusion_tables(DIFF_STEPS)
    t0 = time.perf_counter()
    ret, cond = O.conditioner(W, cpu_in["txt_tokens"], cpu_in["time_mel_masks"], cpu_in["mel2ph"],
                              cpu_in["spk_embed"], cpu_in["ref_mels"], cpu_in["f0"], cpu_in["uv"])
    t_cond = time.perf_counter() - t0
    B = cond.shape[0]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 1, M, T, generator=g)
    times = []
    parity = None
    for k in range(1 + n_timed):
        i = DIFF_STEPS - 1 - k
        tt = torch.full((B,), i, dtype=torch.long)
        eps = torch.randn(B, 1, M, T, generator=g)
        t0 = time.perf_counter()
        x0 = O.diffnet_forward(W, x, tt, cond)
        if parity is None:  # the checker's by-product: the same DiffNet pass on the GPU path, full benchmark size
            dev = next(model.parameters()).device
            with torch.no_grad():
                x0_gpu = model.denoise_fn(x.to(dev), tt.to(dev), cond.to(dev).contiguous())
            parity = float((x0_gpu.cpu().double() - x0.double()).abs().max())
            # (this is iteration 0, the warm-up, which is not part of the timed mean)
        x = O.q_posterior_sample(tab, x0, x, tt, eps)
        times.append(time.perf_counter() - t0)
    t_step = sum(times[1:]) / n_timed
    total = t_cond + DIFF_STEPS * t_step
    return {"value": B * T / total, "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "parity_max_abs_dx0_one_pass": parity,
            "sample": "conditioner once + %d timed DiffNet+posterior steps (after 1 warm-up) at B=%d,T=%d, scaled to "
                      "%d steps; s/step=%.3f, conditioner s=%.3f" % (n_timed, B, T, DIFF_STEPS, t_step, t_cond)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import set_amd  # noqa: F401
    from set_amd import _lib, parallel
    from set_amd.synthetic import synthetic_inputs
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the set_amd hot path has no CPU fallback")
    _lib.build()
    rank, world, local_rank = parallel.init_from_env()
    assert world == args.gpus or world == 1, (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.set_grad_enabled(False)

    model = build_model(dev, DIFF_STEPS)
    # global batch = 32 per rank; every rank takes its utterances r::world of the same synthetic batch
    full = synthetic_inputs(B_PER_GPU * world, T, T_TXT, seed=1234)
    inp = {k: v.to(dev) for k, v in parallel.shard_batch(full, rank, world).items()}

    def step(seed, spans=False):
        return model(inp["txt_tokens"], inp["time_mel_masks"], inp["mel2ph"], inp["spk_embed"], inp["ref_mels"],
                     inp["f0"], inp["uv"], infer=True, seed=seed, want_layer_spans=spans)

    for w in range(args.warmup):
        step(1000 + w)
    torch.cuda.synchronize()
    parallel.barrier()
    spans, loop_ms, groups = [], [], 1
    t0 = time.perf_counter()
    for k in range(args.steps):
        ret = step(k, spans=True)
        spans.extend(ret["layer_span_ms"])
        loop_ms.append(ret["loop_ms"])
        groups = ret["n_groups"]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    parallel.barrier()
    t_max = parallel.max_over_ranks(elapsed, device=dev if world > 1 else "cpu")
    assert torch.isfinite(ret["mel_out"]).all()

    frames = B_PER_GPU * world * T * args.steps
    value = frames / t_max
    # Dominant kernel.  Default path: ONE persistent diffnet_stack_kernel launch per denoise step runs all L residual
    # layers (task queue over (layer, 32-frame tile)); SET_AMD_PERSISTENT=0: L diffnet_layer_kernel launches per step.
    #   launch_ms = mean duration of one launch, from hipEvent pairs on the launch stream over the timed region
    #   achieved  = algorithmic FLOPs of one launch / launch_ms  (x concurrent launches if utterance groups > 1)
    persistent = bool(ret.get("persistent", 0))
    variant = _lib.lib().set_diffnet_stack_variant(B_PER_GPU // groups, T, model.denoise_fn.dilation_cycle_length, 1)
    stack_kernel = "diffnet_stack_wino_kernel" if variant == 2 else "diffnet_stack_kernel"
    # the Winograd kernel issues 3/4 of the algorithmic MACs (k=3 conv as F(2,3): 4 multiplies per output PAIR and
    # input channel instead of 6, i.e. 512x512 instead of 512x768 MACs per frame, plus the 512x256 1x1 conv)
    executed_ratio = (512 * 512 + 512 * 256) / (512 * 768 + 512 * 256) if (persistent and variant == 2) else 1.0
    layers_per_launch = L if persistent else 1
    launch_ms = sum(spans) / len(spans) / (L // layers_per_launch)
    flop_per_launch = FLOP_PER_FRAME_LAYER * (B_PER_GPU / groups) * T * layers_per_launch
    bytes_per_launch = BYTES_PER_FRAME_LAYER * (B_PER_GPU / groups) * T * layers_per_launch
    ach_tflops = groups * flop_per_launch / (launch_ms * 1e-3) / 1e12
    ach_wall = FLOP_PER_FRAME_LAYER * B_PER_GPU * T * L * DIFF_STEPS * len(loop_ms) / (sum(loop_ms) * 1e-3) / 1e12
    ach_gbs = groups * bytes_per_launch / (launch_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_bytes_per_launch.json")
    if os.path.exists(tfile):  # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py), same kernel + shape
        with open(tfile) as f:
            tj = json.load(f)
        traffic = tj.get(stack_kernel if persistent else "diffnet_layer_kernel")
    out = {
        "metric": "diffusion mel-frames/s (100-step p_sample, B=32/GPU, T=800)",
        "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "FluentSpeech spec_denoiser 100-step p_sample inference (conditioner + 100 x "
                               "(DiffNet + posterior)), synthetic 80-mel T=800 batches, B=32 per GPU (shapes of "
                               "BASELINE configs[1]); on-device Philox noise",
                   "B_per_gpu": B_PER_GPU, "T": T, "T_txt": T_TXT, "denoise_steps": DIFF_STEPS,
                   "sharding": "utterances r::N, no collective"},
        "roofline": {"kernel": stack_kernel if persistent else "diffnet_layer_kernel", "bound": "mfma",
                     "achieved": ach_tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "launch_ms": launch_ms,
                     "flop_per_launch": flop_per_launch, "layers_per_launch": layers_per_launch,
                     "executed_mfma_flop_ratio": executed_ratio,
                     "mfma_issue_frac": executed_ratio * ach_tflops / PEAK_F32_MFMA_TFLOPS,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "concurrent_launches": groups,
                     "achieved_wall_lower_bound": ach_wall, "frac_wall_lower_bound": ach_wall / PEAK_F32_MFMA_TFLOPS,
                     "hbm_algorithmic_GBps": ach_gbs, "hbm_frac": ach_gbs / PEAK_HBM_GBS},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, inp)
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
