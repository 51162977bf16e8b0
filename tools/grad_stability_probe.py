"""Which parameter gradients differ between two runs of the same full-size training step (B=32, T=800)?  Diagnosis tool for
tests/test_gpu_training.py::test_full_size_training_step_is_bit_stable: prints every parameter whose gradient is not bit-identical
across REPEAT runs, with the largest difference and the number of differing elements."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_training as T
dev = torch.device("cuda:0")
dtype = os.environ.get("DTYPE", "f32")
task, _ = T._train_setup(dev, 8, 18)
# TRACE=1: every set_conv1d call of a run (forward and input-gradient convs) keeps a copy of its output; the FIRST call whose output differs
# between two runs of the same state is printed with its arguments -- the kernel launch the non-determinism enters through
from set_amd import ops
trace, real_conv = [], ops.conv1d


def traced_conv(x, weight, bias=None, **kw):
    out = real_conv(x, weight, bias, **kw)
    B, Cin, T_in = x.shape
    impl = ops._pick_impl(kw.get("impl"), kw.get("T_iter") or out.shape[2], weight.Cout, Cin, weight.K, kw.get("dil", 1), kw.get("out_stride", 1),
                          kw.get("out_off", 0), kw.get("in_chan_add") is not None, False, kw.get("pad", 0))
    trace.append((dict(B=B, Cin=Cin, Cout=weight.Cout, K=weight.K, T_in=T_in, T_out=out.shape[2], dil=kw.get("dil", 1), pad=kw.get("pad", 0),
                       act=kw.get("act", "none"), res=kw.get("res") is not None, mask=kw.get("mask") is not None,
                       acc=bool(kw.get("accumulate", False)), chan_add=kw.get("in_chan_add") is not None, impl=impl), out.clone()))
    return out


if os.environ.get("TRACE", "0") == "1":
    ops.conv1d = traced_conv
runs, traces = [], []
for r in range(int(os.environ.get("REPEAT", 3))):
    del trace[:]
    T._full_size_step(dev, task, int(os.environ.get("BATCH", 32)), dtype)
    runs.append({n: p.grad.clone() for n, p in task.model.named_parameters() if p.grad is not None})
    traces.append(list(trace))
if traces[0]:
    for r in range(1, len(traces)):
        assert len(traces[r]) == len(traces[0])
        first = next((i for i, ((_, a), (_, b)) in enumerate(zip(traces[0], traces[r])) if not torch.equal(a, b)), None)
        if first is None:
            print("run 0 vs %d: all %d conv outputs bit-identical" % (r, len(traces[0])))
        else:
            cfg, a = traces[0][first]
            b = traces[r][first][1]
            d = (a - b).abs()
            nz = (d > 0).nonzero()
            print("run 0 vs %d: FIRST differing conv output is call %d of %d: %s" % (r, first, len(traces[0]), cfg))
            print("   differing %d of %d elements, max |d| %.3e (max |out| %.3e); first indices %s; rows (channels) touched: %s" % (
                int((d > 0).sum()), d.numel(), float(d.max()), float(a.abs().max()), nz[:4].tolist(),
                sorted(set(nz[:, 1].tolist()))[:12]))
            n_bad = sum(1 for (_, x), (_, y) in zip(traces[0], traces[r]) if not torch.equal(x, y))
            print("   conv calls with differing outputs: %d" % n_bad)
bad = 0
for n in runs[0]:
    for r in range(1, len(runs)):
        a, b = runs[0][n], runs[r][n]
        if not torch.equal(a, b):
            d = (a - b).abs()
            print("run 0 vs %d: %-60s shape %-18s differing %d of %d, max |d| %.3e (max |g| %.3e)" % (
                r, n, tuple(a.shape), int((d > 0).sum()), d.numel(), float(d.max()), float(a.abs().max())))
            bad += 1
print("parameters with run-to-run differences: %d" % bad)
