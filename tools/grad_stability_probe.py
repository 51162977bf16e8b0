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
runs = []
for r in range(int(os.environ.get("REPEAT", 3))):
    T._full_size_step(dev, task, 32, dtype)
    runs.append({n: p.grad.clone() for n, p in task.model.named_parameters() if p.grad is not None})
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
