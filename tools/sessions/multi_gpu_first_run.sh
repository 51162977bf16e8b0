#!/bin/bash
# The FIRST thing to run on a lease with >= 2 GPUs (no such box has run this tree: DESIGN.md section 6).  In this order:
#   1. the RCCL tests that skip on a 1-GPU box (tests/test_gpu_dist.py: 2-rank gradient equivalence and bucket overlap over real RCCL),
#   2. bench.py --gpus N for N = 2 (and 4, 8 when visible), inference (no collective) and training (bucketed all-reduce): every line carries
#      per-rank device name / UUID / PCI id (`devices`, `distinct_devices` must equal N), `rccl_ranks`, `allreduce_bytes_per_step`,
#      `allreduce_exposed_ms_per_step` and `allreduce_launched_under_backward_frac`.
# Outputs under gpurun_out/multi_gpu/.  Nothing here is needed on a 1-GPU box.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"; OUT=gpurun_out/multi_gpu; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $N" | tee $OUT/summary.log
[ "$N" -lt 2 ] && { echo "needs >= 2 GPUs" | tee -a $OUT/summary.log; exit 0; }
python -m pytest tests/test_gpu_dist.py -m gpu -q -rs > $OUT/pytest_gpu_dist.log 2>&1; tail -3 $OUT/pytest_gpu_dist.log | tee -a $OUT/summary.log
for n in 2 4 8; do
  [ "$n" -gt "$N" ] && break
  python bench.py --gpus $n --steps 3 --warmup 1 > $OUT/bench_infer_$n.json 2> $OUT/bench_infer_$n.err
  python bench.py --gpus $n --mode train --dtype bf16 --steps 20 --warmup 5 > $OUT/bench_train_bf16_$n.json 2> $OUT/bench_train_bf16_$n.err
  python bench.py --gpus $n --mode train --model campnet --dtype bf16 --steps 20 --warmup 5 > $OUT/bench_campnet_bf16_$n.json 2> $OUT/bench_campnet_bf16_$n.err
  python - "$n" <<'PY' | tee -a $OUT/summary.log
import json, sys
n = sys.argv[1]
for name in ("infer", "train_bf16", "campnet_bf16"):
    try:
        d = json.loads([l for l in open("gpurun_out/multi_gpu/bench_%s_%s.json" % (name, n)) if l.startswith("{")][-1])
        print("N=%s %-13s value %.0f %s, ms/step %.2f, rccl_ranks %s, distinct devices %s, all-reduce %s B/step exposed %s ms, launched under backward %s" % (
            n, name, d["value"], d["unit"], d["ms_per_step"], d.get("rccl_ranks"), d.get("distinct_devices"), d.get("allreduce_bytes_per_step"),
            d.get("allreduce_exposed_ms_per_step"), d.get("allreduce_launched_under_backward_frac")))
    except Exception as e:
        print("N=%s %s: no line (%r)" % (n, name, e))
PY
done
