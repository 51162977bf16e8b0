#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "groups or full_size or full_inference" 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
for g in 1 2 3 4 6 8; do
  echo "== groups $g"
  SET_AMD_GROUPS=$g timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.0f  ms/step %.1f  launch_ms %.4f  achieved %.1f  wall_lb %.1f' % (d['value'], d['ms_per_step'], r['launch_ms'], r['achieved'], r['achieved_wall_lower_bound']))" | tee -a gpurun_out/groups.log
done
