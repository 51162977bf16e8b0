#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r02/pytest_gpu_full.log 2>&1
tail -4 gpurun_out/r02/pytest_gpu_full.log
grep -E "^FAILED|^ERROR" gpurun_out/r02/pytest_gpu_full.log | head
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
