#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
(cd /tmp && SET_AMD_DTYPE=bf16 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_campnet_bf16" -o c -- python "$GRAFT_REPO_ROOT/tools/campnet_bench.py" > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_campnet_bf16.log" 2>&1)
tail -1 gpurun_out/r02/rocprof_campnet_bf16.log | cut -c1-200
