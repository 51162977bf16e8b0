#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
for k in 1 2; do SIZES=1x800,2x800 timeout 200 python tools/latency_probe.py 2>&1 | grep "B="; done
(cd /tmp && SIZES=1x800 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r02/prof_latency" -o lat -- python "$GRAFT_REPO_ROOT/tools/latency_probe.py" > "$GRAFT_REPO_ROOT/gpurun_out/r02/rocprof_latency.log" 2>&1)
grep "B=" gpurun_out/r02/rocprof_latency.log
ls gpurun_out/r02/prof_latency/*/ | head
python tools/rocpd_summary.py $(ls gpurun_out/r02/prof_latency/*/*.db | head -1) gpurun_out/r02/latency_kernel_stats.csv 2>&1 | tail -2
head -8 gpurun_out/r02/latency_kernel_stats.csv
