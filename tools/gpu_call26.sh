#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout 200 python tools/x3_phase_probe.py 2>&1 | grep "B=\|ticks"
