#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
SIZES=1x800 timeout 200 python tools/split_phase_probe.py 2>&1 | grep "B="
