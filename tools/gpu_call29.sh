#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for m in ${MODES:-f16x2}; do SET_AMD_SPLIT_OPERAND=$m timeout 200 python tools/x3_phase_probe.py 2>&1 | grep "B=\|ticks"; done
