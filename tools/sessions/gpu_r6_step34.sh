#!/bin/bash
# round 6, step 34: is the vocoder at the power cap?
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/hifigan_power_probe.py 8 2>&1 | grep "HiFi-GAN" | tee $OUT/hifigan_power.log
