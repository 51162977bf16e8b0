#!/bin/bash
# round 6, step 42: pair splits (v_cvt_pk_f16_f32) in the HiFi-GAN split-operand kernels: forward time and wav hash, HEAD's library against the new one; then the GPU suite on the new library
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/hg.py <<'P'
import os, sys, time, hashlib, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import set_amd
from set_amd.hifigan import HifiGanGenerator
import hifigan_bench as hb  # prints its own line at import
wav = hb.g(hb.mel); torch.cuda.synchronize()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); wav = hb.g(hb.mel); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("forward ms: min %.2f median %.2f | wav sha256 %s" % (1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2], hashlib.sha256(wav.cpu().numpy().tobytes()).hexdigest()[:16]))
P
for rep in 1 2; do
  for tag in head d2; do
    SET_AMD_LIB=$PWD/build/exp/libset_amd_$tag.so timeout 300 python /tmp/hg.py 2>&1 | grep "forward ms" | sed "s/^/$tag: /"
  done
done | tee $OUT/hifigan_pk_ab.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_step42.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu_step42.log | cut -c1-250
