#!/bin/bash
# round 5, step 8: leaf-stream equivalence test (tiny + full size); where the headline loop loses time BETWEEN its kernels (gaps by kernel pair)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r5s8; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_training.py -q -x -k "leaf_stream" 2>&1 | tail -3 | tee $OUT/pytest.log
rm -rf $OUT/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/$OUT/prof" -o t -- python "$R/bench.py" --no-cpu-baseline --no-native-fp32 --no-bf16x3-loop --no-bf16-loop --no-secondary --no-quality --steps 3 > "$R/$OUT/rocprof.log" 2>&1)
python tools/rocpd_gaps.py $(find $OUT/prof -name "*.db" | head -1) 100 | tee $OUT/headline_gaps.log
rm -rf $OUT/prof
