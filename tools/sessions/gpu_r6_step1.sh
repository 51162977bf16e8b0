#!/bin/bash
# round 6, step 1: the new reference-pinned T=800 x 100-step tests, the whole-loop launch (bit-identity + A/B with power / clock), INTEGRATION.md
# section B, the leaf-stream operand release, host core facts of the box; then the whole GPU suite.
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp
python - > $OUT/host_cores.log 2>&1 <<'PY'
import bench, json
hc = bench.host_cores(); hc["affinity"] = len(hc["affinity"]); print(json.dumps(hc))
PY
cat $OUT/host_cores.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "full800 or whole_loop" > $OUT/step1_new_tests.log 2>&1; echo "new parity tests rc=$?" | tee -a $OUT/step1_new_tests.log
grep -h "infer_full800\|max|dmel|" $OUT/step1_new_tests.log | tail -12
timeout 600 python -m pytest tests/test_integration_doc.py tests/test_gpu_training.py -q -x -s -k "section_b or operands_are_released" > $OUT/step1_misc_tests.log 2>&1; echo "misc rc=$?" | tee -a $OUT/step1_misc_tests.log
grep -h "INTEGRATION\|leaf operand" $OUT/step1_misc_tests.log | tail -4
timeout 600 python tools/loop_ab_probe.py 6 grid=240 grid=288 > $OUT/loop_launch_ab.log 2>&1; cat $OUT/loop_launch_ab.log | tail -12
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_step1.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_step1.log; tail -3 $OUT/pytest_gpu_step1.log
