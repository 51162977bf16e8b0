#!/bin/bash
# round-end evidence: default bench line (full CPU protocol), e2e / latency / vocoder stages, kernel trace, full suite + smoke
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r02; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
(timeout 900 python bench.py 2>$OUT/bench_default.err | tail -1) > $OUT/bench_default.json
python - <<'P'
import json
j=json.loads(open('gpurun_out/r02/bench_default.json').read())
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'], j.get('native_fp32_loop',{}).get('value'), (j.get('bf16_operand_loop') or {}).get('value'))
P
SIZES=1x800,2x800,3x800,4x800,8x800,16x800,32x800,64x800 timeout 300 python tools/latency_probe.py 2>&1 | grep "B=" > $OUT/latency.log; cat $OUT/latency.log
(EB=64 timeout 300 python tools/e2e_bench.py 2>&1 | tail -1) > $OUT/e2e.log; cut -c1-330 $OUT/e2e.log
(EB=1 ESTEPS=8 timeout 200 python tools/e2e_bench.py 2>&1 | tail -1) > $OUT/e2e_b1_8.log
(EB=1 ESTEPS=100 timeout 200 python tools/e2e_bench.py 2>&1 | tail -1) > $OUT/e2e_b1_100.log; cut -c1-330 $OUT/e2e_b1_8.log $OUT/e2e_b1_100.log
HSTAGES=1 timeout 300 python tools/hifigan_bench.py 2>&1 | grep "HiFi-GAN\|stage" > $OUT/hifigan_stages_x2.log; head -3 $OUT/hifigan_stages_x2.log
rm -rf $OUT/prof_bench $OUT/prof_hifigan
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_bench" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-bf16-loop --no-native-fp32 --steps 2 > "$R/$OUT/rocprof_bench.log" 2>&1)
python tools/rocpd_summary.py $(find $OUT/prof_bench -name "*.db" | head -1) $OUT/kernel_stats.csv > /dev/null 2>&1; head -4 $OUT/kernel_stats.csv | cut -c1-160
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_hifigan" -o h -- python "$R/tools/hifigan_bench.py" > "$R/$OUT/rocprof_hifigan.log" 2>&1)
python tools/rocpd_summary.py $(find $OUT/prof_hifigan -name "*.db" | head -1) $OUT/hifigan_kernel_stats.csv > /dev/null 2>&1; head -5 $OUT/hifigan_kernel_stats.csv | cut -c1-160
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu_full.log 2>&1
tail -3 $OUT/pytest_gpu_full.log; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu_full.log | head
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
