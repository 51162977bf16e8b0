#!/bin/bash
# round 6, step 30: V tile rows of 544 bytes (conflict-free 16-wide fragment reads): parity, LDS counters, A/B against 528-byte rows
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x3v or full800" > $OUT/step30_tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/step30_tests.log; tail -3 $OUT/step30_tests.log | cut -c1-200
rm -rf $OUT/pmc_x3_lds
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_lds" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_lds.log" 2>&1)
F=$(find $OUT/pmc_x3_lds -name "*counter_collection.csv" | head -1)
python - "$F" <<'P' | tee $OUT/pmc_x3_lds_summary_544.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "x3v" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-24s %.4g per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
P
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for rep in 1 2; do
  SET_AMD_LIB=$PWD/build/exp/libset_amd_xr528.so timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_xrv_ab_528_$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_xrv_ab_528_$rep.log | sed 's/^/V rows 528 B: /' | cut -c1-340
  timeout 300 python tools/loop_ab_probe.py 5 > $OUT/x3v_xrv_ab_544_$rep.log 2>&1; grep "x3_winograd_default\"" $OUT/x3v_xrv_ab_544_$rep.log | sed 's/^/V rows 544 B: /' | cut -c1-340
done
