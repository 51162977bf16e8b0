#!/bin/bash
# round 6, step 29: LDS conflict counters of the headline kernel
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/pmc_x3_lds
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$R/$OUT/pmc_x3_lds" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_x3_lds.log" 2>&1)
F=$(find $OUT/pmc_x3_lds -name "*counter_collection.csv" | head -1)
python - "$F" <<'P' | tee $OUT/pmc_x3_lds_summary.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "x3v" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-24s %.4g per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
P
tail -3 $OUT/pmc_x3_lds.log
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
