#!/bin/bash
# round 6, step 38: where the x3v kernel's excess fetches come from: FETCH_SIZE with the residual-row re-read (1) / the second staging pass's x re-read (2) / both (3) removed
# (measurement builds: WRONG RESULTS)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
for tag in shipped expm1 expm2 expm3; do
  rm -rf $OUT/pmc_f
  LIB=""; [ $tag != shipped ] && LIB="$R/build/exp/libset_amd_$tag.so"
  (cd /tmp && SET_AMD_LIB=$LIB timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_f" -o pmc -- python "$R/tools/x3_phase_probe.py" > "$R/$OUT/pmc_f_$tag.log" 2>&1)
  F=$(find $OUT/pmc_f -name "*counter_collection.csv" | head -1)
  python - "$F" $tag <<'P'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "x3v" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("%-8s FETCH_SIZE x 2 = %.3f GB per launch (%d launches)" % (sys.argv[2], 2 * 1024 * sum(v) / len(v) / 1e9, len(v)))
P
done | tee $OUT/x3v_fetch_sources.log
rm -rf $OUT/pmc_f
