#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) of the fused bf16 layer-group kernel at B = 32, T = 800 (tools/bf16_layers_probe.py):
# MFMA busy cycles / clock and HBM bytes per launch -> gpurun_out/pmc_bf16_layers.json
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/pmc_bf16; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
rm -rf $OUT/*
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/util" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/util.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/fetch" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/write" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/write.log" 2>&1)
python - <<'PY'
import csv, glob, json, collections
out = {}
def collect(d):
    f = glob.glob("gpurun_out/pmc_bf16/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    kt = glob.glob("gpurun_out/pmc_bf16/%s/**/*kernel_trace.csv" % d, recursive=True)
    for r in csv.DictReader(open(f[0])):
        if "diffnet_layers_fwd_bf16_kernel" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            if "diffnet_layers_fwd_bf16_kernel" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, dur
u, du = collect("util"); f, df = collect("fetch"); w, dw = collect("write")
# the probe launches groups of 1, 2, 4, 8 layers: 11 dispatches each (1 warm-up + 10); take the last dispatch of each group size
ids = sorted(u.keys(), key=int)
for gi, nl in enumerate((1, 2, 4, 8)):
    did = ids[gi * 11 + 10] if len(ids) >= (gi + 1) * 11 else None
    if did is None: continue
    c = {k: sum(v) for k, v in u[did].items()}
    fid, wid = sorted(f.keys(), key=int)[gi * 11 + 10], sorted(w.keys(), key=int)[gi * 11 + 10]
    fetch_kb, write_kb = sum(f[fid]["FETCH_SIZE"]), sum(w[wid]["WRITE_SIZE"])
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out["layers_per_launch_%d" % nl] = {
        "duration_us_under_pmc": du.get(did), "cycles_per_launch": cyc, "sclk_GHz": (cyc / du[did] / 1e3) if did in du else None,
        "mfma_busy_frac_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
        "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
        "algorithmic_bytes_per_launch": 4864 * 25600 * nl,
    }
json.dump(out, open("gpurun_out/pmc_bf16_layers.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -delete
