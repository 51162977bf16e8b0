#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) of the fused bf16 layer-group kernels at B = 32, T = 800 (tools/bf16_layers_probe.py):
# MFMA busy cycles / clock / wait shares / LDS bank conflicts and HBM bytes per launch -> gpurun_out/pmc_bf16_layers_<tile>.json
# usage: TILE=64|128 [VARIANT=0..3] [NLS=5,10] tools/sessions/gpu_pmc_bf16_layers.sh
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/pmc_bf16; mkdir -p $OUT; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"
export SET_AMD_BF16_FUSE_TILE=${TILE:-128} SET_AMD_BF16_REG_VARIANT=${VARIANT:-0} NLS=${NLS:-5,10}
rm -rf $OUT/*
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$R/$OUT/util" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/util.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$R/$OUT/lds" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/lds.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/fetch" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/fetch.log" 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/$OUT/write" -o pmc -- python "$R/tools/bf16_layers_probe.py" > "$R/$OUT/write.log" 2>&1)
python - <<'PY'
import csv, glob, json, collections, os
import hashlib
KS = ["speech-editing-toolkit_amd/csrc/diffnet_bf16.hip", "speech-editing-toolkit_amd/csrc/common.h"]
hh = hashlib.sha256()
for fsrc in KS:
    hh.update(open(fsrc, "rb").read())
out = {"tile": os.environ["SET_AMD_BF16_FUSE_TILE"], "variant": os.environ["SET_AMD_BF16_REG_VARIANT"], "kernel_sources": KS,
       "kernel_source_sha256": hh.hexdigest(), "shape": "B=32, T=800 (tools/bf16_layers_probe.py)",
       "_note": "per launch; hbm_bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 half-count correction of MI355X_MICROARCH.md)"}
nls = [int(v) for v in os.environ["NLS"].split(",")]
def collect(d):
    f = glob.glob("gpurun_out/pmc_bf16/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    kt = glob.glob("gpurun_out/pmc_bf16/%s/**/*kernel_trace.csv" % d, recursive=True)
    if not f: return {}, {}
    for r in csv.DictReader(open(f[0])):
        if "diffnet_layers_" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            if "diffnet_layers_" in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc, dur
u, du = collect("util"); l, dl = collect("lds"); f, df = collect("fetch"); w, dw = collect("write")
def pick(d, gi):  # the probe launches each group size 11 times (1 warm-up + 10): the last dispatch of group gi
    ids = sorted(d.keys(), key=int)
    return ids[gi * 11 + 10] if len(ids) >= (gi + 1) * 11 else None
for gi, nl in enumerate(nls):
    did = pick(u, gi)
    if did is None: continue
    c = u[did]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    o = {"duration_us_under_pmc": du.get(did), "cycles_per_launch": cyc, "sclk_GHz": (cyc / du[did] / 1e3) if did in du else None,
         "mfma_busy_frac_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
         "wave_cycles": c["SQ_WAVE_CYCLES"], "wait_any_frac": c["SQ_WAIT_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]),
         "wait_inst_any_frac": c["SQ_WAIT_INST_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]), "active_inst_any_frac": c["SQ_ACTIVE_INST_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]),
         "algorithmic_bytes_per_launch": 4864 * 25600 * nl}
    lid = pick(l, gi)
    if lid: o.update({k.lower(): v for k, v in l[lid].items()})
    fid, wid = pick(f, gi), pick(w, gi)
    if fid and wid: o["hbm_bytes_per_launch"] = 2 * f[fid]["FETCH_SIZE"] * 1024 + w[wid]["WRITE_SIZE"] * 1024
    out["layers_per_launch_%d" % nl] = o
json.dump(out, open("gpurun_out/pmc_bf16_layers_%s.json" % out["tile"], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -delete
