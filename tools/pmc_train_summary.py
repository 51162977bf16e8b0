"""rocprofv3 passes over `bench.py --mode train` -> profiles/r06_pmc_train.json: per kernel family the HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE KB, the gfx950 half-count correction of MI355X_MICROARCH.md's HBM section; separate --pmc passes) and, from a
plain kernel trace, the launches per step; stamped with the sha256 of the kernel sources (bench.py quotes the figures only while it matches).
usage: python tools/pmc_train_summary.py <out.json> <model>_<dtype> <fetch.csv> <write.csv> <kernel_stats.csv> [more triples of the other model]"""
import collections
import csv
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["speech-editing-toolkit_amd/csrc/diffnet_bf16.hip", "speech-editing-toolkit_amd/csrc/bf16.hip",
                  "speech-editing-toolkit_amd/csrc/common.h", "speech-editing-toolkit_amd/csrc/rows_sum.h",
                  "speech-editing-toolkit_amd/csrc/train.hip", "speech-editing-toolkit_amd/csrc/conv1d.hip",
                  "speech-editing-toolkit_amd/autograd_ops.py"]
FAMILIES = ["diffnet_layer_bwd_bf16_kernel", "diffnet_layer_fwd_bf16_kernel", "conv1d_wgrad3u_bf16_kernel", "conv1d_wgrad3_bf16_kernel", "conv1d_wgrad_bf16_kernel",
            "conv1d_bf16_kernel", "conv1x1_oneshot_bf16_kernel", "attn_bwd_dkv_kernel", "attn_bwd_dq_kernel", "attn_fwd_kernel",
            # fp32 step (round 6: the reference's default precision had no PMC figure): the two generic fp32 MFMA conv kernels, the Winograd stack
            "conv1d_mfma_v2_kernel", "conv1d_mfma_kernel", "diffnet_stack_wino_kernel"]


def source_sha():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()


def collect(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        fam = next((f for f in FAMILIES if f in r["Kernel_Name"]), None)
        if fam:
            acc[fam].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


if __name__ == "__main__":
    out_path = sys.argv[1]
    out = {"_note": "per launch, mean over the launches of a kernel family in a few training steps at the bench shape; traffic_bytes = "
                    "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (KB counters; gfx950 reports half of wide coalesced reads); launches_per_step = "
                    "the kernels of one steady-state step of a rocprofv3 --kernel-trace run (launches_per_step_incl_setup: all kernels / optimizer steps)",
           "kernel_sources": KERNEL_SOURCES, "kernel_source_sha256": source_sha()}
    rest = sys.argv[2:]
    while rest:
        tag, fetch_csv, write_csv, stats_csv = rest[:4]
        rest = rest[4:]
        fetch, write = collect(fetch_csv, "FETCH_SIZE"), collect(write_csv, "WRITE_SIZE")
        ent = {}
        for fam in fetch:
            f, nf = fetch[fam]
            w, nw = write.get(fam, (0.0, 0))
            ent[fam] = {"traffic_bytes": 2.0 * f * 1024.0 + w * 1024.0, "fetch_KB_raw": f, "write_KB_raw": w, "launches_sampled": [nf, nw]}
        rows = list(csv.DictReader(open(stats_csv)))
        steps = next((int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]), 0)
        if steps:
            ent["launches_per_step_incl_setup"] = sum(int(r["Calls"]) for r in rows) / steps  # (first-step flattening / packing launches included)
            ent["launches_per_step"] = ent["launches_per_step_incl_setup"]
            ent["kernel_time_us_per_step"] = sum(float(r["TotalDurationUs"]) for r in rows) / steps
        # steady state: the kernels between two optimizer launches late in the run, counted by tools/rocpd_timeline.py ("step 9: 549 kernels")
        tl = stats_csv.replace("_kernel_stats.csv", "_timeline.log")
        if os.path.exists(tl):
            m = re.search(r"step \d+: (\d+) kernels", open(tl).read())
            if m:
                ent["launches_per_step"] = int(m.group(1))
        out[tag] = ent
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])
