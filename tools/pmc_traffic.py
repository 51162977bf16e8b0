"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch of the dominant kernels.
Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half
of the bytes of wide coalesced reads -> doubled here (stated in the output); WRITE_SIZE is used as reported.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"]
            for key in ("diffnet_stack_x3_kernel", "diffnet_stack_wino_kernel", "diffnet_stack_kernel", "diffnet_layer_kernel",
                        "conv1d_mfma_kernel"):
                if key in name and "pack_" not in name:
                    if key == "diffnet_stack_x3_kernel":  # one entry per splitting
                        key += "<SplitF16x2>" if "SplitF16x2" in name else "<SplitBf16x3>"
                    acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "bytes per launch = 2*FETCH_SIZE*1024 (gfx950 half-count correction) + WRITE_SIZE*1024; mean over launches",
       "_launches": nf}
for k in fetch:
    out[k] = 2.0 * fetch[k] * 1024.0 + write.get(k, 0.0) * 1024.0
    out[k + "_fetch_KB_raw"] = fetch[k]
    out[k + "_write_KB_raw"] = write.get(k, 0.0)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
