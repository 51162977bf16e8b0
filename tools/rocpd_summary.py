"""Export the per-kernel summary (name, calls, total/avg duration, %) of a rocprofv3 rocpd SQLite trace to CSV.
usage: python tools/rocpd_summary.py gpurun_out/prof/trace_results.db profiles/r01_kernel_stats.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
extra = list(db.execute("select name, min(duration), max(duration), vgpr_count, accum_vgpr_count, lds_size, grid_x, grid_y, "
                        "workgroup_x from kernels group by name order by sum(duration) desc limit 6"))
for e in extra:
    print(e)
