"""Per-(kernel, grid) summary of a rocprofv3 rocpd SQLite trace: which SHAPES of a kernel the time goes to.
usage: python tools/rocpd_by_grid.py trace_results.db out.csv [top]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gz = "grid_z" if "grid_z" in cols else "0"
rows = list(db.execute("select name, grid_x, grid_y, %s, workgroup_x, count(*), sum(duration), avg(duration), min(duration) "
                       "from kernels group by name, grid_x, grid_y, %s order by sum(duration) desc" % (gz, gz)))
tot = sum(r[6] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "GridX", "GridY", "GridZ", "Workgroup", "Calls", "TotalUs", "AverageUs", "MinUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], r[4], r[5], "%.1f" % (r[6] / 1e3), "%.2f" % (r[7] / 1e3), "%.2f" % (r[8] / 1e3),
                    "%.3f" % (100.0 * r[6] / tot)])
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-70s grid %7d %5d %5d  n %5d  avg %8.1f us  tot %8.2f ms %5.1f%%" % (r[0][:70], r[1], r[2], r[3], r[5], r[7] / 1e3, r[6] / 1e6,
                                                                                100.0 * r[6] / tot))
