"""Timeline of ONE training step from a rocprofv3 rocpd SQLite trace: every kernel between two consecutive `adamw_kernel` launches with its
start offset, duration and stream / queue, plus per-stream busy time, the union busy time and the idle gaps of the step (is the step waiting for
the host or for kernels?).
usage: python tools/rocpd_timeline.py trace_results.db out.csv [step_index]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("kernels columns:", cols)
sid = next((c for c in ("stream_id", "queue_id", "stream", "queue") if c in cols), None)
start = next((c for c in ("start", "start_ns", "begin") if c in cols), None)
end = next((c for c in ("end", "end_ns", "stop") if c in cols), None)
q = "select name, %s, %s, %s from kernels order by %s" % (start, end, sid or "0", start)
rows = list(db.execute(q))
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) // 2
i0, i1 = marks[k] + 1, marks[k + 1] + 1
step = rows[i0:i1]
t0 = rows[marks[k]][2]  # end of the previous step's optimizer kernel
span = step[-1][2] - t0
streams = {}
for r in step:
    streams.setdefault(r[3], []).append(r)
print("step %d: %d kernels, span %.3f ms" % (k, len(step), span / 1e6))
for s, rs in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[2] - r[1] for r in rs)
    print("  stream %s: %d kernels, busy %.3f ms, first at %.3f ms, last ends at %.3f ms" % (s, len(rs), busy / 1e6, (rs[0][1] - t0) / 1e6, (rs[-1][2] - t0) / 1e6))
# union busy time and gaps
iv = sorted((r[1], r[2]) for r in step)
cur_s, cur_e, union, gaps = iv[0][0], iv[0][1], 0, []
if cur_s > t0:
    gaps.append((t0, cur_s))
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
gap_tot = sum(b - a for a, b in gaps)
print("  union busy %.3f ms, idle %.3f ms in %d gaps (>= 20 us: %d, total %.3f ms)" % (
    union / 1e6, gap_tot / 1e6, len(gaps), sum(1 for a, b in gaps if b - a >= 20000), sum(b - a for a, b in gaps if b - a >= 20000) / 1e6))
# the main stream's kernel-to-kernel gaps (launch latency / host starvation)
main = max(streams.values(), key=len)
mg = [main[i + 1][1] - main[i][2] for i in range(len(main) - 1)]
mg_s = sorted(mg)
print("  main stream: median gap %.2f us, p90 %.2f us, sum of gaps %.3f ms" % (mg_s[len(mg_s) // 2] / 1e3, mg_s[int(0.9 * len(mg_s))] / 1e3, sum(mg) / 1e6))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["StartUs", "DurUs", "Stream", "Name"])
    for r in step:
        w.writerow(["%.2f" % ((r[1] - t0) / 1e3), "%.2f" % ((r[2] - r[1]) / 1e3), r[3], r[0][:120]])
