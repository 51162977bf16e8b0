"""Launch-to-launch gaps of a rocprofv3 kernel trace (rocpd SQLite): for every (previous kernel -> next kernel) pair on the busiest stream the
mean idle time between the end of one and the start of the next, and the mean duration of the next -- where a loop of few large kernels
(the reverse diffusion loop: stack launch, step boundary, flag reset) loses time BETWEEN its kernels.
usage: python tools/rocpd_gaps.py trace_results.db [min_count]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = list(db.execute("select name, start, end, %s from kernels order by start" % sid))
by_stream = collections.Counter(r[3] for r in rows)
main = by_stream.most_common(1)[0][0]
rows = [r for r in rows if r[3] == main]
pairs = collections.defaultdict(lambda: [0, 0.0, 0.0])
short = lambda n: n.split("(")[0].split("::")[-1][:48]
for a, b in zip(rows, rows[1:]):
    p = pairs[(short(a[0]), short(b[0]))]
    p[0] += 1
    p[1] += max(0, b[1] - a[2])
    p[2] += b[2] - b[1]
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 50
print("%-50s -> %-50s %7s %10s %12s" % ("previous kernel", "next kernel", "count", "gap us", "next dur us"))
for (a, b), (n, g, d) in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
    if n >= mn:
        print("%-50s -> %-50s %7d %10.2f %12.2f" % (a, b, n, g / n / 1e3, d / n / 1e3))
