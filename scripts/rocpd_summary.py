"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel stats table we commit under
profiles/ (same columns as rocprofv3 --stats: calls, total, average, min, max, percentage)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'Name':90s} {'Calls':>6s} {'TotalDuration(ns)':>18s} {'Average(ns)':>14s} {'Min(ns)':>12s} {'Max(ns)':>12s} {'Pct':>6s}")
for n, c, t, a, mn, mx in rows:
    print(f"{n[:90]:90s} {c:6d} {t:18d} {a:14.0f} {mn:12d} {mx:12d} {100*t/tot:6.2f}")
