"""Timeline of one repetition out of a rocprofv3 results database: kernels between two launches of a marker kernel.
python tools/prof_timeline.py <results.db> <marker-fragment> [repetition]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
rep = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) // 2
g = rows[marks[rep]:marks[rep + 1]]
t0, prev = g[0][1], g[0][1]
print(f"span {(g[-1][2] - t0) / 1e3:.1f} us, busy {sum(r[2] - r[1] for r in g) / 1e3:.1f} us, {len(g)} kernels")
for name, a, b in g:
    print(f"{(a - t0) / 1e3:8.1f} +{(b - a) / 1e3:6.1f} gap {(a - prev) / 1e3:6.1f}  {name.replace('(anonymous namespace)::', '')[:80]}")
    prev = b
