"""Per-(kernel, grid) durations from a rocprofv3 --kernel-trace CSV: python tools/ktrace.py <trace.csv> [name regex]."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"\w+")
d = collections.defaultdict(list)
for r in rows:
    name = re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    if not pat.search(name):
        continue
    d[(name, r["Grid_Size_X"], r.get("Grid_Size_Y", "1"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v2 = sorted(v)
    print(f"{k[0]:44s} grid {k[1]:>9s}x{k[2]:<3s} n={len(v):5d} mean {sum(v) / len(v):9.1f} med {v2[len(v) // 2]:9.1f} min {v2[0]:9.1f} max {v2[-1]:9.1f} us")
