"""One frame's HIP API calls out of a rocprofv3 --hip-trace csv (frames delimited by torch's D2H copy of the scene cloud):
python tools/hip_api_frame.py <hip_api_trace.csv> [frame]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if not r["Function"].startswith("__hip")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Function"] == "hipMemcpyWithStream"]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 3
g = rows[marks[k]:marks[k + 1]]
t0 = int(g[0]["Start_Timestamp"]); prev = t0
print(f"{len(g)} calls, span {(int(g[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, inside HIP {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in g) / 1e3:.1f} us")
for r in g:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(a - t0) / 1e3:8.1f} +{(b - a) / 1e3:7.1f} gap {(a - prev) / 1e3:7.1f}  {r['Function']}")
    prev = b
