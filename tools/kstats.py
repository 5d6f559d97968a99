"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <dir or csv> [rows]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in list(csv.DictReader(open(p)))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    print(f'{name[:44]:44s} n={r["Calls"]:>6s} tot_ms={float(r["TotalDurationNs"])/1e6:9.3f} avg_us={float(r["AverageNs"])/1e3:9.2f} {float(r["Percentage"]):6.2f}%')
