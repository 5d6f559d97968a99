"""Launch-by-launch timeline of one registration's pass kernels from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/icp_only.py
    python tools/pass_timeline.py gpurun_out/tl
Prints, for the LAST 21 launches of icp_pass_kernel<8, false>, duration and the gap to the previous kernel's end."""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "icp_pass_kernel<8, false>" in r["Kernel_Name"] or "icp_pass_kernelILi8ELb0" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 21
sel = idx[-n:]
t0 = int(rows[sel[0]]["Start_Timestamp"])
dur, gap = [], []
for i in sel:
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    prev_e = int(rows[i - 1]["End_Timestamp"]) if i > 0 else s
    dur.append((e - s) / 1e3); gap.append((s - prev_e) / 1e3)
    print(f"start {(s - t0) / 1e3:9.2f} us  dur {dur[-1]:7.2f}  gap {gap[-1]:6.2f}  grid {r['Grid_Size_X']}  prev {rows[i-1]['Kernel_Name'][:40]}")
print(f"steady passes: mean dur {sum(dur[1:]) / len(dur[1:]):.2f} us, mean gap {sum(gap[1:]) / len(gap[1:]):.2f} us; pass 0 {dur[0]:.2f} us; span {(int(rows[sel[-1]]['End_Timestamp']) - t0) / 1e3:.1f} us")
