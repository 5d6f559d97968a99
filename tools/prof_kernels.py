"""Per-kernel durations out of a rocprofv3 results database (rocprofv3 --kernel-trace writes <name>_results.db on this image):
python tools/prof_kernels.py gpurun_out/x/prof/pp_results.db [name-fragment ...]"""
import sqlite3, sys
import numpy as np

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, end-start, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, scratch_size from kernels"))
by = {}
for r in rows:
    by.setdefault(r[0], []).append(r)
tot = sum(r[1] for r in rows)
pats = sys.argv[2:]
for name, rs in sorted(by.items(), key=lambda kv: -sum(r[1] for r in kv[1])):
    if pats and not any(p in name for p in pats):
        continue
    d = np.array([r[1] for r in rs]) / 1e3
    short = name.replace("(anonymous namespace)::", "")[:60]
    print(f"{short:60s} n={len(d):5d} min {d.min():8.1f} med {np.median(d):8.1f} max {d.max():8.1f} us {100 * d.sum() * 1e3 / tot:5.1f}%  "
          f"grid {rs[-1][2]}x{rs[-1][3]} wg {rs[-1][4]} lds {rs[-1][5]} vgpr {rs[-1][6]} scratch {rs[-1][7]}")
