#!/usr/bin/env python3
"""Developer aid: kernels of the LAST step of a traced bench run (rocprofv3 --kernel-trace), start / end relative to the step's
validate_batch.  usage: python tools/dbg/step_timeline.py <dir>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:], r.get("Queue_Id", ""),
                     r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "validate_batch" in r[2]]
big = max(int(r[5] or 0) for r in rows if r[2].endswith("gen_candidates"))
full = [i for i in starts if any(rows[j][2].endswith("gen_candidates") and int(rows[j][5] or 0) == big for j in range(i, min(i + 3, len(rows))))]
first = full[-1]
t0 = rows[first][0]
for s, e, name, q, wg, grid, lds in rows[first:first + 16]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{q:>3} grid {grid:>9} lds {lds:>6}  {name}")
