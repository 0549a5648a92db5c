#!/usr/bin/env python3
"""Developer aid: timeline of the LAST vbt_tokenize_batch call of a traced run (rocprofv3 --kernel-trace --memory-copy-trace):
kernels and copies with start / end relative to the call's first record.  usage: python tools/dbg/h2h_timeline.py <dir>"""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-48:], r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
rows.sort()
# the last call: walk back from the end to the last validate_batch preceded by a gap of > 300 us
starts = [i for i, r in enumerate(rows) if "validate_batch" in r[3]]
last = starts[-1]
while last > 0 and rows[last][0] - rows[last - 1][1] < 300000 and rows[last][0] - rows[starts[-1]][0] > -6000000:
    last -= 1
t0 = rows[last][0]
for s, e, kind, name, q in rows[last:]:
    if e - s < 3000 and kind == "K":
        continue
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {kind} q{q:>3} {name}")
