#!/usr/bin/env python3
"""Print per-dispatch PMC values of the last full-batch step from gpurun_out/pmc_<tag>/."""
import collections, csv, glob, re, sys
tag = sys.argv[1]
pm = collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}/p*/*counter_collection.csv")):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        m = re.findall(r"(\w+)(?:<[^>]*>)?\(vbt::", r["Kernel_Name"])
        if not m:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": m[0], "grid": int(r["Grid_Size"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ks = sorted(rows)
    if not ks: continue
    big = max(rows[k]["grid"] for k in ks)
    starts = [i for i, k in enumerate(ks) if rows[k]["grid"] == big]
    per = starts[-1] - starts[-2] if len(starts) > 1 else len(ks) - starts[-1]
    for j in range(starts[-1], min(starts[-1] + per, len(ks))):
        d = rows[ks[j]]
        pm.setdefault((j - starts[-1], d["kernel"], d["grid"]), {}).update(d)
for (i, k, g), d in pm.items():
    print(f"[{i}] {k} waves={g//64} dur={d['dur_us']:.1f}us", {c: int(v) for c, v in d.items() if c not in ('kernel','grid','dur_us')})
