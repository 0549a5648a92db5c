#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (tools/profile_round.sh) into profiles/<tag>_*.{csv,md,json}."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
lines = [f"# rocprofv3 summary, round {tag}\n"]
if os.path.exists(f"{src}/bench_line.json") and os.path.getsize(f"{src}/bench_line.json"):
    bench = json.load(open(f"{src}/bench_line.json"))
    lines += ["## bench line of the profiled command\n", "```json", json.dumps(bench, ensure_ascii=False), "```\n"]
stats = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copyfile(stats[0], f"profiles/{tag}_kernel_stats.csv")
    lines += [f"## `rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-pipeline` (profiles/{tag}_kernel_stats.csv)\n",
              "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                     f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    lines.append("")
trace = glob.glob(f"{src}/stats/**/*kernel_trace.csv", recursive=True)
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if "vbt::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: (re.findall(r"(\w+)(?:<[^>]*>)?\(vbt::", r["Kernel_Name"]) or ["?"])[0]
    big = max(int(r["Grid_Size_X"]) for r in rows)
    first = name(next(r for r in rows if int(r["Grid_Size_X"]) == big))
    starts = [i for i, r in enumerate(rows) if int(r["Grid_Size_X"]) == big and name(r) == first]
    spans = []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        step = [r for r in rows[a:b] if int(r["Grid_Size_X"]) > 0]
        lat = [r for r in step if name(r) in ("lattice_lds", "lattice_slim", "lattice_lean")]  # the sweep's two instances run side by side
        if len(step) < 3 or not lat:
            continue
        t0 = int(step[0]["Start_Timestamp"])
        spans.append({"step_us": (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3,
                      "gen_us": sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in step if name(r).startswith("gen_candidates")) / 1e3,
                      "lattice_span_us": (max(int(r["End_Timestamp"]) for r in lat) - min(int(r["Start_Timestamp"]) for r in lat)) / 1e3,
                      "lattice_launches": len(lat)})
    per = sorted(x["lattice_launches"] for x in spans)[len(spans) // 2]  # the last step also holds the parity-sample batch
    spans = [s_ for s_ in spans if s_["lattice_launches"] == per] or spans
    if spans:
        avg = lambda k: sum(s_[k] for s_ in spans) / len(spans)
        lines += ["## Per-step wall spans from the same kernel trace (the per-tier `lattice_lds` launches of a step run concurrently, so the",
                  "   per-launch average above is not a wall time; bench.py times the same span with hipEvents on the launch stream)\n",
                  f"| steps | whole step us | gen_candidates(+large) us | sweep span: lattice_lean + lattice_lds (first start -> last end) us |\n|---|---|---|---|",
                  f"| {len(spans)} full-batch steps | {avg('step_us'):.1f} | {avg('gen_us'):.1f} | {avg('lattice_span_us'):.1f} |\n"]
pm = collections.OrderedDict()
for f in sorted(glob.glob(f"{src}/pmc*/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "tokenize" not in r["Kernel_Name"] and "lattice" not in r["Kernel_Name"] and "candidates" not in r["Kernel_Name"]:
            continue
        d = rows.setdefault(int(r["Dispatch_Id"]), {"kernel": (re.findall(r"(\w+)(?:<[^>]*>)?\(vbt::", r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0], "grid": int(r["Grid_Size"]),
                                                      "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ks = sorted(rows)
    if not ks:
        continue
    big = max(rows[k]["grid"] for k in ks)
    first = rows[[k for k in ks if rows[k]["grid"] == big][0]]["kernel"]
    starts = [i for i, k in enumerate(ks) if rows[k]["grid"] == big and rows[k]["kernel"] == first]
    idx = starts[-1]
    per_step = starts[-1] - starts[-2] if len(starts) > 1 else len(ks) - idx
    for j in range(idx, min(idx + per_step, len(ks))):
        d = rows[ks[j]]
        pm.setdefault((j - idx, d["kernel"], d["grid"]), {}).update(d)
if pm:
    lines += ["## PMC counters, one full-batch step (separate `--pmc` passes; per dispatch)\n"]
    for (i, k, g), d in pm.items():
        lines.append(f"### launch {i}: `{k}` grid={g} ({g // 64} waves), {d.get('dur_us', 0):.1f} us, VGPR {d['vgpr']} AGPR {d['agpr']} SGPR {d['sgpr']}\n")
        lines.append("| counter | value |\n|---|---|")
        for c, v in d.items():
            if c not in ("kernel", "grid", "vgpr", "agpr", "sgpr", "dur_us"):
                lines.append(f"| {c} | {v:,.0f} |")
        if "FETCH_SIZE" in d:
            fb, wb = d["FETCH_SIZE"] * 1024, d.get("WRITE_SIZE", 0) * 1024
            lines.append(f"\nHBM traffic (guide: bytes = (FETCH_SIZE + WRITE_SIZE) * 1024; narrow gathers, not the 2x wide-read case): "
                         f"fetch {fb / 1e6:.1f} MB + write {wb / 1e6:.1f} MB = {(fb + wb) / 1e6:.1f} MB")
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
            lines.append(f"L2 hit rate: {d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']):.3f}")
        lines.append("")
    json.dump({f"{i}:{k}:{g}": d for (i, k, g), d in pm.items()}, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
    # calibration of the two counters on known byte counts (tools/calib/fetch_calib.hip, run by tools/profile_round.sh)
    calib = None
    try:
        known = json.loads([l for l in open(f"{src}/calib_bytes.json") if l.startswith("{")][-1])
        names = list(known)
        meas = {}
        for cnt, sub in (("FETCH_SIZE", "calib_r"), ("WRITE_SIZE", "calib_w")):
            f = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)
            rows_c = sorted((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == cnt and "fillBuffer" not in r["Kernel_Name"])
            for nm, (_, v) in zip(names, rows_c[-len(names):]):
                meas.setdefault(nm, {})[cnt] = v * 1024
        calib = {nm: {"known_bytes": known[nm], "counter_bytes": meas[nm]["FETCH_SIZE" if not nm.startswith("write") else "WRITE_SIZE"],
                      "bytes_per_counted_byte": round(known[nm] / max(meas[nm]["FETCH_SIZE" if not nm.startswith("write") else "WRITE_SIZE"], 1), 3)} for nm in names}
        lines += ["## Calibration of FETCH_SIZE / WRITE_SIZE (x 1024 B) on known byte counts over 1 GiB (tools/calib/fetch_calib.hip)\n",
                  "| pattern | known bytes | counter bytes | true bytes per counted byte |", "|---|---|---|---|"]
        for nm, c in calib.items():
            lines.append(f"| {nm} | {c['known_bytes']:,} | {c['counter_bytes']:,.0f} | {c['bytes_per_counted_byte']} |")
        lines.append("\n(gathers: `known` = useful bytes, so < 1 means whole lines are fetched for a few useful bytes; streaming reads > 1 = the counter under-reports)\n")
    except Exception as e:  # no calibration run in this profile directory
        lines.append(f"(no FETCH_SIZE calibration in {src}: {e})\n")
    if all("FETCH_SIZE" in d for d in pm.values()):
        total = sum((d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024 for d in pm.values())
        by_kernel = collections.OrderedDict()
        for (i, k, g), d in pm.items():
            by_kernel[k] = by_kernel.get(k, 0) + int((d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0)) * 1024)
        bench_cfg = json.load(open(f"{src}/bench_line.json"))["config"]["workload"] if os.path.getsize(f"{src}/bench_line.json") else ""
        insts = collections.OrderedDict()
        for (i, k, g), d in pm.items():
            if "SQ_INSTS_VALU" in d:
                e = insts.setdefault(k, {"valu": 0, "salu": 0, "lds": 0, "vmem_rd": 0, "smem": 0})
                for key, cn in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("smem", "SQ_INSTS_SMEM")):
                    e[key] += int(d.get(cn, 0))
        # where the waves' cycles went (SQ counters of the second PMC pass), per kernel: bench.py reports the shares
        sq = collections.OrderedDict()
        for (i, k, g), d in pm.items():
            if "SQ_WAVE_CYCLES" in d:
                e = sq.setdefault(k, {"wave_cycles": 0, "busy_cycles": 0, "wait_any": 0, "wait_inst_any": 0, "active_inst_any": 0, "active_inst_valu": 0,
                                      "active_inst_lds": 0, "lds_bank_conflict": 0, "lds_idx_active": 0})
                for key, cn in (("wave_cycles", "SQ_WAVE_CYCLES"), ("busy_cycles", "SQ_BUSY_CYCLES"), ("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"),
                                ("active_inst_any", "SQ_ACTIVE_INST_ANY"), ("active_inst_valu", "SQ_ACTIVE_INST_VALU"), ("active_inst_lds", "SQ_ACTIVE_INST_LDS"),
                                ("lds_bank_conflict", "SQ_LDS_BANK_CONFLICT"), ("lds_idx_active", "SQ_LDS_IDX_ACTIVE")):
                    e[key] += int(d.get(cn, 0))
        # upper bound if every read of the step were a wide streaming read (the calibrated under-report of read16): the
        # per-char records (16 B/lane) and the sweep records (8 B/lane) are, the trie / matrix gathers are not
        rf = max((calib or {}).get("read16", {}).get("bytes_per_counted_byte", 1.0), (calib or {}).get("read8", {}).get("bytes_per_counted_byte", 1.0), 1.0)
        upper = sum((d["FETCH_SIZE"] * rf + d.get("WRITE_SIZE", 0)) * 1024 for d in pm.values())
        json.dump({"workload": bench_cfg, "hbm_bytes_per_step": int(total), "hbm_bytes_by_kernel": by_kernel, "wave_insts_by_kernel": insts, "sq_by_kernel": sq,
                   "hbm_bytes_per_step_upper_bound": int(upper), "calibration": calib, "source": f"profiles/{tag}_pmc.json",
                   "method": "sum over the step's kernels of (FETCH_SIZE + WRITE_SIZE) * 1024, separate rocprofv3 --pmc passes; upper bound = "
                             "every read scaled by the calibrated under-report of wide streaming reads"},
                  open(f"profiles/{tag}_traffic.json", "w"), indent=1)
        lines.append(f"## HBM traffic of one step (all kernels): {total / 1e6:.1f} MB\n")
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:70]))
