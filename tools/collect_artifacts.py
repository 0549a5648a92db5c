#!/usr/bin/env python3
"""Copies what tools/round_artifacts.sh <tag> left under gpurun_out/art_<tag>/ into profiles/:
the default bench line, the self-launched 2-rank line and the table of the other BASELINE configurations.
  python tools/collect_artifacts.py r04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
art = os.path.join(ROOT, "gpurun_out", f"art_{tag}")
prof = os.path.join(ROOT, "profiles")


def last_json(path):
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1])


for name in ("bench_default", "bench_gpus2_selflaunch_gloo"):
    d = last_json(os.path.join(art, name + ".json"))
    json.dump(d, open(os.path.join(prof, f"{tag}_{name}.json"), "w"), indent=1)
    print(name, d["value"], d["ms_per_step"])

rows = []
for name in ("cfg2_ipadic", "cfg3_unidic", "cfg5_unidic_user_S_M24_mixed", "dense_unidic", "unidic_short_uniform_5_20"):
    d = last_json(os.path.join(art, name + ".json"))
    r = d["roofline"]
    den = r["lattice_density"]
    rows.append(f"| {name} | {d['value'] / 1e6:.2f} M | {d['ms_per_step']:.3f} | {r['gen_candidates']['kernel_ms']:.3f} | {r['kernel_ms']:.3f} | "
                f"{r['whole_path'].get('pack_ms', 0):.3f} | {den['nodes_per_char']} / {den['dedup_pairs_per_char']} | {r['tiers']} | {d['parity_vs_oracle_sample']} |")
prev = ""
p_prev = os.path.join(prof, f"{tag}_other_configs.md")
notes = ""
if os.path.exists(p_prev):  # keep the hand-written notes under the table
    txt = open(p_prev).read()
    k = txt.find("\nargs:")
    notes = txt[k:] if k >= 0 else ""
open(p_prev, "w").write(
    f"# Round {tag[1:].lstrip('0')}: bench lines of the other BASELINE configurations (tools/round_artifacts.sh {tag}, one MI355X, device-resident)\n\n"
    "`python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 <args>`; full JSON lines in "
    f"gpurun_out/art_{tag}/ (scratch).\nThe default `python bench.py` line (profiles/{tag}_bench_default.json) carries config 5 and the dense law as `suite` legs as well.\n\n"
    "| run | sentences/s | ms per step | generator ms | lattice ms (fork to join) | fallback + packing ms | nodes / dedup pairs per char | "
    "sentences: segment tier, escape launches, fallback | bit-exact sample |\n|---|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n" + notes)
print(open(p_prev).read())
