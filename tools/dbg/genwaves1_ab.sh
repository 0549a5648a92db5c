B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 30 --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("  ", round(d["value"]/1e6,2), d["ms_per_step"], "sweep", r["kernel_ms"], "gen", r["gen_candidates"]["kernel_ms"], d["parity_vs_oracle_sample"])'
for rep in 1 2; do for v in 4 8 16; do echo "waves1=$v"; VBT_GEN_WAVES1=$v $B 2>/dev/null | python -c "$pick"; done; done
for v in 4 8; do echo "cfg5 waves1=$v"; VBT_GEN_WAVES1=$v $B --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>/dev/null | python -c "$pick"; done
