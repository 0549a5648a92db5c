#!/bin/bash
# A/B of VBT_GEN_SWEEP (generator + lean sweep in one wave) on the headline, config 5 and the dense law: alternating runs on one box.
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 20 --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  parity %s tiers %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], d["parity_vs_oracle_sample"], r["tiers"]))'
for rep in 1 2; do
  for v in 0 1; do
    echo "VBT_GEN_SWEEP=$v headline"; VBT_GEN_SWEEP=$v $B 2>/dev/null | python -c "$pick"
  done
done
for v in 0 1; do
  echo "VBT_GEN_SWEEP=$v cfg5"; VBT_GEN_SWEEP=$v $B --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>/dev/null | python -c "$pick"
  echo "VBT_GEN_SWEEP=$v dense"; VBT_GEN_SWEEP=$v $B --dict unidic-dense 2>/dev/null | python -c "$pick"
done
