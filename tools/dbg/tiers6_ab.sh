#!/bin/bash
# Lean instance at six waves per SIMD (80 VGPRs): tier sets with a 6.5 KiB lean tier against the 8 KiB one, alternating on one box.
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps ${STEPS:-40} --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  pack %.4f parity %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], r["whole_path"]["pack_ms"], d["parity_vs_oracle_sample"]))'
SETS=${SETS:-"8192,10240,49152,163840 6656,10240,49152,163840 6656,8192,10240,49152,163840"}
for rep in 1 2 3; do
  for t in $SETS; do
    echo "VBT_TIERS=$t headline"; VBT_TIERS=$t $B 2>/dev/null | python -c "$pick"
  done
done
for t in $SETS; do
  echo "VBT_TIERS=$t cfg5"; VBT_TIERS=$t $B --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>/dev/null | python -c "$pick"
  echo "VBT_TIERS=$t dense"; VBT_TIERS=$t $B --dict unidic-dense 2>/dev/null | python -c "$pick"
  echo "VBT_TIERS=$t short"; VBT_TIERS=$t $B --law uniform_5_20 2>/dev/null | python -c "$pick"
done
