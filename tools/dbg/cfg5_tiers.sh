#!/bin/bash
# config 5 (long sentences in segments): which lean / segment tier pair with the five-wave slim instance?
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 20 --warmup 3 --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  parity %s tiers %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], d["parity_vs_oracle_sample"], r["tiers"]))'
for rep in 1 2; do
  for t in "7680,8192" "7680,8704" "7168,8192" "6144,8192" "7680,7936" "6656,7168" "4096,8192"; do
    seg=${t#*,}
    echo "$t"; VBT_TIERS=$t,49152,163840 VBT_SEG_BYTES=$seg $B 2>/dev/null | python -c "$pick"
  done
done
