#!/bin/bash
# s5 library: which lean tier in front of an 8 KiB segment tier?  headline, short sentences, ipadic.
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 30 --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  parity %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], d["parity_vs_oracle_sample"]))'
for rep in 1 2; do
for leg in "" "--law uniform_5_20"; do
  for t in "7168,8192" "7680,8192" "6144,8192" "6656,8192" "7168,8704"; do
    seg=${t#*,}
    echo "$t | $leg"; VBT_LIB_VARIANT=s5 VBT_TIERS=$t,49152,163840 VBT_SEG_BYTES=$seg $B $leg 2>/dev/null | python -c "$pick"
  done
done
done
