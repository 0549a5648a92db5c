#!/bin/bash
# The slim instance of the sweep at five waves per SIMD (variant library s5: -DVBT_SLIM_WAVES=5, 96 VGPRs) with an 8 KiB segment tier
# (20 per CU) against the default (four waves, 10 KiB, 16 per CU), alternating on one box.
#   build first: python -c "from vibrato_amd.build import build; build(variant='s5', defines=['VBT_SLIM_WAVES=5'])"
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps ${STEPS:-30} --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  parity %s tiers %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], d["parity_vs_oracle_sample"], r["tiers"]))'
run() { echo "$1 | $2 | $3"; env VBT_LIB_VARIANT=$1 $2 $B $3 2>/dev/null | python -c "$pick"; }
for rep in 1 2; do
  for leg in "" "--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000" "--dict unidic-dense"; do
    run "" "A=1" "$leg"
    run s5 "A=1" "$leg"
    run s5 "VBT_TIERS=6656,8192,49152,163840 VBT_SEG_BYTES=8192" "$leg"
    run s5 "VBT_TIERS=7168,8192,49152,163840 VBT_SEG_BYTES=8192" "$leg"
  done
done
