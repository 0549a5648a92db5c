#!/bin/bash
# two lean tiers in front of the 8 KiB segment tier (final build of round 6): does the second lean tier still cost the generator what it saves the sweep?
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 30 --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  pack %.4f parity %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], r["whole_path"]["pack_ms"], d["parity_vs_oracle_sample"]))'
for rep in 1 2 3; do
  for t in "7680,8192" "6144,7680,8192" "5632,7680,8192" "6656,7936,8192"; do
    echo "VBT_TIERS=$t"; VBT_TIERS=$t,49152,163840 VBT_SEG_BYTES=8192 $B 2>/dev/null | python -c "$pick"
  done
done
