#!/bin/bash
# lean tiers on one side stream: which sets?  headline, short sentences, config 5, dense law (dense: 10 KiB segments)
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 30 --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  pack %.4f parity %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], r["whole_path"]["pack_ms"], d["parity_vs_oracle_sample"]))'
run() { echo "VBT_TIERS=$1 seg $2 | $3"; VBT_TIERS=$1,49152,163840 VBT_SEG_BYTES=$2 $B $3 2>/dev/null | python -c "$pick"; }
for rep in 1 2; do
  for t in "6144,7680,8192" "6144,7168,8192" "5120,6144,7680,8192" "5632,6656,7680,8192" "6144,8192"; do run $t 8192 ""; done
done
for t in "6144,8192" "5120,6144,8192" "4096,6144,8192" "6144,7680,8192"; do run $t 8192 "--law uniform_5_20"; done
for t in "7680,8192" "6144,7680,8192" "5632,7680,8192"; do run $t 8192 "--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"; done
for t in "8192,10240" "6144,8192,10240" "6656,8192,10240"; do run $t 10240 "--dict unidic-dense"; done
for t in "7680,8192" "6144,7680,8192"; do run $t 8192 "--dict ipadic"; done
