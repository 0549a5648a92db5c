#!/bin/bash
# A/B of VBT_FILL_ROUNDS (gen_one's hit expansion: rounds of 64 hits requested together): variant libraries f1 / f2 / f4, alternating on one box.
#   build first:  python -c "from vibrato_amd.build import build; [build(variant='f%d' % r, defines=['VBT_FILL_ROUNDS=%d' % r]) for r in (1, 2, 4)]"
B="python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 20 --warmup 3"
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("   %.2f M/s  step %.4f ms  sweep %.4f  gen %.4f  parity %s" % (d["value"]/1e6, d["ms_per_step"], r["kernel_ms"], r["gen_candidates"]["kernel_ms"], d["parity_vs_oracle_sample"]))'
for rep in 1 2; do
  for v in ${VARIANTS:-f1 f2 f4}; do
    echo "$v headline"; VBT_LIB_VARIANT=$v $B 2>/dev/null | python -c "$pick"
  done
done
for v in ${VARIANTS:-f1 f2 f4}; do
  echo "$v cfg5"; VBT_LIB_VARIANT=$v $B --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>/dev/null | python -c "$pick"
  echo "$v dense"; VBT_LIB_VARIANT=$v $B --dict unidic-dense 2>/dev/null | python -c "$pick"
done
