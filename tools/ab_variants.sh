#!/bin/bash
# A/B of build variants on the GPU box: tools/ab_variants.sh "<variant> <variant> ..." [bench args]
VARS=$1; shift
for v in $VARS; do
  if [ "$v" = "base" ]; then export VBT_LIB_VARIANT=; else export VBT_LIB_VARIANT=$v; fi
  echo "== $v"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity_vs_oracle_sample'], d['roofline']['tiers'])"
done
