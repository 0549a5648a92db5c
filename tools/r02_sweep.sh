#!/bin/bash
# developer aid: A/B of library variants x env knobs in one box: tools/r02_sweep.sh "<variant>|ENV=.. ENV=.." ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 100 python bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
for spec in "$@"; do
  v="${spec%%|*}"; e="${spec#*|}"
  echo -n "[$spec]: "
  env VBT_LIB_VARIANT=$v $e timeout 100 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"
done
