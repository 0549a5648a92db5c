#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
CFG5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
for rep in a b; do
run cfg5_prio_$rep $CFG5
VBT_TIER_PRIO=0 run cfg5_noprio_$rep $CFG5
run headline_prio_$rep
VBT_TIER_PRIO=0 run headline_noprio_$rep
run dense_prio_$rep --dict unidic-dense
VBT_TIER_PRIO=0 run dense_noprio_$rep --dict unidic-dense
done
