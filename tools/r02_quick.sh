#!/bin/bash
# quick GPU check: parity tests + headline bench (+ optional extra bench args as further runs)
OUT=gpurun_out/${1:-r02q}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
show() { python -c "
import json,sys
d=json.load(open('$1')); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'frac', r['frac'], d['parity_vs_oracle_sample'], r['tiers'])
"; }
timeout 300 python bench.py --steps 10 --no-cpu-baseline 2>$OUT/cfg3.err | grep '^{' | tail -1 > $OUT/cfg3.json; show $OUT/cfg3.json
timeout 300 python bench.py --steps 10 --no-cpu-baseline --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>$OUT/cfg5.err | grep '^{' | tail -1 > $OUT/cfg5.json; show $OUT/cfg5.json
timeout 300 python bench.py --steps 10 --no-cpu-baseline --dict unidic-dense 2>$OUT/dense.err | grep '^{' | tail -1 > $OUT/dense.json; show $OUT/dense.json
