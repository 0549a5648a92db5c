#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
OUT=gpurun_out/r03h; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
run dense --dict unidic-dense
run headline
run cfg5 --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000
run dense_b --dict unidic-dense
run headline_b
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
