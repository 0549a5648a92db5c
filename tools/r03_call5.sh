#!/bin/bash
# Round 3, call 5: lattice_lds main-loop trims (np / nc in the pass records instead of lane masks, key byte offsets in the candidate
# records, ...): parity + timings.
OUT=gpurun_out/r03e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1 || { echo SMOKE FAILED; exit 1; }
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
CFG5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
for rep in a b; do
run headline_$rep
run cfg5_$rep $CFG5
run dense_$rep --dict unidic-dense
run short_$rep --law uniform_5_20
done
run ipadic --dict ipadic
