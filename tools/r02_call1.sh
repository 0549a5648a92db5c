#!/bin/bash
# round-2 GPU call 1: new tests on the round-1 kernels + baseline numbers of the new workloads
OUT=gpurun_out/r02a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
run() { name=$1; shift; timeout 400 python bench.py --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; }
run cfg3_default
run cfg5 --no-cpu-baseline --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000
run dense --no-cpu-baseline --dict unidic-dense
for spec in "VBT_TIERS=8192,16384,163840" "VBT_TIERS=6144,10240,16384,163840" "VBT_TIERS=12288,16384,163840" "VBT_TIERS=10240,16384,163840"; do
  echo -n "[$spec]: " >> $OUT/sweep.txt
  env $spec timeout 100 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'])" >> $OUT/sweep.txt 2>&1
done
timeout 200 python tools/phase_profile.py > $OUT/phase.txt 2>&1
cat $OUT/sweep.txt; tail -15 $OUT/phase.txt
for f in cfg3_default cfg5 dense; do python -c "
import json,sys
d=json.load(open('$OUT/$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'frac', r['frac'], d['parity_vs_oracle_sample'], d.get('cpu_baseline'), d.get('cpu_baseline_all_cores'))
"; done
