#!/bin/bash
# round-2 GPU call: full parity suite, default bench line, rocprofv3 evidence (stats + PMC passes), reorder A/B
OUT=gpurun_out/r02h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err; tail -c 600 $OUT/bench_default.err
bash tools/profile_round.sh r02
show() { python -c "
import json,sys
d=json.load(open('$1')); r=d['roofline']
print('$1', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'frac', r['frac'], d['parity_vs_oracle_sample'], r['tiers'], d['config']['connection_ids_reordered'])
"; }
for i in 1 2; do
timeout 300 python bench.py --steps 10 --no-cpu-baseline 2>$OUT/a$i.err | grep '^{' | tail -1 > $OUT/plain$i.json; show $OUT/plain$i.json
timeout 300 python bench.py --steps 10 --no-cpu-baseline --reorder 2>$OUT/b$i.err | grep '^{' | tail -1 > $OUT/reorder$i.json; show $OUT/reorder$i.json
done
timeout 300 python bench.py --steps 10 --no-cpu-baseline --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>$OUT/cfg5.err | grep '^{' | tail -1 > $OUT/cfg5.json; show $OUT/cfg5.json
timeout 300 python bench.py --steps 10 --no-cpu-baseline --dict unidic-dense 2>$OUT/dense.err | grep '^{' | tail -1 > $OUT/dense.json; show $OUT/dense.json
timeout 300 python bench.py --steps 10 --no-cpu-baseline --dict unidic-dense --reorder 2>$OUT/dense_r.err | grep '^{' | tail -1 > $OUT/dense_reorder.json; show $OUT/dense_reorder.json
