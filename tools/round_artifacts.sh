#!/bin/bash
# Everything the round's committed artefacts come from, in one GPU call: tools/round_artifacts.sh r02
#   parity suite, default bench line (CPU legs + host-to-host leg), rocprofv3 evidence, the other BASELINE configurations
TAG=${1:-r02}; OUT=gpurun_out/art_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --steps 10 "$@" 2>/dev/null | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
run cfg2_ipadic --dict ipadic
run cfg3_unidic
run cfg5_unidic_user_S_M24_mixed --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000
run dense_unidic --dict unidic-dense
run unidic_short_uniform_5_20 --law uniform_5_20
run cfg3_unidic_reordered --reorder
timeout 200 python tools/phase_profile.py > $OUT/phase.txt 2>&1; tail -12 $OUT/phase.txt
