#!/bin/bash
# Bench lines of the other BASELINE configurations + a kernel trace of config 5 (run through gpurun):
#   tools/other_configs.sh r01   -> gpurun_out/other_<tag>/
TAG=${1:-r01}; OUT=gpurun_out/other_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | grep '^{' | tail -1 > $OUT/$name.json; }
run cfg2_ipadic --dict ipadic
run cfg3_unidic
run cfg5_unidic_user_S_M24_mixed --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 2000
run unidic_short_uniform_5_20 --law uniform_5_20
run unidic_mixed --law mixed
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_stats -o stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 2000 > $OUT/cfg5_stats.log 2>&1
ls $OUT
