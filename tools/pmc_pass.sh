#!/bin/bash
# ad-hoc PMC passes (developer aid): tools/pmc_pass.sh <tag> "<counter set>" ["<counter set>" ...]
# every rocprofv3 run is wrapped in `timeout` (a hung counter set once cost 25 GPU-minutes)
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed/timeout"
done
