#!/bin/bash
# quick look: canary parity + one bench line of the default build (+ optional variants: tools/quick_ab.sh "base lp ...": variants built with vibrato_amd.build.build(variant=, defines=))
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/dbg/first_light.py 2>&1 | grep "mismatching" | cut -c1-150
bash tools/ab_variants.sh "${1:-base}" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee gpurun_out/quick_ab.txt
