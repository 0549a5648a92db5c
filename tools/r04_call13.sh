#!/bin/bash
# the sweep loop in assembly: canary parity, then same-box A/B against the committed loop (old) and the C++ loop at depth 2 (cpp)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/dbg/first_light.py 2>&1 | grep "mismatching" | cut -c1-150
bash tools/ab_variants.sh "${1:-old base cpp old base}" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call13_ab.txt
