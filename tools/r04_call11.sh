#!/bin/bash
# where an iteration of the sweep loop is parked: phase profile of the loop-profiling variants (slot "pass_records"/5 = top wait, "end_lists"/3 = landed wait)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in lp; do
  echo "== variant $v"
  VBT_LIB_VARIANT=$v timeout 200 python tools/phase_profile.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/r04_call11_loopprof.txt
