#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" lp lpg; do
  echo "== variant '$v'"
  VBT_LIB_VARIANT=$v timeout 200 python tools/phase_profile.py --steps 2 2>&1 | grep -v amdgpu | tail -14
done | tee gpurun_out/r04_call14.txt
