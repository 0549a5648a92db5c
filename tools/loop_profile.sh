#!/bin/bash
# developer aid, GPU box: where the assembly sweep loop is parked -- phase profile of a -DVBT_LOOP_PROF=1 variant build ("lp":
# python -c "from vibrato_amd import build as b; b.build(force=True, variant='lp', defines=('VBT_LOOP_PROF=1',))")
# lp  (VBT_LOOP_PROF=1): slot "pass_records" = cycles at the top-of-iteration vmcnt wait, "end_lists" = at the LDS wait, per sentence
# lp2 (VBT_LOOP_PROF=2): "pass_records" / "end_lists" = cycles of the iterations that took the common pass / of the others, per sentence;
#     their counts per sentence ride on top of "decode" / "count" (compare with the plain build's)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "" lp lp2; do
  echo "== variant '$v'"
  VBT_LIB_VARIANT=$v timeout 200 python tools/phase_profile.py --steps 2 "$@" 2>&1 | grep -v amdgpu | tail -12
done | tee gpurun_out/loop_profile.txt
