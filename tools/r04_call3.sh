#!/bin/bash
# round 4, GPU call 3: records through the scalar cache -- parity suite, then A/B of units per pass / depth
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04_call3_pytest.txt
for v in r8 r8d8; do VBT_LIB_VARIANT=$v timeout 300 python tools/dbg/first_light.py 2>&1 | grep "mismatching" | sed "s/^/[$v] /"; done | tee gpurun_out/r04_call3_variants_parity.txt
bash tools/ab_variants.sh "base r8 r8d8 d4 d8 base r8" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call3_ab.txt
