#!/bin/bash
# round 4, GPU call 2: the full parity suite on the rebuilt sweep, then bench A/B of the build variants (units per pass, dummy loads)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04_call2_pytest.txt
bash tools/ab_variants.sh "base r8 lane1 r8d8 base r8" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call2_ab.txt
