#!/bin/bash
# round 4, GPU call 4: canary first (small batches, then one full-size bench), then the parity suite, then A/B
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/dbg/first_light.py 2>&1 | grep "mismatching" | tee gpurun_out/r04_call4_first.txt
bash tools/ab_variants.sh "base" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call4_ab.txt
grep -q "True" gpurun_out/r04_call4_ab.txt || { echo "canary failed: skipping the rest"; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r04_call4_pytest.txt
bash tools/ab_variants.sh "r8 d8 base r8" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee -a gpurun_out/r04_call4_ab.txt
