#!/bin/bash
# gpu suite on the default build (optional: SUITE=1), then an A/B of build variants on the same box: tools/gpu_ab.sh "<variants>" [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "${SUITE:-0}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -q --maxfail=5 > gpurun_out/gpu_suite.log 2>&1; echo "suite rc=$?"
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gpu_suite.log | tail -8
fi
V=$1; shift
bash tools/ab_variants.sh "$V" --no-host-pipeline --no-worker-loop "$@" 2>&1 | grep -v amdgpu | tee gpurun_out/ab.txt
