#!/bin/bash
# the whole gpu suite on the box, log under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout ${SUITE_TIMEOUT:-900} python -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-10} "$@" > gpurun_out/gpu_suite.log 2>&1
echo "suite rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gpu_suite.log | tail -20
