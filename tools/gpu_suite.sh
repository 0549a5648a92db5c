#!/bin/bash
# the GPU parity suite (developer aid, through gpurun): bash tools/gpu_suite.sh [repeats]  -> gpurun_out/gpu_suite_<i>.txt
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in $(seq 1 ${1:-1}); do
  timeout 900 python -X faulthandler -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_suite_$i.txt 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite_$i.txt
  grep -n "passed\|failed\|error\|Fatal\|rc=" gpurun_out/gpu_suite_$i.txt | head -6
done
