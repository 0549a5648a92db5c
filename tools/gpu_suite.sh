#!/bin/bash
# the GPU parity suite (developer aid, through gpurun): bash tools/gpu_suite.sh  -> gpurun_out/gpu_suite.txt
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short > gpurun_out/gpu_suite.txt 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.txt
grep -n "passed\|failed\|error\|Fatal\|rc=" gpurun_out/gpu_suite.txt | head -20
