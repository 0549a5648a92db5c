#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 1 0 1; do TAG=out$m VBT_H2H_OUT=$m python tools/h2h_bench.py 1 1 3 4 3 4 4 4 2>&1 | grep -v amdgpu; done
