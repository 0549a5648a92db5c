#!/bin/bash
# first contact of a new kernel with the GPU: the EXEC=0 / vmcnt probe, then small parity batches with per-sentence diagnostics
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/calib/exec0_vmcnt 2>&1 | tee gpurun_out/exec0_vmcnt.txt
timeout 900 python tools/dbg/first_light.py 2>&1 | grep -v "amdgpu\|^$" | tee gpurun_out/first_light.txt | tail -60
