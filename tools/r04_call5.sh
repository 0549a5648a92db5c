#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/phase_profile.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call5_phase.txt
bash tools/profile_round.sh r04a > gpurun_out/r04_call5_profile.log 2>&1
python tools/pmc_show.py gpurun_out/prof_r04a 2>&1 | tail -40 | tee gpurun_out/r04_call5_pmc.txt
