#!/bin/bash
# GPU box: what the sweep of a long sentence costs per character inside a full batch of them (phase profile of 2 000 sentences of >= 1000 characters,
# law "mixed", config 5's options) -- the figure DESIGN.md section 8.2 quotes
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/phase_profile.py --law mixed --ignore-space --mgl 24 --user 1000 --space-p 0.1 --min-chars 1000 --sentences 100000 --keep 2000 2>&1 | grep -v amdgpu | tee gpurun_out/long_profile.txt
