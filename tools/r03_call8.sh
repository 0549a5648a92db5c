#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
timeout 300 python tools/phase_profile.py --law mixed --min-chars 500 --steps 3 2>&1 | grep -v amdgpu | tail -14
timeout 300 python tools/phase_profile.py --law mixed --max-chars 150 --steps 3 2>&1 | grep -v amdgpu | tail -14
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pool or very_long or all_tiers" 2>&1 | tail -2
