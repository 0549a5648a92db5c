#!/bin/bash
# what the driver runs at round end: the GPU parity suite and the smoke test (developer aid: bash tools/gpu_check.sh through gpurun)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
