#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4" --durations=3 2>&1 | tail -8
