#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30
