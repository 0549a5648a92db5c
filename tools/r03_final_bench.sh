#!/bin/bash
# the default bench line, run AFTER profiles/r03_traffic.json of the same code was committed (so that roofline.traffic / issue agree with it)
OUT=gpurun_out/art_r03; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err; tail -c 200 $OUT/bench_default.json
