#!/bin/bash
OUT=gpurun_out/art_r03; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -1
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $OUT/h2h_trace -o h2h -- python tools/h2h_bench.py 1 1 4 4 > $OUT/h2h_trace.log 2>&1
rm -f $OUT/h2h_trace/*hip_api_trace.csv
grep -v amdgpu $OUT/h2h_trace.log | grep threads
TAG=sdma python tools/h2h_bench.py 1 1 4 4 2>&1 | grep threads
CFG5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_stats -o stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $CFG5 > $OUT/cfg5_stats.log 2>&1
grep '^{' $OUT/cfg5_stats.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('cfg5', d['value'], d['ms_per_step'], r['gen_candidates']['kernel_ms'], r['kernel_ms'])"
timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 $CFG5 2>/dev/null | grep '^{' | tail -1 > $OUT/cfg5_unidic_user_S_M24_mixed.json
python -c "import json; d=json.load(open('$OUT/cfg5_unidic_user_S_M24_mixed.json')); r=d['roofline']; print('cfg5', d['value'], d['ms_per_step'], r['gen_candidates']['kernel_ms'], r['kernel_ms'], d['parity_vs_oracle_sample'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generator_scheduling or very_long or config5 or too_dense" 2>&1 | tail -2
