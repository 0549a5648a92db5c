#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-suite --no-worker-loop --no-host-pipeline --steps 10 --warmup 2"
C5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'tiers', r['tiers'])"; }
{
for t in "10240,49152,163840" "12288,49152,163840" "13312,49152,163840" "10240,49152,163840"; do
  VBT_TIERS=$t timeout 300 $B --dict unidic-dense 2>/dev/null | pick "dense $t"
  VBT_TIERS=$t timeout 300 $B $C5 2>/dev/null | pick "cfg5 $t"
  VBT_TIERS=$t timeout 300 $B 2>/dev/null | pick "headline $t"
done
} > gpurun_out/x.txt 2>&1
cat gpurun_out/x.txt
