#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-suite --no-worker-loop --no-host-pipeline --steps 20 --warmup 3"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'tiers', r['tiers'])"; }
{
for rep in 1 2; do
for e in 0 128 192 256; do
  VBT_LEN_BUCKETS=$e timeout 300 $B 2>/dev/null | pick "headline buckets=$e"
done
done
} > gpurun_out/x.txt 2>&1
cat gpurun_out/x.txt
