#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-suite --no-worker-loop --no-host-pipeline --steps 20 --warmup 3"
C5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'pack', r['whole_path'].get('pack_ms'))"; }
{
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for e in 2 0; do
  VBT_EARLY_LONG=$e timeout 300 $B 2>/dev/null | pick "headline early=$e"
done
done
for e in 2 0; do
  VBT_EARLY_LONG=$e timeout 300 $B $C5 2>/dev/null | pick "cfg5 early=$e"
  VBT_EARLY_LONG=$e timeout 300 $B --dict unidic-dense 2>/dev/null | pick "dense early=$e"
done
bash tools/step_trace.sh z1
} > gpurun_out/x.txt 2>&1
cat gpurun_out/x.txt
