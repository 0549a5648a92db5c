#!/bin/bash
# long sentences first (side streams) against the sequential plan
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-suite --no-worker-loop --no-host-pipeline --steps 20 --warmup 3"
C5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'pack', r['whole_path'].get('pack_ms'), 'tiers', r['tiers'])"; }
{
for cfg in "VBT_EARLY_LONG=1" "VBT_EARLY_LONG=0"; do
  env $cfg timeout 300 $B 2>/dev/null | pick "headline $cfg"
  env $cfg timeout 300 $B $C5 2>/dev/null | pick "cfg5 $cfg"
  env $cfg timeout 300 $B --dict unidic-dense 2>/dev/null | pick "dense $cfg"
done
bash tools/step_trace.sh a1
bash tools/step_trace.sh a5 $C5
} > gpurun_out/early.txt 2>&1
cat gpurun_out/early.txt
