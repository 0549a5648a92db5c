#!/bin/bash
# ceiling experiment (GPU box): the sweep without its connection-matrix gather (build variant "nogather": -DVBT_NO_GATHER=1, WRONG RESULTS by
# design; "samecell": -DVBT_NO_GATHER=2, every lane gathers cell 0) against the real kernel, headline workload and dense law.  python -c "from vibrato_amd import build as b; b.build(force=True,
# variant='nogather', defines=('VBT_NO_GATHER=1',))" first.
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for v in "" nogather samecell; do
  for extra in "" "--dict unidic-dense"; do
    echo -n "variant '$v' $extra: "
    VBT_LIB_VARIANT=$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $extra 2>&1 | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('lattice ms', r['kernel_ms'], 'gen ms', r['gen_candidates']['kernel_ms'], 'step ms', d['ms_per_step'], 'parity', d['parity_vs_oracle_sample'])"
  done
done; done | tee gpurun_out/ceiling_nogather.txt
