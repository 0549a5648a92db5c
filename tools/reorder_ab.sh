#!/bin/bash
# connection ids reordered by usage frequency (the reference's `reorder` + `map`, bench.py --reorder) against the dictionary as it is: headline and dense law
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for extra in "" "--dict unidic-dense"; do for ro in "" "--reorder"; do
    echo -n "$extra $ro: "
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $extra $ro 2>&1 | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('lattice ms', r['kernel_ms'], 'gen ms', r['gen_candidates']['kernel_ms'], 'step ms', d['ms_per_step'], 'parity', d['parity_vs_oracle_sample'])"
done; done | tee gpurun_out/reorder_ab.txt
