#!/bin/bash
# A/B of build variants / env settings on the GPU box: tools/ab_variants.sh "<variant>[:ENV=VAL[:ENV=VAL]] ..." [bench args]
VARS=$1; shift
export GPU_MAX_HW_QUEUES=10
for spec in $VARS; do
  v=${spec%%:*}
  ( IFS=':' read -ra parts <<< "$spec"
    for kv in "${parts[@]:1}"; do export "$kv"; done
    if [ "$v" = "base" ]; then export VBT_LIB_VARIANT=; else export VBT_LIB_VARIANT=$v; fi
    echo -n "== $spec : "; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; su=d.get('suite') or {}; print(d['value'], d['ms_per_step'], d['parity_vs_oracle_sample'], r['tiers'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'pack', r['whole_path']['pack_ms'], {k:(v.get('value'), v.get('gen_ms'), v.get('lattice_ms'), v.get('tiers'), v.get('parity_vs_oracle_sample')) for k,v in su.items()} if isinstance(su, dict) else su)" )
done
