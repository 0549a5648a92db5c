#!/bin/bash
# A/B of build variants / env settings on the GPU box: tools/ab_variants.sh "<variant>[:ENV=VAL[:ENV=VAL]] ..." [bench args]
# (variants: lib/libvibrato_hip_<variant>.so built with vibrato_amd.build.build(variant=, defines=); "base" = the default build;
# the whole output of every run is kept under gpurun_out/ab_logs/)
VARS=$1; shift
export GPU_MAX_HW_QUEUES=10
mkdir -p gpurun_out/ab_logs
i=0
for spec in $VARS; do
  v=${spec%%:*}
  i=$((i+1))
  ( IFS=':' read -ra parts <<< "$spec"
    for kv in "${parts[@]:1}"; do export "$kv"; done
    if [ "$v" = "base" ]; then export VBT_LIB_VARIANT=; else export VBT_LIB_VARIANT=$v; fi
    log=gpurun_out/ab_logs/$i.log
    echo -n "== $spec : "; timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $log 2>&1
    tail -1 $log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; su=d.get('suite') or {}; print(d['value'], d['ms_per_step'], d['parity_vs_oracle_sample'], r['tiers'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'pack', r['whole_path']['pack_ms'], {k:(v.get('value'), v.get('gen_ms'), v.get('lattice_ms'), v.get('tiers'), v.get('parity_vs_oracle_sample')) for k,v in su.items()} if isinstance(su, dict) else su)" 2>/dev/null || { echo "FAILED:"; tail -5 $log | cut -c1-300; } )
done
