#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=10
run() { echo -n "== [$*] : "; env "$@" timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
for rep in 1 2; do
run VBT_LIB_VARIANT=
run VBT_LIB_VARIANT=d4
run VBT_LIB_VARIANT=d8
run VBT_LIB_VARIANT=d10
done
EXTRA="--dict unidic-dense"
run VBT_LIB_VARIANT=
run VBT_LIB_VARIANT=d8
EXTRA="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
run VBT_LIB_VARIANT=
run VBT_LIB_VARIANT=d8
