cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py --no-suite --no-worker-loop --no-cpu-baseline --no-host-pipeline --steps 10 --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'])"; }
run base
VBT_GEN_WAVES=16 run w16
VBT_GEN_WAVES=12 run w12
VBT_GEN_WAVES=16 VBT_GEN_WAVES1=8 run w16_w1_8
VBT_GEN_LEVELS=16384,49152,163840 VBT_GEN_WAVES=16 run l48k_w16
run base
