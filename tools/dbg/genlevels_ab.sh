cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py --no-suite --no-worker-loop --no-cpu-baseline --no-host-pipeline --steps 10 --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'])"; }
run base
VBT_GEN_LEVELS=32768,32800,163840 VBT_GEN_WAVES1=8 run merged32k_w8
VBT_GEN_LEVELS=32768,32800,163840 VBT_GEN_WAVES1=4 run merged32k_w4
VBT_GEN_LEVELS=24576,40960,163840 VBT_GEN_WAVES1=4 run l24k_40k
VBT_GEN_LEVELS=16384,32768,163840 VBT_GEN_WAVES1=8 run w1_8
VBT_GEN_LDS=6144 run genlds6k
VBT_GEN_LDS=8192 run genlds8k
run base
