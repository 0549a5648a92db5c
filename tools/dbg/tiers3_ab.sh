cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py --no-suite --no-worker-loop --no-cpu-baseline --no-host-pipeline --steps 20 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'pack', r['whole_path']['pack_ms'], d['parity_vs_oracle_sample'], r.get('tiers'))"; }
for rep in 1 2; do
for t in "8192,10240,49152,163840" "8192,10240,163840" "8192,10240,65536"; do
  export VBT_TIERS=$t
  run "$t headline"
done
done
export VBT_TIERS=8192,10240,163840
run "3tiers cfg5" --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000
run "3tiers dense" --dict unidic-dense
