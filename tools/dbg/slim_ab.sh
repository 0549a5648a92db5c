cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py --no-suite --no-worker-loop --no-cpu-baseline --no-host-pipeline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r.get('tiers'))"; }
for v in 0 1 0 1; do
  export VBT_SLIM=$v
  run "slim=$v headline"
  run "slim=$v dense" --dict unidic-dense
  run "slim=$v cfg5" --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000
done
