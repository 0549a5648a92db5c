cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; python bench.py --no-suite --no-worker-loop --no-cpu-baseline --no-host-pipeline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r.get('tiers'))"; }
for cfg in "A|8192,10240,49152,163840|10240" "B|8192,16384,16400,49152,163840|16400" "C|8192,12288,12304,49152,163840|12304" "E|8192,16384,49152,163840|16384"; do
  IFS='|' read name tiers seg <<< "$cfg"
  export VBT_TIERS=$tiers VBT_SEG_BYTES=$seg
  run "$name headline"
  run "$name dense" --dict unidic-dense
  run "$name cfg5" --law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000
done
