# developer aid: tools/env_sweep.sh "A=1 B=2" "A=3" ...  (each spec = env assignments for one bench run; first run is a throw-away warm-up)
timeout 100 python bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
for spec in "$@"; do
  echo -n "[$spec]: "
  env $spec timeout 100 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'])"
done
