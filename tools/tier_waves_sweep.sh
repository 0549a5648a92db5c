# developer aid: VBT_TIER_WAVES / VBT_STEAL sweep on the GPU box
for spec in ":0" "1121,588,321,256,71,49,14,25:0" "1121,588,321,256,71,49,14,25:1" "1000,550,320,280,100,70,30,60:0" "1000,550,320,280,100,70,30,60:1" "1400,700,400,300,90,60,20,30:0" ":1"; do
  tw=${spec%%:*}; st=${spec##*:}
  echo -n "tw=[$tw] steal=$st: "
  VBT_STEAL=$st VBT_TIER_WAVES=$tw timeout 100 python bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity_vs_oracle_sample'])"
done
