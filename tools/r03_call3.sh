#!/bin/bash
# Round 3, call 3: gen_long (multi-wavefront generator for long sentences): parity on the tests that exercise it, config 5 timing,
# and the host-to-host timeline (hip + memory-copy + kernel trace).
OUT=gpurun_out/r03c; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generator_scheduling or very_long or config5 or too_dense or all_tiers or worker_single or full_size or edge_cases or group_spans" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log; tail -15 $OUT/pytest_sel.log
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
CFG5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
run cfg5_w4 $CFG5
VBT_GEN_WAVES=8 run cfg5_w8 $CFG5
VBT_GEN_WAVES=2 run cfg5_w2 $CFG5
VBT_GEN_WAVES=1 run cfg5_w1 $CFG5
run cfg5_w4_b $CFG5
run headline
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_stats -o stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $CFG5 > $OUT/cfg5_stats.log 2>&1
python - <<'PY'
import csv,re,glob
f=glob.glob('gpurun_out/r03c/cfg5_stats/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'vbt::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
name=lambda r:(re.findall(r'(\w+)(?:<[^>]*>)?\(vbt::',r['Kernel_Name']) or ['?'])[0]
starts=[i for i,r in enumerate(rows) if name(r)=='validate_batch']
i0=starts[-2]; i1=starts[-1]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    print(f"{name(r):22s} wg={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Workgroup_Size_X']:>4s} lds={r['LDS_Block_Size']:>6s} start={(int(r['Start_Timestamp'])-t0)/1e3:8.1f}us end={(int(r['End_Timestamp'])-t0)/1e3:8.1f}us dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us")
PY
cat > $OUT/h2h.py <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv, device=0)
text, offs = sd.sentences(100000, "lognormal_40")
print(tok.host_pipeline_benchmark(text, offs, threads=1, rounds=1, repeats=2))
print(tok.host_pipeline_benchmark(text, offs, threads=3, rounds=4, repeats=2))
PY
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $OUT/h2h_trace -o h2h -- python $OUT/h2h.py > $OUT/h2h_trace.log 2>&1
grep -v amdgpu.ids $OUT/h2h_trace.log | grep -v rocprofv3 | tail -3
find $OUT/h2h_trace -name '*.csv' | xargs ls -la
python - <<'PY'
import csv,glob,collections
fs=glob.glob('gpurun_out/r03c/h2h_trace/**/*memory_copy_trace.csv', recursive=True)
if fs:
    rows=list(csv.DictReader(open(fs[0])))
    print(rows[0].keys())
    big=[r for r in rows if int(r.get('Size', r.get('Bytes','0')) or 0) > 1000000]
    for r in big[-12:]:
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        sz=int(r.get('Size', r.get('Bytes','0')))
        print(r.get('Direction'), sz, f"{d:.1f}us", f"{sz/d/1e3:.1f} GB/s")
PY
