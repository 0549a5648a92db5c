#!/bin/bash
# dense lexicon law: why sentences end up in the global-memory fallback, and what it costs (developer aid)
OUT=gpurun_out/dense; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VBT_DEBUG=1 timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 2 --warmup 1 --dict unidic-dense 2>&1 | grep "vbt\]" | tail -3
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o tr -- python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 3 --warmup 1 --dict unidic-dense > $OUT/log.txt 2>&1
python - <<'PY'
import csv,re,glob
f=glob.glob('gpurun_out/dense/tr/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'vbt::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
name=lambda r:(re.findall(r'(\w+)(?:<[^>]*>)?\(vbt::',r['Kernel_Name']) or ['?'])[0]
starts=[i for i,r in enumerate(rows) if name(r)=='validate_batch']
i0,i1=starts[-3],starts[-2]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    print(f"{name(r):22s} wg={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Workgroup_Size_X']:>4s} lds={r['LDS_Block_Size']:>6s} start={(int(r['Start_Timestamp'])-t0)/1e3:8.1f} end={(int(r['End_Timestamp'])-t0)/1e3:8.1f} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
PY
