#!/bin/bash
# kernel timeline of one bench step (developer aid)
OUT=gpurun_out/timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
python - <<'PY'
import csv,re
rows=[r for r in csv.DictReader(open('gpurun_out/timeline/tl_kernel_trace.csv')) if 'vbt::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
big=max(int(r['Grid_Size_X']) for r in rows)
starts=[i for i,r in enumerate(rows) if int(r['Grid_Size_X'])==big]
i0=starts[-1]; per=starts[-1]-starts[-2]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+per]:
    name=re.findall(r'(\w+)(?:<[^>]*>)?\(vbt::',r['Kernel_Name'])[0]
    print(f"{name:16s} waves={int(r['Grid_Size_X'])//64:6d} lds={r['LDS_Block_Size']:>6s} start={(int(r['Start_Timestamp'])-t0)/1e3:8.1f}us end={(int(r['End_Timestamp'])-t0)/1e3:8.1f}us dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us")
PY
