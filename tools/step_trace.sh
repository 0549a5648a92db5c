#!/bin/bash
# kernel timeline of one step (developer aid): bash tools/step_trace.sh <tag> [bench args]; env passes through
TAG=$1; shift
OUT=gpurun_out/trace_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o tr -- python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 4 --warmup 1 "$@" > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv,re,glob,sys
f=glob.glob(sys.argv[1]+'/tr/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'vbt::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
name=lambda r:(re.findall(r'(\w+)(?:<[^>]*>)?\(vbt::',r['Kernel_Name']) or ['?'])[0]
starts=[i for i,r in enumerate(rows) if name(r)=='validate_batch']
i0,i1=starts[-3],starts[-2]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1+1]:
    print(f"{name(r):22s} q={r.get('Queue_Id','?'):>3s} wg={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d}x{r['Workgroup_Size_X']:>4s} lds={r['LDS_Block_Size']:>6s} start={(int(r['Start_Timestamp'])-t0)/1e3:8.1f} end={(int(r['End_Timestamp'])-t0)/1e3:8.1f} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
PY
