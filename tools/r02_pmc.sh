#!/bin/bash
# two PMC passes over the default bench (SQ instruction mix / wave time split / LDS counters): tools/r02_pmc.sh <tag>
TAG=${1:-x}; OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_show.py $TAG 2>&1 | tail -60
