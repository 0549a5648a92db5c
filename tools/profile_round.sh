#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01 [extra bench args]
# 1. --kernel-trace --stats of the default bench command (per-kernel durations)
# 2. PMC passes (counters in their own runs, never combined with other trace domains)
# Every rocprofv3 run is wrapped in `timeout` (a hung counter set once cost 25 GPU-minutes).
# Output lands in gpurun_out/prof_<tag>/; tools/summarize_profile.py distils it into profiles/.
TAG=${1:-r01}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $@"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_line.json
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o pmc$i -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $@ > $OUT/pmc$i.log 2>&1
done
# 3. calibration of FETCH_SIZE / WRITE_SIZE on known byte counts in the kernels' access patterns (tools/calib/fetch_calib.hip)
if [ -x tools/calib/fetch_calib ]; then
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_r -o calib_r -- tools/calib/fetch_calib > $OUT/calib_bytes.json 2> $OUT/calib_r.log
  timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_w -o calib_w -- tools/calib/fetch_calib > /dev/null 2> $OUT/calib_w.log
fi
find $OUT -name '*.csv' | head -30
