#!/bin/bash
# Everything the round's committed artefacts come from, in one GPU call: tools/round_artifacts.sh r05
#   parity suite, default bench line (CPU legs, suite, Worker loop, host-to-host), rocprofv3 evidence, the other BASELINE
#   configurations, config-5 kernel timeline, phase profile, host-to-host timeline, the self-launched 2-rank bench line
ulimit -c 0
TAG=${1:-r05}; OUT=gpurun_out/art_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CFG5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
if [ -z "$REST" ]; then   # REST=1: only what QUICK=1 leaves out (the two together = one full run)
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err; tail -c 300 $OUT/bench_default.err
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>/dev/null | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
run cfg2_ipadic --dict ipadic
run cfg3_unidic
run cfg5_unidic_user_S_M24_mixed $CFG5
run dense_unidic --dict unidic-dense
run unidic_short_uniform_5_20 --law uniform_5_20
timeout 200 python tools/phase_profile.py > $OUT/phase.txt 2>&1; tail -12 $OUT/phase.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
fi
[ -n "$QUICK" ] && exit 0   # QUICK=1: suite, default line, rocprofv3 evidence, the other configurations, phase profile, smoke -- and no more
# where the assembly sweep loop is parked (needs the lp variant build in vibrato_amd/lib), Worker latency by length, the EXEC = 0 probe, the RCCL overlap ratio
[ -f vibrato_amd/lib/libvibrato_hip_lp.so ] && bash tools/loop_profile.sh > $OUT/loop_profile.txt 2>&1
timeout 300 python tools/worker_latency.py 2>&1 | grep -v amdgpu > $OUT/worker_latency.txt; cat $OUT/worker_latency.txt
[ -x tools/calib/exec0_vmcnt ] && timeout 60 tools/calib/exec0_vmcnt > $OUT/exec0_vmcnt.txt 2>&1
# what the connection-matrix gather still costs (nogather / samecell variant builds), a batch of long sentences per character, the formatter call by call
[ -f vibrato_amd/lib/libvibrato_hip_nogather.so ] && bash tools/ceiling_nogather.sh > /dev/null 2>&1 && cp gpurun_out/ceiling_nogather.txt $OUT/
bash tools/long_profile.sh > /dev/null 2>&1; cp gpurun_out/long_profile.txt $OUT/ 2>/dev/null
timeout 300 python tools/format_bench.py 32 64 128 2>&1 | grep threads > $OUT/format_bench.txt; cat $OUT/format_bench.txt
# occupancy A/B: the default build (4 waves per SIMD, 10 KiB tier) against 5 waves per SIMD with an 8 KiB tier (the w5 variant build), same box
[ -f vibrato_amd/lib/libvibrato_hip_w5.so ] && bash tools/ab_variants.sh "base w5:VBT_TIERS=8192,49152,163840:VBT_SEG_BYTES=8192 base w5:VBT_TIERS=8192,49152,163840:VBT_SEG_BYTES=8192" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu > $OUT/occupancy_ab.txt; cat $OUT/occupancy_ab.txt
cp gpurun_out/rccl_ws1_overlap.json $OUT/ 2>/dev/null
# config 5, kernel by kernel
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5_stats -o stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop $CFG5 > $OUT/cfg5_stats.log 2>&1
# host-to-host pipeline: hip + memory-copy + kernel trace (no counters)
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $OUT/h2h_trace -o h2h -- python tools/h2h_bench.py 1 1 4 4 > $OUT/h2h_trace.log 2>&1
rm -f $OUT/h2h_trace/*hip_api_trace.csv   # (7 MB of API calls: not needed for the timeline)
TAG=sdma python tools/h2h_bench.py 1 1 4 4 8 4 8 4 2>&1 | grep -v amdgpu > $OUT/h2h_modes.txt
TAG=kernel_stores VBT_H2H_OUT=0 python tools/h2h_bench.py 1 1 3 4 4 4 2>&1 | grep -v amdgpu >> $OUT/h2h_modes.txt
cat $OUT/h2h_modes.txt
# bench.py --gpus 2 started WITHOUT a launcher (it starts its own ranks); one GPU on this box: both ranks on device 0, gloo instead of RCCL
VBT_BENCH_BACKEND=gloo VBT_BENCH_SINGLE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --sentences 200000 > $OUT/bench_gpus2_selflaunch_gloo.json 2>$OUT/bench_gpus2.err; tail -c 300 $OUT/bench_gpus2_selflaunch_gloo.json
