#!/bin/bash
# Round 3, call 2: (a) what occupancy buys lattice_lds (build variants with fewer VGPRs / shallower gather ring + smaller LDS tier,
# on the short-sentence law where nothing is segmented, and on the headline); (b) Worker loop before / after; (c) timeline of the
# host-to-host pipeline (hip + memory-copy + kernel trace, no counters).
OUT=gpurun_out/r03b; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
for rep in a b; do
  for law in uniform_5_20 lognormal_40; do
    VBT_TIERS=10240,49152,163840 run ${law}_base_t10_$rep --law $law
    VBT_TIERS=8192,49152,163840 VBT_SEG_BYTES=8192 run ${law}_base_t8_$rep --law $law
    VBT_LIB_VARIANT=w5d3 VBT_TIERS=8192,49152,163840 VBT_SEG_BYTES=8192 run ${law}_w5d3_t8_$rep --law $law
    VBT_LIB_VARIANT=w5d3 VBT_TIERS=10240,49152,163840 run ${law}_w5d3_t10_$rep --law $law
    VBT_LIB_VARIANT=w6d2 VBT_TIERS=6656,49152,163840 VBT_SEG_BYTES=6656 run ${law}_w6d2_t6_$rep --law $law
  done
done
python - <<'PY' > $OUT/worker_before_after.txt 2>&1
import os
import numpy as np
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv, device=0)
text, offs = sd.sentences(3000, "lognormal_40")
for single in ("0", "1"):
    os.environ["VBT_WORKER_SINGLE"] = single
    w = tok.new_worker()
    w.loop_benchmark(text[:int(offs[200])], offs[:201])
    for rep in range(2):
        print("VBT_WORKER_SINGLE", single, w.loop_benchmark(text, offs), w.path_stats())
PY
grep -v amdgpu.ids $OUT/worker_before_after.txt
cat > /tmp/h2h.py <<'PY'
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
timeout 300 rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --output-format csv -d $OUT/h2h_trace -o h2h -- python /tmp/h2h.py > $OUT/h2h_trace.log 2>&1
grep -v amdgpu.ids $OUT/h2h_trace.log | tail -3
find $OUT/h2h_trace -name '*.csv' | xargs ls -la
