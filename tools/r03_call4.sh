#!/bin/bash
# Round 3, call 4: full parity suite (host batch path now packs straight into pinned memory; generator levels side by side),
# host-to-host leg, config 5 / dense law with smaller first escape tiers.
OUT=gpurun_out/r03d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv, device=0)
text, offs = sd.sentences(100000, "lognormal_40")
for th, rounds in ((1, 1), (2, 4), (3, 4), (4, 4), (3, 4), (1, 1)):
    r = tok.host_pipeline_benchmark(text, offs, threads=th, rounds=rounds, repeats=3)
    print(th, rounds, r["sentences_per_s"], r["ms_per_batch"], r["pool"])
PY
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
CFG5="--law mixed --ignore-space --max-grouping-len 24 --user-lexicon 1000"
for rep in a b; do
run cfg5_t48_$rep $CFG5
VBT_TIERS=10240,32768,163840 run cfg5_t32_$rep $CFG5
VBT_TIERS=10240,24576,163840 run cfg5_t24_$rep $CFG5
VBT_TIERS=10240,16384,163840 run cfg5_t16_$rep $CFG5
run dense_t48_$rep --dict unidic-dense
VBT_TIERS=10240,32768,163840 run dense_t32_$rep --dict unidic-dense
VBT_TIERS=10240,24576,163840 run dense_t24_$rep --dict unidic-dense
done
run headline
