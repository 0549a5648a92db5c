#!/bin/bash
OUT=gpurun_out/r02i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 10 --no-cpu-baseline --host-pipeline 2>$OUT/h2h.err | grep '^{' | tail -1 > $OUT/h2h.json
python -c "
import json; d=json.load(open('$OUT/h2h.json')); print(d['value'], d['ms_per_step'], d['host_to_host'])"
for spec in "4 2" "8 2" "8 3" "8 4" "16 4" "16 6" "32 8" "2 2" "1 1"; do set -- $spec
python - <<PY
import numpy as np, vibrato_amd as V
from tools import synth
sd=synth.SynthDict("unidic")
tok=V.Tokenizer(V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk), device=0)
text,offs=sd.sentences(100000,"lognormal_40")
print("$spec", tok.host_pipeline_benchmark(text, offs, n_batches=$1, threads=$2, repeats=5))
PY
done 2>&1 | grep -v Warning | tee $OUT/h2h_sweep.txt
timeout 300 python bench.py --steps 10 --no-cpu-baseline --dict ipadic 2>$OUT/ipadic.err | grep '^{' | tail -1 > $OUT/ipadic.json
python -c "
import json; d=json.load(open('$OUT/ipadic.json')); r=d['roofline']; print('ipadic', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'])"
