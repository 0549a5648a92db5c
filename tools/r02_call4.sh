#!/bin/bash
OUT=gpurun_out/r02j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 10 --no-cpu-baseline --host-pipeline 2>$OUT/h2h.err | grep '^{' | tail -1 > $OUT/h2h.json
python -c "
import json; d=json.load(open('$OUT/h2h.json')); print(d['value'], d['ms_per_step'], json.dumps(d['host_to_host'], indent=1))"
for spec in "1 4" "2 4" "3 4" "4 4"; do set -- $spec
python - <<PY
import numpy as np, vibrato_amd as V
from tools import synth
sd=synth.SynthDict("unidic")
tok=V.Tokenizer(V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk), device=0)
text,offs=sd.sentences(100000,"lognormal_40")
print("$spec", tok.host_pipeline_benchmark(text, offs, threads=$1, rounds=$2, repeats=4))
PY
done 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $OUT/h2h_sweep.txt
timeout 200 python tools/phase_profile.py > $OUT/phase.txt 2>&1; tail -25 $OUT/phase.txt
