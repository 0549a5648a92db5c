#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > $OUT/h2h2.py <<'PY'
import sys, os; sys.path.insert(0, '.')
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv, device=0)
text, offs = sd.sentences(100000, "lognormal_40")
for th, rounds in ((1, 1), (3, 4), (3, 4)):
    r = tok.host_pipeline_benchmark(text, offs, threads=th, rounds=rounds, repeats=3)
    print(os.environ.get("TAG"), th, r["sentences_per_s"], r["ms_per_batch"])
PY
for w in 8 32 128 391; do TAG=packwgs$w VBT_PACK_WGS=$w python $OUT/h2h2.py 2>&1 | grep -v amdgpu; done
