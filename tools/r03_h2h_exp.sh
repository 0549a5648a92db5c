#!/bin/bash
# host-to-host leg: what bounds it?  (a) hardware queues, (b) without the packing kernel (results stay on the device: timing only)
OUT=gpurun_out/r03f; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > $OUT/h2h.py <<'PY'
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
# half-size batches: more, shorter pipeline stages
t2, o2 = sd.sentences(50000, "lognormal_40")
r = tok.host_pipeline_benchmark(t2, o2, threads=4, rounds=8, repeats=3)
print(os.environ.get("TAG"), "50k x4", r["sentences_per_s"], r["ms_per_batch"])
PY
for q in 4 10 24; do TAG=queues$q GPU_MAX_HW_QUEUES=$q python $OUT/h2h.py 2>&1 | grep -v amdgpu; done
TAG=nopack VBT_H2H_NO_PACK=1 python $OUT/h2h.py 2>&1 | grep -v amdgpu
TAG=sdma0 HSA_ENABLE_SDMA=0 python $OUT/h2h.py 2>&1 | grep -v amdgpu
