#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "worker or golden" 2>&1 | tail -8
for p in 2000 0 200 20000; do VBT_WORKER_IDLE_POLLS=$p timeout 200 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv)
text, offs = sd.sentences(10000, "lognormal_40", seed=synth.SEED)
w = tok.new_worker()
w.loop_benchmark(text[:int(offs[500])], offs[:501])
r = w.loop_benchmark(text, offs, rounds=1)
print("idle_polls", os.environ["VBT_WORKER_IDLE_POLLS"], "us_per_call", round(r["us_per_call"], 2), w.path_stats())
PY
done 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call7_worker.txt
