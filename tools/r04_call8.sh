#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/dbg/first_light.py 2>&1 | grep "mismatching" | cut -c1-150
timeout 200 python tools/phase_profile.py 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call8_phase.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "worker" 2>&1 | tail -3
for p in 2000 0; do VBT_WORKER_IDLE_POLLS=$p timeout 200 python - <<'PY'
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
if os.environ["VBT_WORKER_IDLE_POLLS"] == "2000":
    lens = np.diff(offs)
    for lo, hi in ((0, 40), (40, 100), (100, 200), (200, 400), (400, 10**9)):
        idx = np.nonzero((lens >= lo) & (lens < hi))[0][:1500]
        if len(idx) < 20: continue
        parts = [text[int(offs[i]):int(offs[i + 1])] for i in idx]
        t = np.concatenate(parts); o = np.zeros(len(idx) + 1, dtype=np.uint64); o[1:] = np.cumsum([len(p) for p in parts])
        r = w.loop_benchmark(t, o, rounds=1)
        print("   bytes", lo, hi, "n", len(idx), "mean chars", round(float(np.mean([len(p) for p in parts])) / 3, 1), "us_per_call", round(r["us_per_call"], 2))
    b = tok.tokenize_batch(text=text, offsets=offs)
    import time
    for th in ("8", "16", "32", "64", "128"):
        os.environ["VBT_FORMAT_THREADS"] = th
        b.format_bytes("mecab")
        ts = [b.format_bytes("mecab")[1] for _ in range(5)]
        print("   format threads", th, "ms for 10k sentences", round(min(ts) * 1e3, 3))
PY
done 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call8_worker.txt
