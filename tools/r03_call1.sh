#!/bin/bash
# Round 3, first GPU call: parity suite, default bench line (suite + worker loop + host-to-host), the "gather removed" ceiling
# (VBT_NO_GATHER build variant, wrong results by design) on the headline and on the dense law, Worker loop with / without the spin.
OUT=gpurun_out/r03a; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err; echo "bench rc=$?"; tail -c 400 $OUT/bench_default.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r03a/bench_default.json') if l.startswith('{')][-1])
    r=d['roofline']
    print('value', d['value'], 'ms', d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], 'frac', r['frac'], 'parity', d['parity_vs_oracle_sample'])
    print('suite', json.dumps(d.get('suite'))[:1500])
    print('worker_loop', d.get('worker_loop'))
    print('h2h', d.get('host_to_host'))
    print('cpu', d.get('cpu_baseline'))
except Exception as e:
    print('no bench line', e)
PY
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-host-pipeline --no-suite --no-worker-loop --steps 10 "$@" 2>$OUT/$name.err | grep '^{' | tail -1 > $OUT/$name.json; python -c "
import json; d=json.load(open('$OUT/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], 'gen', r['gen_candidates']['kernel_ms'], 'lat', r['kernel_ms'], d['parity_vs_oracle_sample'], r['tiers'])"; }
run headline_a
VBT_LIB_VARIANT=nogather run headline_nogather_a
run headline_b
VBT_LIB_VARIANT=nogather run headline_nogather_b
run dense_a --dict unidic-dense
VBT_LIB_VARIANT=nogather run dense_nogather_a --dict unidic-dense
run dense_b --dict unidic-dense
VBT_LIB_VARIANT=nogather run dense_nogather_b --dict unidic-dense
python - <<'PY' > gpurun_out/r03a/worker_loop.txt 2>&1
import os, json, time
import numpy as np
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv, device=0)
text, offs = sd.sentences(10000, "lognormal_40")
for spin in ("1", "0"):
    os.environ["VBT_WORKER_SPIN"] = spin
    w = tok.new_worker()
    w.loop_benchmark(text[:int(offs[500])], offs[:501])
    for rep in range(3):
        print("spin", spin, w.loop_benchmark(text, offs), w.path_stats())
# latency by sentence length
for lo, hi in ((1, 10), (10, 20), (20, 40), (40, 80), (80, 160), (160, 400)):
    lens = np.diff(offs).astype(np.int64) / 2.85
    keep = np.nonzero((lens >= lo) & (lens < hi))[0][:1500]
    parts = [text[int(offs[i]):int(offs[i + 1])] for i in keep]
    t = np.concatenate(parts); o = np.zeros(len(keep) + 1, dtype=np.uint64); o[1:] = np.cumsum([len(p) for p in parts])
    os.environ["VBT_WORKER_SPIN"] = "1"
    w = tok.new_worker(); w.loop_benchmark(t[:int(o[50])], o[:51])
    print("chars", lo, hi, "n", len(keep), w.loop_benchmark(t, o))
PY
cat gpurun_out/r03a/worker_loop.txt | grep -v amdgpu.ids
