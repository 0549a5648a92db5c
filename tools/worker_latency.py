#!/usr/bin/env python3
"""GPU box: latency of Worker::tokenize through the resident kernel (and the launch-per-call form, VBT_WORKER_IDLE_POLLS=0) by
sentence length, and the MeCab formatter by thread count.  usage: python tools/worker_latency.py  (profiles/r04_worker_latency.txt)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(polls):
    os.environ["VBT_WORKER_IDLE_POLLS"] = polls
    import vibrato_amd as V
    from tools import synth
    sd = synth.SynthDict("unidic")
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tok = V.Tokenizer(dv)
    text, offs = sd.sentences(10000, "lognormal_40", seed=synth.SEED)
    w = tok.new_worker()
    w.loop_benchmark(text[:int(offs[500])], offs[:501])
    r = w.loop_benchmark(text, offs, rounds=1)
    print("idle_polls", polls, "us_per_call", round(r["us_per_call"], 2), "(single-launch / resident path, batch pipeline):", w.path_stats())
    if polls == "0":
        return
    lens = np.diff(offs)
    for lo, hi in ((0, 40), (40, 100), (100, 200), (200, 400), (400, 10**9)):
        idx = np.nonzero((lens >= lo) & (lens < hi))[0][:1500]
        if len(idx) < 20:
            continue
        parts = [text[int(offs[i]):int(offs[i + 1])] for i in idx]
        t = np.concatenate(parts)
        o = np.zeros(len(idx) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(p) for p in parts])
        r = w.loop_benchmark(t, o, rounds=1)
        print("   bytes", lo, hi, "n", len(idx), "mean chars", round(float(np.mean([len(p) for p in parts])) / 3, 1), "us_per_call", round(r["us_per_call"], 2))
    b = tok.tokenize_batch(text=text, offsets=offs)
    for th in ("8", "32", "128"):
        os.environ["VBT_FORMAT_THREADS"] = th
        b.format_bytes("mecab")
        ts = [b.format_bytes("mecab")[1] for _ in range(5)]
        print("   format threads", th, "ms for 10k sentences", round(min(ts) * 1e3, 3))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(sys.argv[1])
    else:
        for p in ("2000", "0"):
            subprocess.call([sys.executable, os.path.abspath(__file__), p])
