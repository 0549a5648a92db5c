#!/usr/bin/env python3
"""Host-to-host leg of the batched entry point (developer aid): TAG=<label> python tools/h2h_bench.py [threads rounds ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vibrato_amd as V  # noqa: E402
from tools import synth  # noqa: E402

sd = synth.SynthDict("unidic")
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv, device=0)
text, offs = sd.sentences(100000, "lognormal_40")
tok.calibrate(text=text, offsets=offs)  # (left to the first batch it would run in the background, next to the first measurement)
for _ in range(3):  # the first call sets the tokens-per-KiB estimate, the second one creates the chunk workspaces
    tok.tokenize_batch(text=text, offsets=offs)
plan = [int(x) for x in sys.argv[1:]] or [1, 1, 3, 4, 3, 4]
for th, rounds in zip(plan[0::2], plan[1::2]):
    r = tok.host_pipeline_benchmark(text, offs, threads=th, rounds=rounds, repeats=3)
    print(os.environ.get("TAG", ""), "threads", th, r["sentences_per_s"], "sentences/s", r["ms_per_batch"], "ms per batch", flush=True)
