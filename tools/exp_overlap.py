#!/usr/bin/env python3
"""Experiments on overlapping batches on one GPU (negative and positive results quoted in DESIGN.md 3.5):
  parts=N : ONE batch cut into N shards, each on its own workspace + stream (gen of one under the sweep of another)
  ring=N  : N workspaces, full batches issued round-robin on N streams (consecutive batches overlap)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")
import numpy as np, torch
import vibrato_amd as V
from tools import synth
from vibrato_amd import sharding

sd = synth.SynthDict("unidic")
tok = V.Tokenizer(V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk), device=0)
text, offs = sd.sentences(100000, "lognormal_40")
d_text = torch.from_numpy(text).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()


def bench(parts, ring=1, steps=24):
    b = sharding.shard_bounds(offs, parts)
    wss, streams, args = [], [], []
    for r in range(ring):
        for i in range(parts):
            lo, hi = b[i], b[i + 1]
            nb = int(offs[hi] - offs[lo])
            wss.append(tok.workspace(hi - lo, nb)); streams.append(torch.cuda.Stream())
            args.append((d_text.data_ptr(), d_offs.data_ptr() + 8 * lo, hi - lo, nb))
    def step(k):
        r = k % ring
        for i in range(parts):
            j = r * parts + i
            wss[j].run(args[j][0], args[j][1], args[j][2], args[j][3], streams[j].cuda_stream)
    for k in range(2 * ring): step(k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps): step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    return dt * 1e3

for parts, ring in ((1, 1), (1, 2), (1, 3), (2, 1), (1, 1), (1, 2), (1, 4)):
    ms = bench(parts, ring)
    print(f"parts={parts} ring={ring}: {ms:.3f} ms/batch  {100000 / ms / 1e3:.2f} M sentences/s", flush=True)
