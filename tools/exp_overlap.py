#!/usr/bin/env python3
"""Experiment: does running the generator of one half-batch under the lattice sweep of the other pay?
Two workspaces on two streams, each fed half of the headline batch, against one workspace with the whole batch."""
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


def bench(parts, steps=20, stagger=True):
    b = sharding.shard_bounds(offs, parts)
    wss, streams, args = [], [], []
    for i in range(parts):
        lo, hi = b[i], b[i + 1]
        nb = int(offs[hi] - offs[lo])
        ws = tok.workspace(hi - lo, nb)
        wss.append(ws); streams.append(torch.cuda.Stream())
        args.append((d_text.data_ptr(), d_offs.data_ptr() + 8 * lo, hi - lo, nb))
    def step():
        for ws, st, a in zip(wss, streams, args):
            ws.run(a[0], a[1], a[2], a[3], st.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    tot = sum(ws.stats()["n_tokens"] for ws in wss)
    return dt * 1e3, tot

for parts in (1, 2, 3, 4, 1, 2, 4, 8):
    ms, tot = bench(parts)
    print(f"parts={parts}: {ms:.3f} ms/step  {100000 / ms / 1e3:.2f} M sentences/s  tokens={tot}")
