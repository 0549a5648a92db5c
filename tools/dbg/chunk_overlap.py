#!/usr/bin/env python3
"""Experiment (round 6): what would cutting ONE batch into K chunks on alternating streams buy?  The generator (latency / request
bound, VALU idle) and the sweep (VALU bound) run back to back inside a batch; with chunks on two streams the generator of chunk
c + 1 runs next to the sweep of chunk c.  Emulated here with K workspaces over sentence ranges of the resident batch -- the same
launch sequences a chunked Workspace::run would enqueue -- against the one-batch step, alternating in one process.
usage (GPU box): python tools/dbg/chunk_overlap.py [shape] [n] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import vibrato_amd as V
    from tools import synth
    shape = sys.argv[1] if len(sys.argv) > 1 else "unidic"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    law = sys.argv[4] if len(sys.argv) > 4 else "lognormal_40"
    sd = synth.SynthDict(shape)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tok = V.Tokenizer(dv)
    text, offs = sd.sentences(n, law)
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    main_stream = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(2)]

    def plan(k):
        # contiguous chunks balanced by bytes
        bounds = [0]
        for c in range(1, k):
            bounds.append(int(np.searchsorted(offs, offs[-1] * c // k)))
        bounds.append(n)
        chunks = []
        for c in range(k):
            lo, hi = bounds[c], bounds[c + 1]
            nb = int(offs[hi] - offs[lo])
            chunks.append((tok.workspace(hi - lo, nb), lo, hi - lo, nb))
        return chunks

    plans = {k: plan(k) for k in (1, 2, 3, 4, 8)}

    def step(k):
        chunks = plans[k]
        if k == 1:
            ws, lo, cn, nb = chunks[0]
            ws.run(d_text.data_ptr(), d_offs.data_ptr(), cn, nb, main_stream.cuda_stream)
            return
        fork = torch.cuda.Event()
        fork.record(main_stream)
        for s in side:
            s.wait_event(fork)
        for c, (ws, lo, cn, nb) in enumerate(chunks):
            ws.run(d_text.data_ptr(), d_offs.data_ptr() + 8 * lo, cn, nb, side[c & 1].cuda_stream)
        for s in side:
            e = torch.cuda.Event()
            e.record(s)
            main_stream.wait_event(e)

    for k in plans:
        for _ in range(3):
            step(k)
    torch.cuda.synchronize()
    tot = {k: sum(int(ws.stats()["n_tokens"]) for ws, _, _, _ in plans[k]) for k in plans}
    assert len(set(tot.values())) == 1, tot
    print(f"{shape} n={n} law={law} steps={steps}: tokens per step {tot[1]}")
    for rep in range(3):
        row = []
        for k in plans:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(k)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            row.append(f"K={k}: {dt * 1e3:.4f} ms ({n / dt / 1e6:.1f} M/s)")
        print("  " + "   ".join(row))


if __name__ == "__main__":
    main()
