#!/usr/bin/env python3
"""Lattice shape statistics of a bench workload, from the oracle (CPU): predecessors per sweep step (np), candidates per step
(nc), steps and nodes per sentence.  What DESIGN.md's sizing of the sweep kernel's rounds (R predecessors x 64 candidates) rests on.
usage: python tools/stats/lattice_shape.py [unidic|unidic-dense|cfg5] [n_sentences]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as ora  # noqa: E402
from tools import synth  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "unidic"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    cfg5 = which == "cfg5"
    sd = synth.SynthDict("unidic" if cfg5 else which)
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    if cfg5:
        do.reset_user_lexicon(sd.user_csv(1000))
    to = ora.Tokenizer(do, cfg5, 24 if cfg5 else 0)
    w = to.new_worker()
    text, offs = sd.sentences(n, "mixed" if cfg5 else "lognormal_40", space_p=0.1 if cfg5 else 0.0, seed=synth.SEED)
    L = ora.lib()
    L.ora_worker_lattice_shape.restype = C.c_uint32
    L.ora_worker_lattice_shape.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    nps, ncs, steps, nodes, chars, dead_sent = [], [], [], [], [], 0
    buf = text.tobytes()
    for s in range(n):
        sent = buf[int(offs[s]):int(offs[s + 1])]
        if not sent:
            continue
        w.reset_sentence(sent)
        w.tokenize()
        cap = len(sent) + 2
        a = np.zeros(cap, dtype=np.uint32)
        b = np.zeros(cap, dtype=np.uint32)
        lc = L.ora_worker_lattice_shape(w._h, a.ctypes.data, b.ctypes.data)
        vis = b[:lc + 1] > 0
        nps.extend(a[:lc + 1][vis].tolist())
        ncs.extend(b[:lc + 1][vis].tolist())
        steps.append(int(vis.sum()) + 1)
        nodes.append(int(a[:lc + 1].sum()))
        chars.append(lc)
        dead_sent += int(vis[:lc].sum() < lc)
    nps, ncs = np.array(nps), np.array(ncs)
    print(f"{which}: {len(chars)} sentences, {np.mean(chars):.1f} chars, {np.mean(nodes):.1f} nodes, {np.mean(steps):.1f} steps per sentence; "
          f"sentences with an unvisited position: {dead_sent / len(chars):.3f}")
    for name, v in (("np (predecessors per step)", nps), ("nc (candidates per step)", ncs)):
        q = np.percentile(v, [50, 75, 90, 95, 99, 100])
        print(f"  {name}: mean {v.mean():.2f}  p50/75/90/95/99/max = {q}")
    for R in (4, 6, 8, 12, 16):
        rounds = np.ceil(nps / R) * np.ceil(ncs / 64)
        print(f"  R={R:2d}: rounds per step {rounds.mean():.3f}, per sentence {rounds.sum() / len(chars):.1f}; ring slot use {nps.sum() / (rounds.sum() * R):.2f}; "
              f"units per sentence {((nps) * np.ceil(ncs / 64)).sum() / len(chars):.1f}")
    pairs = (nps * ncs).sum() / len(chars)
    print(f"  reference pairs per sentence {pairs:.0f}")


if __name__ == "__main__":
    main()
