#!/usr/bin/env python3
"""Developer aid: sentences of 8 000 - 11 000 characters (the `exact` instance of lattice_lds) against the oracle, with the first
differing token of each and the tiers they took.  usage (GPU box): python tools/dbg/long_exact.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vibrato_amd as V  # noqa: E402
from oracle import oracle as ora  # noqa: E402
from tools import synth  # noqa: E402

sd = synth.SynthDict(os.environ.get("DICT", "small"))
do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
to, tv = ora.Tokenizer(do), V.Tokenizer(dv)
base, offs0 = sd.sentences(1200, "lognormal_40")
raw = bytes(base)
chars = np.cumsum([len(raw[int(offs0[i]):int(offs0[i + 1])].decode("utf-8")) for i in range(1200)])
targets = [int(x) for x in (sys.argv[1:] or ["7000", "7900", "8200", "9000", "10500"])]
w0 = to.new_worker()
enc = [raw[:int(offs0[int(np.searchsorted(chars, c)) + 1])] for c in targets]
offs = np.zeros(len(enc) + 1, dtype=np.uint64)
offs[1:] = np.cumsum([len(e) for e in enc])
text = np.frombuffer(b"".join(enc), dtype=np.uint8)
exp, eoff = to.new_worker().tokenize_batch(text, offs)
got, goff = tv.tokenize_batch(text=text, offsets=offs).tokens_in_order()
ws = tv.workspace(len(enc), len(text))
dt = torch.from_numpy(text.copy()).cuda()
do_ = torch.from_numpy(offs.astype(np.int64)).cuda()
ws.run(dt.data_ptr(), do_.data_ptr(), len(enc), len(text), 0)
print("stats", ws.stats())
for s in range(len(enc)):
    e = exp[int(eoff[s]):int(eoff[s + 1])]
    g = got[int(goff[s]):int(goff[s + 1])]
    k = 0
    while k < min(len(e), len(g)) and e[k].tobytes() == g[k].tobytes():
        k += 1
    n = len(enc[s].decode("utf-8"))
    w0.reset_counters(); w0.tokenize_batch(np.frombuffer(enc[s], dtype=np.uint8), np.array([0, len(enc[s])], dtype=np.uint64), counted=True)
    print("   oracle nodes", w0.counters()["n_nodes"], end=" ")
    print(f"sentence {s}: {n} chars, tokens exp {len(e)} got {len(g)}; first diff at token {k}" + (f": exp {e[k]} got {g[k]}" if k < min(len(e), len(g)) else " (none within the common prefix)"))
