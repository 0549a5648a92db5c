"""Stress vbt_tokenize_batch from several host threads (stderr visible, outside pytest)."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.enable()
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import vibrato_amd as V
from tools import synth

sd = synth.SynthDict("small")
tv = V.Tokenizer(V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk), device=0)
text, offs = sd.sentences(6000, "lognormal_40")
ref, ref_off = tv.tokenize_batch(text=text, offsets=offs).tokens_in_order()
bounds = [0, 700, 701, 2500, 2500, 4100, 6000]

def one(i):
    lo, hi = bounds[i], bounds[i + 1]
    got, got_off = tv.tokenize_batch(text=text, offsets=offs[lo:hi + 1]).tokens_in_order()
    assert got.tobytes() == ref[int(ref_off[lo]):int(ref_off[hi])].tobytes()
    return 1

for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(one, range(len(bounds) - 1)))
    if rep % 20 == 0:
        print("rep", rep, tv.pool_stats(), flush=True)
print("ok")
