"""Developer aid: find the sentences whose GPU tokens differ from the oracle and print their shape."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vibrato_amd as V
from oracle import oracle as ora
from tools import synth

shape, n, law = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sd = synth.SynthDict(shape)
do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
to = ora.Tokenizer(do, False, 0)
tv = V.Tokenizer(dv)
text, offs = sd.sentences(n, law)
exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
batch = tv.tokenize_batch(text=text, offsets=offs)
got_tok, got_off = batch.tokens_in_order()
ec, gc = np.diff(exp_off.astype(np.int64)), np.diff(got_off.astype(np.int64))
bad = np.nonzero(ec != gc)[0]
print("sentences with different token counts:", len(bad), "of", n)
for s in bad[:12]:
    nb = int(offs[s + 1] - offs[s])
    print(f"  sid {s}: bytes {nb} exp {ec[s]} got {gc[s]}")
    e = exp_tok[exp_off[s]:exp_off[s + 1]]; g = got_tok[got_off[s]:got_off[s + 1]]
    k = 0
    while k < min(len(e), len(g)) and e[k] == g[k]: k += 1
    print("    first diff at token", k, "exp", e[k] if k < len(e) else None, "got", g[k] if k < len(g) else None)
same = np.nonzero(ec == gc)[0]
nd = 0
for s in same:
    e = exp_tok[exp_off[s]:exp_off[s + 1]]; g = got_tok[got_off[s]:got_off[s + 1]]
    if not np.array_equal(e, g):
        nd += 1
        if nd <= 5:
            k = int(np.nonzero(e != g)[0][0]); print(f"  sid {s}: same count, diff at {k}: exp {e[k]} got {g[k]} bytes {int(offs[s+1]-offs[s])}")
print("same-count but different:", nd)
