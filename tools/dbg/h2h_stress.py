#!/usr/bin/env python3
"""Stress of the host-to-host entry point with small batches (developer aid, round 6): vbt_tokenize_batch over random slices of a
sentence set, every result compared with the oracle's records for the same slice; a fresh tokenizer every `renew` calls.
usage (GPU box): python tools/dbg/h2h_stress.py [calls] [renew]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import vibrato_amd as V
    from oracle import oracle as ora
    from tools import synth
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    renew = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    sd = synth.SynthDict("small")
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    to = ora.Tokenizer(do, False, 0)
    text, offs = sd.sentences(3000, "lognormal_40")
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    raw = text.tobytes()
    lines = [raw[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    rng = np.random.default_rng(5)
    bad = 0
    tv = None
    for c in range(calls):
        if c % renew == 0:
            dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
            tv = V.Tokenizer(dv)
        n = int(rng.choice([1, 2, 7, 50, 150, 250, 400, 1000]))
        lo = int(rng.integers(0, len(lines) - n + 1))
        b = tv.tokenize_batch(sentences=lines[lo:lo + n])
        for i in range(n):
            e = exp_tok[int(exp_off[lo + i]):int(exp_off[lo + i + 1])]
            ok = b.num_tokens(i) == len(e)
            if ok:
                r = b.records(i)
                ok = all(np.array_equal(r[f], e[f]) for f in V.TOKEN_DTYPE.names if f not in ("start_byte", "end_byte")) and \
                    np.array_equal(r["end_byte"] - r["start_byte"], e["end_byte"] - e["start_byte"])
            if not ok:
                bad += 1
                print(f"call {c} (since renew {c % renew}) n={n} lo={lo} sentence {i}: got {b.num_tokens(i)} tokens, expected {len(e)}", flush=True)
                if b.num_tokens(i):
                    print("   got", b.records(i)[:4], "\n   exp", e[:4], flush=True)
                break
    print(f"mismatching calls: {bad} of {calls}")


if __name__ == "__main__":
    main()
