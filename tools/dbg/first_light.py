#!/usr/bin/env python3
"""Developer aid for a new sweep kernel: small batches through the HIP path vs the oracle, with a per-sentence summary of what
differs (which sentences, how long, single- or multi-segment, first differing token), for several tier settings in ONE GPU call.
usage (GPU box): python tools/dbg/first_light.py [dict] [n]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(shape, n, env, law, ignore_space, mgl):
    for k, v in env.items():
        os.environ[k] = v
    import vibrato_amd as V
    from oracle import oracle as ora
    from tools import synth
    sd = synth.SynthDict(shape)
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    to = ora.Tokenizer(do, ignore_space, mgl)
    tv = V.Tokenizer(dv).ignore_space(ignore_space).max_grouping_len(mgl)
    text, offs = sd.sentences(n, law, space_p=0.05 if ignore_space else 0.0)
    exp, eoff = to.new_worker().tokenize_batch(text, offs)
    b = tv.tokenize_batch(text=text, offsets=offs)
    got, goff = b.tokens_in_order()
    st = b.stats() if hasattr(b, "stats") else None
    bad = []
    raw = bytes(text)
    for s in range(n):
        e = exp[int(eoff[s]):int(eoff[s + 1])]
        g = got[int(goff[s]):int(goff[s + 1])] if s + 1 < len(goff) else got[0:0]
        if len(e) != len(g) or e.tobytes() != g.tobytes():
            k = 0
            while k < min(len(e), len(g)) and e[k].tobytes() == g[k].tobytes():
                k += 1
            nchar = len(raw[int(offs[s]):int(offs[s + 1])].decode("utf-8"))
            bad.append((s, nchar, len(e), len(g), k, e[k] if k < len(e) else None, g[k] if k < len(g) else None))
    print(f"[{shape} n={n} law={law} space={ignore_space} env={env}] mismatching sentences: {len(bad)} / {n}; stats={st}")
    for row in bad[:6]:
        print("    sid %d chars %d  tokens exp %d got %d  first diff at token %d\n      exp %s\n      got %s" % row)
    if bad:
        lens = np.array([r[1] for r in bad])
        print(f"    chars of the mismatching sentences: min {lens.min()} median {int(np.median(lens))} max {lens.max()}")
    return len(bad)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        import json
        a = json.loads(sys.argv[2])
        sys.exit(1 if one(**a) else 0)
    import json
    cases = [
        dict(shape="tiny", n=300, env={}, law="lognormal_40", ignore_space=False, mgl=0),
        dict(shape="small", n=2000, env={}, law="lognormal_40", ignore_space=False, mgl=0),
        dict(shape="small", n=2000, env={"VBT_TIERS": "2048,163840", "VBT_SEG_BYTES": "2048"}, law="lognormal_40", ignore_space=False, mgl=0),
        dict(shape="small", n=2000, env={}, law="mixed", ignore_space=True, mgl=24),
        dict(shape="small-dense", n=1500, env={"VBT_TIERS": "2048,163840", "VBT_SEG_BYTES": "2048"}, law="mixed", ignore_space=True, mgl=24),
    ]
    rc = 0
    for c in cases:  # one process per case: the environment is read when the library creates a workspace, and a fault must not hide the others
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", json.dumps(c)], env=dict(os.environ, VBT_DEBUG="1"), timeout=600)
        rc |= r.returncode
    sys.exit(rc)
