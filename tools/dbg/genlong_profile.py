#!/usr/bin/env python3
"""Developer aid: where a workgroup of gen_long (the generator of sentences the bulk generator's LDS does not hold) spends its
wall time, phase by phase.  Needs the variant build `glprof` (vibrato_amd.build.build(variant="glprof", defines=("VBT_GENLONG_PROF=1",)));
run on the GPU box: VBT_LIB_VARIANT=glprof python tools/dbg/genlong_profile.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["VBT_PROFILE"] = "1"
os.environ["VBT_PROF_HALF"] = "1"
os.environ.setdefault("VBT_LIB_VARIANT", "glprof")

PHASES = ["count+scan", "decode", "groupable", "trie walks", "csr scans+fence", "expand hits", "far ends", "records", "route"]


def main():
    import ctypes as C
    import torch
    import vibrato_amd as V
    from vibrato_amd import _native as N
    from tools import synth
    sd = synth.SynthDict("unidic")
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    dv.reset_user_lexicon_from_reader(sd.user_csv(1000))
    tok = V.Tokenizer(dv, device=0).ignore_space(True).max_grouping_len(24)
    n = 100000
    text, offs = sd.sentences(n, "mixed", space_p=0.1, seed=synth.SEED)
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tok.workspace(n, len(text))
    ws.set_timing(True)
    for i in range(4):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = (C.c_uint64 * 12)()
        N.check(N.lib().vbt_workspace_profile(ws._h, out, 1))
    w = [int(x) for x in out]
    ns, chars, hits = w[9], w[10], w[11]
    tot = sum(w[:9])
    print(f"gen_long: {ns} sentences, {chars} characters ({chars / max(ns, 1):.0f} each), {hits / max(chars, 1):.2f} hits per character; waves per workgroup:"
          f" {os.environ.get('VBT_GEN_WAVES', 'default')}; stats {ws.stats()}")
    print(f"wall cycles of a workgroup per character: {tot / max(chars, 1):.1f} (100 MHz clock64 ticks x 21 = shader cycles at 2.1 GHz)" if False else
          f"clock64 ticks per character: {tot / max(chars, 1):.2f}")
    for name, v in zip(PHASES, w[:9]):
        print(f"  {name:16s} {v / max(chars, 1):8.3f} per character  {100.0 * v / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    main()
