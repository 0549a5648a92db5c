#!/usr/bin/env python3
"""Timing probes of the bulk generator (round 6): what would its expansion phase cost without its candidate stores / without its entry loads /
with perfectly coalesced stores?  Variant libraries abl1 / abl2 / abl3 (-DVBT_ABLATE=n: results are WRONG, the sweep is skipped with
VBT_SKIP_SWEEP=1) against the default build with the sweep skipped as well.
   build:  python -c "from vibrato_amd.build import build; [build(variant='abl%d' % r, defines=['VBT_ABLATE=%d' % r]) for r in (1, 2, 3)]"
   run  :  for v in '' abl1 abl2 abl3; do VBT_LIB_VARIANT=$v VBT_SKIP_SWEEP=1 python tools/dbg/gen_ablate.py; done"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import vibrato_amd as V
    from tools import synth
    shape = sys.argv[1] if len(sys.argv) > 1 else "unidic"
    n = 100000
    sd = synth.SynthDict(shape)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tok = V.Tokenizer(dv)
    text, offs = sd.sentences(n, "lognormal_40")
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tok.workspace(n, len(text))
    ws.set_timing(True)
    st = torch.cuda.current_stream().cuda_stream
    gen = []
    for it in range(23):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), st)
        torch.cuda.synchronize()
        s = ws.stats()
        if it >= 3:
            gen.append(s["ms_tier0"])
    print(f"variant '{os.environ.get('VBT_LIB_VARIANT', '')}' {shape}: generator phase (validate .. gen_long) {np.mean(gen):.4f} ms (min {np.min(gen):.4f})")


if __name__ == "__main__":
    main()
