#!/usr/bin/env python3
"""Timing probes of the lean sweep (round 6): where does lattice_lean's time go?  Variant libraries lab1..lab4 (-DVBT_ABLATE_LEAN=n: 1 = header, loads and
LDS fill only; 2 = + the loop; 3 = + back-trace, no token records; 4 = everything but the loop: results are WRONG) with only the lean tier launched
(VBT_SKIP_SWEEP=2), against the default build under the same setting.
   build:  python -c "from vibrato_amd.build import build; [build(variant='lab%d' % r, defines=['VBT_ABLATE_LEAN=%d' % r]) for r in (1, 2, 3, 4)]"
   run  :  for v in '' lab1 lab2 lab3 lab4; do VBT_LIB_VARIANT=$v VBT_SKIP_SWEEP=2 python tools/dbg/lean_ablate.py; done"""
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
    tok.calibrate(text=text, offsets=offs)  # (the counting run is exempt from VBT_SKIP_SWEEP; its counts come from the general instance, which the probes leave alone)
    ws = tok.workspace(n, len(text))
    ws.set_timing(True)
    st = torch.cuda.current_stream().cuda_stream
    sw = []
    for it in range(23):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), st)
        torch.cuda.synchronize()
        s = ws.stats()
        if it >= 3:
            sw.append(s["ms_tier12"])
    print(f"variant '{os.environ.get('VBT_LIB_VARIANT', '')}' skip={os.environ.get('VBT_SKIP_SWEEP', '0')} {shape}: sweep span {np.mean(sw):.4f} ms (min {np.min(sw):.4f})  connid {tok.connid_reorder_info()['state']}")


if __name__ == "__main__":
    main()
