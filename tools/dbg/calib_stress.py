#!/usr/bin/env python3
"""Developer aid: the background calibration triggered over and over on fresh tokenizers (with a Worker's resident kernel next to it,
as tests/test_connid_mapping.py has it); prints every outcome that is not "done, epoch 1".  VBT_DEBUG=1 shows why one was given up."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import vibrato_amd as V  # noqa: E402
from tools import synth  # noqa: E402

sd = synth.SynthDict("small")
text, offs = sd.sentences(4000, "lognormal_40", space_p=0.1)
d_text = torch.from_numpy(text).cuda()
d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
st = torch.cuda.current_stream().cuda_stream
bad = 0
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for it in range(n_iter):
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tok = V.Tokenizer(dv).ignore_space(True)
    if it % 2 == 0:
        w = tok.new_worker()
        w.reset_sentence("東京都に行く")
        w.tokenize()
    ws = tok.workspace(4000, len(text))
    if it % 3 == 0:
        ws.count_connids(True)
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), 1000, int(offs[1000]), st)
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), 4000, len(text), st)
    torch.cuda.synchronize()
    ok = tok.wait_connid_reorder(60)
    info = tok.connid_reorder_info()
    if not ok or info["epoch"] != 1 or info["state"] != "done":
        bad += 1
        print("iteration", it, "wait", ok, info, "stats", ws.stats(), flush=True)
print("bad outcomes:", bad, "of", n_iter)
