"""Developer aid: fallback reasons of the segmented sweep (VBT_DEBUG=1 VBT_SEG_BYTES=... python tools/dbg/seg_stats.py small 3000 lognormal_40)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VBT_DEBUG"] = "1"
import torch
import vibrato_amd as V
from tools import synth
shape, n, law = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sd = synth.SynthDict(shape)
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tok = V.Tokenizer(dv).ignore_space(len(sys.argv) > 4)
text, offs = sd.sentences(n, law, space_p=0.15 if len(sys.argv) > 4 else 0.0)
ws = tok.workspace(n, len(text))
ws.set_timing(True)
d_text = torch.from_numpy(text).cuda(); d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
for _ in range(3):
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
print(ws.stats())
