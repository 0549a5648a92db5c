import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["VBT_DEBUG"] = "1"
import numpy as np, torch
import vibrato_amd as V
from tools import synth
for shape, law in (("unidic", "lognormal_40"), ("unidic", "mixed")):
    sd = synth.SynthDict(shape)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tok = V.Tokenizer(dv)
    text, offs = sd.sentences(100000, law)
    d_text = torch.from_numpy(text).cuda(); d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tok.workspace(100000, len(text))
    for _ in range(2):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), 100000, len(text), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    print(shape, law, "density", tok.lattice_density())
    print(ws.stats())
