import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vibrato_amd as V
from tools import synth
sd = synth.SynthDict(os.environ.get("DICT", "small"))
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tv = V.Tokenizer(dv)
base, offs0 = sd.sentences(1200, "lognormal_40")
raw = bytes(base)
chars = np.cumsum([len(raw[int(offs0[i]):int(offs0[i + 1])].decode("utf-8")) for i in range(1200)])
sent = raw[:int(offs0[int(np.searchsorted(chars, int(sys.argv[1]))) + 1])]
text = np.frombuffer(sent, dtype=np.uint8); offs = np.array([0, len(sent)], dtype=np.uint64)
ws = tv.workspace(1, len(text))
dt = torch.from_numpy(text.copy()).cuda(); do_ = torch.from_numpy(offs.astype(np.int64)).cuda()
ws.run(dt.data_ptr(), do_.data_ptr(), 1, len(text), 0)
torch.cuda.synchronize()
print(ws.stats())
