import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["VBT_PROFILE"] = "1"
import torch
import vibrato_amd as V
from oracle import oracle as ora
from tools import synth
sd = synth.SynthDict("small")
do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
to, tv = ora.Tokenizer(do), V.Tokenizer(dv)
base, offs0 = sd.sentences(1200, "lognormal_40")
raw = bytes(base)
chars = np.cumsum([len(raw[int(offs0[i]):int(offs0[i + 1])].decode("utf-8")) for i in range(1200)])
for c in (6200, 7000):
    sent = raw[:int(offs0[int(np.searchsorted(chars, c)) + 1])]
    w = to.new_worker(); w.reset_counters()
    text = np.frombuffer(sent, dtype=np.uint8); offs = np.array([0, len(sent)], dtype=np.uint64)
    exp, eoff = w.tokenize_batch(text, offs, counted=True)
    cnt = w.counters()
    ws = tv.workspace(1, len(text))
    dt = torch.from_numpy(text.copy()).cuda(); do_ = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws.run(dt.data_ptr(), do_.data_ptr(), 1, len(text), 0)
    torch.cuda.synchronize()
    pr = ws.profile()
    print(len(sent.decode()), "chars; oracle", {k: cnt[k] for k in ("n_chars", "n_nodes", "n_tokens", "n_lex_matches", "n_unk_nodes")}, "device", pr["counts"], "sentences profiled", pr["sentences"], ws.stats()["n_tokens"])
