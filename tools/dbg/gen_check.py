#!/usr/bin/env python3
"""Developer aid: consistency of what the generator leaves for the sweep (per-character records, candidate records, header) for
one long sentence; needs the `dbgapi` library variant (-DVBT_DEBUG_API=1).  usage: VBT_LIB_VARIANT=dbgapi python tools/dbg/gen_check.py [chars]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vibrato_amd as V  # noqa: E402
from vibrato_amd import _native as N, sharding  # noqa: E402
from tools import synth  # noqa: E402

sd = synth.SynthDict(os.environ.get("DICT", "small"))
dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
tv = V.Tokenizer(dv)
base, offs0 = sd.sentences(1200, "lognormal_40")
raw = bytes(base)
chars = np.cumsum([len(raw[int(offs0[i]):int(offs0[i + 1])].decode("utf-8")) for i in range(1200)])
target = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
sent = raw[:int(offs0[int(np.searchsorted(chars, target)) + 1])]
n = len(sent.decode("utf-8"))
text = np.frombuffer(sent, dtype=np.uint8)
offs = np.array([0, len(sent)], dtype=np.uint64)
ws = tv.workspace(1, len(text))
dt = torch.from_numpy(text.copy()).cuda()
do_ = torch.from_numpy(offs.astype(np.int64)).cuda()
ws.run(dt.data_ptr(), do_.data_ptr(), 1, len(text), 0)
torch.cuda.synchronize()
L = N.lib()
L.vbt_debug_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
out = (C.c_uint64 * 8)()
L.vbt_debug_ptrs(ws._h, out)
hdr = sharding.device_view(out[0], 16).cpu().numpy().view(np.uint32)
print("chars", n, "bytes", len(sent), "hdr: n", hdr[0] & 0xFFFF, "nb", hdr[0] >> 16, "C", hdr[1] & 0xFFFF, "tier", (hdr[1] >> 16) & 0xFF, "passes", hdr[2], "rel", hdr[3])
Cn = int(hdr[1] & 0xFFFF)
nf = int(out[5])
pc = sharding.device_view(out[1], 16 * (n + 1)).cpu().numpy().view(np.uint32).reshape(-1, 4)
cand = sharding.device_view(out[2], 16 * Cn).cpu().numpy().view(np.uint32).reshape(-1, 4)
co = pc[:, 0] & 0xFFFF
eo = pc[:, 0] >> 16
print("cand_off monotone:", bool(np.all(np.diff(co.astype(np.int64)) >= 0)), "last", co[-1], "eo monotone:", bool(np.all(np.diff(eo.astype(np.int64)) >= 0)), "eo[n]", eo[-1], "terminator z (eo(n+1))", pc[n, 2])
bad = np.nonzero(np.diff(co.astype(np.int64)) < 0)[0]
print("  first non-monotone cand_off at position", bad[:5], co[bad[:5]], co[bad[:5] + 1])
bad = np.nonzero(np.diff(eo.astype(np.int64)) < 0)[0]
print("  first non-monotone eo at position", bad[:5], eo[bad[:5]], eo[bad[:5] + 1])
# candidates: start position from cand_off ranges
start = np.searchsorted(co[:n + 1], np.arange(Cn), side="right") - 1
end = cand[:, 3] & 0xFFFF
slot = cand[:, 1] >> 16
eo_full = np.concatenate([eo, [pc[n, 2]]])  # eo(0..n), eo(n+1)
ok_end = (end > start) & (end <= n)
lo, hi = eo_full[np.minimum(end, n)], eo_full[np.minimum(end + 1, n + 1)]
ok_slot = (slot >= lo) & (slot < hi)
print("candidates", Cn, "end ok", int(ok_end.sum()), "slot in its end list", int(ok_slot.sum()), "distinct slots", len(np.unique(slot)))
b = np.nonzero(~ok_slot)[0]
if len(b):
    print("  first bad slots: cand", b[:8], "start", start[b[:8]], "end", end[b[:8]], "slot", slot[b[:8]], "list", lo[b[:8]], hi[b[:8]])
wend = (pc[:n, 1] >> 14) & 0xFFFF
print("window end monotone:", bool(np.all(np.diff(wend.astype(np.int64)) >= 0)), "max", wend.max())
