#!/usr/bin/env python3
"""Developer aid: vbt_batch_format call by call on one headline-sized batch (cold / warm buffer, thread counts).
usage (GPU box): python tools/format_bench.py [threads ...]"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import vibrato_amd as V
    from vibrato_amd import _native as N
    from tools import synth
    sd = synth.SynthDict("unidic")
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tok = V.Tokenizer(dv)
    text, offs = sd.sentences(100000, "lognormal_40")
    b = tok.tokenize_batch(text=text, offsets=offs)
    L = N.lib()
    ts = []
    for i in range(8):
        p, n = C.c_void_p(), C.c_size_t()
        t = time.perf_counter()
        N.check(L.vbt_batch_format(b._h, 0, C.byref(p), C.byref(n)))
        ts.append((time.perf_counter() - t) * 1e3)
        L.vbt_free(p)
    print(f"threads={os.environ.get('VBT_FORMAT_THREADS', 'default')}: {n.value / 1e6:.1f} MB; ms per call: " + " ".join(f"{x:.2f}" for x in ts)
          + f"; best {n.value / min(ts) / 1e6:.1f} GB/s")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one()
    else:
        for t in (sys.argv[1:] or ["16", "32", "64", "128"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, VBT_FORMAT_THREADS=t))
