#!/usr/bin/env python3
"""Several host threads on ONE fresh tokenizer while its background calibration runs and publishes the renumbered image (developer aid, round 6):
large host batches (one of them triggers the calibration; one is big enough for the pipelined path), small host batches, a Worker loop and
device-resident steps on a torch stream -- every result compared with the oracle's records.  usage (GPU box): python tools/dbg/mt_stress.py [iterations]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import vibrato_amd as V
    from oracle import oracle as ora
    from tools import synth
    from vibrato_amd import sharding
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sd = synth.SynthDict("small")
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    to = ora.Tokenizer(do, False, 0)
    text, offs = sd.sentences(24000, "lognormal_40")
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    raw = text.tobytes()
    names = [f for f in V.TOKEN_DTYPE.names if f not in ("start_byte", "end_byte")]
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    bad = []

    def check_batch(tag, b, lo, n):
        for i in range(n):
            e = exp_tok[int(exp_off[lo + i]):int(exp_off[lo + i + 1])]
            if b.num_tokens(i) != len(e):
                bad.append((tag, lo, i, "count", b.num_tokens(i), len(e)))
                return
            r = b.records(i)
            if not all(np.array_equal(r[f], e[f]) for f in names):
                bad.append((tag, lo, i, "records"))
                return

    def host_batches(tok, tag, n, rounds, seed):
        rng = np.random.default_rng(seed)
        for _ in range(rounds):
            lo = int(rng.integers(0, 24000 - n + 1))
            sub_offs = offs[lo:lo + n + 1]
            b = tok.tokenize_batch(text=text, offsets=sub_offs)
            check_batch(tag, b, lo, n)

    def worker_loop(tok, rounds, seed):
        rng = np.random.default_rng(seed)
        w = tok.new_worker()
        for _ in range(rounds):
            i = int(rng.integers(0, 24000))
            w.reset_sentence(raw[int(offs[i]):int(offs[i + 1])])
            w.tokenize()
            e = exp_tok[int(exp_off[i]):int(exp_off[i + 1])]
            ok = w.num_tokens() == len(e)
            for k in range(len(e) if ok else 0):
                t = w.token(k)
                ok = ok and ((t.lex_type << 30) | t.word_id) == int(e["word_idx"][k]) and t.total_cost == int(e["total_cost"][k]) and t.range_char == (int(e["start_char"][k]), int(e["end_char"][k]))
            if not ok:
                bad.append(("worker", i))
                return

    def device_steps(tok, rounds):
        st = torch.cuda.Stream()
        ws = tok.workspace(6000, int(offs[6000]))
        for _ in range(rounds):
            ws.run(d_text.data_ptr(), d_offs.data_ptr(), 6000, int(offs[6000]), st.cuda_stream)
            s = ws.stats()
            if s["error_flags"] or s["n_tokens"] != int(exp_off[6000]):
                bad.append(("device", s))
                return
            v = sharding.workspace_views(ws, 6000, s["n_tokens"])
            got, _ = sharding.tokens_in_sentence_order(v["tok_off"].cpu().numpy().view(np.uint32), v["tok_cnt"].cpu().numpy().view(np.uint32), v["tokens"].cpu().numpy().view(V.TOKEN_DTYPE))
            if got.tobytes() != exp_tok[:int(exp_off[6000])].tobytes():
                bad.append(("device", "records"))
                return

    for it in range(iters):
        dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        tok = V.Tokenizer(dv)
        th = [threading.Thread(target=host_batches, args=(tok, "big", 20000, 6, it)),
              threading.Thread(target=host_batches, args=(tok, "mid", 5000, 10, 100 + it)),
              threading.Thread(target=host_batches, args=(tok, "small", 150, 40, 200 + it)),
              threading.Thread(target=worker_loop, args=(tok, 300, 300 + it)),
              threading.Thread(target=device_steps, args=(tok, 15))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        tok.wait_connid_reorder(60)
        info = tok.connid_reorder_info()
        # one more of each, alone, on the final image
        host_batches(tok, "big-after", 20000, 1, 999)
        print(f"iteration {it}: calibration {info['state']} epoch {info['epoch']}, mismatches so far {len(bad)}", flush=True)
        if bad:
            print(bad[:5], flush=True)
            break
    print(f"mismatches: {len(bad)}")


if __name__ == "__main__":
    main()
