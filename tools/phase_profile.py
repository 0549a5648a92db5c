#!/usr/bin/env python3
"""Developer aid: per-phase cycle breakdown of the fused tokenize kernel (VBT_PROFILE=1)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VBT_PROFILE"] = "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dict", default="unidic")
    ap.add_argument("--sentences", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--min-chars", type=int, default=0, help="keep only sentences with at least this many bytes/3")
    ap.add_argument("--max-chars", type=int, default=0)
    ap.add_argument("--law", default="lognormal_40")
    ap.add_argument("--user", type=int, default=0, help="user lexicon entries (BASELINE config 5: 1000)")
    ap.add_argument("--ignore-space", action="store_true")
    ap.add_argument("--mgl", type=int, default=0)
    ap.add_argument("--space-p", type=float, default=0.0)
    ap.add_argument("--keep", type=int, default=0, help="after filtering keep only the first N sentences")
    args = ap.parse_args()
    import torch
    import vibrato_amd as V
    from tools import synth
    sd = synth.SynthDict(args.dict)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    if args.user:
        dv.reset_user_lexicon_from_reader(sd.user_csv(args.user))
    tok = V.Tokenizer(dv, device=0).ignore_space(args.ignore_space).max_grouping_len(args.mgl)
    text, offs = sd.sentences(args.sentences, args.law, space_p=args.space_p, seed=synth.SEED)
    if args.min_chars or args.max_chars:
        lens = np.diff(offs).astype(np.int64)
        keep = np.nonzero((lens >= 2.85 * args.min_chars) & ((lens <= 2.85 * args.max_chars) if args.max_chars else True))[0]
        if args.keep:
            keep = keep[:args.keep]
        parts = [text[int(offs[i]):int(offs[i + 1])] for i in keep]
        text = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        offs = np.zeros(len(keep) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(p) for p in parts])
        args.sentences = len(keep)
        print(f"filtered to {len(keep)} sentences, {len(text)} bytes")
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tok.workspace(args.sentences, len(text))
    ws.set_timing(True)
    for i in range(args.steps + 1):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), args.sentences, len(text), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if i == 0:
            ws.profile(reset=True)
    st = ws.stats()
    pr = ws.profile()
    ns = pr.pop("sentences")
    counts = pr.pop("counts", {})
    print("per sentence:", {k: round(v / max(ns, 1), 1) for k, v in counts.items()})
    tot = sum(pr.values())
    print(f"tiers={os.environ.get('VBT_TIERS', 'default')} stats={st}")
    print(f"sentences profiled: {ns}; mean cycles/sentence {tot / max(ns, 1):.0f} (~{tot / max(ns, 1) / 2.4e3:.1f} us @2.4GHz)")
    for k, v in pr.items():
        print(f"  {k:11s} {v / max(ns, 1):9.0f} cyc/sentence  {100.0 * v / max(tot, 1):5.1f}%")


if __name__ == "__main__":
    main()
