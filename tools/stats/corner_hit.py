"""Which share of the connection-matrix cells the sweep gathers would a hot k x k corner (staged in LDS) catch?

The deciding number for north_star's "connection-cost matrix staged through LDS" (DESIGN.md 3.6): connection ids are renumbered
by measured usage exactly as the tokenizer does it internally (engine.hip Tokenizer::calibrate: the counts of
Lattice::add_connid_counts, lattice.rs:170-183, over the first VBT_CONNID_SAMPLE = 16 384 sentences; ids by descending count,
then id -- ConnIdCounter::compute_probs, mapper.rs:108-146; id 0 stays), then every (node, predecessor) pair search_min_node
evaluates over the WHOLE batch (lattice.rs:137-147: the cells `lattice_lds` gathers) is classified by the smallest corner that
holds its cell.  Runs on the CPU oracle (oracle/vibrato_oracle.c: ora_worker_add_corner_hist); no GPU involved.

    python tools/stats/corner_hit.py > profiles/r06_corner_hit.md
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as ora  # noqa: E402
from tools import synth  # noqa: E402

BOUNDS = np.array([32, 64, 128, 181, 256, 362, 512, 724, 1024, 2048, 4096], dtype=np.uint32)
SAMPLE = 16384


def ranks(cnt):
    """dictionary id -> device id: ids 1.. by descending count, then ascending id; id 0 stays (engine.hip calibrate())."""
    ids = np.arange(1, len(cnt))
    order = ids[np.argsort(-cnt[1:].astype(np.int64), kind="stable")]
    r = np.zeros(len(cnt), dtype=np.uint32)
    r[order] = np.arange(1, len(cnt), dtype=np.uint32)
    return r


def measure(shape, n, law, ignore_space=False, mgl=0, space_p=0.0, user=0):
    sd = synth.SynthDict(shape)
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    if user:
        do.reset_user_lexicon(sd.user_csv(user))
    w = ora.Tokenizer(do, ignore_space, mgl).new_worker()
    text, offs = sd.sentences(n, law, space_p=space_p)
    raw = bytes(text)
    lid = np.zeros(sd.num_left, dtype=np.uint64)
    rid = np.zeros(sd.num_right, dtype=np.uint64)
    for i in range(min(SAMPLE, n)):
        w.reset_sentence(raw[int(offs[i]):int(offs[i + 1])])
        w.tokenize()
        w.add_connid_counts(lid, rid)
    out = {}
    for name, rl, rr in (("renumbered by usage", ranks(lid), ranks(rid)),
                         ("dictionary's own numbering", np.arange(sd.num_left, dtype=np.uint32), np.arange(sd.num_right, dtype=np.uint32))):
        hist = np.zeros(len(BOUNDS) + 1, dtype=np.uint64)
        for i in range(n):
            w.reset_sentence(raw[int(offs[i]):int(offs[i + 1])])
            w.tokenize()
            w.add_corner_hist(rl, rr, BOUNDS, hist)
        out[name] = hist
    used_l, used_r = int((lid > 0).sum()), int((rid > 0).sum())
    return sd, out, used_l, used_r


def main():
    print("# Share of gathered connection-matrix cells inside the top-k x top-k corner (round 6)\n")
    print("Produced by `python tools/stats/corner_hit.py` on the CPU oracle (no GPU): connection ids renumbered by the usage counts of the")
    print("first 16 384 sentences -- what `Tokenizer::calibrate` (engine.hip) does on the device -- then every (node, predecessor) pair of")
    print("`search_min_node` (lattice.rs:137-147; the cells `lattice_lds` gathers, one per pair) over the whole batch classified by the")
    print("smallest corner holding its cell.  A k x k corner of i16 cells takes 2 k^2 bytes of LDS: k = 128 -> 32 KiB, 181 -> 64 KiB,")
    print("256 -> 128 KiB, 362 -> 256 KiB (more than a CU's 160 KiB).\n")
    cases = [("headline (BASELINE config 3)", "unidic", 100000, "lognormal_40", {}),
             ("dense law", "unidic-dense", 30000, "lognormal_40", {}),
             ("BASELINE config 5 (-S -M 24, user.csv, mixed lengths)", "unidic", 30000, "mixed", dict(ignore_space=True, mgl=24, space_p=0.10, user=1000)),
             ("BASELINE config 2 (ipadic shape, 1316 ids)", "ipadic", 100000, "lognormal_40", {})]
    for title, shape, n, law, kw in cases:
        sd, out, ul, ur = measure(shape, n, law, **kw)
        print(f"## {title}: syn-{shape}, {n} sentences, matrix {sd.num_right} x {sd.num_left}\n")
        print(f"ids with a non-zero count in the calibration sample: {ul} left, {ur} right\n")
        print("| k | LDS for the corner | " + " | ".join(out.keys()) + " |")
        print("|---|---|" + "---|" * len(out))
        tot = {k: int(v.sum()) for k, v in out.items()}
        for bi, b in enumerate(BOUNDS):
            cells = [f"{100.0 * int(v[:bi + 1].sum()) / tot[k]:.1f} %" for k, v in out.items()]
            print(f"| {int(b)} | {2 * int(b) * int(b) / 1024:.0f} KiB | " + " | ".join(cells) + " |")
        print(f"\npairs classified: {list(tot.values())[0]:,}\n")


if __name__ == "__main__":
    main()
