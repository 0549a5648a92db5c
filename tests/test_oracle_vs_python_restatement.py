"""The C oracle against a second, independent restatement of the reference path (tests/pyref.py) on random small dictionaries.

The oracle is pinned to the reference's own golden vectors (test_oracle_golden.py); those are 21 hand-picked cases.  This test widens
the pin: dictionaries with a handful of connection ids and costs drawn from a few values (so that equal-cost paths -- the `<=` rule of
lattice.rs:141-146 -- are the norm, not the exception), every invoke / group / length combination in char.def, duplicate surfaces, user
entries that shadow system entries, spaces anywhere, characters outside every range and outside the BMP.  Two restatements written
from the Rust sources by different routes (a double array + packed records in C; dicts and lists in Python) have to agree on every token.
"""
import random

import numpy as np
import pytest

from oracle import oracle as ora
from tests import pyref

HIRA = "あいうえ"
KATA = "アイウ"
ALPHA = "abc"
NUM = "12"
OTHER = "。漢𠮷☃"  # no range line covers these (the last but one is outside the BMP: chr2inf[0], character.rs:112-116)


def make_dictionary(rng):
    n_ids = rng.randint(2, 5)  # connection ids 0 .. n_ids - 1 on both sides
    costs = [0, 0, 1, 2, -1, 3]

    def flags():
        return rng.randint(0, 1), rng.randint(0, 1), rng.choice([0, 0, 1, 2, 3])

    cats = ["DEFAULT", "SPACE", "HIRAGANA", "KATAKANA", "ALPHA", "NUMERIC", "KANJINUMERIC"]
    lines = [f"{c} {i} {g} {ln}" for c in cats for i, g, ln in [flags()]]
    lines += ["0x0020 SPACE", "0x3041..0x3096 HIRAGANA", "0x30A1..0x30FA KATAKANA", "0x0061..0x007A ALPHA", "0x0030..0x0039 NUMERIC",
              "0x0031 NUMERIC KANJINUMERIC  # two categories: they chain with both neighbours", "0x3042 HIRAGANA KATAKANA"]
    rng.shuffle(lines)  # range lines may come before the category lines (character.rs collects them first)
    char_def = "\n".join(lines) + "\n"
    unk_rows = []
    for c in cats:
        for _ in range(rng.randint(1, 3)):
            unk_rows.append(f"{c},{rng.randrange(n_ids)},{rng.randrange(n_ids)},{rng.choice(costs) + 2},unk-{c}")
    rng.shuffle(unk_rows)
    alphabet = HIRA + KATA + ALPHA + NUM + " "

    def surface():
        return "".join(rng.choice(alphabet) for _ in range(rng.choice([1, 1, 2, 2, 3, 4])))

    lex_rows = []
    for k in range(rng.randint(15, 60)):
        s = surface() if not lex_rows or rng.random() > 0.2 else rng.choice(lex_rows).split(",")[0]  # duplicate surfaces: several ids
        lex_rows.append(f"{s},{rng.randrange(n_ids)},{rng.randrange(n_ids)},{rng.choice(costs)},sys-{k}")
    user_rows = [f"{surface()},{rng.randrange(n_ids)},{rng.randrange(n_ids)},{rng.choice(costs) - 1},user-{k}" for k in range(rng.randint(0, 8))]
    matrix = [rng.choice(costs) for _ in range(n_ids * n_ids)]
    matrix_def = f"{n_ids} {n_ids}\n" + "".join(f"{r} {l} {matrix[l * n_ids + r]}\n" for l in range(n_ids) for r in range(n_ids))
    return {"lex": "\n".join(lex_rows) + "\n", "user": ("\n".join(user_rows) + "\n") if user_rows else None, "matrix_def": matrix_def,
            "matrix": matrix, "n_ids": n_ids, "char_def": char_def, "unk": "\n".join(unk_rows) + "\n"}


def make_sentence(rng):
    alphabet = HIRA * 3 + KATA * 2 + ALPHA * 2 + NUM + "   " + OTHER
    return "".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 2, 5, 9, 14, 23])))


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_python_restatement(seed):
    rng = random.Random(20260925 + seed)
    d = make_dictionary(rng)
    sentences = [make_sentence(rng) for _ in range(60)] + ["", " ", "  あ  ", "あ ", " a1 ", "𠮷", "1111", "ああああああああ"]
    for ignore_space in (False, True):
        for mgl in (0, 1, 3):
            do = ora.Dictionary.from_sources(d["lex"], d["matrix_def"], d["char_def"], d["unk"])
            if d["user"] is not None:
                do.reset_user_lexicon(d["user"])
            wo = ora.Tokenizer(do, ignore_space, mgl).new_worker()
            py = pyref.PyTokenizer(d["lex"], d["n_ids"], d["n_ids"], d["matrix"], d["char_def"], d["unk"], user_csv=d["user"],
                                   ignore_space=ignore_space, max_grouping_len=mgl)
            wo.reset_counters()
            for s in sentences:
                wo.reset_sentence(s)
                wo.tokenize(counted=True)
                got = []
                for i in range(wo.num_tokens()):
                    rec = np.zeros(1, dtype=ora.TOKEN_DTYPE)
                    ora.lib().ora_worker_token(wo._h, i, rec.ctypes.data)
                    got.append({k: int(rec[0][k]) for k in ora.TOKEN_DTYPE.names})
                assert got == py.tokenize(s), (seed, ignore_space, mgl, s)
            # the event counters the roofline's algorithmic bytes are made of, counted by two routes
            assert wo.counters() == py.counters, (seed, ignore_space, mgl)


def _matrix_from_def(text):
    lines = [ln for ln in text.split("\n") if ln.strip()]
    num_right, num_left = (int(x) for x in lines[0].split())
    m = [0] * (num_right * num_left)
    for ln in lines[1:]:
        r, l, c = (int(x) for x in ln.split())
        m[l * num_right + r] = c
    return num_right, num_left, m


def test_python_restatement_on_the_reference_golden_vectors(tokenize_golden, fixture_sources):
    """The second restatement is itself held to the reference's own end-to-end vectors (the 21 cases of vibrato/src/tests/tokenizer.rs,
    tokenizer.rs:208-361, token.rs, lib.rs that test_oracle_golden.py pins the oracle with): what the two agree on elsewhere is anchored."""
    def text(x):
        return x.decode("utf-8") if isinstance(x, (bytes, bytearray)) else x
    n_checked = 0
    for case in tokenize_golden:
        src = {k: text(v) for k, v in fixture_sources.items()} if case["dict"] == "fixture" else \
            {"lex.csv": case["dict"]["lex"], "matrix.def": case["dict"]["matrix"], "char.def": case["dict"]["char"], "unk.def": case["dict"]["unk"]}
        num_right, num_left, m = _matrix_from_def(text(src["matrix.def"]))
        lex_rows = pyref.parse_rows(text(src["lex.csv"]))
        py = pyref.PyTokenizer(text(src["lex.csv"]), num_right, num_left, m, text(src["char.def"]), text(src["unk.def"]),
                               user_csv=text(fixture_sources["user.csv"]) if case["user"] else None,
                               ignore_space=case["ignore_space"], max_grouping_len=case["max_grouping_len"])
        user_rows = pyref.parse_rows(text(fixture_sources["user.csv"])) if case["user"] else []
        unk_feature = {}  # unknown word id -> feature: unk.def rows in category order (unknown.rs:238-261)
        by_cate = {}
        for name, _, _, _, feat in pyref.parse_rows(text(src["unk.def"])):
            by_cate.setdefault(py.cp.cate_map[name], []).append(feat)
        k = 0
        for cid in range(len(py.cp.cate_map)):
            for feat in by_cate.get(cid, []):
                unk_feature[k] = feat
                k += 1
        for sent in case["sentences"]:
            toks = py.tokenize(sent["text"])
            assert len(toks) == sent["num_tokens"], (case["name"], sent["text"])
            raw = sent["text"].encode("utf-8")
            for exp in sent["tokens"]:
                t = toks[exp["index"]]
                lex, wid = t["word_idx"] >> 30, t["word_idx"] & 0x3FFFFFFF
                got = {"surface": raw[t["start_byte"]:t["end_byte"]].decode("utf-8"), "range_char": [t["start_char"], t["end_char"]],
                       "range_byte": [t["start_byte"], t["end_byte"]], "total_cost": t["total_cost"],
                       "feature": lex_rows[wid][4] if lex == pyref.LEX_SYSTEM else user_rows[wid][4] if lex == pyref.LEX_USER else unk_feature[wid]}
                for key in ["surface", "range_char", "range_byte", "feature", "total_cost"]:
                    if key in exp:
                        assert got[key] == exp[key], (case["name"], sent["text"], exp["index"], key)
        n_checked += 1
    assert n_checked == 21
