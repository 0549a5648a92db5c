"""Compact connectors: RawConnector / DualConnector / Scorer of the reference
(vibrato/src/dictionary/connector/raw_connector.rs, raw_connector/scorer.rs, dual_connector.rs) behind
SystemDictionaryBuilder::from_readers_with_bigram_info (builder.rs:111-160).

Golden vectors: the reference's own unit tests, parsed by tests/golden/make_golden.py.  The product evaluates the
same cost function on the host (vbt_dict_conn_cost) and, on the GPU, expands it once into the dense matrix the
sweep reads (engine.hip expand_connector); the GPU tests check the tokenization against the oracle, whose
search_min_node calls the raw cost function per pair like the reference (raw_connector.rs:153-161)."""
import json
import os
import random

import numpy as np
import pytest

import vibrato_amd as V
from oracle import oracle as ora
from tools import synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "unit_golden.json"), encoding="utf-8"))
CHAR = "DEFAULT 0 1 0"
UNK = "DEFAULT,0,0,0,*"


def _dict(kind, right, left, cost, lex="a,1,1,0,x", char=CHAR, unk=UNK, dual=False):
    if kind == "oracle":
        return ora.Dictionary.from_sources_bigram(lex, right, left, cost, char, unk, dual_connector=dual)
    return V.SystemDictionaryBuilder.from_readers_with_bigram_info(lex, right, left, cost, char, unk, dual_connector=dual)


def _err(kind):
    return ora.OracleError if kind == "oracle" else V.VibratoError


# ------------------------------------------------------------------ golden vectors

@pytest.mark.parametrize("case", GOLD["scorer"], ids=lambda c: c["source"].split("::")[-1])
def test_oracle_scorer_golden(case):
    """scorer.rs:348-480: double-array hash built by ScorerBuilder, retrieve_cost / accumulate_cost."""
    s = ora.Scorer(case["insert"])
    for k1, k2, exp in case["retrieve"]:
        assert s.retrieve_cost(k1, k2) == exp
    for k1, k2, exp in case["accumulate"]:
        assert s.accumulate_cost(k1, k2) == exp


@pytest.mark.parametrize("kind", ["oracle", "product"])
@pytest.mark.parametrize("case", [c for c in GOLD["bigram_connector"] if c["map"] is None], ids=lambda c: c["source"].split("/")[-1])
def test_from_readers_golden(kind, case):
    """raw_connector.rs::from_readers_test, dual_connector.rs::from_readers_test: product and oracle each build the connector
    kind the vector names (RawConnector / DualConnector restatement) and answer the reference's costs."""
    d = _dict(kind, case["right"], case["left"], case["cost"], dual=case["dual"])
    assert (d.num_right, d.num_left) == (3, 3)
    for r, l, c in case["costs"]:
        assert d.conn_cost(r, l) == c
    if kind == "product":
        assert d.connector_kind == ("Dual" if case["dual"] else "Raw")


@pytest.mark.parametrize("case", [c for c in GOLD["bigram_connector"] if c["map"] is not None], ids=lambda c: c["source"].split("/")[-1])
def test_oracle_mapping_golden(case):
    """raw_connector.rs::mapping_test, dual_connector.rs::mapping_test (ConnIdMapper::new moves id 0 too, which
    Dictionary::map_connection_ids_from_iter never does: oracle connector object only)."""
    c = (ora.DualConnector if case["dual"] else ora.RawConnector)(case["right"], case["left"], case["cost"])
    c.map_connection_ids(case["map"][0], case["map"][1])
    for r, l, exp in case["costs"]:
        assert c.cost(r, l) == exp


@pytest.mark.parametrize("kind", ["oracle", "product"])
def test_parse_cost_golden(kind):
    """raw_connector.rs::parse_cost_test: feature ids in order of first appearance.  Through a connector: each right /
    left feature sits alone in its own template position, so cost(i, j) isolates one bigram.cost line."""
    lines = GOLD["parse_cost"][0]["lines"]
    rights = sorted({(r, ln.split("\t")[0].split("/")[0]) for ln, r, _, _ in lines})
    lefts = sorted({(l, ln.split("\t")[0].split("/")[1]) for ln, _, l, _ in lines})
    right = "\n".join(f"{i + 1}\t{name}" for i, (_, name) in enumerate(rights))
    left = "\n".join(f"{i + 1}\t{name}" for i, (_, name) in enumerate(lefts))
    d = _dict(kind, right, left, "\n".join(ln for ln, _, _, _ in lines))
    expect = {(r, l): c for _, r, l, c in lines}
    for i in range(len(rights)):
        for j in range(len(lefts)):
            assert d.conn_cost(i + 1, j + 1) == expect.get((rights[i][0], lefts[j][0]), 0)


@pytest.mark.parametrize("kind", ["oracle", "product"])
def test_parse_errors_golden(kind):
    """raw_connector.rs::parse_cost_invalid_*_test, parse_feature_invalid_id_test, from_readers' ascending-id check"""
    ok = GOLD["bigram_connector"][0]
    for e in GOLD["parse_cost_errors"]:
        with pytest.raises(_err(kind)):
            _dict(kind, ok["right"], ok["left"], e["line"])
    for e in GOLD["parse_features_errors"]:
        with pytest.raises(_err(kind)):
            _dict(kind, e["line"], ok["left"], ok["cost"])
        with pytest.raises(_err(kind)):
            _dict(kind, ok["right"], e["line"], ok["cost"])
    with pytest.raises(_err(kind)):  # raw_connector.rs:222-226 "must be ascending order"
        _dict(kind, "2\tx\n1\ty", ok["left"], ok["cost"])
    with pytest.raises(_err(kind)):  # builder.rs:24-29: the lexicon's ids must exist in the connector
        _dict(kind, ok["right"], ok["left"], ok["cost"], lex="a,3,1,0,x")


@pytest.mark.parametrize("kind", ["oracle", "product"])
def test_parse_features_golden(kind):
    """raw_connector.rs::parse_feature_test: quoted CSV fields, '*' (absent from the id map) = INVALID_FEATURE_ID.
    Through a connector: one bigram.cost line per known feature with a distinct power-of-two cost."""
    g = GOLD["parse_features"][0]
    names = sorted(g["id_map"], key=g["id_map"].get)
    row = g["line"].split("\t", 1)[1]
    right = "1\t" + row
    left = "1\t" + ",".join("L" for _ in g["features"])
    cost = "\n".join(f"{n}/L\t{1 << g['id_map'][n]}" for n in names)
    d = _dict(kind, right, left, cost)
    assert d.conn_cost(1, 1) == sum(1 << f for f in g["features"] if f != 0x7FFFFFFF)


# ------------------------------------------------------------------ synthetic compact model

def synth_bigram(num_right, num_left, seed=7, templates=10, vocab=12, density=0.5, max_abs=300, empty_pair=True):
    """bigram.right / bigram.left / bigram.cost of a random compact model: `templates` positions (the builder rounds
    the width up to 16), a small vocabulary per position, '*' and quoted fields, BOS/EOS ("") pairs."""
    rng = random.Random(seed)

    def rows(n, side):
        out = []
        for i in range(1, n):
            feats = []
            for t in range(templates if i % 7 else templates - 3):  # ragged rows are padded with INVALID
                v = rng.randrange(vocab + 2)
                feats.append("*" if v >= vocab else (f'"{side}{t},{v}"' if v == 3 else f"{side}{t}_{v}"))
            out.append(f"{i}\t" + ",".join(feats))
        return "\n".join(out) + "\n"

    cost = []
    for t in range(templates):
        for a in range(vocab):
            for b in range(vocab):
                if rng.random() < density:
                    ra = f"R{t},{a}" if a == 3 else f"R{t}_{a}"
                    lb = f"L{t},{b}" if b == 3 else f"L{t}_{b}"
                    cost.append(f"{ra}/{lb}\t{rng.randint(-max_abs, max_abs)}")
        cost.append(f"/L{t}_1\t{rng.randint(-50, 50)}")   # BOS -> left feature
        cost.append(f"R{t}_2/\t{rng.randint(-50, 50)}")   # right feature -> EOS
    if empty_pair:
        cost.append("/\t3")  # BOS/EOS pair: counted once per position of the padded width
    rng.shuffle(cost)
    return rows(num_right, "R"), rows(num_left, "L"), "\n".join(cost) + "\n"


def test_product_host_cost_function_matches_oracle_with_mapping():
    """every (right, left) pair of a random model, before and after Dictionary::map_connection_ids_from_iter
    (raw_connector.rs:118-146)"""
    sd = synth.SynthDict("tiny")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left)
    dv = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk)
    do = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk)
    assert (dv.num_right, dv.num_left) == (do.num_right, do.num_left) == (sd.num_right, sd.num_left)
    grid = [(r, l) for r in range(sd.num_right) for l in range(sd.num_left)]
    a = [dv.conn_cost(r, l) for r, l in grid]
    assert a == [do.conn_cost(r, l) for r, l in grid]
    assert len(set(a)) > 100 and dv.conn_cost(0, 0) == 3 * 16
    rng = random.Random(5)
    lmap = list(range(1, sd.num_left)); rng.shuffle(lmap)
    rmap = list(range(1, sd.num_right)); rng.shuffle(rmap)
    dv.map_connection_ids_from_iter(lmap, rmap)
    do.map_connection_ids_from_iter(lmap, rmap)
    b = [dv.conn_cost(r, l) for r, l in grid]
    assert b == [do.conn_cost(r, l) for r, l in grid]
    assert b != a and sorted(b) == sorted(a)


def test_dual_connector_builder_gives_the_raw_cost_function():
    """DualConnector::from_readers (dual_connector.rs:145-199): matrix over the classes of the kept templates + the eight removed
    templates through a pruned scorer = the RawConnector's cost for every id pair (what the reference's own dual tests assert on
    their vectors), as long as no matrix cell is clamped and the model prices no ("" , "") feature pair (the dual matrix counts
    that pair once per padded position, U31x8::to_simd_vec pads with feature 0)."""
    sd = synth.SynthDict("tiny")
    for templates, seed in ((10, 3), (16, 4), (19, 5)):
        right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=seed, templates=templates, empty_pair=False)
        raw = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=False)
        dual = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)
        ora_d = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk)
        assert (raw.connector_kind, dual.connector_kind) == ("Raw", "Dual")
        for r in range(sd.num_right):
            for l in range(sd.num_left):
                c = raw.conn_cost(r, l)
                assert dual.conn_cost(r, l) == c == ora_d.conn_cost(r, l)
        # id mapping renumbers the small matrix by first use (dual_connector.rs:211-264) and keeps the function
        rng = random.Random(seed)
        lmap = list(range(1, sd.num_left)); rng.shuffle(lmap)
        rmap = list(range(1, sd.num_right)); rng.shuffle(rmap)
        raw.map_connection_ids_from_iter(lmap, rmap)
        dual.map_connection_ids_from_iter(lmap, rmap)
        assert all(dual.conn_cost(r, l) == raw.conn_cost(r, l) for r in range(sd.num_right) for l in range(sd.num_left))
        # and the container holds a Dual connector that reads back
        back = V.Dictionary.read(dual.write())
        assert back.connector_kind == "Dual"
        assert all(back.conn_cost(r, l) == raw.conn_cost(r, l) for r in range(0, sd.num_right, 3) for l in range(sd.num_left))
    with pytest.raises(V.VibratoError):  # fewer than eight templates: nothing to split off (the reference underflows there)
        right, left, cost = synth_bigram(sd.num_right, sd.num_left, templates=5, empty_pair=False)
        _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)


def dual_reference_costs(right, left, cost, num_right, num_left):
    """Pure-Python restatement of DualConnector::from_readers + cost (dual_connector.rs:25-199, 267-279; small cases only):
    feature ids in order of first appearance in bigram.cost ("" = 0), greedy removal of eight templates (ties: highest index,
    the product's documented convention -- the reference's hash-set order is not reproducible), class matrix with cells
    clamped to i16 and rows padded with feature 0 to a multiple of 8, the removed templates through the pair -> cost table."""
    import csv as _csv
    rid, lid, table = {"": 0}, {"": 0}, {}
    for line in cost.splitlines():
        feats, c = line.split("\t")
        a, b = feats.split("/")
        table[(rid.setdefault(a, len(rid)), lid.setdefault(b, len(lid)))] = int(c)
    INV = 0x7FFFFFFF

    def rows(text, ids):
        out = []
        for line in text.splitlines():
            out.append([ids.get(f, INV) for f in next(_csv.reader([line.split("\t", 1)[1]]))])
        return out
    R, L = rows(right, rid), rows(left, lid)
    T = max(len(r) for r in R + L)
    keep = list(range(T))

    def distinct(rws, without):
        return len({tuple(r[i] for i in keep if i != without and i < len(r)) for r in rws})
    for _ in range(8):
        best, best_size = 0, len(R) * len(L)
        for trial in keep:
            size = distinct(R, trial) * distinct(L, trial)
            if size <= best_size:
                best, best_size = trial, size
        keep.remove(best)
    raw_idx = [i for i in range(T) if i not in keep]

    def acc(a, b):
        return sum(table.get((x, y), 0) for x, y in zip(a, b))

    def pad8(v):
        return list(v) + [0] * (-len(v) % 8)

    def side(rws):
        classes, id_map = {tuple([0] * (T - 8)): 0}, [0]
        for r in rws:
            id_map.append(classes.setdefault(tuple(r[i] if i < len(r) else INV for i in keep), len(classes)))
        return [pad8(k) for k in classes], id_map  # dict order = class id order
    rc, rmap = side(R)
    lc, lmap = side(L)
    matrix = [[max(-32768, min(32767, acc(rc[r], lc[l]))) for r in range(len(rc))] for l in range(len(lc))]
    rraw = [[0] * 8] + [[r[i] if i < len(r) else INV for i in raw_idx] for r in R]
    lraw = [[0] * 8] + [[r[i] if i < len(r) else INV for i in raw_idx] for r in L]
    return [[matrix[lmap[l]][rmap[r]] + acc(rraw[r], lraw[l]) for l in range(num_left)] for r in range(num_right)]


def test_dual_connector_builder_matches_a_python_restatement_including_its_quirks():
    """The cases where a Dual connector is NOT the Raw cost function: a priced ("", "") pair is counted once per padded
    position of the class rows, and class-matrix cells are clamped to i16."""
    sd = synth.SynthDict("tiny")
    for templates, seed, max_abs in ((10, 21, 300), (13, 22, 300), (19, 23, 9000)):
        right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=seed, templates=templates, max_abs=max_abs, empty_pair=True)
        dual = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)
        raw = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=False)
        exp = dual_reference_costs(right, left, cost, sd.num_right, sd.num_left)
        got = [[dual.conn_cost(r, l) for l in range(sd.num_left)] for r in range(sd.num_right)]
        assert got == exp
        # ... and the oracle's C restatement of DualConnector (oracle/dual_connector.c), before and after an id mapping
        od = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)
        assert [[od.conn_cost(r, l) for l in range(sd.num_left)] for r in range(sd.num_right)] == exp
        rng = random.Random(seed)
        lmap = list(range(1, sd.num_left)); rng.shuffle(lmap)
        rmap = list(range(1, sd.num_right)); rng.shuffle(rmap)
        dual2 = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)
        dual2.map_connection_ids_from_iter(lmap, rmap)
        od.map_connection_ids_from_iter(lmap, rmap)
        assert all(dual2.conn_cost(r, l) == od.conn_cost(r, l) for r in range(sd.num_right) for l in range(sd.num_left))
        differs = sum(got[r][l] != raw.conn_cost(r, l) for r in range(sd.num_right) for l in range(sd.num_left))
        assert differs > 0  # the quirks are exercised: these models are where Dual and Raw part ways


# ------------------------------------------------------------------ GPU: tokenization through the expanded matrix

def _assert_same(batch, exp, exp_off):
    got, got_off = batch.tokens_in_order()
    assert np.array_equal(got_off, exp_off)
    for f in V.TOKEN_DTYPE.names:
        assert np.array_equal(got[f], exp[f]), f


@pytest.mark.gpu
@pytest.mark.parametrize("dual", [False, True])
def test_tokenize_with_compact_connector_matches_oracle(dual):
    sd = synth.SynthDict("small")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=11, empty_pair=not dual)
    dv = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=dual)
    do = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=dual)
    text, offs = sd.sentences(1500, "lognormal_40")
    exp, exp_off = ora.Tokenizer(do).new_worker().tokenize_batch(text, offs)
    tok = V.Tokenizer(dv, device=0)
    _assert_same(tok.tokenize_batch(text=text, offsets=offs), exp, exp_off)
    assert len(set(exp["total_cost"].tolist())) > 500


@pytest.mark.gpu
@pytest.mark.parametrize("templates,seed,max_abs", [(10, 21, 300), (19, 23, 9000)])
def test_tokenize_with_a_dual_connector_that_is_not_the_raw_cost_function(templates, seed, max_abs):
    """Models on which DualConnector and RawConnector part ways (a priced ("", "") pair counted per padded class-row position,
    class-matrix cells clamped to i16 -- dual_connector.rs:103): the GPU sweeps the Dual cost function (expanded into the device
    matrix) and agrees with the oracle's DualConnector restatement, token for token, while the Raw connector of the same model
    gives different best paths."""
    sd = synth.SynthDict("small")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=seed, templates=templates, max_abs=max_abs, empty_pair=True)
    dv = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)
    do = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=True)
    d_raw = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=False)
    assert sum(do.conn_cost(r, l) != d_raw.conn_cost(r, l) for r in range(0, sd.num_right, 7) for l in range(sd.num_left)) > 0
    text, offs = sd.sentences(1500, "lognormal_40")
    exp, exp_off = ora.Tokenizer(do).new_worker().tokenize_batch(text, offs)
    exp_raw, _ = ora.Tokenizer(d_raw).new_worker().tokenize_batch(text, offs)
    tok = V.Tokenizer(dv, device=0)
    _assert_same(tok.tokenize_batch(text=text, offsets=offs), exp, exp_off)
    assert exp.tobytes() != exp_raw.tobytes()


@pytest.mark.gpu
def test_compact_connector_after_id_mapping_and_in_mecab_compat_mode():
    sd = synth.SynthDict("small")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=13)
    dv = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk)
    do = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk)
    rng = random.Random(17)
    lmap = list(range(1, sd.num_left)); rng.shuffle(lmap)
    rmap = list(range(1, sd.num_right)); rng.shuffle(rmap)
    for d in (dv, do):
        d.map_connection_ids_from_iter(lmap, rmap)
    user = sd.user_csv(200)
    dv.reset_user_lexicon_from_reader(user)
    do.reset_user_lexicon(user)
    text, offs = sd.sentences(800, "mixed", space_p=0.05)
    exp, exp_off = ora.Tokenizer(do, ignore_space=True, max_grouping_len=24).new_worker().tokenize_batch(text, offs)
    tok = V.Tokenizer(dv, device=0).ignore_space(True).max_grouping_len(24)
    _assert_same(tok.tokenize_batch(text=text, offsets=offs), exp, exp_off)


@pytest.mark.gpu
@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("fused", ["0", "1"])
def test_compact_connector_costs_outside_i16_use_the_i32_device_matrix(dual, fused, monkeypatch):
    """ConnectorCost::cost of a Raw / Dual connector is an i32 sum (raw_connector.rs:153-161, dual_connector.rs:267-279): a model
    whose costs leave i16 is expanded into an i32 device matrix and swept by the wide instances of the kernels (batch pipeline,
    fused fallback, Worker's single launch) -- bit-exact against the oracle, whose search_min_node adds the same i32 costs."""
    monkeypatch.setenv("VBT_FUSED", fused)
    sd = synth.SynthDict("small")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=31, templates=12, max_abs=30000, empty_pair=not dual)
    dv = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=dual)
    do = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk, dual=dual)
    wide = [(r, l) for r in range(sd.num_right) for l in range(0, sd.num_left, 5) if not -32768 <= do.conn_cost(r, l) <= 32767]
    assert wide and dv.conn_cost(*wide[0]) == do.conn_cost(*wide[0])
    text, offs = sd.sentences(1500, "mixed", space_p=0.05)
    exp, exp_off = ora.Tokenizer(do, ignore_space=True).new_worker().tokenize_batch(text, offs)
    tok = V.Tokenizer(dv, device=0).ignore_space(True)
    _assert_same(tok.tokenize_batch(text=text, offsets=offs), exp, exp_off)
    assert max(abs(int(c)) for c in exp["total_cost"][:2000]) > 40000  # paths whose costs could not come from i16 cells
    if fused == "0":  # the per-sentence path on the same dictionary
        w = tok.new_worker()
        raw = bytes(text)
        k = 0
        for i in range(60):
            w.reset_sentence(raw[int(offs[i]):int(offs[i + 1])])
            w.tokenize()
            n = w.num_tokens()
            assert n == int(exp_off[i + 1] - exp_off[i])
            assert [w.token(t).total_cost for t in range(n)] == exp["total_cost"][k:k + n].tolist()
            k += n
