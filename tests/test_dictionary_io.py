"""Dictionary::read / Dictionary::write: vibrato's `system.dic` (magic + bincode, dictionary.rs:27,142-197,
common.rs:5-9) and the zstd frame the released dictionaries and the reference's CLIs put around it.

The reference cannot be run here and holds no `.dic` fixture, so the layout is checked against an INDEPENDENT pure-Python
decoder written from the reference's struct definitions (below, each with its file:line), and the trie blob against a
pure-Python restatement of crawdad 0.3's common-prefix search (the crate is not under /root/reference: its layout is
restated from the published source, "format parity unpinned" -- DESIGN.md section 9)."""
import io
import json
import os
import random
import struct
import subprocess
import sys

import numpy as np
import pytest

import vibrato_amd as V
from oracle import oracle as ora
from tools import synth

HERE = os.path.dirname(__file__)
RES = os.path.join(HERE, "golden", "resources")
GOLD = json.load(open(os.path.join(HERE, "golden", "unit_golden.json"), encoding="utf-8"))
TOK_GOLD = json.load(open(os.path.join(HERE, "golden", "tokenize_golden.json"), encoding="utf-8"))
MAGIC = b"VibratoTokenizer 0.5\n"  # dictionary.rs:27


def _src(name):
    return open(os.path.join(RES, name), "rb").read()


def fixture_dict(user=False):
    d = V.SystemDictionaryBuilder.from_readers(_src("lex.csv"), _src("matrix.def"), _src("char.def"), _src("unk.def"))
    if user:
        d.reset_user_lexicon_from_reader(_src("user.csv"))
    return d


# ------------------------------------------------------------------ independent decoder (bincode fixed-int LE, common.rs:5-9)

class Dec:
    def __init__(self, b):
        self.b, self.p = b, 0

    def num(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.p)[0]
        self.p += struct.calcsize("<" + fmt)
        return v

    def vec(self, fmt):
        n = self.num("Q")
        v = list(struct.unpack_from(f"<{n}{fmt}", self.b, self.p))
        self.p += n * struct.calcsize("<" + fmt)
        return v

    def string(self):
        n = self.num("Q")
        s = self.b[self.p:self.p + n].decode("utf-8")
        self.p += n
        return s

    def strings(self):
        return [self.string() for _ in range(self.num("Q"))]

    def lexicon(self):  # dictionary/lexicon.rs:23-29, lexicon/map.rs:13-17, map/trie.rs:14-19, map/posting.rs:6-13, param.rs:5-27, feature.rs:3-6
        n = self.num("Q")
        blob = self.b[self.p:self.p + n]
        self.p += n
        postings = self.vec("I")
        params = [(self.num("H"), self.num("H"), self.num("h")) for _ in range(self.num("Q"))]
        features = self.strings()
        return {"trie": blob, "postings": postings, "params": params, "features": features, "lex_type": self.num("I")}

    def matrix(self):  # connector/matrix_connector.rs:11-15
        return {"data": self.vec("h"), "num_right": self.num("Q"), "num_left": self.num("Q")}

    def u31x8(self):  # raw_connector/scorer.rs:62-101: eight u32 per item
        n = self.num("Q")
        v = list(struct.unpack_from(f"<{8 * n}I", self.b, self.p))
        self.p += 32 * n
        return v

    def scorer(self):  # scorer.rs:230-237
        return {"bases": self.vec("I"), "checks": self.vec("I"), "costs": self.vec("i")}

    def dictionary(self):  # dictionary.rs:43-51
        assert self.b[:len(MAGIC)] == MAGIC
        self.p = len(MAGIC)
        d = {"system": self.lexicon()}
        d["user"] = self.lexicon() if self.num("B") else None
        kind = self.num("I")  # connector.rs:30-35
        if kind == 0:
            d["connector"] = ("Matrix", self.matrix())
        elif kind == 1:  # raw_connector.rs:22-27
            d["connector"] = ("Raw", {"right": self.u31x8(), "left": self.u31x8(), "blocks": self.num("Q"), "scorer": self.scorer()})
        else:  # dual_connector.rs:16-23
            d["connector"] = ("Dual", {"matrix": self.matrix(), "right_map": self.vec("H"), "left_map": self.vec("H"),
                                       "right": self.u31x8(), "left": self.u31x8(), "scorer": self.scorer()})
        d["mapper"] = (self.vec("H"), self.vec("H")) if self.num("B") else None  # mapper.rs:9-12
        d["chr2inf"] = self.vec("I")  # character.rs:105-108
        d["categories"] = self.strings()
        d["unk_offsets"] = self.vec("Q")  # unknown.rs:63-66
        d["unk_entries"] = [(self.num("H"), self.num("H"), self.num("H"), self.num("h"), self.string()) for _ in range(self.num("Q"))]
        assert self.p == len(self.b), "trailing bytes"
        return d


def crawdad_common_prefix_search(blob, text):
    """crawdad 0.3 Trie::common_prefix_search restated: yields (value, end_char)."""
    table_len = struct.unpack_from("<I", blob, 0)[0]
    table = struct.unpack_from(f"<{table_len}I", blob, 4)
    alphabet, n_nodes = struct.unpack_from("<II", blob, 4 + 4 * table_len)
    assert len(blob) == 12 + 4 * table_len + 8 * n_nodes
    nodes = struct.unpack_from(f"<{2 * n_nodes}I", blob, 12 + 4 * table_len)
    MASK = 0x7FFFFFFF
    out, node = [], 0
    for pos, ch in enumerate(text):
        c = ord(ch)
        if c >= table_len or table[c] == 0xFFFFFFFF:
            break
        if nodes[2 * node] >> 31:  # a leaf has no children
            break
        child = (nodes[2 * node] & MASK) ^ table[c]
        if child >= n_nodes or (nodes[2 * child + 1] & MASK) != node:
            break
        node = child
        if nodes[2 * node] >> 31:
            out.append((nodes[2 * node] & MASK, pos + 1))
            break
        if nodes[2 * node + 1] >> 31:  # has_leaf: the value sits in the child on the end code 0
            leaf = (nodes[2 * node] & MASK) ^ 0
            assert (nodes[2 * leaf + 1] & MASK) == node and nodes[2 * leaf] >> 31
            out.append((nodes[2 * leaf] & MASK, pos + 1))
    return out


def packed(ci):
    """CharInfo bit layout, character.rs:10-24"""
    return ci["cate_idset"] | ci["base_id"] << 18 | ci["invoke"] << 26 | ci["group"] << 27 | ci["length"] << 28


def parse_lex(csv):
    rows = []
    for line in csv.decode("utf-8").splitlines():
        if line:
            f = line.split(",", 4)
            rows.append((f[0], int(f[1]), int(f[2]), int(f[3]), f[4]))
    return rows


# ------------------------------------------------------------------ what Dictionary::write emits

def test_written_bytes_decode_field_by_field_to_the_sources():
    d = fixture_dict(user=True)
    raw = d.write()
    got = Dec(raw).dictionary()
    for lex, name, ty in ((got["system"], "lex.csv", 0), (got["user"], "user.csv", 1)):
        rows = parse_lex(_src(name))
        assert lex["lex_type"] == ty
        assert lex["params"] == [(l, r, c) for _, l, r, c, _ in rows]
        assert lex["features"] == [f for *_, f in rows]
        # WordMapBuilder::build (map.rs:57-73): BTreeMap order of the surfaces, postings = [len, ids...] per surface
        by_surface = {}
        for i, (s, *_rest) in enumerate(rows):
            by_surface.setdefault(s, []).append(i)
        expect_post = []
        offsets = {}
        for s in sorted(by_surface, key=lambda s: s.encode("utf-8")):
            offsets[s] = len(expect_post)
            expect_post += [len(by_surface[s])] + by_surface[s]
        assert lex["postings"] == expect_post
        for s, off in offsets.items():  # every key is found by crawdad's search, with its postings offset and all shorter keys
            hits = crawdad_common_prefix_search(lex["trie"], s)
            assert hits[-1] == (off, len(s))
            assert hits == [(offsets[s[:k]], k) for k in range(1, len(s) + 1) if s[:k] in offsets]
        assert crawdad_common_prefix_search(lex["trie"], "zz") == []
    kind, m = got["connector"]
    lines = _src("matrix.def").decode().split("\n")
    nr, nl = map(int, lines[0].split())
    assert kind == "Matrix" and (m["num_right"], m["num_left"]) == (nr, nl) and len(m["data"]) == nr * nl
    for ln in lines[1:]:
        if ln.strip():
            r, l, c = map(int, ln.split())
            assert m["data"][l * nr + r] == c  # matrix_connector.rs:47
    assert got["mapper"] is None
    assert len(got["chr2inf"]) == 65536 and got["categories"][0] == "DEFAULT"
    for cp in (0, 0x20, 0x30, 0x41, 0x3042, 0x30A2, 0x4E00, 0xFFFF):
        assert got["chr2inf"][cp] == packed(d.char_info(cp))
    unk = parse_lex(_src("unk.def"))
    assert got["unk_offsets"][0] == 0 and got["unk_offsets"][-1] == len(unk) == len(got["unk_entries"])
    assert len(got["unk_offsets"]) == len(got["categories"]) + 1
    assert sorted((got["categories"][c], l, r, w, f) for c, l, r, w, f in got["unk_entries"]) == sorted(unk)
    for i, (c, *_r) in enumerate(got["unk_entries"]):
        assert got["unk_offsets"][c] <= i < got["unk_offsets"][c + 1]


def test_crawdad_blob_answers_the_reference_enumeration_vectors():
    """vibrato/src/tests/lexicon.rs:8-57 through the written blob + postings + params"""
    got = Dec(fixture_dict().write()).dictionary()["system"]
    for case in GOLD["lexicon_common_prefix"]:
        if case.get("dict") != "fixture":
            continue
        out = []
        for value, end in crawdad_common_prefix_search(got["trie"], case["input"]):
            n = got["postings"][value]
            for wid in got["postings"][value + 1:value + 1 + n]:
                out.append([wid, end, *got["params"][wid]])
        assert out == case["expect"]


# ------------------------------------------------------------------ read(write(x)) == x

def _same_dictionary(a, b, n_conn=None):
    assert (a.num_left, a.num_right, a.connector_kind) == (b.num_left, b.num_right, b.connector_kind)
    for ty in (0, 1, 2):
        assert a.num_words(ty) == b.num_words(ty)
        for w in range(a.num_words(ty)):
            assert a.word_param(ty, w) == b.word_param(ty, w) and a.word_feature(ty, w) == b.word_feature(ty, w)
    rng = random.Random(1)
    pairs = [(r, l) for r in range(a.num_right) for l in range(a.num_left)]
    for r, l in (pairs if n_conn is None else rng.sample(pairs, n_conn)):
        assert a.conn_cost(r, l) == b.conn_cost(r, l)
    for cp in list(range(0, 0x3100, 7)) + [0x4E00, 0x9FFF, 0xFF21, 0xFFFF]:
        assert a.char_info(cp) == b.char_info(cp)


@pytest.mark.parametrize("user", [False, True])
@pytest.mark.parametrize("mapped", [False, True])
def test_round_trip_fixture(user, mapped):
    d = fixture_dict(user)
    if mapped:
        rng = random.Random(9)
        lmap = list(range(1, d.num_left)); rng.shuffle(lmap)
        rmap = list(range(1, d.num_right)); rng.shuffle(rmap)
        d.map_connection_ids_from_iter(lmap, rmap)
    raw = d.write()
    e = V.Dictionary.read(raw)
    _same_dictionary(d, e)
    assert e.write() == raw
    for text in ("東京都に行く", "X", "京都東京都京都", "自然言語処理"):
        for ty in ((0, 1) if user else (0,)):
            assert d.common_prefix(text, ty) == e.common_prefix(text, ty)
    if mapped:  # the stored mapper survives: a user lexicon attached after read() goes through it (dictionary.rs:214-217)
        assert Dec(raw).dictionary()["mapper"] is not None
        d.reset_user_lexicon_from_reader(_src("user.csv"))
        e.reset_user_lexicon_from_reader(_src("user.csv"))
        assert [d.word_param(1, w) for w in range(d.num_words(1))] == [e.word_param(1, w) for w in range(e.num_words(1))]


def test_round_trip_synthetic_lexicon_30k_words():
    sd = synth.SynthDict("small")
    d = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    d.reset_user_lexicon_from_reader(sd.user_csv(300))
    raw = d.write()
    e = V.Dictionary.read(raw)
    _same_dictionary(d, e, n_conn=2000)
    assert e.write() == raw
    text, offs = sd.sentences(50, "lognormal_40")
    for i in range(50):
        s = bytes(text[int(offs[i]):int(offs[i + 1])]).decode("utf-8")
        for k in range(0, len(s), 5):
            assert d.common_prefix(s[k:], 0) == e.common_prefix(s[k:], 0)
            assert d.common_prefix(s[k:], 1) == e.common_prefix(s[k:], 1)


# ------------------------------------------------------------------ zstd frame, errors

def test_zstd_frames():
    import pyarrow as pa
    d = fixture_dict()
    raw = d.write()
    z19 = d.write(zstd_level=19)
    assert z19[:4] == b"\x28\xb5\x2f\xfd" and len(z19) < len(raw) // 20
    assert V.Dictionary.read(z19).write() == raw
    assert bytes(pa.decompress(z19, decompressed_size=len(raw), codec="zstd")) == raw  # a zstd frame other tools read
    other = bytes(pa.compress(raw, codec="zstd", asbytes=True))  # and one written by another compressor
    assert V.Dictionary.read(io.BytesIO(other)).write() == raw
    with pytest.raises(V.VibratoError):
        V.Dictionary.read(z19[:len(z19) // 2])


def test_read_errors():
    raw = fixture_dict(user=True).write()
    with pytest.raises(V.VibratoError, match="magic number"):  # dictionary.rs:188-193
        V.Dictionary.read(b"VibratoTokenizer 0.4\n" + raw[21:])
    with pytest.raises(V.VibratoError):
        V.Dictionary.read(b"")
    rng = random.Random(4)
    for cut in [21, 22, 29, 100, len(raw) // 3, len(raw) - 1]:
        with pytest.raises(V.VibratoError):
            V.Dictionary.read(raw[:cut])
    with pytest.raises(V.VibratoError):
        V.Dictionary.read(raw + b"\0")
    bad = 0
    for _ in range(200):  # corrupted bytes either fail loudly or decode to a self-consistent dictionary; never crash
        b = bytearray(raw)
        for _ in range(rng.randint(1, 4)):
            b[rng.randrange(21, min(len(b), 40000))] ^= 1 << rng.randrange(8)
        try:
            V.Dictionary.read(bytes(b))
        except V.VibratoError:
            bad += 1
    assert bad > 20


# ------------------------------------------------------------------ compact connectors in the container

def _enc_u64(v):
    return struct.pack("<Q", v)


def _enc_vec(fmt, v):
    return _enc_u64(len(v)) + struct.pack(f"<{len(v)}{fmt}", *v)


def test_raw_connector_round_trip():
    from test_compact_connector import synth_bigram
    sd = synth.SynthDict("tiny")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=3)
    d = V.SystemDictionaryBuilder.from_readers_with_bigram_info(sd.lex, right, left, cost, sd.char_def, sd.unk)
    raw = d.write()
    kind, c = Dec(raw).dictionary()["connector"]
    assert kind == "Raw" and c["blocks"] == 2 and len(c["right"]) == sd.num_right * 16 and c["right"][:16] == [0] * 16
    e = V.Dictionary.read(raw)
    _same_dictionary(d, e)
    assert e.write() == raw


def test_dual_connector_read_cost_map_write():
    """A DualConnector container assembled by hand from the struct definitions (dual_connector.rs:16-23; the product has no
    dual *builder*: the split into matrix + raw part is a memory layout of the same cost function, dual_connector.rs:141-198):
    cost(r, l) = matrix[left_map[l]][right_map[r]] + scorer(right_feats[r], left_feats[l]) (dual_connector.rs:267-279)."""
    base = Dec(V.SystemDictionaryBuilder.from_readers("a,1,1,0,x\nb,2,3,5,y", "4 4\n" + "\n".join(f"{r} {l} 0" for r in range(4) for l in range(4)),
                                                      "DEFAULT 0 1 0", "DEFAULT,0,0,0,*").write())
    full = base.b
    base.dictionary()
    # locate the connector: re-encode everything around it
    d0 = Dec(full)
    d0.p = len(MAGIC)
    d0.lexicon()
    assert d0.num("B") == 0
    conn_at = d0.p
    assert d0.num("I") == 0
    d0.matrix()
    conn_end = d0.p
    rng = random.Random(2)
    mat = [rng.randint(-500, 500) for _ in range(3 * 2)]  # small matrix: 3 right classes x 2 left classes, data[l * 3 + r]
    right_map, left_map = [0, 1, 2, 1], [0, 1, 1, 0]
    scorer_pairs = {(5, 6): 40, (7, 6): -15, (5, 9): 3}
    bases = [0] * 8
    bases[7] = 8  # (7, 6) -> slot 14, (5, 6) -> 6, (5, 9) -> 9
    checks, costs = [0xFFFFFFFF] * 16, [0] * 16
    for (k1, k2), c in scorer_pairs.items():
        checks[bases[k1] ^ k2], costs[bases[k1] ^ k2] = k1, c
    INV = 0x7FFFFFFF
    right_feats = [[0] * 8, [5, INV, 7, INV, INV, INV, INV, INV], [7, 7, INV, INV, INV, INV, INV, INV], [INV] * 8]
    left_feats = [[0] * 8, [6, INV, 6, INV, INV, INV, INV, INV], [9, 6, INV, INV, INV, INV, INV, INV], [6] * 8]
    dual = (struct.pack("<I", 2) + _enc_vec("h", mat) + _enc_u64(3) + _enc_u64(2) + _enc_vec("H", right_map) + _enc_vec("H", left_map)
            + _enc_u64(4) + b"".join(struct.pack("<8I", *r) for r in right_feats)
            + _enc_u64(4) + b"".join(struct.pack("<8I", *r) for r in left_feats)
            + _enc_vec("I", bases) + _enc_vec("I", checks) + _enc_vec("i", costs))
    raw = full[:conn_at] + dual + full[conn_end:]
    d = V.Dictionary.read(raw)
    assert d.connector_kind == "Dual" and (d.num_right, d.num_left) == (4, 4)

    def expect(r, l):
        s = sum(scorer_pairs.get((a, b), 0) for a, b in zip(right_feats[r], left_feats[l]))
        return mat[left_map[l] * 3 + right_map[r]] + s
    table = {(r, l): expect(r, l) for r in range(4) for l in range(4)}
    assert {(r, l): d.conn_cost(r, l) for r in range(4) for l in range(4)} == table
    assert table[(1, 1)] == mat[1 * 3 + 1] + 40 - 15
    assert d.write() == raw
    d.map_connection_ids_from_iter([3, 1, 2], [2, 3, 1])  # new left 1 <- old 3 ...; the small matrix is renumbered by first use
    lnew, rnew = {0: 0, 3: 1, 1: 2, 2: 3}, {0: 0, 2: 1, 3: 2, 1: 3}
    assert {(rnew[r], lnew[l]): c for (r, l), c in table.items()} == {(r, l): d.conn_cost(r, l) for r in range(4) for l in range(4)}
    e = V.Dictionary.read(d.write())
    assert {(r, l): e.conn_cost(r, l) for r in range(4) for l in range(4)} == {(r, l): d.conn_cost(r, l) for r in range(4) for l in range(4)}
    _kind, c = Dec(d.write()).dictionary()["connector"]
    assert c["left_map"] == [0, 0, 1, 1] and c["right_map"] == [0, 1, 2, 2]  # dual_connector.rs:237-262


# ------------------------------------------------------------------ GPU: a dictionary that went through the container

@pytest.mark.gpu
def test_golden_vectors_through_a_read_dictionary():
    for case in TOK_GOLD["cases"]:
        if case["dict"] != "fixture":
            continue
        d = V.Dictionary.read(fixture_dict(case["user"]).write(zstd_level=3))
        tok = V.Tokenizer(d, device=0).ignore_space(case["ignore_space"]).max_grouping_len(case["max_grouping_len"])
        w = tok.new_worker()
        for s in case["sentences"]:
            w.reset_sentence(s["text"])
            w.tokenize()
            assert w.num_tokens() == s["num_tokens"], case["name"]
            for exp in s["tokens"]:
                got = w.token(exp["index"])
                for k in ("surface", "feature", "total_cost"):
                    if k in exp:
                        assert getattr(got, k) == exp[k], (case["name"], k)
                for k in ("range_char", "range_byte"):
                    if k in exp:
                        assert list(getattr(got, k)) == exp[k], (case["name"], k)


@pytest.mark.gpu
def test_read_dictionary_matches_oracle_on_synthetic_batch_and_cli_takes_system_dic_zst(tmp_path):
    sd = synth.SynthDict("small")
    d = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    path = tmp_path / "system.dic.zst"
    path.write_bytes(d.write(zstd_level=3))
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    text, offs = sd.sentences(1500, "lognormal_40")
    exp, exp_off = ora.Tokenizer(do).new_worker().tokenize_batch(text, offs)
    tok = V.Tokenizer(V.Dictionary.read(str(path)), device=0)
    batch = tok.tokenize_batch(text=text, offsets=offs)
    got, got_off = batch.tokens_in_order()
    assert np.array_equal(got_off, exp_off)
    for f in V.TOKEN_DTYPE.names:
        assert np.array_equal(got[f], exp[f]), f
    lines = b"".join(bytes(text[int(offs[i]):int(offs[i + 1])]) + b"\n" for i in range(200))
    out = subprocess.run([sys.executable, "-m", "vibrato_amd.cli", "-i", str(path), "-O", "detail"], input=lines, capture_output=True,
                         cwd=os.path.dirname(HERE), check=True).stdout
    sub_offs = offs[:201]
    assert out.decode("utf-8") == tok.tokenize_batch(text=text[:int(sub_offs[-1])], offsets=sub_offs).format("detail")


# ------------------------------------------------------------------ loader hardening (no reference-written file exists offline)

def _split_system_trie(raw):
    """(prefix, trie blob, suffix) of a written container: the system lexicon's trie is its first field (Vec<u8>)."""
    at = len(MAGIC)
    n = struct.unpack_from("<Q", raw, at)[0]
    return raw[:at], raw[at + 8:at + 8 + n], raw[at + 8 + n:]


def test_unbounded_alphabet_field_is_rejected_before_any_allocation():
    """a u32 alphabet_size of 0xFFFFFFFF used to size a 16 GiB inverse table; it is refused from the field alone."""
    import resource
    raw = fixture_dict().write()
    head, blob, tail = _split_system_trie(raw)
    table_len = struct.unpack_from("<I", blob, 0)[0]
    at = 4 + 4 * table_len
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for bogus in (0xFFFFFFFF, 0x7FFFFFFF, table_len + 2):
        b = bytearray(blob)
        struct.pack_into("<I", b, at, bogus)
        with pytest.raises(V.VibratoError, match="alphabet"):
            V.Dictionary.read(head + struct.pack("<Q", len(b)) + bytes(b) + tail)
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 200 * 1024  # KiB: nothing alphabet-sized was allocated


def _assemble_crawdad_blob(keys, direct_leaves):
    """A crawdad 0.3 blob assembled by hand (independent of csrc/dictio.cpp's writer): first-fit XOR double array.
    direct_leaves=True: a key that is no proper prefix of another key ends AT the node reached by its last character (is_leaf on
    that node, value in its base); only keys with continuations use the has_leaf flag + a leaf child on the end code 0.
    direct_leaves=False: every key ends in a leaf child on the end code."""
    MASK, TOP = 0x7FFFFFFF, 0x80000000
    chars = sorted({c for k in keys for c in k})
    code = {c: i + 1 for i, c in enumerate(chars)}  # 0 = end code
    table = [0xFFFFFFFF] * (max(map(ord, chars)) + 1)
    for c, v in code.items():
        table[ord(c)] = v
    base, check, used = {0: 0}, {0: MASK}, {0}
    prefixes = {k[:i] for k in keys for i in range(len(k))}  # proper prefixes of some key

    def place(node, prefix):
        labels = sorted({code[k[len(prefix)]] for k in keys if k.startswith(prefix) and len(k) > len(prefix)})
        ends_here = prefix in keys
        if ends_here:
            labels = [0] + labels
        if not labels:
            return
        b = 0
        while any((b ^ l) in used for l in labels):
            b += 1
        base[node] = b | (base.get(node, 0) & TOP)
        for l in labels:
            used.add(b ^ l)
            check[b ^ l] = node
            base[b ^ l] = 0
        if ends_here:
            base[b] = keys[prefix] | TOP
            check[node] |= TOP
        inv = {v: c for c, v in code.items()}
        for l in labels:
            if l == 0:
                continue
            child, p2 = b ^ l, prefix + inv[l]
            if direct_leaves and p2 in keys and p2 not in prefixes:
                base[child] = keys[p2] | TOP  # the key ends at this very node
            else:
                place(child, p2)

    place(0, "")
    n_nodes = max(used) + 1
    nodes = []
    for i in range(n_nodes):
        nodes += [base.get(i, MASK), check.get(i, MASK)]
    return struct.pack("<I", len(table)) + struct.pack(f"<{len(table)}I", *table) + struct.pack("<II", len(chars) + 1, n_nodes) + \
        struct.pack(f"<{2 * n_nodes}I", *nodes)


@pytest.mark.parametrize("direct_leaves", [False, True])
def test_reader_accepts_both_terminal_encodings_of_a_hand_assembled_trie(direct_leaves):
    """crawdad keeps a key's value either in the node its last character reaches (is_leaf) or, when the key has continuations, in
    a leaf child on the end code (has_leaf): a container whose system trie is assembled by hand in either convention reads back
    to the same dictionary (every enumeration vector of vibrato/src/tests/lexicon.rs, every word id)."""
    d0 = fixture_dict()
    raw = d0.write()
    head, blob0, tail = _split_system_trie(raw)
    got = Dec(raw).dictionary()["system"]
    rows = parse_lex(_src("lex.csv"))
    by_surface = {}
    for i, (s, *_r) in enumerate(rows):
        by_surface.setdefault(s, []).append(i)
    keys, off = {}, 0
    for s in sorted(by_surface, key=lambda s: s.encode("utf-8")):
        keys[s] = off
        off += 1 + len(by_surface[s])
    assert off == len(got["postings"])
    blob = _assemble_crawdad_blob(keys, direct_leaves)
    for s, v in keys.items():  # the independent search over the hand-assembled blob finds every key
        assert crawdad_common_prefix_search(blob, s)[-1] == (v, len(s))
    if direct_leaves:
        n_direct = sum(1 for s in keys if not any(o != s and o.startswith(s) for o in keys))
        assert 0 < n_direct < len(keys)  # both encodings occur in one trie
    d1 = V.Dictionary.read(head + struct.pack("<Q", len(blob)) + blob + tail)
    for case in GOLD["lexicon_common_prefix"]:
        if case.get("dict") == "fixture":
            assert d1.common_prefix(case["input"]) == case["expect"] == d0.common_prefix(case["input"])
    for s in keys:
        assert d1.common_prefix(s) == d0.common_prefix(s)
    assert d1.write() == raw  # re-written by this project's writer: the same bytes as the original


@pytest.mark.parametrize("kind", ["matrix", "raw", "dual"])
def test_verify_dic_tool_walks_and_checks_a_container(kind, tmp_path):
    """tools/verify_dic.py (the one command that pins the format the day a reference-written file is at hand) passes on
    containers of every connector kind written here, and fails on a blob whose length field is off."""
    if kind == "matrix":
        d = fixture_dict(user=True)
    else:
        from test_compact_connector import synth_bigram
        sd = synth.SynthDict("tiny")
        right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=5)
        d = V.SystemDictionaryBuilder.from_readers_with_bigram_info(sd.lex, right, left, cost, sd.char_def, sd.unk, dual_connector=(kind == "dual"))
    path = tmp_path / "system.dic.zst"
    path.write_bytes(d.write(zstd_level=1))
    tool = os.path.join(os.path.dirname(HERE), "tools", "verify_dic.py")
    p = subprocess.run([sys.executable, tool, str(path), "東京都"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ALL CHECKS PASSED" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    if kind == "matrix":
        assert "word        6 end_char 3 left 6 right 8 cost 5320" in p.stdout  # vibrato/src/tests/lexicon.rs vector
        raw = bytearray(d.write())
        struct.pack_into("<I", raw, len(MAGIC) + 8, struct.unpack_from("<I", raw, len(MAGIC) + 8)[0] + 1)  # table_len + 1
        bad = tmp_path / "bad.dic"
        bad.write_bytes(bytes(raw))
        p = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and "blob length" in p.stdout
