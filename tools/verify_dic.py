#!/usr/bin/env python3
"""Pins the container format against a REAL dictionary in one command:  python tools/verify_dic.py system.dic[.zst] [text ...]

No reference-written `system.dic` exists offline, so the byte layout of the trie blob inside it (crate crawdad 0.3, not under
/root/reference) is "format parity unpinned" (DESIGN.md section 9).  Given any compiled dictionary -- e.g. the released
ipadic-mecab-2_7_0/system.dic.zst -- this tool

  1. unwraps the zstd frame (system libzstd through ctypes) and walks the bincode container field by field with an independent
     pure-Python decoder written from the reference's struct definitions (dictionary.rs:43-51, lexicon.rs:23-29, map.rs:13-17,
     map/trie.rs:14-19, posting.rs:6-13, param.rs:5-27, connector/*.rs, mapper.rs:9-12, character.rs:105-108, unknown.rs:20-27,62-66),
     printing every size it meets, so that a layout mismatch shows WHERE the stream stops making sense;
  2. checks the crawdad blob structurally (blob length == 12 + 4 * table_len + 8 * n_nodes, code table injective, alphabet within
     the table, every postings offset reached exactly once by walking the double array from the root, ascending word ids);
  3. runs the enumeration contract of vibrato/src/tests/lexicon.rs on the file's own data: common-prefix search (pure-Python
     restatement of crawdad's search) of the given texts -- default: a few Japanese probes -- printing word id, end, left/right id,
     cost and feature, ready to be compared with `vibrato`'s own output for the same file;
  4. reads the same bytes with the product's reader (vbt_dict_read) and compares: word counts, connector kind and dimensions, and
     the same common-prefix results through the product's own double array.

Exit code 0 = every check passed.  Needs no GPU.
"""
import ctypes
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MAGIC = b"VibratoTokenizer 0.5\n"  # dictionary.rs:27
MASK, TOP, INVALID = 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF


def unzstd(data):
    if data[:4] != b"\x28\xb5\x2f\xfd":
        return data, False
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_createDStream.restype = ctypes.c_void_p
    z.ZSTD_decompressStream.restype = ctypes.c_size_t
    z.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    z.ZSTD_freeDStream.argtypes = [ctypes.c_void_p]

    class Buf(ctypes.Structure):
        _fields_ = [("p", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]
    ds = z.ZSTD_createDStream()
    src = ctypes.create_string_buffer(data, len(data))
    inb = Buf(ctypes.cast(src, ctypes.c_void_p), len(data), 0)
    out, chunk = bytearray(), ctypes.create_string_buffer(1 << 24)
    while True:
        ob = Buf(ctypes.cast(chunk, ctypes.c_void_p), len(chunk), 0)
        rc = z.ZSTD_decompressStream(ds, ctypes.byref(ob), ctypes.byref(inb))
        if z.ZSTD_isError(rc):
            raise SystemExit("zstd: corrupt frame")
        out += chunk.raw[:ob.pos]
        if rc == 0 and inb.pos == inb.size:
            break
        if ob.pos == 0 and inb.pos == inb.size:
            raise SystemExit("zstd: truncated frame")
    z.ZSTD_freeDStream(ds)
    return bytes(out), True


class Dec:
    """bincode 2, little endian, fixed-width integers, u64 lengths (common.rs:5-9)."""

    def __init__(self, b, at=0):
        self.b, self.p = b, at

    def num(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.p)[0]
        self.p += struct.calcsize("<" + fmt)
        return v

    def length(self, what, each=1):
        n = self.num("Q")
        if n * each > len(self.b) - self.p:
            raise SystemExit(f"  !! {what}: length {n} x {each} B does not fit the remaining {len(self.b) - self.p} bytes at offset {self.p - 8}")
        return n

    def vec(self, fmt, what):
        each = struct.calcsize("<" + fmt)
        n = self.length(what, each)
        v = struct.unpack_from(f"<{n}{fmt}", self.b, self.p)
        self.p += n * each
        return v

    def blob(self, what):
        n = self.length(what)
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def strings(self, what):
        out = []
        for _ in range(self.length(what, 8)):
            out.append(self.blob(what + " item").decode("utf-8"))
        return out


def lexicon(d, name):
    print(f"{name} @ {d.p}")
    lx = {"trie": d.blob(name + ".map.trie")}
    print(f"  trie blob      {len(lx['trie']):>12,} B")
    lx["postings"] = d.vec("I", name + ".map.postings")
    print(f"  postings       {len(lx['postings']):>12,} u32")
    n = d.length(name + ".params", 6)
    raw = d.b[d.p:d.p + 6 * n]
    d.p += 6 * n
    lx["params"] = list(struct.iter_unpack("<HHh", raw))
    print(f"  params         {n:>12,} x (u16, u16, i16)")
    lx["features"] = d.strings(name + ".features")
    print(f"  features       {len(lx['features']):>12,} strings")
    lx["lex_type"] = d.num("I")
    print(f"  lex_type       {lx['lex_type']}")
    return lx


def check_trie(lx, name):
    """structural self-checks of the crawdad blob; returns (table, nodes, ok)"""
    b = lx["trie"]
    ok = True

    def fail(msg):
        nonlocal ok
        ok = False
        print(f"  !! {name} trie: {msg}")
    table_len = struct.unpack_from("<I", b, 0)[0]
    if 4 + 4 * table_len + 8 > len(b):
        fail(f"table_len {table_len} exceeds the blob ({len(b)} B)")
        return None, None, False
    table = struct.unpack_from(f"<{table_len}I", b, 4)
    alphabet, n_nodes = struct.unpack_from("<II", b, 4 + 4 * table_len)
    print(f"  crawdad: table_len {table_len:,}  alphabet_size {alphabet:,}  nodes {n_nodes:,}  (blob {len(b):,} B, expected {12 + 4 * table_len + 8 * n_nodes:,})")
    if len(b) != 12 + 4 * table_len + 8 * n_nodes:
        fail("blob length != 12 + 4 * table_len + 8 * n_nodes -> the serialized layout differs from the restated one")
        return None, None, False
    nodes = struct.unpack_from(f"<{2 * n_nodes}I", b, 12 + 4 * table_len)
    mapped = [c for c in table if c != INVALID]
    if len(set(mapped)) != len(mapped):
        fail("two code points share a code")
    if mapped and max(mapped) >= alphabet:
        fail(f"a code ({max(mapped)}) is outside the alphabet ({alphabet})")
    print(f"  code table: {len(mapped):,} mapped code points, U+0000 -> {table[0] if table_len else None} (the end code is 0)")
    # every postings offset must be the value of exactly one key: walk the double array from the root
    inv = {}
    for cp, c in enumerate(table):
        if c != INVALID:
            inv[c] = cp
    children = {}
    for i in range(1, n_nodes):
        p = nodes[2 * i + 1] & MASK
        if p < n_nodes and p != i and not (nodes[2 * p] >> 31):
            c = (nodes[2 * p] & MASK) ^ i
            if c == 0 or c in inv:
                children.setdefault(p, []).append((c, i))
    values, direct, via_end, stack = [], 0, 0, [(0, 0)]
    seen = {0}
    while stack:
        n, code = stack.pop()
        if nodes[2 * n] >> 31:
            values.append(nodes[2 * n] & MASK)
            if code == 0:
                via_end += 1
            else:
                direct += 1
            continue
        for c, ch in children.get(n, ()):
            if ch in seen:
                fail("a node is reachable twice")
                continue
            seen.add(ch)
            stack.append((ch, c))
    print(f"  keys: {len(values):,} ({direct:,} end at the node of their last character, {via_end:,} in a leaf child on the end code)")
    expect, off, post = [], 0, lx["postings"]
    while off < len(post):
        expect.append(off)
        off += 1 + post[off]
    if off != len(post):
        fail("postings do not parse as [len, ids...] runs")
    if sorted(values) != expect:
        fail(f"trie values ({len(values):,}) are not exactly the postings offsets ({len(expect):,})")
    ids = [w for o in expect for w in post[o + 1:o + 1 + post[o]]]
    if sorted(ids) != list(range(len(lx["params"]))):
        fail("postings do not name every word id exactly once")
    return table, nodes, ok


def common_prefix(table, nodes, lx, text):
    """crawdad Trie::common_prefix_search restated + Lexicon::common_prefix_iterator (lexicon.rs:33-46)"""
    out, node, n_nodes = [], 0, len(nodes) // 2
    for pos, ch in enumerate(text):
        c = ord(ch)
        if c >= len(table) or table[c] == INVALID or nodes[2 * node] >> 31:
            break
        child = (nodes[2 * node] & MASK) ^ table[c]
        if child >= n_nodes or (nodes[2 * child + 1] & MASK) != node:
            break
        node = child
        value = None
        if nodes[2 * node] >> 31:
            value = nodes[2 * node] & MASK
        elif nodes[2 * node + 1] >> 31:
            value = nodes[2 * (nodes[2 * node] & MASK)] & MASK
        if value is not None:
            k = lx["postings"][value]
            for wid in lx["postings"][value + 1:value + 1 + k]:
                out.append([wid, pos + 1, *lx["params"][wid]])
    return out


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    data = open(sys.argv[1], "rb").read()
    probes = sys.argv[2:] or ["東京都", "自然言語処理", "すもももももももものうち", "京都東京都京都", "abc"]
    raw, was_zstd = unzstd(data)
    print(f"{sys.argv[1]}: {len(data):,} B" + (f" (zstd frame) -> {len(raw):,} B" if was_zstd else ""))
    if raw[:len(MAGIC)] != MAGIC:
        raise SystemExit(f"  !! magic: {raw[:len(MAGIC)]!r}, expected {MAGIC!r} (dictionary.rs:188-193)")
    d = Dec(raw, len(MAGIC))
    ok = True
    system = lexicon(d, "system_lexicon")
    user = lexicon(d, "user_lexicon") if d.num("B") else None
    tag = d.num("I")
    kind = {0: "Matrix", 1: "Raw", 2: "Dual"}.get(tag)
    print(f"connector @ {d.p - 4}: tag {tag} = {kind}")
    if kind == "Matrix":
        m = d.vec("h", "matrix.data")
        nr, nl = d.num("Q"), d.num("Q")
        print(f"  matrix {nr} x {nl} (num_right x num_left), {len(m):,} cells")
        ok &= len(m) == nr * nl
        dims = (nr, nl)
    elif kind in ("Raw", "Dual"):
        if kind == "Dual":
            m = d.vec("h", "dual.matrix.data")
            mr, ml = d.num("Q"), d.num("Q")
            rmap, lmap = d.vec("H", "dual.right_conn_id_map"), d.vec("H", "dual.left_conn_id_map")
            print(f"  dual: matrix {mr} x {ml}, id maps {len(rmap):,} / {len(lmap):,}")
        right = d.vec("I", "right_feat_ids")  # Vec<U31x8>: the length counts 8-word items
        d.p += 7 * 4 * len(right)
        left = d.vec("I", "left_feat_ids")
        d.p += 7 * 4 * len(left)
        width = d.num("Q") if kind == "Raw" else 8
        bases, checks, costs = d.vec("I", "scorer.bases"), d.vec("I", "scorer.checks"), d.vec("i", "scorer.costs")
        print(f"  feature rows {len(right):,} right / {len(left):,} left x 8 lanes, template size {width}; scorer {len(bases):,} bases, {len(checks):,} cells")
        dims = None
    else:
        raise SystemExit("  !! unknown connector tag")
    if d.num("B"):
        lm, rm = d.vec("H", "mapper.left"), d.vec("H", "mapper.right")
        print(f"mapper: {len(lm):,} left / {len(rm):,} right ids")
    chr2inf = d.vec("I", "char_prop.chr2inf")
    cats = d.strings("char_prop.categories")
    print(f"char_prop: chr2inf {len(chr2inf):,} entries, categories {cats}")
    ok &= len(chr2inf) == 65536
    offs = d.vec("Q", "unk_handler.offsets")
    n_unk = d.length("unk_handler.entries", 16)
    for _ in range(n_unk):
        d.p += 8
        d.blob("unk feature")
    print(f"unk_handler: {len(offs)} offsets, {n_unk} entries")
    if d.p != len(raw):
        print(f"  !! {len(raw) - d.p} trailing bytes")
        ok = False
    tries = {}
    for name, lx in (("system", system), ("user", user)):
        if lx:
            table, nodes, good = check_trie(lx, name)
            ok &= good
            tries[name] = (table, nodes, lx)
    res = {}
    if tries.get("system", (None,))[0] is not None:
        table, nodes, lx = tries["system"]
        for t in probes:
            res[t] = common_prefix(table, nodes, lx, t)
            print(f"common_prefix({t!r}):")
            for wid, end, l, r, c in res[t]:
                print(f"    word {wid:>8} end_char {end} left {l} right {r} cost {c}  {lx['features'][wid]}")
    # the product's reader on the same bytes
    import vibrato_amd as V
    try:
        dv = V.Dictionary.read(data)
    except V.VibratoError as e:
        print(f"  !! product reader (vbt_dict_read): {e}")
        ok = False
    else:
        print(f"product reader: {dv.num_words(0):,} system words, {dv.num_words(1):,} user words, connector {dv.connector_kind} "
              f"{dv.num_right} x {dv.num_left}")
        ok &= dv.num_words(0) == len(system["params"]) and dv.connector_kind == kind
        if dims:
            ok &= (dv.num_right, dv.num_left) == dims
        for t, exp in res.items():
            got = dv.common_prefix(t)
            if got != exp:
                print(f"  !! common_prefix({t!r}) differs between the product's double array and the file's: {got[:3]} vs {exp[:3]}")
                ok = False
    print("ALL CHECKS PASSED" if ok else "CHECKS FAILED")
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
