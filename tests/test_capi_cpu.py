"""Host-side tests of libvibrato_hip.so that need no GPU: the library loads, exports every
symbol include/vibrato_hip.h declares, builds dictionaries exactly like the reference's
builder (golden vectors) and agrees with the oracle's lexicon lookups.  No compute calls."""
import os
import random
import re

import numpy as np
import pytest

import vibrato_amd as V
from vibrato_amd import _native as N
from oracle import oracle as ora
from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture_dict(src):
    return V.SystemDictionaryBuilder.from_readers(src["lex.csv"], src["matrix.def"], src["char.def"], src["unk.def"])


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vibrato_hip.h")).read()
    declared = set(re.findall(r"VBT_API[^;(]*?\b(vbt_\w+)\s*\(", header))
    assert len(declared) >= 35
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    L = N.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_rust_sys_crate_declares_every_entry_point_and_the_stats_struct():
    """rust/vibrato-hip-sys is what a vibrato maintainer binds; it cannot be compiled here, so at least its extern
    block has to list exactly the header's entry points and its mirror of vbt_call_stats the header's fields, in order."""
    header = open(os.path.join(ROOT, "include", "vibrato_hip.h")).read()
    sys_rs = open(os.path.join(ROOT, "rust", "vibrato-hip-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"VBT_API[^;(]*?\b(vbt_\w+)\s*\(", header))
    assert declared == set(re.findall(r"pub fn (vbt_\w+)\s*\(", sys_rs))
    body = re.search(r"typedef struct vbt_call_stats \{(.*?)\} vbt_call_stats;", header, re.S).group(1)
    c_fields = [f.strip() for decl in body.split(";") if decl.strip() for f in decl.strip().split(None, 1)[1].split(",")]
    rs_body = re.search(r"pub struct vbt_call_stats \{(.*?)\}", sys_rs, re.S).group(1)
    assert c_fields == re.findall(r"pub (\w+):", rs_body)
    assert c_fields == [f for f, _ in N.CallStats._fields_]


def test_lexicon_golden(unit_golden, fixture_sources):
    for c in unit_golden["lexicon_common_prefix"]:
        if c.get("dict") == "fixture":
            d = _fixture_dict(fixture_sources)
        else:
            d = V.SystemDictionaryBuilder.from_readers(c["lex"], "12 12\n", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")
        assert d.common_prefix(c["input"]) == c["expect"], c["source"]
    d = _fixture_dict(fixture_sources)
    assert d.num_words(0) == 46
    for c in unit_golden["word_feature"]:
        assert d.word_feature(0, c["word_id"]) == c["feature"]


def test_connector_golden(unit_golden, fixture_sources):
    for c in unit_golden["connector"]:
        m = fixture_sources["matrix.def"] if c["matrix"] == "fixture" else c["matrix"]
        d = V.SystemDictionaryBuilder.from_readers("a,0,0,0,x", m, "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")
        assert (d.num_left, d.num_right) == (c["num_left"], c["num_right"])
        for r, l, cost in c["costs"]:
            assert d.conn_cost(r, l) == cost
    for c in unit_golden["connector_errors"]:
        with pytest.raises(V.VibratoError):
            V.SystemDictionaryBuilder.from_readers("a,0,0,0,x", c["matrix"], "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")


def test_char_def_golden(unit_golden):
    for c in unit_golden["char_info"]:
        d = V.SystemDictionaryBuilder.from_readers("a,0,0,0,x", "1 1\n0 0 0", c["char_def"], "DEFAULT,0,0,0,*")
        ci = d.char_info(c["cp"])
        for k in ["cate_idset", "base_id", "invoke", "group", "length"]:
            assert ci[k] == c[k]
    for c in unit_golden["char_def_errors"]:
        with pytest.raises(V.VibratoError):
            V.SystemDictionaryBuilder.from_readers("a,0,0,0,x", "1 1\n0 0 0", c["char_def"], "DEFAULT,0,0,0,*")
    for c in unit_golden["char_def_ok"]:
        V.SystemDictionaryBuilder.from_readers("a,0,0,0,x", "1 1\n0 0 0", c["char_def"], "DEFAULT,0,0,0,*")


def test_error_behaviour(fixture_sources):
    s = fixture_sources
    with pytest.raises(V.VibratoError) as e:  # builder.rs:24-29
        V.SystemDictionaryBuilder.from_readers("a,10,0,0,x", s["matrix.def"], s["char.def"], s["unk.def"])
    assert e.value.code == 1
    with pytest.raises(V.VibratoError) as e:  # lexicon.rs:170-176
        V.SystemDictionaryBuilder.from_readers("a,0,0,0", s["matrix.def"], s["char.def"], s["unk.def"])
    assert e.value.code == 2
    with pytest.raises(V.VibratoError) as e:
        V.SystemDictionaryBuilder.from_readers("a,x,0,0,f", s["matrix.def"], s["char.def"], s["unk.def"])
    assert e.value.code == 4
    with pytest.raises(V.VibratoError):  # unknown.rs:240-244
        V.SystemDictionaryBuilder.from_readers(s["lex.csv"], s["matrix.def"], s["char.def"], "NOPE,0,0,0,*")
    d = _fixture_dict(s)
    with pytest.raises(V.VibratoError):  # dictionary.rs:218-223
        d.reset_user_lexicon_from_reader("a,0,10,0,x")
    d.reset_user_lexicon_from_reader(s["user.csv"])
    assert d.num_words(1) == 3
    d.reset_user_lexicon_from_reader(None)
    assert d.num_words(1) == 0
    d2 = V.SystemDictionaryBuilder.from_readers("a,0,0,0,x", "1 1\n0 0 0", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")
    with pytest.raises(V.VibratoError):  # tokenizer.rs:44-49
        V.Tokenizer(d2).ignore_space(True)


def test_csv_dialect():
    """csv_core defaults used by Lexicon::parse_csv (lexicon.rs:111-200)."""
    lex = '"a,b",1,2,3,"x,y",z\r\n\r\n,0,0,0,skipped\n"q""r",0,0,-5,f1\nlast,0,0,7,tail'
    for mk in (lambda: V.SystemDictionaryBuilder.from_readers(lex, "3 3\n", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*"),
               lambda: ora.Dictionary.from_sources(lex, "3 3\n", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")):
        d = mk()
        assert d.num_words(0) == 3
        assert d.word_feature(0, 0) == '"x,y",z'
        assert d.word_param(0, 0) == (1, 2, 3)
        assert d.word_feature(0, 1) == "f1"
        assert d.word_param(0, 1) == (0, 0, -5)
        assert d.word_feature(0, 2) == "tail"
        assert [m[:2] for m in d.common_prefix("a,b")] == [[0, 3]]
        assert [m[:2] for m in d.common_prefix('q"r')] == [[1, 3]]


def test_lone_carriage_return_at_the_end_of_a_definition_file_is_not_stripped():
    """BufRead::lines() strips '\\r' only as part of "\\r\\n": an unterminated last line that ends in '\\r' keeps it, and the
    reference's integer parse then fails (matrix_connector.rs:27-77) -- product and oracle alike; "\\r\\n" everywhere parses."""
    lex, ch, unk = "a,0,0,0,x", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*"
    for mk, err in ((lambda m: V.SystemDictionaryBuilder.from_readers(lex, m, ch, unk), V.VibratoError),
                    (lambda m: ora.Dictionary.from_sources(lex, m, ch, unk), Exception)):
        mk("1 1\r\n0 0 5\r\n")
        mk("1 1\n0 0 5")
        with pytest.raises(err):
            mk("1 1\n0 0 5\r")


def test_host_trie_matches_oracle_on_synthetic_lexicon():
    sd = synth.SynthDict("small")
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    assert dv.num_words(0) == do.num_words(0) == sd.n_words
    assert dv.num_words(2) == do.num_words(2) == 40
    text, offs = sd.sentences(300)
    rng = random.Random(7)
    n_match = 0
    for s in range(300):
        sent = bytes(text[offs[s]:offs[s + 1]]).decode("utf-8")
        for _ in range(4):
            i = rng.randrange(len(sent))
            a, b = dv.common_prefix(sent[i:i + 16]), do.common_prefix(sent[i:i + 16])
            assert a == b
            n_match += len(a)
    assert n_match > 1000
    for wid in rng.sample(range(sd.n_words), 200):
        assert dv.word_feature(0, wid) == do.word_feature(0, wid)
        assert dv.word_param(0, wid) == do.word_param(0, wid)
    for u in range(40):
        assert dv.word_feature(2, u) == do.word_feature(2, u)
        assert dv.word_param(2, u) == do.word_param(2, u)
    for cp in [0x20, 0x41, 0x3042, 0x30A2, 0x4E00, 0x4E8C, 0x9FA5, 0x3002, 0x1F600]:
        assert dv.char_info(cp) == do.char_info(cp)
    m = sd.matrix
    for _ in range(100):
        r, l = rng.randrange(sd.num_right), rng.randrange(sd.num_left)
        assert dv.conn_cost(r, l) == do.conn_cost(r, l) == int(m[l, r])


def test_tokenizer_fails_loudly_without_gpu(fixture_sources):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    d = _fixture_dict(fixture_sources)
    with pytest.raises(V.VibratoError) as e:
        V.Tokenizer(d).new_worker()
    assert e.value.code == 100  # VBT_ERR_DEVICE: no silent CPU fallback
    assert d.num_words(0) == 46  # the dictionary handle survives a failed Tokenizer::new


def test_vbt_free_ignores_what_is_not_a_live_buffer_of_the_library(fixture_sources):
    """vbt_free releases only pointers the library handed out and that are still live (round-5 advisor: a 16-byte header in front of
    the pointer used to be trusted -- a second free put one block into the output cache twice, a foreign pointer had the bytes in front
    of it read and was then freed): a second vbt_free of the same buffer and a pointer from somewhere else are ignored, a big buffer
    goes through the cache and comes back for the next request, and the cache is released with the tokenizer's pools."""
    import ctypes as C
    from vibrato_amd import _native as N
    s = fixture_sources
    d = V.SystemDictionaryBuilder.from_readers(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    L = N.lib()
    out, n = C.c_void_p(), C.c_size_t()
    N.check(L.vbt_dict_write(d._handle(), -1, C.byref(out), C.byref(n)))
    assert n.value > 1000 and C.string_at(out.value, 21) == b"VibratoTokenizer 0.5\n"
    L.vbt_free.argtypes = [C.c_void_p]
    L.vbt_free(out)
    L.vbt_free(out)            # again: ignored (it is not live any more)
    foreign = C.create_string_buffer(b"x" * 64)
    L.vbt_free(C.cast(C.byref(foreign, 32), C.c_void_p))  # never ours: ignored, nothing in front of it is read or freed
    assert foreign.raw[:64] == b"x" * 64
    L.vbt_free(None)
    # the library still hands out good buffers afterwards
    a, b = C.c_void_p(), C.c_void_p()
    N.check(L.vbt_dict_write(d._handle(), -1, C.byref(a), C.byref(n)))
    N.check(L.vbt_dict_write(d._handle(), -1, C.byref(b), C.byref(n)))
    assert a.value != b.value and C.string_at(a.value, n.value) == C.string_at(b.value, n.value)
    L.vbt_free(a)
    L.vbt_free(b)


def test_utf8_validity_matches_python_strict_decoder():
    """vbt_utf8_valid == Rust `str` validity == Python's strict 'utf-8' codec, on hand-picked and random byte strings."""
    from vibrato_amd.api import utf8_valid
    cases = [b"", b"abc", "東京都".encode(), "\U0001F600".encode(), b"\x80", b"\xc0\xaf", b"\xc2", b"\xe3\x81", b"\xed\xa0\x80",
             b"\xed\x9f\xbf", b"\xee\x80\x80", b"\xf4\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf0\x8f\xbf\xbf", b"\xf0\x90\x80\x80",
             b"\xe0\x9f\xbf", b"\xe0\xa0\x80", b"a" * 9 + b"\xff", b"a" * 8 + "あ".encode(), b"\xf8\x88\x80\x80\x80", b"\xc2\x80\x80"]
    rng = random.Random(5)
    pool = [b"a", b"\x7f", b"\x80", b"\xbf", b"\xc2", b"\xdf", b"\xe0", b"\xed", b"\xef", b"\xf0", b"\xf4", b"\xf5", b"\xa0", b"\x90", "あ".encode(), "\U0001F600".encode()]
    for _ in range(3000):
        cases.append(b"".join(rng.choice(pool) for _ in range(rng.randrange(1, 12))))
    for c in cases:
        try:
            c.decode("utf-8")
            ok = True
        except UnicodeDecodeError:
            ok = False
        assert utf8_valid(c) == ok, c


def test_gather_ring_registers_are_untouched_in_the_compiled_isa():
    """The sweep kernel issues its connection-cost gathers as inline assembly and waits for them with a hand-placed s_waitcnt, so the
    compiler does not know that their destination registers are written asynchronously.  tools/check_ring_isa.py compiles lattice.hip
    for gfx950 (hipcc cross-compiles without a GPU) and proves on the ISA, along every path of the control-flow graph, that nothing
    touches such a register before a wait at which the gather has landed -- for every instance of lattice_lds, lattice_lean, lattice_slim and tokenize_serve."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ring_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" 0 violations") == 14, r.stdout  # lattice_lds x 4, lattice_lean x 2, lattice_slim x 2, gen_sweep x 2, tokenize_serve x 4


def test_ring_checker_flags_seeded_violations():
    """The checker itself: on a hand-written kernel body it accepts a ring whose consumer sits behind an exact wait, and flags (a) a
    consumer in front of the wait, (b) a wait that lets one load too many stay in flight, (c) a copy of the destination register on
    one of two paths, (d) a record load (register pair) whose second register is read early."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_ring_isa", os.path.join(ROOT, "tools", "check_ring_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)

    def run(lines):
        body = ["\t;;#ASMSTART"] + ["\t" + ln if not ln.startswith(".L") else ln for ln in lines] + ["\t;;#ASMEND", "\ts_endpgm"]
        return m.check_kernel("k", body)

    good = ["buffer_load_sshort v40, v1, s[4:7], 0 idxen",      # G
            "global_load_dwordx2 v[76:77], v24, s[60:61]",       # one load behind G
            "buffer_load_sshort v41, v2, s[4:7], 0 idxen",       # two
            "s_waitcnt vmcnt(2)",                                # all but the 2 youngest: G has landed
            "v_add_u32 v3, v3, v40",
            "s_waitcnt vmcnt(0)",
            "v_add_u32 v3, v3, v41",
            "v_add_u32 v3, v3, v76",
            "v_add_u32 v3, v3, v77"]
    n, errs = run(good)
    assert n == 4 and errs == [], errs                            # (4 destination registers: v40, v76, v77, v41)
    # (a) consumer in front of the wait
    n, errs = run(good[:3] + ["v_add_u32 v3, v3, v40"] + good[3:])
    assert any("v40" in e for e in errs), errs
    # (b) the wait leaves three loads in flight, G may be one of them
    bad = list(good)
    bad[3] = "s_waitcnt vmcnt(3)"
    n, errs = run(bad)
    assert any("v40" in e for e in errs), errs
    # (c) a branch around the wait on which the register is copied
    n, errs = run(["buffer_load_sshort v40, v1, s[4:7], 0 idxen",
                   "s_cbranch_scc1 .LBBx_1",
                   "s_waitcnt vmcnt(0)",
                   "s_branch .LBBx_2",
                   ".LBBx_1:",
                   "v_mov_b32 v9, v40",
                   ".LBBx_2:",
                   "s_waitcnt vmcnt(0)"])
    assert any("v_mov_b32 v9, v40" in e for e in errs), errs
    # (d) the second register of a record load read before the wait
    n, errs = run(["global_load_dwordx2 v[76:77], v24, s[60:61]",
                   "v_readfirstlane_b32 s45, v77",
                   "s_waitcnt vmcnt(0)"])
    assert any("v77" in e for e in errs) and not any("v76:" in e for e in errs), errs

    # (e) LDS reads (the pass records of the default build, the loop's own reads): landed behind s_waitcnt lgkmcnt(0) only
    n, errs = run(["ds_read_b64 v[76:77], v59 offset:48",
                   "ds_read_b32 v54, v53",
                   "s_waitcnt lgkmcnt(0)",
                   "v_add_u32 v3, v54, v76",
                   "v_add_u32 v3, v3, v77"])
    assert n == 3 and errs == [], errs
    n, errs = run(["ds_read_b64 v[76:77], v59 offset:48",
                   "s_waitcnt vmcnt(0)",                      # the wrong counter
                   "v_readfirstlane_b32 s45, v77",
                   "s_waitcnt lgkmcnt(0)"])
    assert any("v77" in e for e in errs), errs
    n, errs = run(["ds_read_b64 v[76:77], v59 offset:48",   # a path around the wait
                   "s_cbranch_scc1 .LBBy_1",
                   "s_waitcnt lgkmcnt(0)",
                   ".LBBy_1:",
                   "v_mov_b32 v9, v76"])
    assert any("v_mov_b32 v9, v76" in e for e in errs), errs
    # (f) the block that holds the ring must begin by draining the compiler's own loads
    ring = ["buffer_load_sshort v40, v1, s[4:7], 0 idxen", "s_waitcnt vmcnt(0)", "v_add_u32 v3, v3, v40"] + ["s_nop 0"] * 20
    n, errs = run(["s_waitcnt vmcnt(0) lgkmcnt(0)"] + ring)
    assert errs == [], errs
    n, errs = run(["v_mov_b32 v24, 0"] + ring)
    assert any("begins with" in e for e in errs), errs
