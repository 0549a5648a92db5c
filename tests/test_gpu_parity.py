"""Parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the
C ABI, against (a) the reference's golden vectors and (b) the CPU oracle on seeded
synthetic inputs -- bit-exact token ranges, word ids and total costs."""
import os

import numpy as np
import pytest

import vibrato_amd as V
from oracle import oracle as ora
from tools import synth

pytestmark = pytest.mark.gpu


def _build(case, src):
    if case["dict"] == "fixture":
        d = V.SystemDictionaryBuilder.from_readers(src["lex.csv"], src["matrix.def"], src["char.def"], src["unk.def"])
    else:
        c = case["dict"]
        d = V.SystemDictionaryBuilder.from_readers(c["lex"], c["matrix"], c["char"], c["unk"])
    if case["user"]:
        d.reset_user_lexicon_from_reader(src["user.csv"])
    return V.Tokenizer(d).ignore_space(case["ignore_space"]).max_grouping_len(case["max_grouping_len"])


def test_golden_vectors_worker_api(tokenize_golden, fixture_sources):
    """The reference's own tests (vibrato/src/tests/tokenizer.rs etc.) through Worker."""
    for case in tokenize_golden:
        worker = _build(case, fixture_sources).new_worker()
        for sent in case["sentences"]:
            worker.reset_sentence(sent["text"])
            worker.tokenize()
            assert worker.num_tokens() == sent["num_tokens"], (case["name"], sent["text"])
            for exp in sent["tokens"]:
                got = worker.token(exp["index"])
                for k in ["surface", "feature", "total_cost"]:
                    if k in exp:
                        assert getattr(got, k) == exp[k], (case["name"], sent["text"], exp["index"], k)
                for k in ["range_char", "range_byte"]:
                    if k in exp:
                        assert list(getattr(got, k)) == exp[k], (case["name"], sent["text"], exp["index"], k)


def test_golden_vectors_batch_api(tokenize_golden, fixture_sources):
    for case in tokenize_golden:
        tok = _build(case, fixture_sources)
        batch = tok.tokenize_batch([s["text"] for s in case["sentences"]])
        for si, sent in enumerate(case["sentences"]):
            assert batch.num_tokens(si) == sent["num_tokens"]
            for exp in sent["tokens"]:
                got = batch.token(si, exp["index"])
                for k in ["surface", "feature", "total_cost"]:
                    if k in exp:
                        assert getattr(got, k) == exp[k]
                for k in ["range_char", "range_byte"]:
                    if k in exp:
                        assert list(getattr(got, k)) == exp[k], (case["name"], sent["text"], exp["index"], k)


def _oracle_and_product(sd, user_csv=None, ignore_space=False, max_grouping_len=0):
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    if user_csv is not None:
        do.reset_user_lexicon(user_csv)
        dv.reset_user_lexicon_from_reader(user_csv)
    to = ora.Tokenizer(do, ignore_space, max_grouping_len)
    tv = V.Tokenizer(dv).ignore_space(ignore_space).max_grouping_len(max_grouping_len)
    return to, tv


def _assert_batch_equal(to, tv, text, offs):
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    batch = tv.tokenize_batch(text=text, offsets=offs)
    got_tok, got_off = batch.tokens_in_order()
    assert np.array_equal(got_off, exp_off)
    assert len(got_tok) == len(exp_tok)
    for f in V.TOKEN_DTYPE.names:
        bad = np.nonzero(got_tok[f] != exp_tok[f])[0]
        assert bad.size == 0, (f, int(bad[0]), got_tok[bad[0]], exp_tok[bad[0]])
    return batch, len(exp_tok)


@pytest.mark.parametrize("shape,n", [("tiny", 3000), ("small", 10000)])
def test_differential_default_options(shape, n):
    sd = synth.SynthDict(shape)
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(n, "lognormal_40")
    _, ntok = _assert_batch_equal(to, tv, text, offs)
    assert ntok > n


def test_differential_mecab_compat_mode_with_user_lexicon():
    """BASELINE config 5: -S -M 24 + user.csv, mixed lengths, injected spaces."""
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd, user_csv=sd.user_csv(1000), ignore_space=True, max_grouping_len=24)
    text, offs = sd.sentences(4000, "mixed", space_p=0.10)
    batch, _ = _assert_batch_equal(to, tv, text, offs)
    toks, _, _ = batch.arrays()
    assert (toks["word_idx"] >> 30 == 1).sum() > 0  # user-lexicon words on best paths
    assert (toks["word_idx"] >> 30 == 2).sum() > 0  # unknown words too


def test_differential_spaces_without_ignore_space():
    sd = synth.SynthDict("tiny")
    to, tv = _oracle_and_product(sd, max_grouping_len=3)
    text, offs = sd.sentences(2000, "uniform_5_20", space_p=0.3)
    _assert_batch_equal(to, tv, text, offs)


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("tiers", ["2048,8192", "1024", "4096,6144,8192,12288,16384,65536"])
def test_all_tiers_agree(tiers, fused, monkeypatch):
    """Force sentences through the larger LDS tiers and the global-scratch tier
    (and through multi-block connection-cost staging when LDS is tight)."""
    monkeypatch.setenv("VBT_TIERS", tiers)
    monkeypatch.setenv("VBT_FUSED", fused)  # 1 = the single fused kernel (kept as the global-scratch fallback)
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd, ignore_space=True)
    text, offs = sd.sentences(3000, "mixed", space_p=0.05)
    _assert_batch_equal(to, tv, text, offs)


@pytest.mark.parametrize("env", [{"VBT_GEN_LDS": "2048"}, {"VBT_GEN_LDS": "1024", "VBT_GEN_WAVES": "2"}, {"VBT_GEN_LDS": "3072", "VBT_GEN_WAVES": "8"},
                                 {"VBT_GEN_LDS": "2048", "VBT_GEN_LEVELS": "4096,8192,163840", "VBT_GEN_WAVES": "1"},
                                 {"VBT_SEG_BYTES": "0"}, {"VBT_SEG_BYTES": "8192"}, {"VBT_TIERS": "4096,16384", "VBT_SEG_BYTES": "4096"},
                                 {"VBT_TIERS": "3072", "VBT_SEG_BYTES": "2048"},
                                 {"VBT_TIERS": "1536,163840", "VBT_SEG_BYTES": "1536"}, {"VBT_TIERS": "2048", "VBT_SEG_BYTES": "2048", "VBT_GEN_LDS": "1024", "VBT_GEN_LEVELS": "4096,8192,163840"},
                                 {"VBT_PACK_SCAN": "1"}, {"VBT_FB_WGS": "3", "VBT_TIERS": "1024"},
                                 {"VBT_LEAN": "0"}, {"VBT_TIERS": "3072,5120,10240,163840"}, {"VBT_TIERS": "6144,8192", "VBT_SEG_BYTES": "8192", "VBT_LEAN": "0"},
                                 {"VBT_GEN_SWEEP": "1"}, {"VBT_GEN_SWEEP": "1", "VBT_TIERS": "3072,10240,163840"}])
def test_generator_scheduling_variants_agree(env, monkeypatch):
    """A tiny bulk-generator LDS (most sentences then go through gen_long, the multi-wavefront generator, with 1 / 2 / 4 / 8
    wavefronts per workgroup and through its small levels), the segmented sweep of sentences that do not fit the segment tier
    (cut anywhere, the window of open end lists handed over; down to segments of 8 positions), the tile-prefix kernel in front
    of the packing, a fallback launch of three waves, and the lean instance of the sweep (lattice_lean: the tiers in front of the segment
    tier, whole sentences with the generator's pass records) switched off or spread over two small tiers, and the generator's wave sweeping
    its own sentence (VBT_GEN_SWEEP=1, a measured negative result kept as a knob) must not change a single token."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd, ignore_space=True)
    text, offs = sd.sentences(4000, "mixed", space_p=0.05)
    _assert_batch_equal(to, tv, text, offs)


@pytest.mark.parametrize("seed", range(6))
def test_tie_heavy_random_dictionaries_match_oracle(seed):
    """Equal-cost paths as the norm: 2-5 connection ids, word and connection costs from six values, duplicate surfaces, user entries
    shadowing system entries, every invoke / group / length combination in char.def, spaces and characters outside every range (the
    dictionaries of tests/test_oracle_vs_python_restatement.py, where the oracle is checked against an independent restatement).  The
    synthetic BASELINE dictionaries almost never tie; here the `<=` of lattice.rs:141-146 -- the last inserted predecessor wins -- decides
    most nodes, in whole sentences and across the cuts of segmented ones."""
    import random
    from tests.test_oracle_vs_python_restatement import make_dictionary, HIRA, KATA, ALPHA, NUM, OTHER
    rng = random.Random(977 + seed)
    d = make_dictionary(rng)
    alphabet = HIRA * 3 + KATA * 2 + ALPHA * 2 + NUM + "   " + OTHER
    sents = ["".join(rng.choice(alphabet) for _ in range(rng.choice([0, 1, 2, 5, 9, 14, 23, 40, 80, 200, 500, 1200]))) for _ in range(600)]
    raw = [x.encode("utf-8") for x in sents]
    text = np.frombuffer(b"".join(raw), dtype=np.uint8)
    offs = np.zeros(len(raw) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in raw])
    for ignore_space, mgl in ((False, 0), (True, 1), (True, 3)):
        do = ora.Dictionary.from_sources(d["lex"], d["matrix_def"], d["char_def"], d["unk"])
        dv = V.SystemDictionaryBuilder.from_readers(d["lex"], d["matrix_def"], d["char_def"], d["unk"])
        if d["user"] is not None:
            do.reset_user_lexicon(d["user"])
            dv.reset_user_lexicon_from_reader(d["user"])
        to = ora.Tokenizer(do, ignore_space, mgl)
        tv = V.Tokenizer(dv).ignore_space(ignore_space).max_grouping_len(mgl)
        _assert_batch_equal(to, tv, text, offs)


@pytest.mark.parametrize("shape", ["ipadic", "unidic"])
def test_full_size_batch_bit_exact(shape):
    """BASELINE configs 2 and 3 at full size: 100k sentences over the ipadic- / unidic-shaped
    synthetic dictionary (458.6 MiB matrix), every token record compared with the oracle."""
    sd = synth.SynthDict(shape)
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(100000, "lognormal_40")
    batch, ntok = _assert_batch_equal(to, tv, text, offs)
    assert ntok > 2_000_000
    # size-independent property: tokens tile every sentence (no spaces are skipped in this mode)
    toks, off, cnt = batch.arrays()
    first = off[cnt > 0]
    last = (off + cnt - 1)[cnt > 0]
    lens = np.diff(offs).astype(np.uint32)
    assert np.all(toks["start_byte"][first] == 0)
    assert np.array_equal(toks["end_byte"][last], lens[cnt > 0])


def test_config4_sized_batch_on_one_gpu_equals_its_shards():
    """BASELINE config 4's corpus (1 M sentences, 140 MB of text) as ONE batch on one GPU (a ~56 GB workspace): the records equal
    those of the same corpus tokenized as 8 contiguous shards (what 8 ranks would each do), the first 20 000 sentences equal the
    oracle's, and tokens tile every sentence (no spaces are skipped in this mode)."""
    from vibrato_amd import sharding
    sd = synth.SynthDict("ipadic")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(1000000, "lognormal_40")
    _config4_checks(sd, to, tv, text, offs)


def test_config4_on_unidic_through_the_multi_device_tokenizer():
    """BASELINE config 4 as it is named -- unidic, 1 M sentences, sharded -- behind the C ABI: one tokenizer over two replicas of the
    458.6 MiB image (vbt_tokenizer_new_multi with device list {0, 0}: what a one-GPU box admits), every batch cut into shards whose
    results land in the caller's one pinned block.  Same checks as on ipadic: the 1 M-sentence batch equals its 8 contiguous shards,
    the first 20 000 sentences equal the oracle's, tokens tile every sentence."""
    sd = synth.SynthDict("unidic")
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    to, tv = ora.Tokenizer(do), V.Tokenizer(dv, devices=[0, 0])
    text, offs = sd.sentences(1000000, "lognormal_40")
    _config4_checks(sd, to, tv, text, offs)
    assert tv.num_devices() == 2 and tv.wait_connid_reorder(120) and tv.connid_reorder_info()["epoch"] == 1


def _config4_checks(sd, to, tv, text, offs):
    from vibrato_amd import sharding
    whole = tv.tokenize_batch(text=text, offsets=offs)
    toks, off, cnt = whole.arrays()
    assert len(cnt) == 1000000 and int(cnt.sum()) == len(toks) > 20_000_000
    ordered, ends = sharding.tokens_in_sentence_order(off, cnt, toks)
    bnd = sharding.shard_bounds(offs, 8)
    for r in range(8):
        ltext, loffs, (lo, hi) = sharding.local_shard(text, offs, r, 8)
        part, _ = tv.tokenize_batch(text=np.ascontiguousarray(ltext), offsets=loffs).tokens_in_order()
        assert part.tobytes() == ordered[int(ends[lo]):int(ends[hi])].tobytes(), r
        del part
    exp, exp_off = to.new_worker().tokenize_batch(text[:int(offs[20000])], offs[:20001])
    assert ordered[:len(exp)].tobytes() == exp.tobytes()
    first, last = ends[:-1][cnt > 0], ends[1:][cnt > 0] - 1
    assert np.all(ordered["start_byte"][first] == 0)
    assert np.array_equal(ordered["end_byte"][last], np.diff(offs).astype(np.uint32)[cnt > 0])
    tv.trim_pool()


def test_edge_cases(fixture_sources):
    s = fixture_sources
    d = V.SystemDictionaryBuilder.from_readers(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    tok = V.Tokenizer(d).ignore_space(True)
    do = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    to = ora.Tokenizer(do, True, 0)
    sents = ["", " ", "   ", "東京都", "", "a", "\U0001F600東京\U0001F600", "東京 都 ", "0" * 300, "x" * 5000, "京都" * 700, ""]
    enc = [x.encode() for x in sents]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    text = np.frombuffer(b"".join(enc), dtype=np.uint8)
    _assert_batch_equal(to, tok, text, offs)
    empty = tok.tokenize_batch([])
    assert len(empty) == 0 and empty.total_tokens() == 0
    assert empty.format("mecab") == ""


@pytest.mark.parametrize("ignore_space", [False, True])
def test_group_spans_at_length_mask_boundaries(fixture_sources, ignore_space):
    """Grouped unknown words of exactly 32 / 33 / 63 / 64 / 65 characters (bits 31 and 63 of the per-position
    length mask, and the > 64 overflow path) whose interior positions are unreachable, around skipped spaces."""
    s = fixture_sources
    d = V.SystemDictionaryBuilder.from_readers(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    tok = V.Tokenizer(d).ignore_space(ignore_space)
    do = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    to = ora.Tokenizer(do, ignore_space, 0)
    sents = []
    for k in (31, 32, 33, 63, 64, 65, 127, 128):
        sents += ["a" * k + "1" * 25 + "a" * k + "京都", "東京" + "a" * k, "a" * k + " " + "1" * k + "  " + "a" * k,
                  " " * k + "a" * k + " " * k, "京都 " + " " * (k - 1) + "東京都" + "a" * k + " "]
    enc = [x.encode() for x in sents]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    text = np.frombuffer(b"".join(enc), dtype=np.uint8)
    _assert_batch_equal(to, tok, text, offs)


def test_very_long_sentences_take_every_generator_level_and_the_fallback():
    """~600 / ~2 400 / ~3 300 / ~17 000-character sentences (sentence-aligned prefixes of a mixed batch, spaces kept):
    the 32 KiB, 64 KiB and whole-CU generator levels, the segmented sweep with a windowed back-trace and -- 17 000
    characters fit no generator level -- the global-memory kernel."""
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd, ignore_space=True)
    base, offs0 = sd.sentences(600, "mixed", space_p=0.05)
    enc = [bytes(base[:int(offs0[k])]) for k in (14, 20, 40, 160)]
    enc = [enc[0], b"", enc[1], enc[3], "京都".encode() * 3, enc[2]]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    text = np.frombuffer(b"".join(enc), dtype=np.uint8)
    assert max(len(e) for e in enc) > 50000
    batch, _ = _assert_batch_equal(to, tv, text, offs)


@pytest.mark.parametrize("ignore_space", [False, True])
def test_sentences_of_8000_to_11000_characters_take_the_exact_instance_of_the_sweep(ignore_space):
    """Sentences of ~8 200, ~9 000 and ~10 500 characters: the whole-CU level of gen_long generates them and lattice_lds sweeps them in
    its segment tier, but with >= 8 000 characters the dead-predecessor sentinel of the common build no longer bounds a live cost, so the
    `exact` instance of the C++ loop runs (every predecessor's own field is tested) -- on i16 cells, which no other test reaches (round-4
    review).  None of them may fall through to the global-memory kernel.  (What this test found when it was written: every sentence of
    more than 32 767 lattice nodes -- 6 500+ characters here -- lost the tail of its path: the end-list offset of the EOS step was
    shifted as a signed int.  The oracle counts 38 000 - 50 000 nodes for these.)"""
    import torch
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd, ignore_space=ignore_space)
    base, offs0 = sd.sentences(1200, "lognormal_40", space_p=0.05 if ignore_space else 0.0)
    raw = bytes(base)
    chars = np.cumsum([len(raw[int(offs0[i]):int(offs0[i + 1])].decode("utf-8")) for i in range(1200)])
    cuts = [int(np.searchsorted(chars, c)) for c in (8200, 9000, 10500)]
    assert all(8000 <= chars[k] < 11000 for k in cuts)
    enc = [raw[:int(offs0[k + 1])] for k in cuts] + [raw[int(offs0[5]):int(offs0[6])]]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    text = np.frombuffer(b"".join(enc), dtype=np.uint8)
    _assert_batch_equal(to, tv, text, offs)
    ws = tv.workspace(len(enc), len(text))
    d_text = torch.from_numpy(text.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), len(enc), len(text), torch.cuda.current_stream().cuda_stream)
    st = ws.stats()
    # (the long ones are routed to the segment tier and escalate from there to the escape tier for their back-trace window: counted in both)
    assert st["error_flags"] == 0 and st["n_tier2"] == 0 and st["n_tier0"] == len(enc) and st["n_tier1"] >= 3, st


def test_the_cpp_sweep_loop_on_common_shapes():
    """lattice_lds states its recurrence twice: the assembly loop (the default build's common shapes) and a C++ loop that the default
    build only runs on rare shapes (i32 cells, >= 8000 characters, connection-id counting).  The `cpploop` library variant
    (-DVBT_ASM_LOOP=0, built by build()) runs the C++ loop on EVERYTHING: the differential, tie-heavy, segmented and config-5-shaped
    tests once more against it, in a process of its own (the library is chosen when it is loaded)."""
    import subprocess
    import sys
    sel = "differential or tie_heavy or dense_stretches or dense_lattice_law or group_spans or edge_cases or very_long or golden_vectors_batch"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", sel, "-p", "no:cacheprovider"],
                       env=dict(os.environ, VBT_LIB_VARIANT="cpploop"), capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-1500:]
    assert "libvibrato_hip_cpploop.so" in subprocess.run(
        [sys.executable, "-c", "from vibrato_amd import _native; print(_native.lib()._name)"], env=dict(os.environ, VBT_LIB_VARIANT="cpploop"),
        capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout


def test_cli_output_formats(fixture_sources):
    """Byte-identical tokenize CLI output (tokenize/src/main.rs:83-127) vs the oracle's formatter."""
    s = fixture_sources
    d = V.SystemDictionaryBuilder.from_readers(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    d.reset_user_lexicon_from_reader(s["user.csv"])
    tok = V.Tokenizer(d)
    do = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"]).reset_user_lexicon(s["user.csv"])
    w = ora.Tokenizer(do).new_worker()
    sents = ["京都東京都京都", "kampersanda", "", "東京県に行く", "一橋大学大学院", "東京 都"]
    batch = tok.tokenize_batch(sents)
    for mode in ["mecab", "wakati", "detail"]:
        exp = ""
        for x in sents:
            w.reset_sentence(x)
            w.tokenize()
            exp += ora.format_tokens(w, mode)
        assert batch.format(mode) == exp
    assert batch.format("mecab").startswith("京都東京都\tカスタム名詞\n京都\t京都,名詞,固有名詞,地名,一般,*,*,キョウト,京都,*,A,*,*,*,1/5\nEOS\n")


def test_tokenize_cli_matches_reference_cli_format(tmp_path):
    """python -m vibrato_amd.cli mirrors tokenize/src/main.rs (-i -u -O -S -M), incl. \\r\\n and empty lines."""
    import io
    import shutil
    from conftest import GOLDEN
    from vibrato_amd import cli
    res = os.path.join(GOLDEN, "resources")
    lines = ["京都東京都京都", "kampersanda", "", "東京 都  ", "一橋大学大学院"]
    stdin = ("\n".join(lines[:2]) + "\r\n" + "\n".join(lines[2:]) + "\n").encode()
    src = {k: open(os.path.join(res, k), "rb").read() for k in ["lex.csv", "matrix.def", "char.def", "unk.def", "user.csv"]}
    for mode, S, M in [("mecab", False, None), ("wakati", True, 24), ("detail", True, 9)]:
        do = ora.Dictionary.from_sources(src["lex.csv"], src["matrix.def"], src["char.def"], src["unk.def"]).reset_user_lexicon(src["user.csv"])
        w = ora.Tokenizer(do, S, M or 0).new_worker()
        exp = ""
        for x in lines:
            w.reset_sentence(x)
            w.tokenize()
            exp += ora.format_tokens(w, mode)
        out = io.BytesIO()
        argv = ["-i", res, "-u", os.path.join(res, "user.csv"), "-O", mode, "--block", "2"] + (["-S"] if S else []) + (["-M", str(M)] if M else [])
        assert cli.main(argv, stdin=io.BytesIO(stdin), stdout=out) == 0
        assert out.getvalue().decode() == exp, mode


def test_workspace_device_api_and_roundtrip():
    """Device-resident API: tokens reproduce the input exactly when ignore_space is off
    (concatenated surfaces == sentence), a size-independent property usable at full size."""
    import torch
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    n = 20000
    text, offs = sd.sentences(n, "lognormal_40")
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tv.workspace(n, len(text))
    ws.set_timing(True)
    for _ in range(2):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = ws.stats()
    assert st["error_flags"] == 0 and st["n_sentences"] == n
    assert st["n_tier0"] + st["n_tier1"] + st["n_tier2"] >= n  # re-routed sentences are counted in both lists
    assert st["ms_tier0"] > 0
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    assert st["n_tokens"] == len(exp_tok)
    # read results straight from device memory through torch (plumbing only)
    import ctypes
    ptrs = ws.result_ptrs()
    host = np.empty(st["n_tokens"], dtype=V.TOKEN_DTYPE)
    cnt = np.empty(n, dtype=np.uint32)
    off = np.empty(n, dtype=np.uint32)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(host.ctypes.data, ptrs["tokens"], host.nbytes, 2) == 0
    assert hip.hipMemcpy(cnt.ctypes.data, ptrs["tok_cnt"], cnt.nbytes, 2) == 0
    assert hip.hipMemcpy(off.ctypes.data, ptrs["tok_off"], off.nbytes, 2) == 0
    assert np.array_equal(np.diff(exp_off).astype(np.uint32), cnt)
    # every sentence: tokens tile [0, len) contiguously
    order = np.argsort(off, kind="stable")
    for s in order[:2000]:
        t = host[off[s]:off[s] + cnt[s]]
        assert t["start_byte"][0] == 0 and t["end_byte"][-1] == offs[s + 1] - offs[s]
        assert np.array_equal(t["start_byte"][1:], t["end_byte"][:-1])


def test_config5_full_size_bit_exact():
    """BASELINE config 5 at full size: unidic-shaped dictionary (458.6 MiB matrix) + user.csv of 1000 compounds, -S -M 24,
    100k mixed-length sentences (5 % of 500-2000 characters) with injected spaces; every token record vs the oracle."""
    sd = synth.SynthDict("unidic")
    to, tv = _oracle_and_product(sd, user_csv=sd.user_csv(1000), ignore_space=True, max_grouping_len=24)
    text, offs = sd.sentences(100000, "mixed", space_p=0.10)
    batch, ntok = _assert_batch_equal(to, tv, text, offs)
    toks, off, cnt = batch.arrays()
    assert ntok > 4_000_000
    assert (toks["word_idx"] >> 30 == 1).sum() > 0 and (toks["word_idx"] >> 30 == 2).sum() > 0
    # size-independent property: inside a sentence, tokens never overlap and the gaps between them are spaces only
    sample = np.nonzero(cnt > 1)[0][:3000]
    for s in sample:
        t = toks[off[s]:off[s] + cnt[s]]
        assert np.all(t["start_byte"][1:] >= t["end_byte"][:-1])
        base = int(offs[s])
        for a, b in zip(t["end_byte"][:-1], t["start_byte"][1:]):
            assert bytes(text[base + a:base + b]).strip(b" ") == b""


@pytest.mark.parametrize("shape,n", [("small-dense", 8000), ("unidic-dense", 30000)])
def test_dense_lattice_law_bit_exact(shape, n):
    """The denser synthetic law (>= 12 lattice nodes and >= 80 deduplicated connection pairs per character, what
    SURVEY.md 8(a) estimates for the real unidic; the default law has 5.7 / 29): more groups per position, more
    multi-pass steps, more sentences in the segment tier."""
    sd = synth.SynthDict(shape)
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(n, "lognormal_40")
    w = to.new_worker()
    w.reset_counters()
    w.tokenize_batch(text[:int(offs[500])], offs[:501], counted=True, want_tokens=False)
    c = w.counters()
    assert c["n_nodes"] / c["n_chars"] >= 12 and c["n_pairs_dedup"] / c["n_chars"] >= 80
    _assert_batch_equal(to, tv, text, offs)


BAD_UTF8 = [b"\x80abc", b"abc\xe3\x81", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xe0\x80\xaf", b"a\xffb",
            b"\xe3\x81\x82\x81", b"\xf0\x82\x82\xac", b"\xf8\x88\x80\x80\x80"]


def test_invalid_utf8_is_rejected_like_the_reference(fixture_sources):
    """The reference takes &str and its CLI fails on an invalid line; the product must not decode garbage silently:
    host entry points return VBT_ERR_UTF8 for exactly the inputs the oracle rejects, the device entry point flags them."""
    import torch
    s = fixture_sources
    tok = V.Tokenizer(V.SystemDictionaryBuilder.from_readers(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"]))
    do = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    wo = ora.Tokenizer(do).new_worker()
    wv = tok.new_worker()
    good = ["東京都".encode(), b"", b"abc", "\U0001F600".encode(), "京都 ".encode()]
    for sent in BAD_UTF8 + good:
        try:
            wo.reset_sentence(sent)
            oracle_ok = True
        except ora.OracleError:
            oracle_ok = False
        assert oracle_ok == (sent in good)
        assert V.api.utf8_valid(sent) == oracle_ok
        if oracle_ok:
            wv.reset_sentence(sent)
            wv.tokenize()
        else:
            with pytest.raises(V.VibratoError) as e:
                wv.reset_sentence(sent)
            assert e.value.code == 5
    for bad in BAD_UTF8:
        enc = [good[0], bad, good[2]]
        offs = np.zeros(4, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in enc])
        text = np.frombuffer(b"".join(enc), dtype=np.uint8)
        with pytest.raises(V.VibratoError) as e:
            tok.tokenize_batch(text=text, offsets=offs)
        assert e.value.code == 5
        # device API: flagged (16), batch skipped
        ws = tok.workspace(3, len(text))
        d_text = torch.from_numpy(text.copy()).cuda()
        d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), 3, len(text), torch.cuda.current_stream().cuda_stream)
        assert ws.stats()["error_flags"] & 16
    # a sentence boundary inside a character is invalid even though the stream as a whole is valid
    text = np.frombuffer("東京".encode(), dtype=np.uint8)
    offs = np.array([0, 2, 6], dtype=np.uint64)
    with pytest.raises(V.VibratoError):
        tok.tokenize_batch(text=text, offsets=offs)
    ws = tok.workspace(2, 6)
    d_text = torch.from_numpy(text.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), 2, 6, torch.cuda.current_stream().cuda_stream)
    assert ws.stats()["error_flags"] & 16


def test_device_api_accepts_a_window_and_rejects_bad_offsets():
    """offsets[0] != 0 (a window into a larger device text buffer) gives the same records as the rebased batch;
    decreasing offsets or a span beyond total_bytes are flagged (8) and the batch is skipped."""
    import torch
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(3000, "lognormal_40")
    lo, hi = 700, 2900
    nloc, nbytes = hi - lo, int(offs[hi] - offs[lo])
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tv.workspace(nloc, nbytes)
    stream = torch.cuda.current_stream().cuda_stream
    ws.run(d_text.data_ptr(), d_offs.data_ptr() + 8 * lo, nloc, nbytes, stream)  # window: offsets[lo..hi] of the big buffer
    st = ws.stats()
    assert st["error_flags"] == 0
    from vibrato_amd import sharding
    v = sharding.workspace_views(ws, nloc, st["n_tokens"])
    got, _ = sharding.tokens_in_sentence_order(v["tok_off"].cpu().numpy().view(np.uint32), v["tok_cnt"].cpu().numpy().view(np.uint32),
                                               v["tokens"].cpu().numpy().view(V.TOKEN_DTYPE))
    exp, _ = to.new_worker().tokenize_batch(text[int(offs[lo]):int(offs[hi])], offs[lo:hi + 1] - offs[lo])
    assert got.tobytes() == exp.tobytes()
    bad = offs.astype(np.int64).copy()
    bad[10], bad[11] = bad[11], bad[10] + 1
    d_bad = torch.from_numpy(bad).cuda()
    ws2 = tv.workspace(3000, len(text))
    ws2.run(d_text.data_ptr(), d_bad.data_ptr(), 3000, len(text), stream)
    assert ws2.stats()["error_flags"] & 8
    ws2.run(d_text.data_ptr(), d_offs.data_ptr(), 3000, len(text) - 1, stream)  # declared span too small
    assert ws2.stats()["error_flags"] & 8
    ws2.run(d_text.data_ptr(), d_offs.data_ptr(), 3000, len(text), stream)  # and the workspace still works afterwards
    st = ws2.stats()
    assert st["error_flags"] == 0 and st["n_tokens"] == len(to.new_worker().tokenize_batch(text, offs)[0])


def test_pool_trim_and_budget(monkeypatch):
    """Idle workspaces / pinned blocks are released by trim_pool and never pooled beyond VBT_POOL_MAX_MB."""
    sd = synth.SynthDict("tiny")
    _, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(500, "lognormal_40")
    for _ in range(3):
        tv.tokenize_batch(text=text, offsets=offs)
    created, reused, idle = tv.pool_stats()
    assert created == 1 and reused == 2 and idle == 1
    tv.trim_pool()
    assert tv.pool_stats()[2] == 0
    tv.tokenize_batch(text=text, offsets=offs)
    assert tv.pool_stats()[0] == 2  # re-created on demand


def test_host_batches_reuse_pooled_workspaces():
    sd = synth.SynthDict("tiny")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(500, "lognormal_40")
    for _ in range(4):
        _assert_batch_equal(to, tv, text, offs)
    created, reused, idle = tv.pool_stats()
    assert (created, reused, idle) == (1, 3, 1)
    text2, offs2 = sd.sentences(5000, "lognormal_40")  # larger: a second size class
    _assert_batch_equal(to, tv, text2, offs2)
    _assert_batch_equal(to, tv, text, offs)
    created, reused, idle = tv.pool_stats()
    assert created == 2 and reused == 4 and idle == 2


def test_concurrent_host_batches_from_threads_match_oracle():
    """vbt_tokenize_batch is thread-safe per tokenizer (SURVEY.md 8(b) "threading"): blocks of one corpus pushed by several
    host threads at once (own pooled workspace, pinned staging and stream each) give the oracle's tokens, and the pools stop
    growing; the pipeline helper bench.py reports host-to-host throughput with returns the same token total."""
    from concurrent.futures import ThreadPoolExecutor
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(6000, "lognormal_40")
    exp, exp_off = to.new_worker().tokenize_batch(text, offs)
    bounds = [0, 700, 701, 2500, 2500, 4100, 6000]

    def one(i):
        lo, hi = bounds[i], bounds[i + 1]
        got, got_off = tv.tokenize_batch(text=text, offsets=offs[lo:hi + 1]).tokens_in_order()
        return lo, hi, got, got_off

    for _ in range(3):
        with ThreadPoolExecutor(4) as ex:
            for lo, hi, got, got_off in ex.map(one, range(len(bounds) - 1)):
                assert np.array_equal(got_off, exp_off[lo:hi + 1] - exp_off[lo])
                assert got.tobytes() == exp[int(exp_off[lo]):int(exp_off[hi])].tobytes()
    created = tv.pool_stats()[0]
    r = tv.host_pipeline_benchmark(text, offs, threads=3, rounds=2, repeats=2)
    assert r["tokens_per_batch"] == len(exp) and r["sentences_per_s"] > 0
    assert tv.pool_stats()[0] <= created + 3


@pytest.mark.parametrize("env", [{"VBT_TIERS": "2048,3072,163840", "VBT_SEG_BYTES": "3072"}, {"VBT_TIERS": "1024,163840", "VBT_SEG_BYTES": "1024"},
                                 {"VBT_TIERS": "2048,4096,32768,163840", "VBT_SEG_BYTES": "4096", "VBT_GEN_LDS": "2048", "VBT_GEN_LEVELS": "4096,8192,163840"}])
def test_dense_stretches_are_cut_anywhere_and_handed_over(env, monkeypatch):
    """lattice_lds cuts a sentence that does not fit the segment tier at ANY position and hands the window of end lists behind
    the cut to the next segment (no clean cut needed, nothing pre-routed): with a tiny segment tier on the dense lexicon law
    nearly every sentence is swept in many short segments whose windows hold nodes of several segments; what even a segment of
    8 positions cannot hold escalates to the escape tier at run time (none may be lost, none swept twice); the generator levels
    behind the bulk one keep their length masks / candidate offsets in global memory."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sd = synth.SynthDict("small-dense")
    to, tv = _oracle_and_product(sd, ignore_space=True, max_grouping_len=24)
    text, offs = sd.sentences(3000, "mixed", space_p=0.05)
    _assert_batch_equal(to, tv, text, offs)
    ws = tv.workspace(3000, len(text))
    d_text = torch.from_numpy(text.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), 3000, len(text), torch.cuda.current_stream().cuda_stream)
    st = ws.stats()
    assert st["error_flags"] == 0 and st["n_tier0"] + st["n_tier1"] + st["n_tier2"] >= 3000
    assert st["n_tier2"] < 30  # (the fused global-memory fallback stays the exception)


def _worker_records(worker, text, offs):
    """Token records of every sentence through the reference's 3-call loop (reset_sentence / tokenize / token(i))."""
    raw = bytes(text)
    out = []
    for i in range(len(offs) - 1):
        worker.reset_sentence(raw[int(offs[i]):int(offs[i + 1])])
        worker.tokenize()
        for k in range(worker.num_tokens()):
            t = worker.token(k)
            out.append((t.range_char[0], t.range_char[1], t.range_byte[0], t.range_byte[1], (t.lex_type << 30) | t.word_id, t.total_cost))
    return np.array(out, dtype=np.int64).reshape(-1, 6)


@pytest.mark.parametrize("ignore_space", [False, True])
def test_worker_single_launch_path_matches_oracle(ignore_space):
    """Worker::tokenize is served by a resident kernel (Workspace::serve): mixed lengths (short ones whole, 150+ characters in
    segments, 1000+ characters beyond the single wavefront's generator arrays or not), spaces, user lexicon -- every record equal to
    the oracle's, and the sentences the single launch cannot take come back through the batch pipeline with the same records."""
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd, user_csv=sd.user_csv(300), ignore_space=ignore_space, max_grouping_len=24 if ignore_space else 0)
    text, offs = sd.sentences(1500, "mixed", space_p=0.08 if ignore_space else 0.0)
    exp_tok, _ = to.new_worker().tokenize_batch(text, offs)
    w = tv.new_worker()
    got = _worker_records(w, text, offs)
    exp = np.stack([exp_tok[f].astype(np.int64) for f in V.TOKEN_DTYPE.names], axis=1)
    assert got.shape == exp.shape
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, (int(bad[0]), got[bad[0]], exp[bad[0]])
    fast, slow = w.path_stats()
    assert fast + slow == sum(1 for i in range(1500) if offs[i + 1] > offs[i])
    assert fast > 0.9 * (fast + slow), (fast, slow)
    # empty sentence, then a sentence longer than anything before (the worker's buffers grow), then a short one again
    w.reset_sentence("")
    w.tokenize()
    assert w.num_tokens() == 0
    big = bytes(text[int(offs[0]):int(offs[400])])
    for s in (big, bytes(text[int(offs[3]):int(offs[4])])):
        e, _ = to.new_worker().tokenize_batch(np.frombuffer(s, dtype=np.uint8), np.array([0, len(s)], dtype=np.uint64))
        g = _worker_records(w, np.frombuffer(s, dtype=np.uint8), np.array([0, len(s)], dtype=np.uint64))
        assert np.array_equal(g, np.stack([e[f].astype(np.int64) for f in V.TOKEN_DTYPE.names], axis=1))


def test_worker_loop_benchmark_counts_tokens():
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(500, "lognormal_40")
    exp_tok, _ = to.new_worker().tokenize_batch(text, offs)
    # the resident kernel (default), one launch per call (the round-3 form), and a kernel that leaves after every few polls, so that
    # calls keep finding it gone or on its way out (the doorbell / "has left" handshake)
    for polls in (None, "0", "3"):
        if polls is not None:
            os.environ["VBT_WORKER_IDLE_POLLS"] = polls
        try:
            r = tv.new_worker().loop_benchmark(text, offs, rounds=2)
        finally:
            os.environ.pop("VBT_WORKER_IDLE_POLLS", None)
        assert r["tokens"] == 2 * len(exp_tok) and r["us_per_call"] > 0


def test_tier_set_follows_the_measured_lattice_density(monkeypatch):
    """The sweep's default tier set is chosen per batch from the lattice density the tokenizer measured on the batches before it
    (vbt_tokenizer_lattice_density: candidates per input byte, reported by build_lists into pinned memory): running text gets 8 KiB
    segments behind a 7.5 KiB (short sentences: 6 KiB) lean tier, dense lattices the 8 / 10 KiB set.  The density is what the oracle
    counts for the same sentences, and no choice changes a record: dense and ordinary dictionaries, first batch (nothing measured yet)
    and later ones, adaptation on and off."""
    import torch
    for shape, law in (("small", "lognormal_40"), ("small-dense", "lognormal_40"), ("small", "uniform_5_20")):
        sd = synth.SynthDict(shape)
        do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        to = ora.Tokenizer(do)
        text, offs = sd.sentences(3000, law)
        w = to.new_worker()
        w.reset_counters()
        w.tokenize_batch(text, offs, counted=True, want_tokens=False)
        cnt = w.counters()
        want = cnt["n_nodes"] / len(text)  # the oracle's lattice nodes: every candidate of a visited start position
        for adapt in ("1", "0"):
            monkeypatch.setenv("VBT_TIER_ADAPT", adapt)
            dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
            tv = V.Tokenizer(dv)
            assert tv.lattice_density() == 0.0
            for _ in range(3):  # the first batch runs before anything is known, the next ones behind its report
                _assert_batch_equal(to, tv, text, offs)
                torch.cuda.synchronize()
            d = tv.lattice_density()
            if adapt == "0":
                assert d == 0.0
                continue
            # (the generator lists the candidates of every position, the oracle counts the nodes of the visited ones: the device's figure
            # is the larger one, by the positions no path reaches)
            assert (d > 3.0) == (shape == "small-dense"), (shape, law, d, want)
            assert want * 0.95 <= d <= want * 1.6, (shape, law, d, want)


def test_tokenize_lines_batches_behind_an_iterator():
    """Tokenizer.tokenize_lines: the per-line loop over an iterable, batched behind the scenes (three batches here: the line limit, the
    byte limit, the rest) -- every sentence comes back in input order with the oracle's records."""
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(700, "lognormal_40")
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    buf = text.tobytes()
    lines = [buf[int(offs[i]):int(offs[i + 1])].decode("utf-8") for i in range(len(offs) - 1)]
    seen, batches, last = 0, 0, None
    for s, (b, i) in enumerate(tv.tokenize_lines(iter(lines), batch_bytes=int(offs[300]), batch_lines=250)):
        batches += b is not last  # (not a set of id(b): a freed batch's address is free for the next but one -- seen once in ~15 runs)
        last = b
        e = exp_tok[int(exp_off[s]):int(exp_off[s + 1])]
        assert b.num_tokens(i) == len(e)
        r = b.records(i)
        for f in V.TOKEN_DTYPE.names:
            assert np.array_equal(r[f], e[f]), (s, f)
        seen += 1
    assert seen == len(lines) and batches >= 3


def test_worker_resident_kernel_handshake_under_pauses():
    """A Worker whose resident kernel leaves quickly (VBT_WORKER_IDLE_POLLS=3) with pauses between the calls -- some calls find the
    kernel resident, some find it gone, some catch it leaving -- and a second Worker interleaved with it and with batch calls (which
    allocate and free device memory, i.e. synchronise the device): every record equals the oracle's."""
    import time
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(300, "lognormal_40")
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    exp = np.stack([exp_tok[f].astype(np.int64) for f in V.TOKEN_DTYPE.names], axis=1)
    os.environ["VBT_WORKER_IDLE_POLLS"] = "3"
    try:
        w1 = tv.new_worker()
    finally:
        os.environ.pop("VBT_WORKER_IDLE_POLLS", None)
    w2 = tv.new_worker()
    raw = bytes(text)
    got1, got2 = [], []
    for i in range(300):
        sent = raw[int(offs[i]):int(offs[i + 1])]
        for w, out in ((w1, got1), (w2, got2)):
            w.reset_sentence(sent)
            w.tokenize()
            for k in range(w.num_tokens()):
                t = w.token(k)
                out.append((t.range_char[0], t.range_char[1], t.range_byte[0], t.range_byte[1], (t.lex_type << 30) | t.word_id, t.total_cost))
        if i % 7 == 0:
            time.sleep(0.0005 * (i % 5))
        if i % 50 == 0:
            b = tv.tokenize_batch(text=text[:int(offs[20])], offsets=offs[:21])
            assert b.total_tokens() == int(exp_off[20])
    for got in (got1, got2):
        assert np.array_equal(np.array(got, dtype=np.int64).reshape(-1, 6), exp)
    f1, s1 = w1.path_stats()
    assert f1 == sum(1 for i in range(300) if offs[i + 1] > offs[i]) and s1 == 0


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_multi_device_tokenizer_equals_single_device_and_oracle(devices):
    """vbt_tokenizer_new_multi (one replica of the image per listed device; a one-GPU box lists device 0 more than once): every
    batch is split into contiguous shards balanced by bytes, the shards run side by side and every shard's results land in the
    ONE pinned block at its offset.  Records, offsets and counts must equal the single-device tokenizer's and the oracle's --
    also with fewer sentences than devices, empty sentences at the shard boundaries and an empty batch."""
    sd = synth.SynthDict("small")
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    to = ora.Tokenizer(do, True, 24)
    mk = lambda **kw: V.Tokenizer(V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk),
                                  **kw).ignore_space(True).max_grouping_len(24)
    one, multi = mk(), mk(devices=devices)
    assert multi.num_devices() == len(devices) and one.num_devices() == 1
    text, offs = sd.sentences(5000, "mixed", space_p=0.05)
    _assert_batch_equal(to, multi, text, offs)
    a, b = one.tokenize_batch(text=text, offsets=offs).arrays(), multi.tokenize_batch(text=text, offsets=offs).arrays()
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    # fewer sentences than devices, and empty sentences around the shard boundaries
    t2 = "猫が好き".encode()
    for sents in ([t2], [b"", t2, b"", b""], [b"", b""], []):
        raw = np.frombuffer(b"".join(sents), dtype=np.uint8)
        o = np.zeros(len(sents) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(x) for x in sents])
        _assert_batch_equal(to, multi, raw, o)
    # the Worker of a multi-device tokenizer runs on its first device
    w = multi.new_worker()
    w.reset_sentence(t2.decode())
    w.tokenize()
    assert w.num_tokens() > 0


def test_results_through_the_packing_kernel_when_sdma_is_off(tmp_path):
    """VBT_H2H_OUT=0 (also what the library falls back to when no HSA agent matches the device): tok_tile_scan, then
    compact_tokens_out stores tok_off / tok_cnt / the records straight into the mapped pinned block.  out_mode is latched per
    tokenizer, so the path runs in a process of its own; a batch of more than one packing tile, with empty sentences, on one and
    on two replicas."""
    import subprocess
    import sys
    code = '''
import numpy as np, sys
sys.path.insert(0, %r)
import vibrato_amd as V
from oracle import oracle as ora
from tools import synth
sd = synth.SynthDict("small")
do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
to = ora.Tokenizer(do, False, 0)
text, offs = sd.sentences(3000, "lognormal_40")
offs = np.concatenate([offs[:1500], offs[1500:1501], offs[1500:1501], offs[1500:]]).astype(np.uint64)  # two empty sentences in the middle
exp, eoff = to.new_worker().tokenize_batch(text, offs)
for devs in (None, [0, 0]):
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    tv = V.Tokenizer(dv, devices=devs)
    got, goff = tv.tokenize_batch(text=text, offsets=offs).tokens_in_order()
    assert np.array_equal(goff, eoff) and got.tobytes() == exp.tobytes(), devs
print("OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VBT_H2H_OUT="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_one_call_pipelined_in_chunks_matches_oracle_and_survives_a_wrong_estimate():
    """vbt_tokenize_batch on a single device cuts a batch of >= 4 MiB into chunks that alternate between two workspaces (copy in,
    kernels and copy out of neighbouring chunks overlap; tokenize/src/main.rs:76-95 is a single-threaded caller).  The result block
    is sized from the tokens per KiB seen so far: the tokenizer's first batch runs unpipelined, the second one in chunks -- same
    records, offsets ascending over the chunk borders, empty sentences at the borders -- and a text that yields five times the
    tokens per byte (single letters between spaces) outgrows the estimate and is redone unpipelined, again with the oracle's records."""
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(45000, "lognormal_40")
    assert len(text) > (4 << 20)
    # empty sentences where the chunk borders will fall (the bounds are cut by bytes)
    mid = int(np.searchsorted(offs, offs[-1] // 2))
    offs = np.concatenate([offs[:mid + 1], offs[mid:mid + 1], offs[mid:mid + 1], offs[mid + 1:]]).astype(np.uint64)
    created0 = tv.pool_stats()[0]
    _assert_batch_equal(to, tv, text, offs)  # unpipelined: one workspace
    assert tv.pool_stats()[0] == created0 + 1
    batch, _ = _assert_batch_equal(to, tv, text, offs)  # in chunks, over two workspaces: the idle one and a new one
    assert tv.pool_stats()[0] == created0 + 2
    toks, off, cnt = batch.arrays()
    live = cnt > 0
    assert np.all(np.diff(off[live].astype(np.int64)) == cnt[live][:-1])  # packed in sentence order across the chunk borders
    _assert_batch_equal(to, tv, text, offs)
    assert tv.pool_stats()[0] == created0 + 2  # steady state: nothing new
    dense = ("a " * 100).encode()
    enc = [dense] * 24000
    o2 = np.zeros(len(enc) + 1, dtype=np.uint64)
    o2[1:] = np.cumsum([len(e) for e in enc])
    t2 = np.frombuffer(b"".join(enc), dtype=np.uint8)
    assert len(t2) > (4 << 20)
    _, ntok = _assert_batch_equal(to, tv, t2, o2)  # estimate too low: redone unpipelined
    assert ntok > 4 * 24000 * 40
    _assert_batch_equal(to, tv, t2, o2)  # and now the estimate holds
    _assert_batch_equal(to, tv, text, offs)


def test_a_packed_result_slot_that_is_too_small_says_so_in_its_header():
    """vbt_workspace_set_packed_output: a consumer of gathered slots on another rank sees nothing of the writing rank but the slot.  A
    slot that holds the batch carries flags 0 and the batch's token total; one that is too small carries the records that were written
    (never more), error flag 1 in its header -- and in the workspace's statistics -- and sharding.unpack_results refuses it (round-5 advisor:
    the header used to report the full total over a truncated slot, bytes 16..31 were never written)."""
    import torch
    from vibrato_amd import sharding
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(3000, "lognormal_40")
    exp, exp_off = to.new_worker().tokenize_batch(text, offs)
    n, ntok = 3000, len(exp)
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tv.workspace(n, len(text))
    stream = torch.cuda.current_stream().cuda_stream
    for cap, ok in ((ntok + 10, True), (ntok // 2, False), (ntok, True)):
        slot = torch.full((sharding.packed_bytes(n, cap),), 0xAB, dtype=torch.uint8, device="cuda")
        ws.set_packed_output(slot.data_ptr(), slot.numel(), n)
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), stream)
        st = ws.stats()
        head = slot[:32].cpu().numpy()
        assert int(head[:8].view(np.int64)[0]) == n and not head[16:].any()  # all 32 header bytes are written
        if ok:
            assert st["error_flags"] == 0 and int(head[12:16].view(np.uint32)[0]) == 0 and int(head[8:12].view(np.uint32)[0]) == ntok
            n_s, n_t, off, cnt, tk = sharding.unpack_results(slot, n)
            got, _ = sharding.tokens_in_sentence_order(off, cnt, tk)
            assert got.tobytes() == exp.tobytes()
        else:
            assert st["error_flags"] & 1 and int(head[12:16].view(np.uint32)[0]) & 1 and int(head[8:12].view(np.uint32)[0]) == cap
            with pytest.raises(RuntimeError):
                sharding.unpack_results(slot, n)
    ws.set_packed_output(None, 0, 0)


def test_large_batches_from_several_threads_and_from_one_give_the_same_records():
    """Batches big enough to be pipelined in chunks (>= 4 MiB), pushed by three host threads at once (calls that see each other run
    unpipelined), then by one thread alone (pipelined again after two quiet calls): every result equals the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    sd = synth.SynthDict("small")
    to, tv = _oracle_and_product(sd)
    text, offs = sd.sentences(40000, "lognormal_40")
    assert len(text) > (4 << 20)
    exp, exp_off = to.new_worker().tokenize_batch(text, offs)

    def one(_):
        got, got_off = tv.tokenize_batch(text=text, offsets=offs).tokens_in_order()
        return np.array_equal(got_off, exp_off) and got.tobytes() == exp.tobytes()

    assert one(0)  # the first batch: unpipelined, sets the estimate
    with ThreadPoolExecutor(3) as ex:
        assert all(ex.map(one, range(9)))
    assert all(one(k) for k in range(5))
