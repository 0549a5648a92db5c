"""Parity at the integer boundaries of the device lattice (run with -m gpu on an MI355X).

The reference has no length limit (vibrato/src/common.rs:15 MAX_SENTENCE_LENGTH = usize::MAX; tokenizer.rs:94-139 loops over
len_char; lattice.rs:9-10 INVALID_IDX is u16::MAX only for *connection ids*).  The device pipeline keeps positions, candidate
indices, end-list slots and LDS addresses in 16 bits and hands everything beyond them to the global-memory kernel
(gen_device.hpp: `nb64 >= 65535`, `C >= 65532`; lattice.hip: `lds0 + lds_bytes <= 65536`).  Every test here puts a sentence ON
one of those edges, one step to either side of it, and far beyond it, and compares every token record with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import vibrato_amd as V
from oracle import oracle as ora
from tools import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack(sents):
    enc = [s if isinstance(s, bytes) else s.encode("utf-8") for s in sents]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    return np.frombuffer(b"".join(enc), dtype=np.uint8), offs


def _assert_equal(to, tv, text, offs):
    exp_tok, exp_off = to.new_worker().tokenize_batch(text, offs)
    got_tok, got_off = tv.tokenize_batch(text=text, offsets=offs).tokens_in_order()
    assert np.array_equal(got_off, exp_off)
    for f in V.TOKEN_DTYPE.names:
        bad = np.nonzero(got_tok[f] != exp_tok[f])[0]
        assert bad.size == 0, (f, int(bad[0]), got_tok[bad[0]], exp_tok[bad[0]])
    return exp_tok, exp_off


def _routing(tv, text, offs):
    """(sentences filed in the pipeline's first sweep tier, sentences handed to the global-memory kernel, error flags) of one device-API
    run on the default tiers (a sentence that escalates to an escape tier is counted there as well: not added here)."""
    import torch
    n = len(offs) - 1
    ws = tv.workspace(n, len(text))
    d_text = torch.from_numpy(text.copy()).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, len(text), torch.cuda.current_stream().cuda_stream)
    st = ws.stats()
    return st["n_tier0"], st["n_tier2"], st["error_flags"]


def _small(ignore_space, mgl=0):
    sd = synth.SynthDict("small")
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    return sd, ora.Tokenizer(do, ignore_space, mgl), V.Tokenizer(dv).ignore_space(ignore_space).max_grouping_len(mgl)


def _running_text(sd, n_sentences, space_p):
    """One long valid UTF-8 byte string: synthetic sentences end to end (spaces injected when asked for)."""
    base, _ = sd.sentences(n_sentences, "lognormal_40", space_p=space_p)
    return bytes(base)


def _from(raw, start):
    """`raw` from the first character boundary at or behind `start`."""
    while (raw[start] & 0xC0) == 0x80:
        start += 1
    return raw[start:]


def _exact_bytes(raw, target):
    """A prefix of `raw` cut on a character boundary and padded with ASCII letters to exactly `target` bytes."""
    cut = target
    while cut > 0 and (raw[cut] & 0xC0) == 0x80:
        cut -= 1
    out = raw[:cut] + b"a" * (target - cut)
    assert len(out) == target
    out.decode("utf-8")
    return out


# ------------------------------------------------------------------ bytes: the u16 byte offsets of the per-character arrays

@pytest.mark.parametrize("ignore_space", [False, True])
def test_sentences_of_65534_65535_65536_and_70000_bytes(ignore_space):
    """`nb64 >= 65535` (gen_device.hpp / gen.hip): 65 534 bytes is the last length whose byte offsets fit the u16 arrays, 65 535 the
    first that does not; all four are far more characters than any generator level holds, so every one of them ends in the
    global-memory kernel -- none may be lost or cut on the way there, and the short sentences around them stay in the pipeline."""
    sd, to, tv = _small(ignore_space, 24 if ignore_space else 0)
    raw = _running_text(sd, 900, 0.08 if ignore_space else 0.0)
    assert len(raw) > 75000
    sents = [_exact_bytes(raw, 65534), _exact_bytes(raw, 300), _exact_bytes(_from(raw, 1000), 65535), b"",
             _exact_bytes(_from(raw, 2000), 65536), _exact_bytes(_from(raw, 500), 70000), "京都".encode()]
    text, offs = _pack(sents)
    exp, exp_off = _assert_equal(to, tv, text, offs)
    assert int(exp_off[1] - exp_off[0]) > 5000
    kept, fallback, flags = _routing(tv, text, offs)
    assert flags == 0 and fallback == 4 and kept == 2  # (the empty sentence is in no list)


@pytest.mark.parametrize("ignore_space", [False, True])
def test_sentences_of_30000_and_100000_characters(ignore_space):
    sd, to, tv = _small(ignore_space)
    chars = _running_text(sd, 3200, 0.05 if ignore_space else 0.0).decode("utf-8")
    assert len(chars) > 100000
    text, offs = _pack([chars[:30000], chars[7:100007], chars[:11]])
    exp, exp_off = _assert_equal(to, tv, text, offs)
    assert int(exp_off[2] - exp_off[1]) > 20000


def test_a_sentence_of_more_than_one_megabyte():
    """One sentence of 1.1 MB (~370 000 characters, ~2 M lattice nodes): u32 indices everywhere in the global-memory kernel, the
    slab grown from the scratch arena; tokens tile the sentence (ignore_space is off)."""
    sd, to, tv = _small(False)
    raw = _running_text(sd, 11000, 0.0)
    assert len(raw) > 1_150_000
    big = _exact_bytes(raw, 1_100_003)
    text, offs = _pack([_exact_bytes(raw, 120), big])
    exp, exp_off = _assert_equal(to, tv, text, offs)
    t = exp[int(exp_off[1]):]
    assert len(t) > 150000 and t["start_byte"][0] == 0 and t["end_byte"][-1] == len(big)
    assert np.array_equal(t["start_byte"][1:], t["end_byte"][:-1])


def test_worker_takes_a_70000_byte_sentence_through_the_batch_pipeline():
    """Worker::tokenize (worker.rs:49-55) on a sentence no single wavefront can take: served by the batch pipeline, same records."""
    sd, to, tv = _small(True, 24)
    raw = _running_text(sd, 900, 0.08)
    big = _exact_bytes(raw, 70000)
    text, offs = _pack([big])
    exp, _ = to.new_worker().tokenize_batch(text, offs)
    w = tv.new_worker()
    for sent in (b"\xe4\xba\xac\xe9\x83\xbd", big, b"abc"):
        w.reset_sentence(sent)
        w.tokenize()
        if sent is big:
            assert w.num_tokens() == len(exp)
            for k in (0, 1, len(exp) // 2, len(exp) - 1):
                t = w.token(k)
                assert (t.range_char[0], t.range_char[1], t.range_byte[0], t.range_byte[1], (t.lex_type << 30) | t.word_id, t.total_cost) == \
                    tuple(int(exp[f][k]) for f in V.TOKEN_DTYPE.names)
    assert w.path_stats()[1] >= 1  # the long one took the slow path


def test_a_sentence_the_scratch_arena_cannot_hold_fails_loudly(monkeypatch):
    """What the device cannot represent is an error (VBT_ERR_UNSUPPORTED = 101), never a wrong or truncated path: a 70 000-byte
    sentence needs a few MB of global scratch for its lattice; with a 1 MiB arena the batch call reports it."""
    monkeypatch.setenv("VBT_SCRATCH_MB", "1")
    sd, to, tv = _small(False)
    text, offs = _pack([_exact_bytes(_running_text(sd, 900, 0.0), 70000), "京都".encode()])
    with pytest.raises(V.VibratoError) as e:
        tv.tokenize_batch(text=text, offsets=offs)
    assert e.value.code == 101
    _, _, flags = _routing(tv, text, offs)
    assert flags & 2


# ------------------------------------------------------------------ nodes: the u16 candidate indices, slots and sequence numbers

# A dictionary on which the candidate count of a sentence is a closed formula: 11 homographs of "あ" with different ids and costs
# (a dense end list at every position), one word "ああ", one word "京", unknown words only where nothing matches (invoke 0) --
# a run of k spaces adds one grouped unknown word per space position.  "あ" x L has 11 L + (L - 1) candidates.
_DENSE_LEX = "".join(f"あ,{1 + i % 7},{1 + (3 * i) % 7},{900 + 137 * i},h{i}\n" for i in range(11)) + "ああ,3,5,1500,pair\n京,2,6,800,kyo\n"
_DENSE_CHAR = "DEFAULT 0 1 0\nSPACE 0 1 0\nHIRAGANA 0 0 0\nKANJI 0 0 0\n0x0020 SPACE\n0x3041..0x3096 HIRAGANA\n0x4E00..0x9FFF KANJI\n"
_DENSE_UNK = "DEFAULT,1,1,5000,*\nSPACE,2,2,4000,*\nHIRAGANA,3,3,6000,*\nKANJI,4,4,7000,*\n"


def _dense_matrix():
    rows = ["8 8"]
    x = 12345
    for r in range(8):
        for l in range(8):
            x = (x * 1103515245 + 12345) & 0x7FFFFFFF
            rows.append(f"{r} {l} {(x >> 8) % 6001 - 3000}")
    return "\n".join(rows) + "\n"


def _dense(ignore_space):
    m = _dense_matrix()
    do = ora.Dictionary.from_sources(_DENSE_LEX, m, _DENSE_CHAR, _DENSE_UNK)
    dv = V.SystemDictionaryBuilder.from_readers(_DENSE_LEX, m, _DENSE_CHAR, _DENSE_UNK)
    return ora.Tokenizer(do, ignore_space, 0), V.Tokenizer(dv).ignore_space(ignore_space)


def _dense_candidates(s):
    c = 0
    for i, ch in enumerate(s):
        if ch == "あ":
            c += 11 + (1 if s[i + 1:i + 2] == "あ" else 0)
        else:
            assert ch in "京 "
            c += 1
    return c


@pytest.mark.parametrize("ignore_space", [False, True])
def test_sentences_of_65531_65532_65533_and_72000_lattice_nodes(ignore_space):
    """`C >= 65532` (gen_device.hpp, gen.hip): candidate indices, end-list slots (C + 1 with BOS) and the sequence numbers of the
    tie-break (0xFFFE - sequence) are 16 bits in the LDS lattice.  65 531 candidates is the largest lattice the pipeline sweeps
    itself -- slot 65 531, BOS sequence 65 532 -- and must stay there; 65 532 and 65 533 must be handed to the global-memory kernel;
    72 000 is far beyond.  The routing is asserted, so the engineered counts are known to sit exactly on the edge."""
    to, tv = _dense(ignore_space)
    if ignore_space:  # spaces at the front, inside (single and double) and in front of the tail; あ adds 12 inside a run, 京 adds 1
        head = " " + "あ" * 2000 + " " + "あ" * 3000 + "  "
        k = 0
        while _dense_candidates(head + "あ" * (k + 1) + " ") <= 65531:
            k += 1
        edge = head + "あ" * k + " "
        edge += "京" * (65531 - _dense_candidates(edge))
    else:
        edge = "あ" * 5461
    assert _dense_candidates(edge) == 65531
    over1, over2 = edge + "京", edge + "京京"
    far = "あ" * 6100 + ("  " if ignore_space else "") + "京"
    assert _dense_candidates(far) > 72000
    for sents, kept, fb in (([edge, "あ京"], 2, 0), ([over1, "あ京", over2, far], 1, 3)):
        text, offs = _pack(sents)
        exp, exp_off = _assert_equal(to, tv, text, offs)
        assert len(set(exp["word_idx"].tolist())) >= 2  # more than one of the homographs is on the best paths
        k, f, flags = _routing(tv, text, offs)
        assert (k, f, flags) == (kept, fb, 0), (k, f, flags)


def test_a_dense_sentence_of_100000_characters_and_1_2_million_nodes():
    to, tv = _dense(True)
    text, offs = _pack(["あ" * 40000 + " 京 " + "あ" * 60000, "あ"])
    _assert_equal(to, tv, text, offs)


# ------------------------------------------------------------------ i32 matrix cells, 16-bit LDS addresses

def test_long_sentences_on_an_i32_matrix():
    """A Raw connector whose costs leave i16 (kWide instances: raw_connector.rs:153-161): 9 000 characters stay in the LDS pipeline
    (segments, the exact instance of the C++ loop), 30 000 take the global-memory kernel."""
    from tests.test_compact_connector import synth_bigram, _dict
    sd = synth.SynthDict("small")
    right, left, cost = synth_bigram(sd.num_right, sd.num_left, seed=31, templates=12, max_abs=30000, empty_pair=True)
    dv = _dict("product", right, left, cost, sd.lex, sd.char_def, sd.unk)
    do = _dict("oracle", right, left, cost, sd.lex, sd.char_def, sd.unk)
    chars = _running_text(sd, 1200, 0.05).decode("utf-8")
    text, offs = _pack([chars[:9000], chars[:30000], chars[:40]])
    to, tv = ora.Tokenizer(do, True, 0), V.Tokenizer(dv).ignore_space(True)
    exp, _ = _assert_equal(to, tv, text, offs)
    assert max(abs(int(c)) for c in exp["total_cost"][:3000]) > 40000
    kept, fallback, flags = _routing(tv, text, offs)
    assert (kept, fallback, flags) == (2, 1, 0)


@pytest.mark.parametrize("env", [{"VBT_TIERS": "65536", "VBT_SEG_BYTES": "65536"}, {"VBT_TIERS": "10240,65536", "VBT_SEG_BYTES": "65536"},
                                 {"VBT_TIERS": "65536,163840", "VBT_SEG_BYTES": "0"}])
def test_an_lds_tier_whose_arena_ends_at_65536(env, monkeypatch):
    """`lds0 + lds_bytes <= 65536` (lattice.hip): the assembly loop's records hold 16-bit LDS addresses, and a 64 KiB tier is the
    largest they reach -- its last slot record, candidate record and pass record sit just under address 65 536.  Dense sentences
    of a few hundred characters fill such a tier whole; longer ones are swept there in 64 KiB segments."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sd = synth.SynthDict("small-dense")
    do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    to, tv = ora.Tokenizer(do, True, 24), V.Tokenizer(dv).ignore_space(True).max_grouping_len(24)
    text, offs = sd.sentences(1500, "mixed", space_p=0.05)
    _assert_equal(to, tv, text, offs)
    kept, fallback, flags = _routing(tv, text, offs)
    # (without segmentation a lattice of more than 160 KiB has nowhere to go but the global-memory kernel)
    assert flags == 0 and (fallback < 15 or env["VBT_SEG_BYTES"] == "0")


def test_the_boundaries_with_the_cpp_sweep_loop():
    """The node-count, i32 and 64 KiB-tier tests once more on the `cpploop` library variant (the second statement of the recurrence)."""
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "lattice_nodes or i32 or arena",
                        "-p", "no:cacheprovider"], env=dict(os.environ, VBT_LIB_VARIANT="cpploop"), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-1500:]


# ------------------------------------------------------------------ scheduling knobs x sentence shapes, drawn

def test_drawn_scheduling_knobs_and_sentence_shapes_agree_with_the_oracle():
    """hypothesis draws (length law, sentence count, space density, ignore_space, max_grouping_len, dictionary density, VBT_TIERS,
    VBT_SEG_BYTES, VBT_GEN_LDS, VBT_GEN_LEVELS, VBT_GEN_WAVES, VBT_LEAN) together -- the hand-picked points of test_gpu_parity.py cover
    each knob alone -- with a fixed example budget and a fixed seed (derandomize): every record equals the oracle's."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    tier_sets = ["default", "default", "8192,10240,49152,163840", "10240,49152,163840", "2048,8192", "1024", "4096,6144,8192,12288,16384,65536", "1536,163840", "3072", "2048,4096,32768,163840",
                 "65536", "8192,12288,16384,24576,32768,49152,65536,163840", "512,163840"]
    level_sets = ["16384,32768,163840", "4096,8192,163840", "8192,65536,131072"]
    dicts = {}

    def dictionary(shape):
        if shape not in dicts:
            sd = synth.SynthDict(shape)
            dicts[shape] = (sd, ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk))
        return dicts[shape]

    saved = {k: os.environ.get(k) for k in ("VBT_TIERS", "VBT_SEG_BYTES", "VBT_GEN_LDS", "VBT_GEN_LEVELS", "VBT_GEN_WAVES", "VBT_GEN_WAVES1", "VBT_LEAN")}

    @settings(max_examples=30, derandomize=True, deadline=None, suppress_health_check=list(HealthCheck))
    @given(shape=st.sampled_from(["small", "small-dense", "tiny"]), law=st.sampled_from(["uniform_5_20", "lognormal_40", "mixed"]),
           n=st.integers(1, 1200), space_p=st.sampled_from([0.0, 0.03, 0.1, 0.3]), ignore_space=st.booleans(), mgl=st.sampled_from([0, 1, 3, 24]),
           tiers=st.sampled_from(tier_sets), seg=st.sampled_from(["default", "0", "first", "last"]), gen_lds=st.sampled_from([1024, 2048, 3072, 4096, 8192]),
           levels=st.sampled_from(level_sets), waves=st.sampled_from([1, 2, 4, 8]), lean=st.booleans(), seed=st.integers(1, 1 << 30))
    def run(shape, law, n, space_p, ignore_space, mgl, tiers, seg, gen_lds, levels, waves, lean, seed):
        sd, do = dictionary(shape)
        env = {"VBT_GEN_LDS": str(gen_lds), "VBT_GEN_LEVELS": levels, "VBT_GEN_WAVES": str(waves), "VBT_GEN_WAVES1": str(waves),
               "VBT_LEAN": "1" if lean else "0"}
        if tiers != "default":  # ("default": the library's own choice per batch, by bytes per sentence and measured density)
            sizes = tiers.split(",")
            env["VBT_TIERS"] = tiers
            if seg != "default":
                env["VBT_SEG_BYTES"] = "0" if seg == "0" else sizes[0] if seg == "first" else sizes[-1]
        for k in saved:
            os.environ.pop(k, None)
        os.environ.update(env)
        dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        to, tv = ora.Tokenizer(do, ignore_space, mgl), V.Tokenizer(dv).ignore_space(ignore_space).max_grouping_len(mgl)
        text, offs = sd.sentences(n, law, space_p=space_p, seed=seed)
        _assert_equal(to, tv, text, offs)
        if tiers == "default":
            _assert_equal(to, tv, text, offs)  # (the second batch runs behind the first one's density report)

    try:
        run()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
