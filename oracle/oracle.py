"""ctypes wrapper around oracle/libvibrato_oracle.so (the CPU ORACLE).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (vibrato_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvibrato_oracle.so")

TOKEN_DTYPE = np.dtype([("start_char", "<u4"), ("end_char", "<u4"), ("start_byte", "<u4"),
                        ("end_byte", "<u4"), ("word_idx", "<u4"), ("total_cost", "<i4")])

COUNTER_FIELDS = ["n_sentences", "n_bytes", "n_chars", "n_trie_steps", "n_trie_hits", "n_lex_matches",
                  "n_unk_nodes", "n_nodes", "n_pairs_ref", "n_pairs_dedup", "n_tokens"]


def build(force=False):
    src = os.path.join(_HERE, "vibrato_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvibrato_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_native(out_dir):
    """The same source compiled for the host it runs on (-O3 -march=native), for the cpu_baseline leg of
    bench.py: the committed Makefile targets x86-64-v3 because its .so travels to other machines.  Returns the path, or None
    when no compiler is available (the prebuilt library is used then)."""
    so = os.path.join(out_dir, "libvibrato_oracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu11", "-fPIC", "-fvisibility=hidden", "-shared", "-o", so,
                               os.path.join(_HERE, "vibrato_oracle.c")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.CalledProcessError):
        return None
    return so


_lib = None
_so_override = None  # set by use_library() before the first lib() call


def use_library(path):
    """Load the oracle from `path` (a build_native() result) instead of the in-tree library."""
    global _so_override
    if _lib is not None:
        raise RuntimeError("the oracle library is already loaded")
    _so_override = path


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_so_override or _SO)
        vp, u8p, sz, u32, u64, i32 = C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int32
        L.ora_dict_from_sources.restype = vp
        L.ora_dict_from_sources.argtypes = [u8p, sz, u8p, sz, u8p, sz, u8p, sz, u8p, sz]
        L.ora_dict_from_sources_binmatrix.restype = vp
        L.ora_dict_from_sources_binmatrix.argtypes = [u8p, sz, vp, u32, u32, u8p, sz, u8p, sz, u8p, sz]
        L.ora_dict_from_sources_bigram.restype = vp
        L.ora_dict_from_sources_bigram.argtypes = [u8p, sz, u8p, sz, u8p, sz, u8p, sz, u8p, sz, u8p, sz, u8p, sz]
        L.ora_scorer_new.restype = vp
        L.ora_scorer_new.argtypes = [vp, u32]
        L.ora_scorer_free.argtypes = [vp]
        L.ora_scorer_retrieve.restype = C.c_int
        L.ora_scorer_retrieve.argtypes = [vp, u32, u32, C.POINTER(i32)]
        L.ora_scorer_accumulate.restype = i32
        L.ora_scorer_accumulate.argtypes = [vp, vp, vp, u32]
        L.ora_dict_from_sources_bigram2.restype = vp
        L.ora_dict_from_sources_bigram2.argtypes = [u8p, sz, u8p, sz, u8p, sz, u8p, sz, u8p, sz, u8p, sz, C.c_int, u8p, sz]
        L.ora_dual_connector_new.restype = vp
        L.ora_dual_connector_new.argtypes = [u8p, sz, u8p, sz, u8p, sz, u8p, sz]
        L.ora_dual_connector_free.argtypes = [vp]
        L.ora_dual_connector_cost.restype = i32
        L.ora_dual_connector_cost.argtypes = [vp, u32, u32]
        L.ora_dual_connector_num.restype = u32
        L.ora_dual_connector_num.argtypes = [vp, C.c_int]
        L.ora_dual_connector_map.argtypes = [vp, vp, vp]
        L.ora_dual_connector_matrix_shape.argtypes = [vp, vp, vp]
        L.ora_raw_connector_new.restype = vp
        L.ora_raw_connector_new.argtypes = [u8p, sz, u8p, sz, u8p, sz, u8p, sz]
        L.ora_raw_connector_free.argtypes = [vp]
        L.ora_raw_connector_cost.restype = i32
        L.ora_raw_connector_cost.argtypes = [vp, u32, u32]
        L.ora_raw_connector_num.restype = u32
        L.ora_raw_connector_num.argtypes = [vp, C.c_int]
        L.ora_raw_connector_map.argtypes = [vp, vp, vp]
        L.ora_dict_set_user_lexicon.restype = C.c_int
        L.ora_dict_set_user_lexicon.argtypes = [vp, u8p, sz, u8p, sz]
        L.ora_dict_map_connection_ids.restype = C.c_int
        L.ora_dict_map_connection_ids.argtypes = [vp, vp, sz, vp, sz, u8p, sz]
        L.ora_dict_free.argtypes = [vp]
        for f in ["ora_dict_num_left", "ora_dict_num_right", "ora_dict_num_categories"]:
            getattr(L, f).restype = u32
            getattr(L, f).argtypes = [vp]
        L.ora_dict_num_words.restype = u32
        L.ora_dict_num_words.argtypes = [vp, C.c_int]
        L.ora_dict_trie_nodes.restype = u32
        L.ora_dict_trie_nodes.argtypes = [vp, C.c_int]
        L.ora_dict_conn_cost.restype = i32
        L.ora_dict_conn_cost.argtypes = [vp, u32, u32]
        L.ora_dict_char_info.restype = u32
        L.ora_dict_char_info.argtypes = [vp, u32]
        L.ora_dict_cate_id.restype = C.c_int
        L.ora_dict_cate_id.argtypes = [vp, u8p]
        L.ora_dict_word_feature.restype = C.POINTER(C.c_char)
        L.ora_dict_word_feature.argtypes = [vp, C.c_int, u32, C.POINTER(u32)]
        L.ora_dict_word_param.argtypes = [vp, C.c_int, u32, C.POINTER(i32)]
        L.ora_dict_common_prefix.restype = u32
        L.ora_dict_common_prefix.argtypes = [vp, C.c_int, vp, u32, vp, u32]
        L.ora_tokenizer_new.restype = vp
        L.ora_tokenizer_new.argtypes = [vp, C.c_int, u32, u8p, sz]
        L.ora_tokenizer_free.argtypes = [vp]
        L.ora_worker_new.restype = vp
        L.ora_worker_new.argtypes = [vp]
        L.ora_worker_free.argtypes = [vp]
        L.ora_worker_reset_sentence.restype = C.c_int
        L.ora_worker_reset_sentence.argtypes = [vp, u8p, sz]
        L.ora_worker_add_connid_counts.argtypes = [vp, vp, vp]
        L.ora_worker_add_corner_hist.argtypes = [vp, vp, vp, vp, C.c_uint32, vp]
        L.ora_worker_tokenize.argtypes = [vp]
        L.ora_worker_tokenize_counted.argtypes = [vp]
        L.ora_worker_num_tokens.restype = u32
        L.ora_worker_num_tokens.argtypes = [vp]
        L.ora_worker_eos_cost.restype = i32
        L.ora_worker_eos_cost.argtypes = [vp]
        L.ora_worker_token.argtypes = [vp, u32, vp]
        L.ora_worker_token_ids.argtypes = [vp, u32, C.POINTER(i32)]
        L.ora_worker_counters.argtypes = [vp, vp]
        L.ora_worker_reset_counters.argtypes = [vp]
        L.ora_tokenize_batch.restype = u64
        L.ora_tokenize_batch.argtypes = [vp, vp, vp, u64, vp, u64, vp, C.c_int]
        L.ora_tokenize_format_batch.restype = u64
        L.ora_tokenize_format_batch.argtypes = [vp, vp, vp, u64, C.c_int, vp, u64]
        _lib = L
    return _lib


class OracleError(Exception):
    pass


def _b(x):
    return x.encode("utf-8") if isinstance(x, str) else bytes(x)


class Dictionary:
    """Mirror of vibrato::Dictionary / SystemDictionaryBuilder (dictionary.rs, builder.rs:64-89)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_sources(cls, lex, matrix, char_def, unk):
        lex, matrix, char_def, unk = _b(lex), _b(matrix), _b(char_def), _b(unk)
        err = C.create_string_buffer(512)
        h = lib().ora_dict_from_sources(lex, len(lex), matrix, len(matrix), char_def, len(char_def), unk, len(unk), err, 512)
        if not h:
            raise OracleError(err.value.decode("utf-8", "replace"))
        d = cls(h)
        d._keep = (lex, unk)
        return d

    @classmethod
    def from_sources_binmatrix(cls, lex, matrix_i16, num_right, num_left, char_def, unk):
        lex, char_def, unk = _b(lex), _b(char_def), _b(unk)
        m = np.ascontiguousarray(matrix_i16, dtype=np.int16)
        assert m.size == num_right * num_left
        err = C.create_string_buffer(512)
        h = lib().ora_dict_from_sources_binmatrix(lex, len(lex), m.ctypes.data, num_right, num_left,
                                                  char_def, len(char_def), unk, len(unk), err, 512)
        if not h:
            raise OracleError(err.value.decode("utf-8", "replace"))
        return cls(h)

    @classmethod
    def from_sources_bigram(cls, lex, bigram_right, bigram_left, bigram_cost, char_def, unk, dual_connector=False):
        """SystemDictionaryBuilder::from_readers_with_bigram_info (builder.rs:111-160): a RawConnector, or with dual_connector=True
        a DualConnector (dual_connector.c: its matrix cells are clamped to i16 and its padded class rows price ("", "") pairs, so
        the two are NOT the same cost function on every model)."""
        a = [_b(x) for x in (lex, bigram_right, bigram_left, bigram_cost, char_def, unk)]
        err = C.create_string_buffer(512)
        args = []
        for x in a:
            args += [x, len(x)]
        h = lib().ora_dict_from_sources_bigram2(*args, int(bool(dual_connector)), err, 512)
        if not h:
            raise OracleError(err.value.decode("utf-8", "replace"))
        return cls(h)

    def reset_user_lexicon(self, csv):
        err = C.create_string_buffer(512)
        if csv is None:
            ok = lib().ora_dict_set_user_lexicon(self._h, None, 0, err, 512)
        else:
            csv = _b(csv)
            ok = lib().ora_dict_set_user_lexicon(self._h, csv, len(csv), err, 512)
        if not ok:
            raise OracleError(err.value.decode("utf-8", "replace"))
        return self

    def map_connection_ids_from_iter(self, lmap, rmap):
        l = np.ascontiguousarray(list(lmap), dtype=np.uint16)
        r = np.ascontiguousarray(list(rmap), dtype=np.uint16)
        err = C.create_string_buffer(512)
        if not lib().ora_dict_map_connection_ids(self._h, l.ctypes.data, len(l), r.ctypes.data, len(r), err, 512):
            raise OracleError(err.value.decode("utf-8", "replace"))
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ora_dict_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown
            pass

    def num_words(self, lex_type=0):
        return lib().ora_dict_num_words(self._h, lex_type)

    @property
    def num_left(self):
        return lib().ora_dict_num_left(self._h)

    @property
    def num_right(self):
        return lib().ora_dict_num_right(self._h)

    def conn_cost(self, right_id, left_id):
        return lib().ora_dict_conn_cost(self._h, right_id, left_id)

    def char_info(self, cp):
        x = lib().ora_dict_char_info(self._h, cp)
        return {"cate_idset": x & 0x3FFFF, "base_id": (x >> 18) & 0xFF, "invoke": (x >> 26) & 1,
                "group": (x >> 27) & 1, "length": x >> 28}

    def word_feature(self, lex_type, word_id):
        n = C.c_uint32()
        p = lib().ora_dict_word_feature(self._h, lex_type, word_id, C.byref(n))
        return C.string_at(p, n.value).decode("utf-8")

    def word_param(self, lex_type, word_id):
        out = (C.c_int32 * 3)()
        lib().ora_dict_word_param(self._h, lex_type, word_id, out)
        return tuple(out)

    def common_prefix(self, text, lex_type=0):
        cps = np.array([ord(c) for c in text], dtype=np.uint32)
        out = np.zeros((256, 5), dtype=np.int32)
        n = lib().ora_dict_common_prefix(self._h, lex_type, cps.ctypes.data, len(cps), out.ctypes.data, 256)
        return out[:n].tolist()


LEX_NAMES = ["System", "User", "Unknown"]


class Tokenizer:
    """Mirror of vibrato::Tokenizer (tokenizer.rs:13-84)."""

    def __init__(self, dictionary, ignore_space=False, max_grouping_len=0):
        self.dict = dictionary
        err = C.create_string_buffer(512)
        self._h = lib().ora_tokenizer_new(dictionary._h, int(ignore_space), max_grouping_len, err, 512)
        if not self._h:
            raise OracleError(err.value.decode("utf-8", "replace"))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ora_tokenizer_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown
            pass

    def new_worker(self):
        return Worker(self)


class Worker:
    """Mirror of vibrato::tokenizer::worker::Worker (worker.rs:13-75)."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        self._h = lib().ora_worker_new(tokenizer._h)
        self._text = b""

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ora_worker_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown
            pass

    def reset_sentence(self, text):
        self._text = _b(text)
        if not lib().ora_worker_reset_sentence(self._h, self._text, len(self._text)):
            raise OracleError("invalid utf-8")

    def tokenize(self, counted=False):
        (lib().ora_worker_tokenize_counted if counted else lib().ora_worker_tokenize)(self._h)

    def num_tokens(self):
        return lib().ora_worker_num_tokens(self._h)

    def eos_cost(self):
        return lib().ora_worker_eos_cost(self._h)

    def token(self, i):
        rec = np.zeros(1, dtype=TOKEN_DTYPE)
        lib().ora_worker_token(self._h, i, rec.ctypes.data)
        r = rec[0]
        ids = (C.c_int32 * 2)()
        lib().ora_worker_token_ids(self._h, i, ids)
        lex_type, word_id = int(r["word_idx"]) >> 30, int(r["word_idx"]) & 0x3FFFFFFF
        d = self.tokenizer.dict
        return {
            "surface": self._text[r["start_byte"]:r["end_byte"]].decode("utf-8"),
            "range_char": [int(r["start_char"]), int(r["end_char"])],
            "range_byte": [int(r["start_byte"]), int(r["end_byte"])],
            "feature": d.word_feature(lex_type, word_id),
            "lex_type": lex_type, "word_id": word_id,
            "left_id": ids[0], "right_id": ids[1],
            "word_cost": d.word_param(lex_type, word_id)[2],
            "total_cost": int(r["total_cost"]),
        }

    def add_connid_counts(self, lid, rid):
        """Worker::update_connid_counts for the sentence just tokenized; lid/rid are np.uint64 arrays."""
        lib().ora_worker_add_connid_counts(self._h, lid.ctypes.data, rid.ctypes.data)

    def add_corner_hist(self, rank_left, rank_right, bounds, hist):
        """Statistics (not part of the reference): histogram of max(rank_left[left id], rank_right[right id]) over the (node, predecessor)
        pairs of the sentence just tokenized; np.uint32 ranks and bounds, np.uint64 hist of len(bounds) + 1."""
        lib().ora_worker_add_corner_hist(self._h, rank_left.ctypes.data, rank_right.ctypes.data, bounds.ctypes.data, len(bounds), hist.ctypes.data)

    def counters(self):
        buf = (C.c_uint64 * len(COUNTER_FIELDS))()
        lib().ora_worker_counters(self._h, buf)
        return dict(zip(COUNTER_FIELDS, [int(x) for x in buf]))

    def reset_counters(self):
        lib().ora_worker_reset_counters(self._h)

    def tokenize_format_batch(self, text_u8, offsets_u64, mode="mecab", out=None):
        """Text in, `tokenize` output out (tokenize/src/main.rs:78-127), one thread, into a caller-owned np.uint8 buffer
        (sized by a first call when None).  Returns (bytes, buffer)."""
        text = np.ascontiguousarray(text_u8, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        m = {"mecab": 0, "wakati": 1, "detail": 2}[mode]
        if out is None:
            need = lib().ora_tokenize_format_batch(self._h, text.ctypes.data, offs.ctypes.data, len(offs) - 1, m, None, 0)
            if need == 2**64 - 1:
                raise OracleError("invalid utf-8")
            out = np.empty(int(need), dtype=np.uint8)
        got = lib().ora_tokenize_format_batch(self._h, text.ctypes.data, offs.ctypes.data, len(offs) - 1, m, out.ctypes.data, len(out))
        if got == 2**64 - 1:
            raise OracleError("invalid utf-8")
        return int(got), out

    def tokenize_batch(self, text_u8, offsets_u64, counted=False, want_tokens=True):
        """text_u8: np.uint8 array; offsets_u64: n+1 offsets. Returns (tokens, tok_off)."""
        text = np.ascontiguousarray(text_u8, dtype=np.uint8)
        offs = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        n = len(offs) - 1
        tok_off = np.zeros(n + 1, dtype=np.uint64)
        if want_tokens:
            cap = int(offs[-1] - offs[0]) + 1
            toks = np.zeros(cap, dtype=TOKEN_DTYPE)
            total = lib().ora_tokenize_batch(self._h, text.ctypes.data, offs.ctypes.data, n, toks.ctypes.data, cap,
                                             tok_off.ctypes.data, int(counted))
        else:
            toks = None
            total = lib().ora_tokenize_batch(self._h, text.ctypes.data, offs.ctypes.data, n, None, 0,
                                             tok_off.ctypes.data, int(counted))
        if total == 2**64 - 1:
            raise OracleError("oracle batch failed (invalid utf-8 or capacity)")
        return (toks[:total] if want_tokens else None), tok_off


def format_tokens(worker, mode="mecab"):
    """tokenize/src/main.rs:83-127 output formats."""
    n = worker.num_tokens()
    toks = [worker.token(i) for i in range(n)]
    if mode == "mecab":
        return "".join(f"{t['surface']}\t{t['feature']}\n" for t in toks) + "EOS\n"
    if mode == "wakati":
        return " ".join(t["surface"] for t in toks) + "\n"
    if mode == "detail":
        return "".join(
            f"{t['surface']}\t{t['feature']}\tlex_type={LEX_NAMES[t['lex_type']]}\tleft_id={t['left_id']}\t"
            f"right_id={t['right_id']}\tword_cost={t['word_cost']}\ttotal_cost={t['total_cost']}\n" for t in toks) + "EOS\n"
    raise ValueError(mode)


class Scorer:
    """ScorerBuilder + Scorer (connector/raw_connector/scorer.rs:103-282), for the reference's unit vectors."""

    def __init__(self, triples):
        t = np.ascontiguousarray(np.array(triples, dtype=np.int64).astype(np.uint32).reshape(-1, 3))
        self._h = lib().ora_scorer_new(t.ctypes.data, len(t))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().ora_scorer_free(self._h)
                self._h = None
        except Exception:
            pass

    def retrieve_cost(self, key1, key2):
        out = C.c_int32()
        return out.value if lib().ora_scorer_retrieve(self._h, key1, key2, C.byref(out)) else None

    def accumulate_cost(self, keys1, keys2):
        a = np.ascontiguousarray(keys1, dtype=np.uint32)
        b = np.ascontiguousarray(keys2, dtype=np.uint32)
        return lib().ora_scorer_accumulate(self._h, a.ctypes.data, b.ctypes.data, len(a))


class RawConnector:
    """RawConnector::from_readers / cost / map_connection_ids (connector/raw_connector.rs:45-161)."""
    _kind = "raw"

    def __init__(self, bigram_right, bigram_left, bigram_cost):
        r, l, c = _b(bigram_right), _b(bigram_left), _b(bigram_cost)
        err = C.create_string_buffer(512)
        self._h = getattr(lib(), f"ora_{self._kind}_connector_new")(r, len(r), l, len(l), c, len(c), err, 512)
        if not self._h:
            raise OracleError(err.value.decode("utf-8", "replace"))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                getattr(lib(), f"ora_{self._kind}_connector_free")(self._h)
                self._h = None
        except Exception:
            pass

    def cost(self, right_id, left_id):
        return getattr(lib(), f"ora_{self._kind}_connector_cost")(self._h, right_id, left_id)

    @property
    def num_left(self):
        return getattr(lib(), f"ora_{self._kind}_connector_num")(self._h, 1)

    @property
    def num_right(self):
        return getattr(lib(), f"ora_{self._kind}_connector_num")(self._h, 0)

    def map_connection_ids(self, left, right):
        """ConnIdMapper::new(left, right): new id = left[old id] (mapper.rs:14-17)."""
        l = np.ascontiguousarray(left, dtype=np.uint16)
        r = np.ascontiguousarray(right, dtype=np.uint16)
        getattr(lib(), f"ora_{self._kind}_connector_map")(self._h, l.ctypes.data, r.ctypes.data)


class DualConnector(RawConnector):
    """DualConnector::from_readers / cost / map_connection_ids (connector/dual_connector.rs:145-279)."""
    _kind = "dual"

    @property
    def matrix_shape(self):
        """(num_right, num_left) of the class matrix"""
        a, b = C.c_uint32(), C.c_uint32()
        lib().ora_dual_connector_matrix_shape(self._h, C.byref(a), C.byref(b))
        return a.value, b.value
