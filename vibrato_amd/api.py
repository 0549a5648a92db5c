"""Host-side mirror of vibrato's public API for the tokenize() path
(vibrato/src/lib.rs:52-78): Dictionary / SystemDictionaryBuilder, Tokenizer, Worker,
Token, plus the batched entry points this project adds.  Everything computes on the
MI355X through libvibrato_hip.so; Python only moves handles and bytes."""
import ctypes as C
import os
import threading

import numpy as np

from . import _native as N
from ._native import VibratoError  # noqa: F401

LEX_NAMES = ("System", "User", "Unknown")  # LexType, dictionary.rs:30-40
TOKEN_DTYPE = np.dtype([("start_char", "<u4"), ("end_char", "<u4"), ("start_byte", "<u4"),
                        ("end_byte", "<u4"), ("word_idx", "<u4"), ("total_cost", "<i4")])


def _b(x):
    return x.encode("utf-8") if isinstance(x, str) else bytes(x)


class Dictionary:
    """vibrato::Dictionary (dictionary.rs:53-259)."""

    def __init__(self, handle, owned=True, owner=None):
        self._h = handle
        self._owned = owned
        self._owner = owner  # a borrowed view keeps the Tokenizer that owns the memory alive

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._owned:
                N.lib().vbt_dict_free(self._h)
            self._h = None
        except Exception:  # interpreter teardown: module globals may be gone
            pass

    def _handle(self):
        if not self._h:
            raise VibratoError(3, "dictionary was moved into a Tokenizer")
        return self._h

    @classmethod
    def read(cls, rdr):
        """Dictionary::read (dictionary.rs:173-197).  `rdr`: bytes, a binary file object or a path; `system.dic` as written
        by Dictionary::write, or the released `system.dic.zst` (a zstd frame around it is unwrapped)."""
        if isinstance(rdr, (str, os.PathLike)):
            with open(rdr, "rb") as fh:
                data = fh.read()
        elif hasattr(rdr, "read"):
            data = rdr.read()
        else:
            data = bytes(rdr)
        buf = np.frombuffer(data, dtype=np.uint8)
        h = C.c_void_p()
        N.check(N.lib().vbt_dict_read(buf.ctypes.data if len(buf) else None, len(buf), C.byref(h)))
        return cls(h)

    def write(self, wtr=None, zstd_level=None):
        """Dictionary::write (dictionary.rs:142-150); zstd_level=19 gives what the reference's `compile` CLI writes
        (compile/src/main.rs:98).  Returns the bytes (and writes them to `wtr` if given)."""
        out = C.c_void_p()
        n = C.c_size_t()
        N.check(N.lib().vbt_dict_write(self._handle(), -1 if zstd_level is None else int(zstd_level), C.byref(out), C.byref(n)))
        try:
            data = C.string_at(out, n.value)
        finally:
            N.lib().vbt_free(out)
        if wtr is not None:
            wtr.write(data)
        return data

    def reset_user_lexicon_from_reader(self, csv):
        """Dictionary::reset_user_lexicon_from_reader (dictionary.rs:209-229); csv=None clears."""
        if csv is None:
            N.check(N.lib().vbt_dict_set_user_lexicon(self._handle(), None, 0))
        else:
            csv = _b(csv)
            N.check(N.lib().vbt_dict_set_user_lexicon(self._handle(), csv, len(csv)))
        return self

    def map_connection_ids_from_iter(self, lmap, rmap):
        """Dictionary::map_connection_ids_from_iter (dictionary.rs:245-259)."""
        l = np.ascontiguousarray(list(lmap), dtype=np.uint16)
        r = np.ascontiguousarray(list(rmap), dtype=np.uint16)
        N.check(N.lib().vbt_dict_map_connection_ids(self._handle(), l.ctypes.data, len(l), r.ctypes.data, len(r)))
        return self

    def num_words(self, lex_type=0):
        return N.lib().vbt_dict_num_words(self._handle(), lex_type)

    @property
    def connector_kind(self):
        """"Matrix" | "Raw" | "Dual" (ConnectorWrapper, connector.rs:30-35)."""
        return ("Matrix", "Raw", "Dual")[N.lib().vbt_dict_connector_kind(self._handle())]

    @property
    def num_left(self):
        return N.lib().vbt_dict_num_left(self._handle())

    @property
    def num_right(self):
        return N.lib().vbt_dict_num_right(self._handle())

    def word_feature(self, lex_type, word_id):
        """Dictionary::word_feature (dictionary.rs:108-114)."""
        p, n = C.c_void_p(), C.c_size_t()
        N.check(N.lib().vbt_dict_word_feature(self._handle(), lex_type, word_id, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value).decode("utf-8")

    def word_param(self, lex_type, word_id):
        out = (C.c_int32 * 3)()
        N.check(N.lib().vbt_dict_word_param(self._handle(), lex_type, word_id, out))
        return tuple(out)

    def conn_cost(self, right_id, left_id):
        """ConnectorCost::cost(right_id, left_id) (connector.rs:25-28)."""
        out = C.c_int32()
        N.check(N.lib().vbt_dict_conn_cost(self._handle(), right_id, left_id, C.byref(out)))
        return out.value

    def char_info(self, cp):
        x = N.lib().vbt_dict_char_info(self._handle(), cp)
        return {"cate_idset": x & 0x3FFFF, "base_id": (x >> 18) & 0xFF, "invoke": (x >> 26) & 1,
                "group": (x >> 27) & 1, "length": x >> 28}

    def cate_id(self, name):
        name = _b(name)
        return N.lib().vbt_dict_cate_id(self._handle(), name, len(name))

    def common_prefix(self, text, lex_type=0):
        """Lexicon::common_prefix_iterator (lexicon.rs:33-46): [(word_id, end_char, left, right, cost)]."""
        cps = np.array([ord(c) for c in text], dtype=np.uint32)
        out = np.zeros((1024, 2), dtype=np.uint32)
        n = N.lib().vbt_dict_common_prefix(self._handle(), lex_type, cps.ctypes.data, len(cps), out.ctypes.data, 1024)
        return [[int(w), int(e), *self.word_param(lex_type, int(w))] for w, e in out[:n]]


class SystemDictionaryBuilder:
    """vibrato::SystemDictionaryBuilder (dictionary/builder.rs:12-89)."""

    @staticmethod
    def from_readers(lex, matrix, char_def, unk):
        lex, matrix, char_def, unk = _b(lex), _b(matrix), _b(char_def), _b(unk)
        h = C.c_void_p()
        N.check(N.lib().vbt_dict_from_sources(lex, len(lex), matrix, len(matrix), char_def, len(char_def), unk, len(unk), C.byref(h)))
        return Dictionary(h)

    @staticmethod
    def from_readers_with_bigram_info(lex, bigram_right, bigram_left, bigram_cost, char_def, unk, dual_connector=False):
        """SystemDictionaryBuilder::from_readers_with_bigram_info (builder.rs:111-160): a RawConnector, or with dual_connector=True a
        DualConnector (small matrix + 8-wide raw part)."""
        a = [_b(x) for x in (lex, bigram_right, bigram_left, bigram_cost, char_def, unk)]
        h = C.c_void_p()
        args = []
        for x in a:
            args += [x, len(x)]
        N.check(N.lib().vbt_dict_from_sources_bigram(*args, int(dual_connector), C.byref(h)))
        return Dictionary(h)

    @staticmethod
    def from_readers_binmatrix(lex, matrix_i16, num_right, num_left, char_def, unk):
        """Same with a binary connection matrix laid out data[left*num_right+right]."""
        lex, char_def, unk = _b(lex), _b(char_def), _b(unk)
        m = np.ascontiguousarray(matrix_i16, dtype=np.int16)
        if m.size != num_right * num_left:
            raise VibratoError(1, "matrix: size must be num_right*num_left")
        h = C.c_void_p()
        N.check(N.lib().vbt_dict_from_sources_binmatrix(lex, len(lex), m.ctypes.data, num_right, num_left, char_def,
                                                        len(char_def), unk, len(unk), C.byref(h)))
        return Dictionary(h)


class Token:
    """vibrato::token::Token (token.rs:8-92)."""

    __slots__ = ("surface", "feature", "range_char", "range_byte", "lex_type", "word_id", "left_id", "right_id",
                 "word_cost", "total_cost")

    def __init__(self, t):
        self.surface = C.string_at(t.surface, t.surface_len).decode("utf-8")
        self.feature = C.string_at(t.feature, t.feature_len).decode("utf-8")
        self.range_char = (t.start_char, t.end_char)
        self.range_byte = (t.start_byte, t.end_byte)
        self.lex_type = t.lex_type
        self.word_id = t.word_id
        self.left_id = t.left_id
        self.right_id = t.right_id
        self.word_cost = t.word_cost
        self.total_cost = t.total_cost

    def __repr__(self):
        return (f"Token(surface={self.surface!r}, range_char={self.range_char}, feature={self.feature!r}, "
                f"lex_type={LEX_NAMES[self.lex_type]}, total_cost={self.total_cost})")


class Tokenizer:
    """vibrato::Tokenizer (tokenizer.rs:13-84). Creating one uploads the dictionary image to the GPU."""

    def __init__(self, dictionary, device=-1, devices=None):
        """device: HIP device of the dictionary image (-1 = current).  devices: a list of HIP devices instead -- one replica of the
        image on each, tokenize_batch splits every batch over them and gathers the results into one host block (new; a device
        may be listed twice)."""
        self._dict_in = dictionary
        self._ignore_space = False
        self._max_grouping_len = 0
        self._device = device
        self._devices = list(devices) if devices is not None else None
        self._h = None
        self._dict = None
        self._build_lock = threading.Lock()

    @classmethod
    def new(cls, dictionary, device=-1, devices=None):
        return cls(dictionary, device, devices)

    def ignore_space(self, yes):
        """Tokenizer::ignore_space (tokenizer.rs:42-55)."""
        self._require_unbuilt()
        if yes and self._dict_in.cate_id("SPACE") < 0:  # checked before anything changes (tokenizer.rs:44-49)
            raise VibratoError(1, "dict: SPACE is not defined in the input dictionary (i.e., char.def).")
        self._ignore_space = bool(yes)
        return self

    def max_grouping_len(self, n):
        """Tokenizer::max_grouping_len (tokenizer.rs:67-74); 0 = unlimited."""
        self._require_unbuilt()
        self._max_grouping_len = int(n)
        return self

    def _require_unbuilt(self):
        if self._h:
            raise VibratoError(3, "tokenizer options must be set before the first use")

    def _handle(self):
        if not self._h:
            with self._build_lock:  # the device image is built once, whichever thread gets here first
                if not self._h:
                    h = C.c_void_p()
                    if self._devices is not None:
                        devs = (C.c_int * len(self._devices))(*self._devices)
                        N.check(N.lib().vbt_tokenizer_new_multi(self._dict_in._handle(), int(self._ignore_space), self._max_grouping_len,
                                                                devs, len(self._devices), C.byref(h)))
                    else:
                        N.check(N.lib().vbt_tokenizer_new(self._dict_in._handle(), int(self._ignore_space), self._max_grouping_len,
                                                          self._device, C.byref(h)))
                    self._dict_in._h = None  # moved (Tokenizer::new consumes the dictionary)
                    self._dict = C.c_void_p(N.lib().vbt_tokenizer_dictionary(h))
                    self._h = h
        return self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                N.lib().vbt_tokenizer_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown: module globals may be gone
            pass

    def num_devices(self):
        """Devices the tokenizer's batches are split over (1 unless created with `devices=[...]`)."""
        return int(N.lib().vbt_tokenizer_num_devices(self._handle()))

    def connid_reorder_info(self):
        """The internal renumbering of the connection ids by measured usage (include/vibrato_hip.h:
        vbt_tokenizer_connid_reorder_info): which image is in use and what the calibration cost."""
        out = (C.c_uint64 * 8)()
        N.check(N.lib().vbt_tokenizer_connid_reorder_info(self._handle(), out))
        return {"epoch": int(out[0]), "state": ("waiting", "running", "done", "off")[min(int(out[1]), 3)], "sample_sentences": int(out[2]),
                "min_sentences": int(out[3]), "ms": out[4] / 1e3, "moved_left": int(out[5]), "moved_right": int(out[6])}

    def wait_connid_reorder(self, timeout_s=None):
        """Blocks while a calibration of the internal renumbering is running (it runs on a background thread behind the tokenizer's
        first large batch); True when none is running any more."""
        idle = C.c_int(0)
        N.check(N.lib().vbt_tokenizer_connid_reorder_wait(self._handle(), -1 if timeout_s is None else int(timeout_s * 1000), C.byref(idle)))
        return bool(idle.value)

    def lattice_density(self):
        """Candidates per input byte of the last batch that reported (vbt_tokenizer_lattice_density); 0.0 before the first."""
        d = C.c_double(0.0)
        N.check(N.lib().vbt_tokenizer_lattice_density(self._handle(), C.byref(d)))
        return float(d.value)

    def calibrate(self, sentences=None, text=None, offsets=None):
        """The internal renumbering of the connection ids done up front, synchronously, from host text (vbt_tokenizer_calibrate)."""
        if sentences is not None:
            enc = [_b(s) for s in sentences]
            offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
            if enc:
                offsets[1:] = np.cumsum([len(e) for e in enc])
            text = np.frombuffer(b"".join(enc), dtype=np.uint8)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        N.check(N.lib().vbt_tokenizer_calibrate(self._handle(), text.ctypes.data if text.size else None, offsets.ctypes.data, len(offsets) - 1))
        return self.connid_reorder_info()

    def dictionary(self):
        """Tokenizer::dictionary (tokenizer.rs:77-79)."""
        self._handle()
        return Dictionary(self._dict, owned=False, owner=self)

    def pool_stats(self):
        """(created, reused, idle) workspaces of the host-buffer entry point's pool."""
        c, r, i = C.c_uint64(), C.c_uint64(), C.c_uint64()
        N.check(N.lib().vbt_tokenizer_pool_stats(self._handle(), C.byref(c), C.byref(r), C.byref(i)))
        return c.value, r.value, i.value

    def trim_pool(self):
        """Releases the idle pooled workspaces and pinned blocks of tokenize_batch (they are re-created on demand)."""
        N.check(N.lib().vbt_tokenizer_trim_pool(self._handle()))

    def new_worker(self):
        """Tokenizer::new_worker (tokenizer.rs:82-84)."""
        return Worker(self)

    def tokenize_batch(self, sentences=None, text=None, offsets=None):
        """Batched tokenization of host data: a list of str, or (uint8 text, uint64 offsets[n+1])."""
        if sentences is not None:
            enc = [_b(s) for s in sentences]
            offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
            if enc:
                offsets[1:] = np.cumsum([len(e) for e in enc])
            text = np.frombuffer(b"".join(enc), dtype=np.uint8)
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        N.check(N.lib().vbt_tokenize_batch(self._handle(), text.ctypes.data if text.size else None, offsets.ctypes.data,
                                           len(offsets) - 1, C.byref(h)))
        return Batch(self, h)

    def tokenize_lines(self, lines, batch_bytes=16 << 20, batch_lines=100000):
        """The per-line loop of the reference's callers (tokenize/src/main.rs:78-95) over an iterable of lines, batched behind the
        scenes: lines are collected until `batch_bytes` of text (or `batch_lines` lines) are in hand, tokenized as ONE device batch and
        yielded one at a time, in input order, as (batch, index) -- batch.num_tokens(index), batch.token(index, i).  Worker.tokenize
        costs ~39 us per call on the GPU, a batch ~17 ns per line: whoever has more than a handful of lines in hand wants this."""
        buf, size = [], 0
        for line in lines:
            e = _b(line)
            buf.append(e)
            size += len(e)
            if size >= batch_bytes or len(buf) >= batch_lines:
                b = self.tokenize_batch(sentences=buf)
                for i in range(len(buf)):
                    yield b, i
                buf, size = [], 0
        if buf:
            b = self.tokenize_batch(sentences=buf)
            for i in range(len(buf)):
                yield b, i

    def host_pipeline_benchmark(self, text, offsets, threads=2, rounds=4, repeats=3):
        """Host-to-host streaming throughput of vbt_tokenize_batch: `threads` host threads each push the whole batch through
        the thread-safe entry point `rounds` times, concurrently -- every call owns a pooled workspace, pinned staging and a
        stream, so the H2D copy and the D2H copy (SDMA engines) of one batch run under the kernels of another.  Includes everything a
        caller of the reference's 3-call loop pays: the copy of the text into the batch, both PCIe directions and the result
        arrays landing in (pinned) host memory.  threads=1, rounds=1 is the latency of one unpipelined call."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        h = self._handle()
        L = N.lib()

        def stream(_):
            nt = 0
            for _r in range(rounds):
                b = C.c_void_p()
                N.check(L.vbt_tokenize_batch(h, text.ctypes.data, offsets.ctypes.data, n, C.byref(b)))
                nt = L.vbt_batch_total_tokens(b)
                L.vbt_batch_free(b)
            return nt

        best, tokens = None, 0
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(stream, range(threads)))  # warm-up: fills the workspace and pinned-block pools
            for _ in range(repeats):
                t = time.perf_counter()
                tokens = list(ex.map(stream, range(threads)))[0]
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
        nbytes = int(offsets[-1] - offsets[0])
        calls = threads * rounds
        return {"sentences_per_s": round(n * calls / best, 1), "input_MB_per_s": round(nbytes * calls / best / 1e6, 2),
                "ms_per_batch": round(best * 1e3 / calls, 3), "batch_sentences": n, "host_threads": threads, "batches_per_thread": rounds,
                "tokens_per_batch": int(tokens), "pool": self.pool_stats(),
                "includes": "host copy of the text into the batch, H2D, kernels, D2H of token records into pinned host memory"}

    def text_pipeline_benchmark(self, text, offsets, batches=8, mode="mecab"):
        """Text in -> `tokenize` output text out (tokenize/src/main.rs:76-95) as a stream of batches: one host thread pushes batch
        k + 1 through vbt_tokenize_batch while another renders batch k with vbt_batch_format (both calls release the GIL), which
        is how vibrato_amd.cli runs.  Wall time of `batches` batches of the given text, output bytes counted, nothing copied
        into Python objects."""
        import queue
        import threading
        import time
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        m = {"mecab": 0, "wakati": 1, "detail": 2}[mode]
        L, h = N.lib(), self._handle()

        def tokenize():
            b = C.c_void_p()
            N.check(L.vbt_tokenize_batch(h, text.ctypes.data, offsets.ctypes.data, n, C.byref(b)))
            return b

        def render(b):
            p, ln = C.c_void_p(), C.c_size_t()
            try:
                N.check(L.vbt_batch_format(b, m, C.byref(p), C.byref(ln)))
                L.vbt_free(p)
            finally:
                L.vbt_batch_free(b)
            return ln.value

        render(tokenize())  # warm-up: pools, the flat feature table
        q = queue.Queue(maxsize=2)
        err = []

        def producer():
            try:
                for _ in range(batches):
                    q.put(tokenize())
            except Exception as e:  # noqa: BLE001 -- handed to the consumer
                err.append(e)
            q.put(None)

        t0 = time.perf_counter()
        th = threading.Thread(target=producer)
        th.start()
        out_bytes = 0
        while True:
            b = q.get()
            if b is None:
                break
            out_bytes += render(b)
        th.join()
        dt = time.perf_counter() - t0
        if err:
            raise err[0]
        return {"sentences_per_s": round(n * batches / dt, 1), "ms_per_batch": round(dt / batches * 1e3, 3), "batches": batches,
                "batch_sentences": n, "output_MB_per_s": round(out_bytes / dt / 1e6, 1), "mode": mode,
                "what": "two host threads: vbt_tokenize_batch of batch k + 1 under vbt_batch_format of batch k"}

    def workspace(self, max_sentences, max_bytes):
        return Workspace(self, max_sentences, max_bytes)


class Worker:
    """vibrato::tokenizer::worker::Worker (worker.rs:13-75): per-sentence API -- ONE kernel launch per tokenize(), text and token
    records through the worker's pinned host block."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer
        h = C.c_void_p()
        N.check(N.lib().vbt_worker_new(tokenizer._handle(), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                N.lib().vbt_worker_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown: module globals may be gone
            pass

    def reset_sentence(self, text):
        text = _b(text)
        N.check(N.lib().vbt_worker_reset_sentence(self._h, text, len(text)))

    def tokenize(self):
        N.check(N.lib().vbt_worker_tokenize(self._h))

    def num_tokens(self):
        return N.lib().vbt_worker_num_tokens(self._h)

    def token(self, i):
        t = N.Token()
        N.check(N.lib().vbt_worker_token(self._h, i, C.byref(t)))
        return Token(t)

    def token_iter(self):
        """Worker::token_iter (worker.rs:71-75)."""
        return (self.token(i) for i in range(self.num_tokens()))

    def path_stats(self):
        """(sentences served by the single-launch latency path, sentences handed to the batch pipeline)."""
        f, s = C.c_uint64(), C.c_uint64()
        N.check(N.lib().vbt_worker_path_stats(self._h, C.byref(f), C.byref(s)))
        return f.value, s.value

    def loop_benchmark(self, text, offsets, rounds=1):
        """The reference's 3-call loop (tokenize/src/main.rs:78-82) over (uint8 text, uint64 offsets[n+1]), timed inside the
        library: {sentences_per_s, us_per_call, tokens}."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        sec, ntok = C.c_double(), C.c_uint64()
        N.check(N.lib().vbt_worker_loop_benchmark(self._h, text.ctypes.data if text.size else None, offsets.ctypes.data, n, int(rounds),
                                                  C.byref(sec), C.byref(ntok)))
        calls = n * int(rounds)
        return {"sentences_per_s": round(calls / sec.value, 1) if sec.value > 0 else None, "us_per_call": round(sec.value / max(calls, 1) * 1e6, 3),
                "sentences": n, "rounds": int(rounds), "tokens": int(ntok.value)}

    def init_connid_counter(self):
        """Worker::init_connid_counter (worker.rs:77-84)."""
        N.check(N.lib().vbt_worker_init_connid_counter(self._h))

    def update_connid_counts(self):
        """Worker::update_connid_counts (worker.rs:86-93)."""
        N.check(N.lib().vbt_worker_update_connid_counts(self._h))

    def connid_counts(self):
        d = self.tokenizer.dictionary()
        lid = np.zeros(d.num_left, dtype=np.uint64)
        rid = np.zeros(d.num_right, dtype=np.uint64)
        N.check(N.lib().vbt_worker_connid_counts(self._h, lid.ctypes.data, rid.ctypes.data))
        return lid, rid

    def compute_connid_probs(self):
        """Worker::compute_connid_probs (worker.rs:95-103): ([(left_id, prob)], [(right_id, prob)])."""
        d = self.tokenizer.dictionary()
        nl, nr = d.num_left - 1, d.num_right - 1
        li, lp = np.zeros(nl, dtype=np.uint32), np.zeros(nl, dtype=np.float64)
        ri, rp = np.zeros(nr, dtype=np.uint32), np.zeros(nr, dtype=np.float64)
        N.check(N.lib().vbt_worker_compute_connid_probs(self._h, li.ctypes.data, lp.ctypes.data, ri.ctypes.data, rp.ctypes.data))
        return list(zip(li.tolist(), lp.tolist())), list(zip(ri.tolist(), rp.tolist()))


class Batch:
    """Results of Tokenizer.tokenize_batch (host copy)."""

    def __init__(self, tokenizer, handle):
        self.tokenizer = tokenizer
        self._h = handle

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                N.lib().vbt_batch_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown: module globals may be gone
            pass

    def __len__(self):
        return N.lib().vbt_batch_num_sentences(self._h)

    def total_tokens(self):
        return N.lib().vbt_batch_total_tokens(self._h)

    def num_tokens(self, s):
        return N.lib().vbt_batch_num_tokens(self._h, s)

    def token(self, s, i):
        t = N.Token()
        N.check(N.lib().vbt_batch_token(self._h, s, i, C.byref(t)))
        return Token(t)

    def records(self, s):
        """Token records of sentence s as a structured numpy array (copy)."""
        n = self.num_tokens(s)
        if n == 0:
            return np.zeros(0, dtype=TOKEN_DTYPE)
        p = N.lib().vbt_batch_records(self._h, s)
        return np.frombuffer(C.string_at(p, n * TOKEN_DTYPE.itemsize), dtype=TOKEN_DTYPE).copy()

    def arrays(self):
        """(tokens[TOKEN_DTYPE], tok_off[u32], tok_cnt[u32]) copies for the whole batch."""
        pt, po, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(N.lib().vbt_batch_arrays(self._h, C.byref(pt), C.byref(po), C.byref(pc)))
        n, t = len(self), self.total_tokens()
        toks = np.frombuffer(C.string_at(pt, t * TOKEN_DTYPE.itemsize), dtype=TOKEN_DTYPE).copy() if t else np.zeros(0, TOKEN_DTYPE)
        off = np.frombuffer(C.string_at(po, n * 4), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32)
        cnt = np.frombuffer(C.string_at(pc, n * 4), dtype=np.uint32).copy() if n else np.zeros(0, np.uint32)
        return toks, off, cnt

    def tokens_in_order(self):
        """All token records concatenated in sentence order + exclusive offsets (n+1)."""
        toks, off, cnt = self.arrays()
        ends = np.zeros(len(cnt) + 1, dtype=np.uint64)
        ends[1:] = np.cumsum(cnt, dtype=np.uint64)
        idx = np.repeat(off.astype(np.int64) - ends[:-1].astype(np.int64), cnt) + np.arange(int(ends[-1]), dtype=np.int64)
        return toks[idx], ends

    def format(self, mode="mecab"):
        """Byte-identical `tokenize` CLI output (tokenize/src/main.rs:83-127)."""
        m = {"mecab": 0, "wakati": 1, "detail": 2}[mode]
        p, n = C.c_void_p(), C.c_size_t()
        N.check(N.lib().vbt_batch_format(self._h, m, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value).decode("utf-8")
        finally:
            N.lib().vbt_free(p)


def _batch_format_bytes(self, mode="mecab"):
    """The same as bytes, plus the seconds the C call took (no UTF-8 decode, one copy out of the library's buffer)."""
    import time
    m = {"mecab": 0, "wakati": 1, "detail": 2}[mode]
    p, n = C.c_void_p(), C.c_size_t()
    t0 = time.perf_counter()
    N.check(N.lib().vbt_batch_format(self._h, m, C.byref(p), C.byref(n)))
    dt = time.perf_counter() - t0
    try:
        return C.string_at(p, n.value), dt
    finally:
        N.lib().vbt_free(p)


Batch.format_bytes = _batch_format_bytes


def _batch_format_into(self, write, mode="mecab"):
    """Renders the batch and hands the library's buffer to `write` (e.g. a binary file's write) as a memoryview: no copy into a
    Python object.  Returns the number of bytes."""
    m = {"mecab": 0, "wakati": 1, "detail": 2}[mode]
    p, n = C.c_void_p(), C.c_size_t()
    N.check(N.lib().vbt_batch_format(self._h, m, C.byref(p), C.byref(n)))
    try:
        if n.value:
            write(memoryview((C.c_char * n.value).from_address(p.value)))
        return n.value
    finally:
        N.lib().vbt_free(p)


Batch.format_into = _batch_format_into


def compute_connid_probs(lid_count, rid_count):
    """ConnIdCounter::compute_probs (mapper.rs:108-146): per side, (id, count / sum) without id 0, sorted by
    probability descending then id ascending -- the content of the reference's *.lmap / *.rmap files."""
    out = []
    for cnt in (lid_count, rid_count):
        cnt = np.ascontiguousarray(cnt, dtype=np.uint64)
        ids, probs = np.zeros(len(cnt) - 1, dtype=np.uint32), np.zeros(len(cnt) - 1, dtype=np.float64)
        N.check(N.lib().vbt_connid_probs(cnt.ctypes.data, len(cnt), ids.ctypes.data, probs.ctypes.data))
        out.append(list(zip(ids.tolist(), probs.tolist())))
    return out[0], out[1]


def utf8_valid(data):
    data = _b(data)
    return bool(N.lib().vbt_utf8_valid(data, len(data)))


class Workspace:
    """Device-resident batch interface (zero host copies): what bench.py times."""

    def __init__(self, tokenizer, max_sentences, max_bytes):
        self.tokenizer = tokenizer
        h = C.c_void_p()
        N.check(N.lib().vbt_workspace_new(tokenizer._handle(), max_sentences, max_bytes, C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                N.lib().vbt_workspace_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown: module globals may be gone
            pass

    def set_timing(self, on=True):
        N.check(N.lib().vbt_workspace_set_timing(self._h, int(on)))

    def run(self, d_text_ptr, d_offsets_ptr, n, total_bytes, stream=0):
        """Enqueue on `stream` (raw hipStream_t as int). Pointers are raw device addresses."""
        N.check(N.lib().vbt_tokenize_batch_device(self._h, d_text_ptr, d_offsets_ptr, n, total_bytes, stream or None))

    def set_packed_output(self, slot_ptr, slot_bytes, max_sentences):
        """Results of every later run() go straight into the caller's device buffer `slot_ptr`, laid out as a rank's slot of the
        final gather (sharding.packed_bytes / unpack_results): no copy kernels between the tokenizer and the collective.
        slot_ptr=None: back to the workspace's own buffers."""
        N.check(N.lib().vbt_workspace_set_packed_output(self._h, slot_ptr or None, int(slot_bytes), int(max_sentences)))

    def result_ptrs(self):
        p = [C.c_void_p() for _ in range(4)]
        N.check(N.lib().vbt_workspace_results(self._h, *[C.byref(x) for x in p]))
        return {"tokens": p[0].value, "tok_off": p[1].value, "tok_cnt": p[2].value, "total": p[3].value}

    def count_connids(self, on=True):
        """Worker::init_connid_counter (worker.rs:77-84): start accumulating connection-id usage."""
        N.check(N.lib().vbt_workspace_count_connids(self._h, int(on)))

    def connid_counts(self, reset=False):
        """(lid_count[num_left], rid_count[num_right]) accumulated by Worker::update_connid_counts."""
        d = self.tokenizer.dictionary()
        lid = np.zeros(d.num_left, dtype=np.uint64)
        rid = np.zeros(d.num_right, dtype=np.uint64)
        N.check(N.lib().vbt_workspace_connid_counts(self._h, lid.ctypes.data, rid.ctypes.data, int(reset)))
        return lid, rid

    PHASES = ("decode", "count", "fill", "end_lists", "prepass", "pass_records", "recurrence", "emit")

    def profile(self, reset=True):
        """Per-phase cycle totals (needs VBT_PROFILE=1 at workspace creation)."""
        out = (C.c_uint64 * 12)()
        N.check(N.lib().vbt_workspace_profile(self._h, out, int(reset)))
        d = dict(zip(self.PHASES, [int(x) for x in out[:8]]))
        d["sentences"] = int(out[8])
        d["counts"] = {"steps": int(out[9]), "passes": int(out[10]), "candidates": int(out[11])}
        return d

    def stats(self):
        st = N.CallStats()
        N.check(N.lib().vbt_workspace_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in N.CallStats._fields_}
