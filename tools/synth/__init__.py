"""Synthetic dictionaries / sentence batches; ctypes wrapper of synth.c.

Neutral test/bench infrastructure: produces byte-identical inputs for the CPU oracle
and for the HIP product.  Real ipadic/unidic dictionaries are not available offline.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvbt_synth.so")

SEED = 20260925

# name -> (n_words, num_right, num_left)
SHAPES = {
    "tiny": (3000, 40, 36),
    "small": (30000, 300, 280),
    "ipadic": (392126, 1316, 1316),       # ipadic-mecab-2.7.0 shape
    "unidic": (876803, 15626, 15388),     # unidic-cwj-3.1.1 shape (458.6 MiB matrix)
}
# lexicon laws (dup_p, len_lambda, kanji_zipf, n_kanji); None = SURVEY.md 8(d)'s default law
# measured with the oracle: unidic-dense 13.3 nodes / 147 deduplicated pairs per char, small-dense 13.7 / 127 (default law: 5.6 / 29)
LAWS = {"unidic-dense": (0.38, 1.75, 0.8, 2500), "small-dense": (0.65, 1.1, 0.9, 300)}
SHAPES["unidic-dense"] = SHAPES["unidic"]
SHAPES["small-dense"] = SHAPES["small"]
LEN_LAWS = {"uniform_5_20": 0, "lognormal_40": 1, "mixed": 2}


def build(force=False):
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-fvisibility=hidden",
                               "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.syn_dict_new.restype = C.c_void_p
        L.syn_dict_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
        L.syn_dict_new_law.restype = C.c_void_p
        L.syn_dict_new_law.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_uint32]
        L.syn_dict_free.argtypes = [C.c_void_p]
        for f in ["syn_dict_lex", "syn_dict_unk"]:
            getattr(L, f).restype = C.POINTER(C.c_char)
            getattr(L, f).argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.syn_char_def.restype = C.POINTER(C.c_char)
        L.syn_char_def.argtypes = [C.POINTER(C.c_size_t)]
        L.syn_fill_matrix.argtypes = [C.c_void_p, C.c_void_p]
        L.syn_user_csv.restype = C.c_void_p
        L.syn_user_csv.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_size_t)]
        L.syn_sentences.restype = C.c_void_p
        L.syn_sentences.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_void_p,
                                    C.POINTER(C.c_size_t)]
        L.syn_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class SynthDict:
    """Text sources + binary matrix of one synthetic dictionary."""

    def __init__(self, shape="ipadic", seed=SEED, law=None):
        if isinstance(shape, str):
            self.name = "syn-" + shape
            law = law or LAWS.get(shape)
            shape = SHAPES[shape]
        else:
            self.name = "syn-custom"
        self.n_words, self.num_right, self.num_left = shape
        self.seed = seed
        L = lib()
        self.law = law
        if law is None:
            self._h = L.syn_dict_new(self.n_words, self.num_right, self.num_left, seed)
        else:
            self._h = L.syn_dict_new_law(self.n_words, self.num_right, self.num_left, seed, *law)
        n = C.c_size_t()
        self.lex = C.string_at(L.syn_dict_lex(self._h, C.byref(n)), n.value)
        self.unk = C.string_at(L.syn_dict_unk(self._h, C.byref(n)), n.value)
        self.char_def = C.string_at(L.syn_char_def(C.byref(n)), n.value)
        self._matrix = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().syn_dict_free(self._h)
                self._h = None
        except Exception:  # interpreter teardown
            pass

    @property
    def matrix(self):
        """np.int16 [num_left, num_right]: data[left*num_right+right] (matrix_connector.rs:47)."""
        if self._matrix is None:
            m = np.empty((self.num_left, self.num_right), dtype=np.int16)
            lib().syn_fill_matrix(self._h, m.ctypes.data)
            self._matrix = m
        return self._matrix

    def matrix_def_text(self):
        """matrix.def text form (only sensible for small shapes)."""
        m = self.matrix
        rows = [f"{self.num_right} {self.num_left}"]
        for r in range(self.num_right):
            for l in range(self.num_left):
                rows.append(f"{r} {l} {int(m[l, r])}")
        return ("\n".join(rows) + "\n").encode()

    def user_csv(self, n=1000, seed=None):
        ln = C.c_size_t()
        p = lib().syn_user_csv(self._h, n, self.seed if seed is None else seed, C.byref(ln))
        out = C.string_at(p, ln.value)
        lib().syn_free(p)
        return out

    def sentences(self, n, law="lognormal_40", space_p=0.0, seed=None):
        """Returns (text np.uint8[total_bytes], offsets np.uint64[n+1])."""
        offs = np.zeros(n + 1, dtype=np.uint64)
        ln = C.c_size_t()
        p = lib().syn_sentences(self._h, n, self.seed if seed is None else seed, LEN_LAWS[law], float(space_p),
                                offs.ctypes.data, C.byref(ln))
        text = np.frombuffer(C.string_at(p, ln.value), dtype=np.uint8).copy()
        lib().syn_free(p)
        return text, offs
