"""ctypes binding of libvibrato_hip.so (the C ABI declared in include/vibrato_hip.h).

There is no Python or CPU fallback: if the shared library is missing, or no gfx950
device is present when a Tokenizer is created, the call fails loudly.
"""
import ctypes as C
import os

from . import build as _build

_lib = None

STATUS_NAMES = {1: "InvalidArgument", 2: "InvalidFormat", 3: "InvalidState", 4: "ParseInt", 5: "Utf8",
                100: "Device", 101: "Unsupported"}


class VibratoError(Exception):
    """Mirror of vibrato::errors::VibratoError (errors.rs:7-42) + device errors."""

    def __init__(self, code, message):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class TokenRec(C.Structure):
    _fields_ = [("start_char", C.c_uint32), ("end_char", C.c_uint32), ("start_byte", C.c_uint32),
                ("end_byte", C.c_uint32), ("word_idx", C.c_uint32), ("total_cost", C.c_int32)]


class Token(C.Structure):
    _fields_ = [("surface", C.c_void_p), ("surface_len", C.c_size_t), ("feature", C.c_void_p), ("feature_len", C.c_size_t),
                ("start_char", C.c_uint32), ("end_char", C.c_uint32), ("start_byte", C.c_uint32), ("end_byte", C.c_uint32),
                ("lex_type", C.c_uint32), ("word_id", C.c_uint32), ("left_id", C.c_uint16), ("right_id", C.c_uint16),
                ("word_cost", C.c_int16), ("total_cost", C.c_int32)]


class CallStats(C.Structure):
    _fields_ = [("n_sentences", C.c_uint64), ("n_tier0", C.c_uint64), ("n_tier1", C.c_uint64), ("n_tier2", C.c_uint64),
                ("n_tokens", C.c_uint64), ("error_flags", C.c_uint32), ("ms_tier0", C.c_float), ("ms_tier12", C.c_float), ("ms_pack", C.c_float)]


# name -> (restype, argtypes); also the list of symbols include/vibrato_hip.h declares.
_vp, _cp, _sz, _u32, _u64, _int = C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "vbt_last_error": (C.c_char_p, []),
    "vbt_utf8_valid": (_int, [_cp, _sz]),
    "vbt_dict_from_sources": (_int, [_cp, _sz, _cp, _sz, _cp, _sz, _cp, _sz, _PP]),
    "vbt_dict_from_sources_binmatrix": (_int, [_cp, _sz, _vp, _u32, _u32, _cp, _sz, _cp, _sz, _PP]),
    "vbt_dict_from_sources_bigram": (_int, [_cp, _sz, _cp, _sz, _cp, _sz, _cp, _sz, _cp, _sz, _cp, _sz, _int, _PP]),
    "vbt_dict_connector_kind": (_int, [_vp]),
    "vbt_dict_read": (_int, [_vp, _sz, _PP]),
    "vbt_dict_write": (_int, [_vp, _int, _PP, C.POINTER(_sz)]),
    "vbt_dict_set_user_lexicon": (_int, [_vp, _cp, _sz]),
    "vbt_dict_map_connection_ids": (_int, [_vp, _vp, _sz, _vp, _sz]),
    "vbt_dict_free": (None, [_vp]),
    "vbt_dict_num_words": (_u32, [_vp, _u32]),
    "vbt_dict_num_left": (_u32, [_vp]),
    "vbt_dict_num_right": (_u32, [_vp]),
    "vbt_dict_word_feature": (_int, [_vp, _u32, _u32, _PP, C.POINTER(_sz)]),
    "vbt_dict_word_param": (_int, [_vp, _u32, _u32, C.POINTER(C.c_int32)]),
    "vbt_dict_conn_cost": (_int, [_vp, _u32, _u32, C.POINTER(C.c_int32)]),
    "vbt_dict_char_info": (_u32, [_vp, _u32]),
    "vbt_dict_cate_id": (_int, [_vp, _cp, _sz]),
    "vbt_dict_common_prefix": (_u32, [_vp, _u32, _vp, _u32, _vp, _u32]),
    "vbt_tokenizer_new": (_int, [_vp, _int, _u32, _int, _PP]),
    "vbt_tokenizer_new_multi": (_int, [_vp, _int, _u32, C.POINTER(C.c_int), _u32, _PP]),
    "vbt_tokenizer_num_devices": (_u32, [_vp]),
    "vbt_tokenizer_connid_reorder_info": (_int, [_vp, C.POINTER(_u64)]),
    "vbt_tokenizer_calibrate": (_int, [_vp, _vp, _vp, _u64]),
    "vbt_tokenizer_connid_reorder_wait": (_int, [_vp, C.c_int64, C.POINTER(C.c_int)]),
    "vbt_tokenizer_lattice_density": (_int, [_vp, C.POINTER(C.c_double)]),
    "vbt_tokenizer_free": (None, [_vp]),
    "vbt_tokenizer_dictionary": (_vp, [_vp]),
    "vbt_tokenizer_trim_pool": (_int, [_vp]),
    "vbt_worker_new": (_int, [_vp, _PP]),
    "vbt_worker_free": (None, [_vp]),
    "vbt_worker_reset_sentence": (_int, [_vp, _cp, _sz]),
    "vbt_worker_tokenize": (_int, [_vp]),
    "vbt_worker_num_tokens": (_u32, [_vp]),
    "vbt_worker_token": (_int, [_vp, _u32, C.POINTER(Token)]),
    "vbt_worker_path_stats": (_int, [_vp, _vp, _vp]),
    "vbt_worker_loop_benchmark": (_int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    "vbt_worker_init_connid_counter": (_int, [_vp]),
    "vbt_worker_update_connid_counts": (_int, [_vp]),
    "vbt_worker_connid_counts": (_int, [_vp, _vp, _vp]),
    "vbt_worker_compute_connid_probs": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "vbt_connid_probs": (_int, [_vp, _sz, _vp, _vp]),
    "vbt_tokenize_batch": (_int, [_vp, _vp, _vp, _u64, _PP]),
    "vbt_tokenizer_pool_stats": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "vbt_batch_free": (None, [_vp]),
    "vbt_batch_num_sentences": (_u64, [_vp]),
    "vbt_batch_total_tokens": (_u64, [_vp]),
    "vbt_batch_num_tokens": (_u32, [_vp, _u64]),
    "vbt_batch_token": (_int, [_vp, _u64, _u32, C.POINTER(Token)]),
    "vbt_batch_records": (C.POINTER(TokenRec), [_vp, _u64]),
    "vbt_batch_arrays": (_int, [_vp, _PP, _PP, _PP]),
    "vbt_batch_format": (_int, [_vp, _int, _PP, C.POINTER(_sz)]),
    "vbt_free": (None, [_vp]),
    "vbt_workspace_new": (_int, [_vp, _u64, _u64, _PP]),
    "vbt_workspace_free": (None, [_vp]),
    "vbt_tokenize_batch_device": (_int, [_vp, _vp, _vp, _u64, _u64, _vp]),
    "vbt_workspace_results": (_int, [_vp, _PP, _PP, _PP, _PP]),
    "vbt_workspace_set_packed_output": (_int, [_vp, _vp, _u64, _u64]),
    "vbt_workspace_set_timing": (_int, [_vp, _int]),
    "vbt_workspace_count_connids": (_int, [_vp, _int]),
    "vbt_workspace_connid_counts": (_int, [_vp, _vp, _vp, _int]),
    "vbt_workspace_profile": (_int, [_vp, C.POINTER(C.c_uint64), _int]),
    "vbt_workspace_stats": (_int, [_vp, C.POINTER(CallStats)]),
}


def lib():
    """Loads (building first if sources are newer) the HIP shared library. Never falls back."""
    global _lib
    if _lib is None:
        # The lattice kernel is launched once per LDS tier on its own stream; the ROCm runtime maps
        # streams onto 4 hardware queues by default, which serialises half of them.  Must be set
        # before the HIP runtime initialises (harmless if it already has).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process
        # cannot both own the GPU.  Importing torch first makes the loader resolve our NEEDED
        # libamdhip64.so.7 to the copy torch already mapped (plumbing only: no torch compute here).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        variant = os.environ.get("VBT_LIB_VARIANT", "")
        so = _build.variant_so(variant)
        if variant:
            if not os.path.exists(so):
                raise ImportError(f"{so} (VBT_LIB_VARIANT={variant}) has not been built")
        elif _build.needs_build():
            if os.path.exists(_build.HIPCC):
                _build.build()
            elif not os.path.exists(so):
                raise ImportError(f"{so} is missing and hipcc is not available to build it; "
                                  "vibrato_amd has no CPU fallback")
        L = C.CDLL(so)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise VibratoError(status, lib().vbt_last_error().decode("utf-8", "replace"))
