"""vibrato_amd: MI355X (gfx950) native batched Viterbi tokenizer, a drop-in for the
tokenize() path of daac-tools/vibrato.  The compute lives in csrc/ (HIP kernels + C ABI,
built into lib/libvibrato_hip.so); this package is the thin host mirror of vibrato's
Dictionary / Tokenizer / Worker / Token API."""
from .api import (Batch, Dictionary, SystemDictionaryBuilder, Token, Tokenizer, VibratoError, Worker,  # noqa: F401
                  Workspace, TOKEN_DTYPE, LEX_NAMES, compute_connid_probs)

__all__ = ["Batch", "Dictionary", "SystemDictionaryBuilder", "Token", "Tokenizer", "VibratoError", "Worker",
           "Workspace", "TOKEN_DTYPE", "LEX_NAMES", "compute_connid_probs"]
