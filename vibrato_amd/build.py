"""Builds vibrato_amd/lib/libvibrato_hip.so in-tree with hipcc for gfx950 (MI355X only): one object per translation unit,
compiled in parallel, then linked."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libvibrato_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# host code, then the kernels: gen.hip (validate_batch, generators, work lists), lattice.hip (the sweep, the resident Worker
# kernel), fused.hip (the single-kernel fallback), pack.hip (token compaction, connector expansion), engine.hip (Tokenizer, Workspace)
SOURCES = ["dict.cpp", "connector.cpp", "dictio.cpp", "capi.cpp", "engine.hip", "gen.hip", "lattice.hip", "fused.hip", "pack.hip"]
# every header under csrc/ (sweep_asm.hpp -- the hottest code -- was once missing from a hand-written list) + the C ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + ["../../include/vibrato_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def variant_so(variant):
    return os.path.join(LIBDIR, f"libvibrato_hip{('_' + variant) if variant else ''}.so")


def build(force=False, verbose=False, variant="", defines=()):
    """variant/defines: developer A/B builds (lib/libvibrato_hip_<variant>.so with -D flags),
    selected at load time with VBT_LIB_VARIANT=<variant>."""
    so = variant_so(variant)
    if not variant and not force and not needs_build():
        return so
    os.makedirs(LIBDIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="vbt_build_") as tmp:
        def compile_one(f):
            obj = os.path.join(tmp, f + ".o")
            cmd = [HIPCC] + FLAGS + ["-D" + d for d in defines] + ["-x", "hip", "-c", os.path.join(CSRC, f), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, SOURCES))
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-lhsa-runtime64", "-o", so]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(SO)
