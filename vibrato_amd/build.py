"""Builds vibrato_amd/lib/libvibrato_hip.so in-tree with hipcc for gfx950 (MI355X only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libvibrato_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["dict.cpp", "connector.cpp", "dictio.cpp", "engine.hip", "capi.cpp"]
HEADERS = ["dict.hpp", "engine.hpp", "../../include/vibrato_hip.h"]


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
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-result", "-x", "hip"]
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    cmd += ["-ldl", "-lhsa-runtime64", "-o", so]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(SO)
