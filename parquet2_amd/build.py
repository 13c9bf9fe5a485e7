"""Builds libpq2hip.so (HIP kernels for gfx950 + the C ABI) in-tree with hipcc."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libpq2hip.so")
SOURCES = ["pq2_kernels.hip", "pq2_noalloc.hip", "pq2_api.cpp", "pq2_host.cpp"]
DEPS = SOURCES + ["pq2_device.h", os.path.join("..", "..", "include", "pq2hip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


SYNTH_SRC = os.path.join(PKG, "synth", "pq2_synth.cpp")
SYNTH_LIB = os.path.join(LIB_DIR, "libpq2synth.so")


def build_synth(force=False, verbose=False):
    """Host-only generator of synthetic pages (tests / bench input)."""
    if not force and os.path.exists(SYNTH_LIB) and os.path.getmtime(SYNTH_SRC) <= os.path.getmtime(SYNTH_LIB):
        return SYNTH_LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", SYNTH_LIB, SYNTH_SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SYNTH_LIB


def build(force=False, verbose=False, extra_flags=()):
    build_synth(force, verbose)
    if not force and up_to_date():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-o", LIB] + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
