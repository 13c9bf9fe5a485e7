"""Builds libpq2hip.so (HIP kernels for gfx950 + the C ABI) in-tree with hipcc.

Every source is compiled to an object of its own (in parallel, and only when it or a header changed), then the
objects are linked: the two large kernel files take minutes each, the rest seconds."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libpq2hip.so")
SOURCES = ["pq2_kernels.hip", "pq2_ring.hip", "pq2_ring_dict.hip", "pq2_scout.hip", "pq2_density.hip", "pq2_optional.hip", "pq2_noalloc.hip", "pq2_api.cpp", "pq2_host.cpp"]
HEADERS = ["pq2_device.h", "pq2_ring.h", os.path.join("..", "..", "include", "pq2hip.h")]
DEPS = SOURCES + HEADERS


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


def build(force=False, verbose=False, extra_flags=(), lib=LIB, obj_dir=OBJ_DIR, only=None):
    """`lib` / `obj_dir` / `extra_flags`: A/B builds (tools/build_variant.sh) next to the product library.
    `only`: the sources an A/B build compiles with its flags; the others are linked as the product build has them."""
    build_synth(force, verbose)
    if not force and lib == LIB and up_to_date():
        return lib
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall"] + list(extra_flags)

    def compile_one(src):
        path = os.path.join(CSRC, src)
        if only is not None and src not in only:
            return os.path.join(OBJ_DIR, src + ".o")
        obj = os.path.join(obj_dir, src + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), newest_header):
            return obj
        cmd = [hipcc()] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    name = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")), None)
    only = next((a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--only=")), None)
    if name:  # python -m parquet2_amd.build --variant=<name> [--only=a.hip,b.hip] -DFLAG...: lib/variants/libpq2hip_<name>.so
        print(build(force=True, verbose=True, extra_flags=flags,
                    lib=os.path.join(LIB_DIR, "variants", "libpq2hip_%s.so" % name),
                    obj_dir=os.path.join(LIB_DIR, "obj_" + name), only=only))
    else:
        print(build(force="--force" in sys.argv, verbose=True, extra_flags=flags))
