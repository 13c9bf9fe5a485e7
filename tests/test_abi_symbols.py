"""The C-ABI libraries load without a GPU and export every symbol that include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_pq2hip_exports_every_declared_symbol():
    from parquet2_amd import _lib

    names = declared("pq2hip.h", "pq2_")
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    # the Python binding covers the same set
    assert sorted(_lib.SIGNATURES) == names


def test_rust_sys_crate_declares_every_export():
    """rust/pq2hip-sys/src/lib.rs (sources only: no rustc here) has an extern "C" item for every export of the
    header, with as many parameters, and nothing the header does not have."""
    header = open(os.path.join(ROOT, "include", "pq2hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    want = {name: 0 if args.strip() in ("", "void") else args.count(",") + 1
            for name, args in re.findall(r"\b(pq2_\w+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S)}
    rust = open(os.path.join(ROOT, "rust", "pq2hip-sys", "src", "lib.rs")).read()
    rust = re.sub(r"//[^\n]*", "", rust)
    have = {name: 0 if not args.strip() else args.count(":")
            for name, args in re.findall(r"pub fn (pq2_\w+)\s*\(([^;]*?)\)\s*(?:->[^;]*)?;", rust, flags=re.S)}
    assert sorted(have) == sorted(want)
    assert have == want
    # the structs of the header, field for field
    for struct, fields in re.findall(r"typedef struct (pq2_\w+) \{(.*?)\}", header, flags=re.S):
        names = [n for decl in fields.split(";") for n in re.findall(r"(\w+)\s*(?:,|$)", decl.strip())]
        body = re.search(r"pub struct %s \{(.*?)\}" % struct, rust, flags=re.S)
        assert body, struct
        assert re.findall(r"pub (\w+):", body.group(1)) == names, struct


def test_pq2synth_exports_every_declared_symbol():
    from parquet2_amd import synth

    names = declared("pq2synth.h", "pq2s_")
    handle = ctypes.CDLL(synth.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    assert sorted(synth.SIGNATURES) == names


def test_no_gpu_means_loud_failure():
    """Without a usable device the context cannot be created: there is no CPU decode path."""
    import pytest
    import torch

    from parquet2_amd import Context
    from parquet2_amd.error import DeviceError

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(DeviceError):
        Context(0)
