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
