"""CPU: librxgpu.so loads, exports every symbol include/rxgpu.h declares, has no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rx_tools_amd as R
from support import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rxgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rxgpu_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = R.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "include/rxgpu.h declares %s but librxgpu.so does not export it" % n
    assert sorted(R._lib.EXPORTS) == names


def test_struct_sizes_match_header():
    """the C structs the library is compiled against == the ctypes mirrors the tests use"""
    from rx_tools_amd.structs import DemodState, SIZEOF_DEMOD_STATE
    assert C.sizeof(DemodState) == SIZEOF_DEMOD_STATE
    assert C.sizeof(R.FmCarry) == 324 and C.sizeof(R.FmParams) == 72
    assert C.sizeof(R.PowerParams) == 28


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = R.lib()
    assert L.rxgpu_device_count() == 0
    assert L.rxgpu_init(-1) == -1
    assert b"no HIP device" in L.rxgpu_last_error()
    with pytest.raises(R.RxGpuError):
        R.FmStream(R.FmParams.wbfm(), 1, 16384)
    with pytest.raises(R.RxGpuError):
        R.PowerScan(R.PowerParams(12, 16384, 1, 0, 1, 0, 0), 1, np.ones(4096, np.int32), np.zeros(3072, np.int16))


def test_product_never_touches_oracle():
    """nothing under rx_tools_amd/ or include/ may import, link or mention oracle/"""
    for base in ("rx_tools_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath or "__pycache__" in dirpath:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".pyc")):
                    continue
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("rx_oracle", "librxoracle", "libref_", "import support", "from support", "rxo_"):
                    assert needle not in text, "%s/%s refers to the oracle (%s)" % (dirpath, f, needle)
