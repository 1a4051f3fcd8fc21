"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/p2phd.h declares, the ctypes table matches the header, and nothing in the product package
imports the oracle or falls back to the CPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "p2phd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p2phd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from pix2pixhdaudiosr_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 8
    assert sorted(_lib.SIGNATURES.keys()) == syms
    assert os.path.isfile(_lib.LIB_PATH), "build the extension first: __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), s
    assert _lib.lib().p2phd_abi_version() >= 1
    # the fp16 build (the same sources, 16-bit type = IEEE half): the same exports, told apart by p2phd_half_type()
    assert os.path.isfile(_lib.LIB_PATH_F16), "build the extension first: __graft_entry__.build()"
    L16 = ctypes.CDLL(_lib.LIB_PATH_F16)
    for s in syms:
        assert hasattr(L16, s), s
    assert (_lib.lib().p2phd_half_type(), _lib.lib("f16").p2phd_half_type()) == (1, 2)


def test_host_only_entry_points():
    """Frame layout and table fill are host arithmetic: usable without a GPU."""
    import numpy as np
    from pix2pixhdaudiosr_amd import _lib
    from pix2pixhdaudiosr_amd.models.mdct import frame_layout
    g = np.load(os.path.join(ROOT, "tests", "golden", "mdct4.npz"))
    for B, T, frames in g["quirk_frames_n1024"]:
        assert frame_layout(int(B), int(T), 512, 1024, True)[2] == int(frames)
    L = _lib.lib()
    n = L.p2phd_mdct4_tables_floats(1024)
    buf = np.zeros(n, dtype=np.float32)
    _lib.check(L.p2phd_mdct4_tables_fill(1024, ctypes.c_void_p(buf.ctypes.data)))
    tw = buf[:512].view(np.complex64)
    assert np.allclose(tw, np.exp(-2j * np.pi * np.arange(256) / 256), atol=1e-7)
    assert L.p2phd_mdct4_tables_fill(100, ctypes.c_void_p(buf.ctypes.data)) != 0
    assert b"n_fft" in L.p2phd_last_error()


def test_product_never_imports_oracle_or_reference():
    pkg = os.path.join(ROOT, "pix2pixhdaudiosr_amd")
    bad = []
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                s = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) or "/root/reference" in s:
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_missing_gpu_raises_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pix2pixhdaudiosr_amd.models.mdct import MDCT4
    from pix2pixhdaudiosr_amd.util.util import kbdwin
    with pytest.raises(Exception):
        MDCT4(n_fft=64, hop_length=32, win_length=64, window=kbdwin, device="cpu")(torch.zeros(2, 96))
