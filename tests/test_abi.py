"""The C-ABI library loads (no GPU needed) and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'deepinteraction_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(di_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from deepinteraction_amd import _lib, build
    build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/deepinteraction_hip.h but not exported'


def test_binding_covers_every_compute_symbol():
    from deepinteraction_amd import _lib
    declared = set(_declared()) - {'di_abi_version', 'di_last_error'}
    bound = set(_lib.SIGNATURES) | set(_lib.VALUE_FUNCS)
    assert declared == bound, declared ^ bound
    assert _lib.lib().di_abi_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch):
    from deepinteraction_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdeepinteraction_hip.so')
    with pytest.raises(_lib.HipLibraryError):
        _lib.lib()


def test_bad_arguments_return_error_codes_not_crashes():
    from deepinteraction_amd import _lib
    L = _lib.lib()
    rc = L.di_local_attn_fwd(None, None, None, None, 1, 8, 8, 100, 9, 9, 1.0, 0, None)   # C not /8
    assert rc == -1 and b'C=100' in L.di_last_error()
    rc = L.di_local_attn_fwd(None, None, None, None, 1, 8, 8, 128, 4, 4, 1.0, 0, None)   # even window
    assert rc == -1 and b'window' in L.di_last_error()
