"""The C-ABI shared library loads without a GPU and exports every symbol that
include/egonet_hip.h declares; the ctypes signature table covers them all."""
import ctypes as C
import os
import re

from egonet_amd import _lib


def _declared():
    src = open(_lib.HEADER_PATH).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(egn_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 25
    assert os.path.isfile(_lib.LIB_PATH), 'build it: python -m egonet_amd.build'
    handle = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_strings():
    L = _lib.lib()
    assert L.egn_version() >> 16 == 1
    assert L.egn_strerror(0) == b'ok'
    assert b'bad argument' in L.egn_strerror(-1)
    assert L.egn_conv_num_configs() >= 8
    tm, tn = C.c_int(), C.c_int()
    assert L.egn_conv_config_info(1, C.byref(tm), C.byref(tn)) == 0 and (tm.value, tn.value) == (256, 48)
    assert L.egn_conv_config_info(99, C.byref(tm), C.byref(tn)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    try:
        _lib.lib()
    except _lib.EgonetHipError as e:
        assert 'no CPU/torch fallback' in str(e)
    else:
        raise AssertionError('expected EgonetHipError')
