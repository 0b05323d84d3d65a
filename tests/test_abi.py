"""CPU: the C-ABI library loads and exports every symbol include/lip.h declares (no compute calls)."""
import ctypes
import os
import re

from lip_amd import _native as nv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "lip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = nv.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"liblip_hip.so does not export {n}"
        assert n in nv.SIGNATURES, f"ctypes binding lacks a signature for {n}"
    assert lib.lip_abi_version() == 8
    assert lib.lip_sizeof_op() == ctypes.sizeof(nv.Op)


def test_errors_are_status_codes_not_crashes():
    lib = nv.load()
    assert lib.lip_bdot(None, None, None, 0, 0, None) != 0
    assert b"bad argument" in lib.lip_last_error()
    h = ctypes.c_void_p()
    assert lib.lip_engine_create(ctypes.byref(h), 0, 0, 0) != 0
    assert lib.lip_engine_create(ctypes.byref(h), 10, 2, 3) == 0
    assert lib.lip_ggn_vp(h, None, None, 1, 1.0, 0.0, None) != 0          # not bound
    assert b"not bound" in lib.lip_last_error()
    assert lib.lip_engine_destroy(h) == 0
