"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/whisprrec_hip.h declares, the ctypes table mirrors the header, and argument errors are reported through
the documented convention (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from whisprrec_amd import abi

HEADER = os.path.join(ROOT, "include", "whisprrec_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int32_t|int64_t|void|const char \*)\s*(wr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        nargs = 0 if args in ("", "void") else len(args.split(","))
        decls[m.group(2)] = (m.group(1), nargs)
    return decls


def test_header_and_binding_agree():
    decls = _declared()
    assert len(decls) >= 20
    assert set(decls) == set(abi.SIGNATURES), set(decls) ^ set(abi.SIGNATURES)
    for name, (ret, nargs) in decls.items():
        res, args = abi.SIGNATURES[name]
        assert len(args) == nargs, name
        assert {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "const char *": ctypes.c_char_p, "void": None}[ret] is res, name


def test_library_exports_every_declared_symbol():
    assert os.path.exists(abi.LIB_PATH), "run __graft_entry__.build() first"
    raw = ctypes.CDLL(abi.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), name
    assert abi.lib().wr_abi_version() == 1


def test_argument_errors_use_the_error_convention():
    L = abi.lib()
    # NULL table -> WR_E_NULL (-1), message available from wr_last_error()
    rc = L.wr_gather_rows(None, 10, 64, None, 4, None, None)
    assert rc == -1 and "NULL" in abi.last_error()
    # D not a multiple of 4 -> WR_E_SHAPE (-2)
    buf = (ctypes.c_float * 64)()
    addr = ctypes.addressof(buf)
    addr16 = (addr + 15) // 16 * 16
    rc = L.wr_axpy(addr16 + 4, addr16, 4, 1.0, 0, None)
    assert rc == -4 and "aligned" in abi.last_error()
    rc = L.wr_bpr_fwd(addr16, 4, addr16, 4, 6, addr16, addr16, addr16, 4, None, None, None, addr16, addr16, 1024, None)
    assert rc == -2 and "multiple of 4" in abi.last_error()
    with pytest.raises(abi.WhisprRecHipError):
        abi.check(rc, "wr_bpr_fwd")


def test_host_wrappers_refuse_cpu_tensors():
    import torch
    from whisprrec_amd import hip_ops
    t = torch.zeros(4, 64)
    i = torch.zeros(4, dtype=torch.int64)
    with pytest.raises(abi.WhisprRecHipError):
        hip_ops.bpr_fwd(t, t, i, i, i)
