"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(exa_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    import exaconstit_amd.lib as L
    lib = C.CDLL(L.LIB_PATH)
    names = _declared("exaconstit_hip.h") + _declared("exaconstit_driver.h")
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported by libexaconstit_hip.so"
        assert hasattr(L, n), f"{n} has no ctypes signature in exaconstit_amd/lib.py"


def test_create_fails_loudly_without_gpu_or_with_bad_args():
    import torch
    import exaconstit_amd.lib as L
    props = np.loadtxt(os.path.join(ROOT, "tests", "golden", "refdata", "props_cp_voce.txt")).ravel()
    err = C.c_int(0)
    cfg = L.ExaConfig(L.EXA_FCC_VOCE, 16, props.ctypes.data_as(C.POINTER(C.c_double)), 298.0, 1, 8, 0, 0, -1)   # wrong nprops
    assert not L.exa_create(C.byref(cfg), C.byref(err)) and err.value == -1
    if not torch.cuda.is_available():
        cfg = L.ExaConfig(L.EXA_FCC_VOCE, 17, props.ctypes.data_as(C.POINTER(C.c_double)), 298.0, 1, 8, 0, 0, -1)
        assert not L.exa_create(C.byref(cfg), C.byref(err)) and err.value == -2   # no CPU fallback: HIP error


def test_product_does_not_reference_the_oracle():
    """The product path must not include, link or import anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "exaconstit_amd")):
        for f in files:
            if f.endswith((".hip", ".hpp", ".cpp", ".h", ".py")) or f == "Makefile":
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle/" not in txt.replace("nothing here includes or links the oracle", "") or f.endswith(".hpp") and "oracle/ecmech_port.hpp restates" in txt, f
    for f in ("exaconstit_hip.h", "exaconstit_driver.h"):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read()
