"""CPU: the C-ABI library loads without a GPU and exports every symbol include/drba_hip.h declares;
host-side weight packing is checked against a plain-python restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from drba_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "drba_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(drba_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/drba_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes prototype in drba_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.drba_abi_version() == 1
    assert lib.drba_error_string(-1) == b"invalid argument"


def test_argument_validation_without_gpu():
    lib = _lib.load()
    assert lib.drba_softsplat(None, None, None, None, None, 1, 1, 4, 4, 1, 0, None) == -1
    assert lib.drba_conv3x3(None, None, None, None, None, None, None, 1, 3, 8, 8, 16, 1, 1, 0.0, 0, 0.0, 0, None) == -1
    assert lib.drba_conv3x3_pick_cfg(3, 16, 8, 8, 3) == -2


@pytest.mark.parametrize("cin,cout,stride", [(3, 16, 2), (52, 16, 2), (39, 96, 2), (32, 32, 1), (192, 192, 1), (16, 16, 1)])
def test_conv_weight_packing_layout(cin, cout, stride):
    lib = _lib.load()
    cfg = lib.drba_conv3x3_pick_cfg(cin, cout, 64, 64, stride)
    n = lib.drba_conv3x3_packed_floats(cin, cout, cfg)
    w = torch.arange(cout * cin * 9, dtype=torch.float32).reshape(cout, cin, 3, 3) + 1
    buf = torch.full((n,), -1.0)
    assert lib.drba_conv3x3_pack(C.c_void_p(w.data_ptr()), C.c_void_p(buf.data_ptr()), cin, cout, cfg) == 0
    p = buf.numpy()
    assert (p >= 0).all()  # fully initialised (zeros in the padding)
    nz = np.sort(p[p > 0])
    assert np.array_equal(nz, np.sort(w.numpy().reshape(-1)))  # every weight exactly once


def test_deconv_weight_packing_layout():
    lib = _lib.load()
    cin, cout = 32, 52
    cfg = lib.drba_deconv4x4_pick_cfg(cin, cout, 68, 120)
    n = lib.drba_deconv4x4_packed_floats(cin, cout, cfg)
    w = torch.arange(cin * cout * 16, dtype=torch.float32).reshape(cin, cout, 4, 4) + 1
    buf = torch.full((n,), -1.0)
    assert lib.drba_deconv4x4_pack(C.c_void_p(w.data_ptr()), C.c_void_p(buf.data_ptr()), cin, cout, cfg) == 0
    p = buf.numpy()
    assert (p >= 0).all()
    assert np.array_equal(np.sort(p[p > 0]), np.sort(w.numpy().reshape(-1)))  # 4 phases x 4 taps = all 16 taps once


def test_product_has_no_cpu_fallback():
    from drba_amd import ops
    with pytest.raises(_lib.DrbaHipError):
        ops.flow_distance(torch.zeros(1, 2, 4, 4))
    if not torch.cuda.is_available():
        with pytest.raises(_lib.DrbaHipError):
            ops.default_device()


def test_oracle_is_not_imported_by_the_product():
    """oracle/ is test infrastructure: nothing under drba_amd/, infer.py or models/ may import it."""
    offenders = []
    for base in ("drba_amd", "models"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp")):
                    txt = open(os.path.join(dp, fn)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt:
                        offenders.append(os.path.join(dp, fn))
    for fn in ("infer.py",):
        p = os.path.join(ROOT, fn)
        if os.path.exists(p) and re.search(r"^\s*(from|import)\s+oracle\b", open(p).read(), flags=re.M):
            offenders.append(p)
    assert not offenders, offenders
