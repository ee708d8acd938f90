"""CPU-side ABI checks: both shared libraries load without a GPU and export exactly the symbols
their headers declare; the evaluator refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from alphapig_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, txt)))


def test_hip_library_exports_every_declared_symbol():
    L = _native.hip()
    names = declared("alphapig_hip.h", "apz_")
    assert set(names) == set(_native.HIP_SYMBOLS)
    for s in names:
        assert hasattr(L, s), s
    assert L.apz_version() >= 1


def test_host_library_exports_every_declared_symbol():
    L = _native.host()
    names = declared("alphapig_host.h", "apzh_")
    assert set(names) == set(_native.HOST_SYMBOLS)
    for s in names:
        assert hasattr(L, s), s


def test_headers_cite_reference_interfaces():
    for h in ("alphapig_hip.h", "alphapig_host.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        assert re.search(r"(policy_value_net_mxnet|mcts_alphaZero|game)\.py:\d+", txt)


def test_evaluator_fails_loudly_without_gpu():
    L = _native.hip()
    if L.apz_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = _native.ApzConfig(15, 15, 9, 128, 10, 0, 16, 0)
    assert not L.apz_create(C.byref(cfg))
    assert b"no CPU fallback" in L.apz_last_error()
    from alphapig_amd.policy_value_net import EvaluatorError, PolicyValueNet
    with pytest.raises(EvaluatorError):
        PolicyValueNet(15, 15, batch_size=4, n_blocks=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "alphapig_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
