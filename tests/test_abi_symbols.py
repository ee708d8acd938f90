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


def test_one_hip_runtime_per_process():
    """Loading the HIP library before `import torch` must not leave two HIP runtimes in the process (the torch wheel
    bundles its own libamdhip64.so; with two, the second finds no device and the trainer cannot run after the
    evaluator): _native.hip() maps torch's copy first when torch is installed."""
    import subprocess
    import sys
    code = ("import alphapig_amd._native as n\n"
            "n.hip()\n"
            "import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(libs), libs)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout
