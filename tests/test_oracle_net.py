"""Cross-checks inside the net oracle (parity unpinned vs MXNet -- see oracle/__init__.py):
the C float32 port (cpu_baseline) vs the NumPy float64 restatement, BN folding identities,
and the parameter / FLOP bookkeeping SURVEY.md quotes."""
import numpy as np
import pytest

from alphapig_amd import weights
from oracle import net_ref


def test_param_count_and_flops_match_survey():
    sh = weights.param_shapes("resnet", 15, 15, 9, 10, 128)
    assert sum(int(np.prod(s)) for s in sh.values()) == 3176902
    assert net_ref.flops_per_leaf("resnet", 15, 15, 9, 10, 128) == 2 * 666260550
    assert net_ref.flops_per_leaf("simple", 8, 8) == 2 * 73584768                    # SURVEY row a9 (73.58 M MAC)


def test_c_port_matches_numpy_oracle():
    from alphapig_amd import build
    build.build_oracle()
    from oracle.net_ref_c import CNet
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=3, style="bench")
    rs = np.random.RandomState(0)
    planes = (rs.rand(3, 9, 15, 15) > 0.7).astype(np.float32)
    o = net_ref.forward(prm, planes, "resnet", 2, np.float64)
    # both implementations in oracle/net_ref.c: the plain loops and the vectorised forward the CPU baseline times
    # (AVX-512 / AVX2+FMA at run time; on a CPU with neither `fast` silently is the plain one)
    for fast in (False, True):
        net = CNet(prm, 15, 15, 9, 128, 2, fast=fast)
        assert (net.isa == "plain C loops") == (not fast) or not fast
        for i in range(3):
            lg, pr, vl, v = net.forward_one(planes[i])
            np.testing.assert_allclose(lg, o[0][i], rtol=0, atol=1e-4)
            np.testing.assert_allclose(pr, o[1][i], rtol=0, atol=2e-5)
            np.testing.assert_allclose(vl, o[2][i], rtol=0, atol=1e-4)
            np.testing.assert_allclose(v, o[3][i], rtol=0, atol=2e-5)


def test_fix_gamma_semantics():
    """gamma is ignored on stem/head BN (fix_gamma=True) and honoured in the blocks."""
    prm = weights.init_params("resnet", 8, 8, 9, 1, 128, seed=1, style="bench")
    rs = np.random.RandomState(2)
    planes = (rs.rand(2, 9, 8, 8) > 0.6).astype(np.float32)
    base = net_ref.forward(prm, planes, "resnet", 1)[0]
    p2 = dict(prm)
    p2["res_conv1_gamma"] = prm["res_conv1_gamma"] * 3.0
    p2["conv3_1_1_gamma"] = prm["conv3_1_1_gamma"] * 0.1
    np.testing.assert_array_equal(net_ref.forward(p2, planes, "resnet", 1)[0], base)
    p3 = dict(prm)
    p3["bnA1_gamma"] = prm["bnA1_gamma"] * 3.0
    assert np.abs(net_ref.forward(p3, planes, "resnet", 1)[0] - base).max() > 1e-3
