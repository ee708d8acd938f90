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


@pytest.mark.parametrize("kind,h,blocks", [("resnet", 15, 3), ("resnet", 8, 2), ("simple", 8, 0)])
def test_oracle_matches_an_independent_operator_library(kind, h, blocks):
    """Second opinion on the operator semantics the oracle restates "from the published definitions" (MXNet is
    absent): the same inference graph written with PyTorch's library operators in float64 -- conv2d (NCHW
    cross-correlation, zero padding 1), batch_norm in evaluation mode with eps 1e-3, linear on the row-major
    flatten, softmax, tanh.  Their definitions coincide with MXNet's Convolution / BatchNorm(use_global_stats) /
    FullyConnected / SoftmaxActivation / tanh; a flipped kernel, a transposed FC weight, a wrong flatten order or a
    wrong eps in the oracle would show here.  Test-only: neither side is the product."""
    import torch
    F = torch.nn.functional
    prm = weights.init_params(kind, h, h, 9, blocks, 128, seed=5, style="bench")
    rs = np.random.RandomState(4)
    planes = (rs.rand(3, 9, h, h) > 0.65).astype(np.float64)
    want = net_ref.forward(prm, planes, kind, blocks, np.float64)
    t = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in prm.items()}

    def bn(x, name, fix_gamma, mean_n, var_n):
        gamma = torch.ones_like(t[name + "_beta"]) if fix_gamma else t[name + "_gamma"]
        return F.batch_norm(x, t[name + mean_n], t[name + var_n], gamma, t[name + "_beta"], training=False, eps=1e-3)

    def conv_act(x, name):
        k = t[name + "_weight"].shape[-1]
        y = F.conv2d(x, t[name + "_weight"], t[name + "_bias"], padding=k // 2)
        return F.relu(bn(y, name, True, "_mean", "_var"))

    x = torch.tensor(planes)
    if kind == "resnet":
        x = conv_act(x, "res_conv1")
        for i in range(1, blocks + 1):
            y = F.conv2d(x, t["convA%d_weight" % i], t["convA%d_bias" % i], padding=1)
            y = F.relu(bn(y, "bnA%d" % i, False, "_moving_mean", "_moving_var"))
            y = F.conv2d(y, t["convB%d_weight" % i], t["convB%d_bias" % i], padding=1)
            x = F.relu(bn(y, "bnB%d" % i, False, "_moving_mean", "_moving_var") + x)
    else:
        for name, _ in net_ref.SIMPLE_LAYERS:
            x = conv_act(x, name)
    logits = F.linear(conv_act(x, "conv3_1_1").flatten(1), t["fc_3_1_1_weight"], t["fc_3_1_1_bias"])
    vlogit = F.linear(conv_act(x, "conv3_2_1").flatten(1), t["fc_3_2_1_weight"], t["fc_3_2_1_bias"])
    got = (logits, torch.softmax(logits, dim=1), vlogit, torch.tanh(vlogit))
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.numpy(), w, rtol=0, atol=1e-10)
