"""The 8x8 path of the evaluator (BASELINE config 2: policy_value_net_mxnet_simple.py:68-92 behind policy_value /
policy_value_fn, 64 concurrent games = 32-board launches; and the 8x8 residual nets): conv8_kernel (work item = board x 16
output channels, the contraction split over the four waves, the first layer decoding the position codes itself) and
head8_kernel (both heads in one launch) against the float64 oracle, across batch shapes, through the planes and the codes
entry points, and after a device-side weight refresh.  Round 6: every test in both arithmetics -- "f32" (conv8_kernel, the fp32
matrix pipe) and "f16x2" (conv8h_kernel, csrc/conv8_split.h: split operands on the fp16 matrix pipe, the default on 8x8 boards)
-- to the same tolerances, plus the split kernel's own properties (batch-independent bits, the overflow repeat)."""
import numpy as np
import pytest

from alphapig_amd import weights
from oracle import net_ref
from test_gpu_net import LOGIT_ATOL, random_positions

pytestmark = pytest.mark.gpu


ARITHS = ["f32", "f16x2"]


def test_auto_is_the_split_arithmetic_on_8x8_boards():
    from alphapig_amd.policy_value_net import PolicyValueNet
    net = PolicyValueNet(8, 8, batch_size=16, net_kind="simple")
    assert net.trunk_arith == "f16x2"
    net.close()


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("n", [1, 3, 32, 33, 100])
def test_simple_net_every_layer_against_the_oracle(n, arith):
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=16, style="bench")
    net = PolicyValueNet(8, 8, batch_size=128, model_params=prm, net_kind="simple", trunk_arith=arith)
    _, planes = random_positions(n, 8, seed=40 + n)
    logits, probs, vlog, vals = net.forward_with_logits(planes)
    o = net_ref.forward(prm, planes, "simple", dtype=np.float64, return_trunk=True)
    np.testing.assert_allclose(logits, o[0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(probs, o[1], rtol=0, atol=2e-5)
    np.testing.assert_allclose(vlog, o[2][:, 0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vals, o[3][:, 0], rtol=0, atol=2e-5)
    assert np.abs(probs.sum(axis=1) - 1).max() < 1e-5
    net.forward_planes(planes)
    np.testing.assert_allclose(net.layer_output(0, n), o[4][0], rtol=0, atol=1e-4)      # 9 -> 64 (three k-steps: wave 0 idle)
    np.testing.assert_allclose(net.layer_output(5, n), o[4][1], rtol=0, atol=1e-4)      # 256 -> 256
    assert net.trunk_overflows() == 0
    net.close()


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("n_filter", [64, 128, 256])
def test_resnet_8x8_residual_layers(n_filter, arith):
    """The residual variant of conv8_kernel / conv8h_kernel (policy_value_net_mxnet.py:77-83 at 8x8) for every filter count."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 8, 8, 9, 2, n_filter, seed=17, style="bench")
    net = PolicyValueNet(8, 8, batch_size=64, n_blocks=2, n_filter=n_filter, model_params=prm, trunk_arith=arith)
    _, planes = random_positions(37, 8, seed=61)
    logits, _, vlog, _ = net.forward_with_logits(planes)
    o = net_ref.forward(prm, planes, "resnet", 2, np.float64, True)
    np.testing.assert_allclose(logits, o[0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog, o[2][:, 0], rtol=0, atol=LOGIT_ATOL)
    net.forward_planes(planes)
    np.testing.assert_allclose(net.layer_output(0, 37), o[4][0], rtol=0, atol=1e-4)
    np.testing.assert_allclose(net.layer_output(4, 37), o[4][1], rtol=0, atol=2e-4)
    net.close()


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("c_in", [9, 4])
def test_codes_entry_point_equals_planes_entry_point(c_in, arith):
    """The self-play path hands over 80-byte position codes and the first convolution decodes them in its LDS tile
    (Board.current_state, game.py:68-94 / the 4-plane encoder :96-115); policy_value hands over float planes.  Same
    positions, same bits; and through the stream-ordered slots as well."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, c_in, seed=18, style="bench")
    net = PolicyValueNet(8, 8, batch_size=64, model_params=prm, net_kind="simple", c_in=c_in, trunk_arith=arith)
    codes, planes = random_positions(50, 8, c=c_in, seed=71)
    p_planes, v_planes = net.forward_planes(planes)
    p_codes, v_codes = net.evaluate_codes(codes)
    np.testing.assert_array_equal(p_codes, p_planes)
    np.testing.assert_array_equal(v_codes, v_planes)
    p_slot, v_slot = net.evaluate_codes_slot(1, codes[:33])
    np.testing.assert_array_equal(p_slot, p_planes[:33])
    np.testing.assert_array_equal(v_slot, v_planes[:33])
    np.testing.assert_allclose(p_planes, net_ref.forward(prm, planes, "simple", dtype=np.float64)[1], rtol=0, atol=2e-5)
    net.close()


@pytest.mark.parametrize("arith", ARITHS)
def test_device_side_weight_refresh_packs_the_same_bits(arith):
    """apz_load_weights_dev (the trainer's refresh after every step, policy_value_net_mxnet.py:295-297) packs conv8's
    weight layout (f16x2: conv8h's two-term layout and its per-channel scales as well) and head8's FullyConnected matrix
    with kernels; apz_load_weights does it on the host.  Same weights in, same evaluator out."""
    torch = pytest.importorskip("torch")
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=19, style="bench")
    other = weights.init_params("simple", 8, 8, 9, seed=20, style="bench")
    a = PolicyValueNet(8, 8, batch_size=32, model_params=prm, net_kind="simple", trunk_arith=arith)
    b = PolicyValueNet(8, 8, batch_size=32, model_params=other, net_kind="simple", trunk_arith=arith)
    _, planes = random_positions(20, 8, seed=81)
    pa, va = a.forward_planes(planes)
    assert not np.array_equal(pa, b.forward_planes(planes)[0])
    dev = {k: torch.tensor(v, device="cuda") for k, v in prm.items()}
    b.load_device_params(dev)
    pb, vb = b.forward_planes(planes)
    np.testing.assert_allclose(pb, pa, rtol=0, atol=1e-6)       # folding in double on the host vs in double on the device: last-bit effects only
    np.testing.assert_allclose(vb, va, rtol=0, atol=1e-6)
    a.close()
    b.close()


def test_split_kernel_bits_do_not_depend_on_the_batch():
    """conv8h_kernel's work item is one board x 16 output channels and its four partial sums are added in wave order: a
    position evaluates to the same bits alone, in a batch of 33 and in a batch of 100 (the reference's predict on one
    state and on a batch agree the same way: policy_value_net_mxnet.py:250-271)."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=23, style="bench")
    net = PolicyValueNet(8, 8, batch_size=128, model_params=prm, net_kind="simple", trunk_arith="f16x2")
    _, planes = random_positions(100, 8, seed=91)
    p100, v100 = net.forward_planes(planes)
    p33, v33 = net.forward_planes(planes[:33])
    p1, v1 = net.forward_planes(planes[7:8])
    np.testing.assert_array_equal(p33, p100[:33])
    np.testing.assert_array_equal(v33, v100[:33])
    np.testing.assert_array_equal(p1[0], p100[7])
    np.testing.assert_array_equal(v1[0], v100[7])
    net.close()


def test_split_kernel_is_fp32_accurate_on_large_activations():
    """Two fp16 terms carry 22 bits of an activation whatever its magnitude inside the fp16 range: the first layer's
    weights times 300 (activations in the hundreds and thousands instead of around one) leave the RELATIVE error against
    the float64 oracle at the fp32 kernel's level, and no forward is repeated."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = dict(weights.init_params("simple", 8, 8, 9, seed=24, style="bench"))
    prm["conv1_weight"] = np.asarray(prm["conv1_weight"], np.float32) * 300.0
    _, planes = random_positions(40, 8, seed=92)
    o = net_ref.forward(prm, planes, "simple", dtype=np.float64, return_trunk=True)
    scale = np.abs(o[4][1]).max()
    assert scale > 50.0
    errs = {}
    for arith in ARITHS:
        net = PolicyValueNet(8, 8, batch_size=64, model_params=prm, net_kind="simple", trunk_arith=arith)
        net.forward_planes(planes)
        errs[arith] = np.abs(net.layer_output(5, 40) - o[4][1]).max() / scale
        assert net.trunk_overflows() == 0
        net.close()
    assert errs["f32"] < 2e-6 and errs["f16x2"] < 2e-6, errs
    assert errs["f16x2"] < 3 * errs["f32"] + 2e-7, errs


def test_f16x2_overflow_repeats_the_forward_on_the_exact_kernel_8x8():
    """An activation beyond the fp16 range (|x| > 65 504; no Winograd transform in front of it here) becomes inf - inf in
    the split: the layer that reads it raises the forward's word and the collecting entry point repeats the forward on
    conv8_kernel -- the bits of a trunk_arith="f32" evaluator, through every entry point, never a silent clamp."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=25, style="bench")
    big = dict(prm)
    big["conv1_weight"] = np.asarray(prm["conv1_weight"], np.float32) * 1.0e6
    codes, planes = random_positions(40, 8, seed=93)
    exact = PolicyValueNet(8, 8, batch_size=64, model_params=big, net_kind="simple", trunk_arith="f32")
    split = PolicyValueNet(8, 8, batch_size=64, model_params=big, net_kind="simple", trunk_arith="f16x2")
    assert split.trunk_overflows() == 0
    a, b = exact.forward_with_logits(planes), split.forward_with_logits(planes)          # apz_forward
    assert split.trunk_overflows() == 1
    for x, y in zip(a, b):
        assert np.isfinite(np.asarray(y)).all()
        np.testing.assert_array_equal(np.asarray(x), np.asarray(y))
    pa, pb = exact.forward_planes(planes), split.forward_planes(planes)                 # apz_forward_host
    assert split.trunk_overflows() == 2
    np.testing.assert_array_equal(pa[0], pb[0])
    ca, cb = exact.evaluate_codes(codes), split.evaluate_codes(codes)                   # apz_forward_codes_host
    assert split.trunk_overflows() == 3
    np.testing.assert_array_equal(ca[0], cb[0])
    np.testing.assert_array_equal(ca[1], cb[1])
    for rep in range(4):                                                                # apz_submit_codes / apz_wait (the 3rd use replays a graph)
        sa, sb = exact.evaluate_codes_slot(1, codes[:32]), split.evaluate_codes_slot(1, codes[:32])
        assert split.trunk_overflows() == 4 + rep
        np.testing.assert_array_equal(sa[0], sb[0])
        np.testing.assert_array_equal(sa[1], sb[1])
    # ordinary weights afterwards: no repeat, the split kernel's own results
    split.set_params(prm)
    q = split.forward_with_logits(planes)
    assert split.trunk_overflows() == 7
    np.testing.assert_allclose(q[0], net_ref.forward(prm, planes, "simple", dtype=np.float64)[0], rtol=0, atol=LOGIT_ATOL)
    exact.close()
    split.close()
