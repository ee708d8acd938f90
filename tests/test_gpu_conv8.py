"""The 8x8 path of the evaluator (BASELINE config 2: policy_value_net_mxnet_simple.py:68-92 behind policy_value /
policy_value_fn, 64 concurrent games = 32-board launches; and the 8x8 residual nets): conv8_kernel (work item = board x 16
output channels, the contraction split over the four waves, the first layer decoding the position codes itself) and
head8_kernel (both heads in one launch) against the float64 oracle, across batch shapes, through the planes and the codes
entry points, and after a device-side weight refresh."""
import numpy as np
import pytest

from alphapig_amd import weights
from oracle import net_ref
from test_gpu_net import LOGIT_ATOL, random_positions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 3, 32, 33, 100])
def test_simple_net_every_layer_against_the_oracle(n):
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=16, style="bench")
    net = PolicyValueNet(8, 8, batch_size=128, model_params=prm, net_kind="simple")
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
    net.close()


@pytest.mark.parametrize("n_filter", [64, 128, 256])
def test_resnet_8x8_residual_layers(n_filter):
    """The residual variant of conv8_kernel (policy_value_net_mxnet.py:77-83 at 8x8) for every filter count."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("resnet", 8, 8, 9, 2, n_filter, seed=17, style="bench")
    net = PolicyValueNet(8, 8, batch_size=64, n_blocks=2, n_filter=n_filter, model_params=prm)
    _, planes = random_positions(37, 8, seed=61)
    logits, _, vlog, _ = net.forward_with_logits(planes)
    o = net_ref.forward(prm, planes, "resnet", 2, np.float64, True)
    np.testing.assert_allclose(logits, o[0], rtol=0, atol=LOGIT_ATOL)
    np.testing.assert_allclose(vlog, o[2][:, 0], rtol=0, atol=LOGIT_ATOL)
    net.forward_planes(planes)
    np.testing.assert_allclose(net.layer_output(0, 37), o[4][0], rtol=0, atol=1e-4)
    np.testing.assert_allclose(net.layer_output(4, 37), o[4][1], rtol=0, atol=2e-4)
    net.close()


@pytest.mark.parametrize("c_in", [9, 4])
def test_codes_entry_point_equals_planes_entry_point(c_in):
    """The self-play path hands over 80-byte position codes and the first convolution decodes them in its LDS tile
    (Board.current_state, game.py:68-94 / the 4-plane encoder :96-115); policy_value hands over float planes.  Same
    positions, same bits; and through the stream-ordered slots as well."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, c_in, seed=18, style="bench")
    net = PolicyValueNet(8, 8, batch_size=64, model_params=prm, net_kind="simple", c_in=c_in)
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


def test_device_side_weight_refresh_packs_the_same_bits():
    """apz_load_weights_dev (the trainer's refresh after every step, policy_value_net_mxnet.py:295-297) packs conv8's
    weight layout and head8's FullyConnected matrix with kernels; apz_load_weights does it on the host.  Same weights in,
    same evaluator out."""
    torch = pytest.importorskip("torch")
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm = weights.init_params("simple", 8, 8, 9, seed=19, style="bench")
    other = weights.init_params("simple", 8, 8, 9, seed=20, style="bench")
    a = PolicyValueNet(8, 8, batch_size=32, model_params=prm, net_kind="simple")
    b = PolicyValueNet(8, 8, batch_size=32, model_params=other, net_kind="simple")
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
