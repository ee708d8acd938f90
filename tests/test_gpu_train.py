"""Hand-written HIP convolution forward / data-gradient / weight-gradient
against torch float64 autograd on the same tensors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("n,ci,co,hw", [(5, 128, 128, 15), (3, 9, 128, 15), (130, 128, 128, 15), (7, 64, 64, 8),
                                        (4, 64, 128, 8), (2, 256, 256, 8), (3, 256, 256, 15), (6, 128, 64, 15)])
def test_conv3x3_fwd_dgrad_wgrad(n, ci, co, hw):
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(n * 1000 + ci + co)
    x = torch.randn(n, ci, hw, hw, generator=g)
    w = (torch.randn(co, ci, 3, 3, generator=g) * float(1.0 / np.sqrt(ci * 9))).float()
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, hw, hw, generator=g)
    need_dx = ci in (64, 128, 256)
    # reference: float64 on the CPU
    x64 = x.double().requires_grad_(need_dx)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, b64, padding=1)
    y64.backward(dy.double())
    # HIP
    xc = x.cuda().requires_grad_(need_dx)
    wc = w.cuda().requires_grad_(True)
    bc = b.cuda().requires_grad_(True)
    assert hipconv.supported(xc, wc)
    y = hipconv.conv3x3(xc, wc, bc)
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    tol = lambda ref: 2e-5 * float(ref.detach().abs().max()) + 1e-6
    assert float((y.detach().cpu().double() - y64.detach()).abs().max()) < tol(y64)
    assert float((wc.grad.cpu().double() - w64.grad).abs().max()) < 5 * tol(w64.grad)
    assert float((bc.grad.cpu().double() - b64.grad).abs().max()) < 5 * tol(b64.grad)
    if need_dx:
        assert float((xc.grad.cpu().double() - x64.grad).abs().max()) < tol(x64.grad)


@pytest.mark.parametrize("n", [1, 6, 131])
def test_trunk_shape_winograd_and_direct_paths_agree(n, monkeypatch):
    """128 -> 128 at 15x15 runs on the self-play path's fused Winograd kernel from 192 boards on
    (APZ_TRAIN_CONV=wino / direct force a path): both against torch float64, forward and data gradient."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(900 + n)
    x = torch.randn(n, 128, 15, 15, generator=g)
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).float()
    b = torch.randn(128, generator=g)
    dy = torch.randn(n, 128, 15, 15, generator=g)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, b64, padding=1)
    y64.backward(dy.double())
    for mode in ("wino", "direct"):
        monkeypatch.setenv("APZ_TRAIN_CONV", mode)
        xc, wc, bc = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = hipconv.conv3x3(xc, wc, bc)
        y.backward(dy.cuda())
        torch.cuda.synchronize()
        tol = lambda ref: 2e-5 * float(ref.detach().abs().max()) + 1e-6
        assert float((y.detach().cpu().double() - y64.detach()).abs().max()) < tol(y64), mode
        assert float((xc.grad.cpu().double() - x64.grad).abs().max()) < tol(x64.grad), mode
        assert float((wc.grad.cpu().double() - w64.grad).abs().max()) < 5 * tol(w64.grad), mode


def test_trainer_with_hip_convs_matches_torch_convs():
    """One optimiser step of the interim trainer with the 3x3 convolutions on the HIP kernels ==
    the same step on torch's convolutions (same dropout stream)."""
    from alphapig_amd import weights
    from alphapig_amd.train import TorchTrainer
    rs = np.random.RandomState(0)
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=1, style="bench")
    states = (rs.rand(16, 9, 15, 15) > 0.6).astype(np.float32)
    pis = rs.dirichlet(np.ones(225), size=16).astype(np.float32)
    zs = rs.choice([-1.0, 1.0], size=16).astype(np.float32)
    out = {}
    for backend in ("torch", "hip"):
        tr = TorchTrainer(prm, "resnet", n_blocks=2, batch_size=16, device="cuda", dropout=0.5, seed=3,
                          conv_backend=backend)
        losses = [tr.train_step(states, pis, zs, 1e-3)[0] for _ in range(3)]
        out[backend] = (losses, tr.get_params())
    np.testing.assert_allclose(out["hip"][0], out["torch"][0], rtol=2e-4)
    # Adam's first updates are ~lr * sign(g): where |g| is at the rounding-noise level the two convolution
    # implementations (fp32 Winograd / direct vs MIOpen) may step in opposite directions, so a few elements
    # per mille differ by up to 2 * lr * steps; everything else agrees to 2e-4.
    for k in ("convA1_weight", "convB2_weight", "res_conv1_weight", "fc_3_1_1_weight", "bnA1_moving_var"):
        d = np.abs(out["hip"][1][k] - out["torch"][1][k])
        assert float((d > 2e-4).mean()) < 5e-3, k
        assert float(d.max()) < 2 * 1e-3 * 3 + 2e-4, k
