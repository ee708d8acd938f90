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


@pytest.mark.parametrize("layout,relu,resid,gamma", [(0, True, True, True), (1, True, True, True), (1, True, False, False),
                                                     (0, False, False, True), (1, False, True, False)])
def test_bn_act_forward_backward(layout, relu, resid, gamma):
    """hipconv.bn_act (training-mode BatchNorm + residual + ReLU, dense and padded-row layouts) against torch
    float64 autograd; the pad column stays zero and takes no part in the statistics."""
    from alphapig_amd import hipconv
    n, c = 7, 128
    g = torch.Generator().manual_seed(11 + layout)
    x = torch.randn(n, c, 15, 15, generator=g) * 1.7 + 0.3
    r = torch.randn(n, c, 15, 15, generator=g)
    ga = torch.rand(c, generator=g) + 0.5
    be = torch.randn(c, generator=g) * 0.2
    dy = torch.randn(n, c, 15, 15, generator=g)
    rm, rv = torch.zeros(c), torch.ones(c)
    # reference
    x64, r64 = x.double().requires_grad_(True), r.double().requires_grad_(True)
    ga64, be64 = ga.double().requires_grad_(gamma), be.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y64 = torch.nn.functional.batch_norm(x64, rm64, rv64, ga64 if gamma else torch.ones(c, dtype=torch.float64), be64,
                                         training=True, momentum=0.1, eps=1e-3)
    if resid:
        y64 = y64 + r64
    if relu:
        y64 = torch.relu(y64)
    y64.backward(dy.double())
    # HIP
    pad = (lambda t: torch.nn.functional.pad(t, (0, 1))) if layout == 1 else (lambda t: t)
    xc, rc = pad(x).cuda().requires_grad_(True), pad(r).cuda().requires_grad_(True)
    gc, bc = ga.cuda().requires_grad_(gamma), be.cuda().requires_grad_(True)
    rmc, rvc = rm.cuda(), rv.cuda()
    y = hipconv.bn_act(xc, gc if gamma else None, bc, rmc, rvc, rc if resid else None, relu, layout, 0.1, 1e-3)
    y.backward(pad(dy).cuda())
    torch.cuda.synchronize()
    cut = (lambda t: t[..., :15]) if layout == 1 else (lambda t: t)
    if layout == 1:
        assert float(y.detach()[..., 15].abs().max()) == 0.0
        assert float(xc.grad[..., 15].abs().max()) == 0.0
    close = lambda a, b, tol: float((a.cpu().double() - b).abs().max()) < tol * (float(b.abs().max()) + 1e-3)
    assert close(cut(y.detach()), y64.detach(), 1e-5)
    assert close(cut(xc.grad), x64.grad, 1e-4)
    assert close(bc.grad, be64.grad, 1e-5)
    if gamma:
        assert close(gc.grad, ga64.grad, 1e-5)
    if resid:
        assert close(cut(rc.grad), r64.grad, 1e-6)
    assert close(rmc, rm64, 1e-5) and close(rvc, rv64, 1e-5)


def test_trainer_hip16_trunk_matches_torch_graph():
    """The residual trunk end to end on HIP kernels in the padded-row layout (trunk_backend="hip16") against the
    torch graph with torch convolutions: losses of three steps and the parameters after them."""
    from alphapig_amd import weights
    from alphapig_amd.train import TorchTrainer
    rs = np.random.RandomState(1)
    prm = weights.init_params("resnet", 15, 15, 9, 2, 128, seed=2, style="bench")
    states = (rs.rand(24, 9, 15, 15) > 0.6).astype(np.float32)
    pis = rs.dirichlet(np.ones(225), size=24).astype(np.float32)
    zs = rs.choice([-1.0, 1.0], size=24).astype(np.float32)
    out = {}
    for name, kw in (("torch", dict(conv_backend="torch")), ("hip16", dict(conv_backend="hip", trunk_backend="hip16"))):
        tr = TorchTrainer(prm, "resnet", n_blocks=2, batch_size=24, device="cuda", dropout=0.5, seed=3, **kw)
        losses = [tr.train_step(states, pis, zs, 1e-3)[0] for _ in range(3)]
        out[name] = (losses, tr.get_params())
    np.testing.assert_allclose(out["hip16"][0], out["torch"][0], rtol=3e-4)
    for k in ("convA1_weight", "convB2_weight", "bnA1_gamma", "bnB2_beta", "res_conv1_weight", "bnA1_moving_var",
              "bnB2_moving_mean"):
        d = np.abs(out["hip16"][1][k] - out["torch"][1][k])
        assert int((d > 2e-4).sum()) <= max(8, 0.01 * d.size), k     # (Adam sign flips on noise-level gradients, see above)
        assert float(d.max()) < 2 * 1e-3 * 3 + 2e-4, k


@pytest.mark.parametrize("n", [64, 130, 513])
def test_winograd_domain_weight_gradient(n, monkeypatch):
    """apz_wgrad_wino (padded-row layout, used from 64 boards on) against torch float64 and against the direct
    weight-gradient kernel on the same tensors."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(4000 + n)
    x = torch.randn(n, 128, 15, 15, generator=g)
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).float()
    dy = torch.randn(n, 128, 15, 15, generator=g)
    w64 = w.double().requires_grad_(True)
    torch.nn.functional.conv2d(x.double(), w64, None, padding=1).backward(dy.double())
    pad = lambda t: torch.nn.functional.pad(t, (0, 1))
    got = {}
    for mode in ("wino", "direct"):
        monkeypatch.setenv("APZ_TRAIN_WGRAD", mode)
        wc = w.cuda().requires_grad_(True)
        y = hipconv.conv3x3(pad(x).cuda(), wc, None, hipconv.ROWS16)
        y.backward(pad(dy).cuda())
        torch.cuda.synchronize()
        got[mode] = wc.grad.cpu().double()
        assert float((got[mode] - w64.grad).abs().max()) < 1e-4 * float(w64.grad.abs().max()), mode
