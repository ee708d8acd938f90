"""The training graph's HIP operators (alphapig_amd/hipconv.py over include/alphapig_hip.h) against
torch float64 autograd on the same tensors, and the product trainer (alphapig_amd/train.py: explicit forward / backward
on those operators, no autograd) against the PyTorch comparator tests/torch_trainer.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


def tol(ref, rel=2e-5):
    return rel * float(ref.detach().abs().max()) + 1e-6


def err(got, ref):
    return float((got.detach().cpu().double() - ref.detach()).abs().max())


def pad16(t):
    return F.pad(t, (0, 1)).contiguous()


@pytest.mark.parametrize("n,ci,co,hw", [(5, 128, 128, 15), (3, 9, 128, 15), (130, 128, 128, 15), (7, 64, 64, 8),
                                        (4, 64, 128, 8), (2, 256, 256, 8), (3, 256, 256, 15), (6, 128, 64, 15),
                                        (200, 128, 128, 15)])
def test_conv3x3_fwd_dgrad_wgrad_bias(n, ci, co, hw):
    """Dense tensors: direct MFMA kernel, and from 192 boards of the trunk shape the Winograd kernel."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(n * 1000 + ci + co)
    x = torch.randn(n, ci, hw, hw, generator=g)
    w = (torch.randn(co, ci, 3, 3, generator=g) * float(1.0 / np.sqrt(ci * 9))).float()
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, hw, hw, generator=g)
    skip = torch.randn(n, ci, hw, hw, generator=g)
    need_dx = ci in (64, 128, 256)
    x64 = x.double().requires_grad_(need_dx)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b64, padding=1)
    y64.backward(dy.double())
    xc, wc, bc, dyc = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    y = hipconv.conv3x3_fwd(xc, wc, bc)
    dw = hipconv.conv3x3_wgrad(xc, dyc)
    db = hipconv.bias_grad(dyc)
    torch.cuda.synchronize()
    assert err(y, y64) < tol(y64)
    assert err(dw, w64.grad) < 5 * tol(w64.grad)
    assert err(db, b64.grad) < 5 * tol(b64.grad)
    yr = hipconv.conv3x3_fwd(xc, wc, bc, relu=True)
    assert err(yr, torch.relu(y64)) < tol(y64)
    if need_dx:
        dx = hipconv.conv3x3_dgrad(dyc, wc)
        assert err(dx, x64.grad) < tol(x64.grad)
        dx2 = hipconv.conv3x3_dgrad(dyc, wc, add=skip.cuda())
        assert err(dx2, x64.grad + skip.double()) < tol(x64.grad)


@pytest.mark.parametrize("n", [1, 6, 64, 131])
def test_trunk_shape_in_the_padded_row_layout(n):
    """128 -> 128 at 15x15 on [n][128][15][16] tensors: the self-play path's fused Winograd kernel forward (with and
    without ReLU) and as data gradient with the skip gradient added in its epilogue; weight gradient through the
    Winograd domain (wgrad_wino3_kernel: every batch size, also slices without boards); pad column zero on every output."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(900 + n)
    x = torch.randn(n, 128, 15, 15, generator=g)
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).float()
    b = torch.randn(128, generator=g)
    dy = torch.randn(n, 128, 15, 15, generator=g)
    skip = torch.randn(n, 128, 15, 15, generator=g)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b64, padding=1)
    y64.backward(dy.double())
    xc, wc, bc, dyc, sc = pad16(x).cuda(), w.cuda(), b.cuda(), pad16(dy).cuda(), pad16(skip).cuda()
    R = hipconv.ROWS16
    y = hipconv.conv3x3_fwd(xc, wc, bc, R)
    yr = hipconv.conv3x3_fwd(xc, wc, bc, R, relu=True)
    dx = hipconv.conv3x3_dgrad(dyc, wc, R)
    dxs = hipconv.conv3x3_dgrad(dyc, wc, R, add=sc)
    dw = hipconv.conv3x3_wgrad(xc, dyc, R)
    db = hipconv.bias_grad(dyc, R)
    torch.cuda.synchronize()
    for t in (y, yr, dx, dxs):
        assert float(t[..., 15].abs().max()) == 0.0
    assert err(y[..., :15], y64) < tol(y64)
    assert err(yr[..., :15], torch.relu(y64)) < tol(y64)
    assert err(dx[..., :15], x64.grad) < tol(x64.grad)
    assert err(dxs[..., :15], x64.grad + skip.double()) < tol(x64.grad)
    assert err(dw, w64.grad) < 1e-4 * float(w64.grad.abs().max())
    assert err(db, b64.grad) < 5 * tol(b64.grad)


def test_winograd_domain_weight_gradient_large_batch():
    from alphapig_amd import hipconv
    n = 513
    g = torch.Generator().manual_seed(4000 + n)
    x = torch.randn(n, 128, 15, 15, generator=g)
    dy = torch.randn(n, 128, 15, 15, generator=g)
    w64 = torch.zeros(128, 128, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, padding=1).backward(dy.double())
    dw = hipconv.conv3x3_wgrad(pad16(x).cuda(), pad16(dy).cuda(), hipconv.ROWS16)
    dwd = hipconv.conv3x3_wgrad(x.cuda(), dy.cuda(), hipconv.DENSE)        # direct kernel on the same data
    torch.cuda.synchronize()
    assert err(dw, w64.grad) < 1e-4 * float(w64.grad.abs().max())
    assert err(dwd, w64.grad) < 1e-4 * float(w64.grad.abs().max())


@pytest.mark.parametrize("layout,relu,resid,gamma,c", [(0, True, True, True, 128), (1, True, True, True, 128),
                                                       (1, True, False, False, 128), (0, False, False, True, 128),
                                                       (1, False, True, False, 128), (0, True, False, False, 4),
                                                       (0, True, False, False, 2)])
def test_bn_forward_backward(layout, relu, resid, gamma, c):
    """Training-mode BatchNorm (+ residual) (+ ReLU), dense and padded-row layouts (and the 4- / 2-channel head
    shapes), against torch float64 autograd; the pad column stays zero and takes no part in the statistics."""
    from alphapig_amd import hipconv
    n = 7
    g = torch.Generator().manual_seed(11 + layout + c)
    x = torch.randn(n, c, 15, 15, generator=g) * 1.7 + 0.3
    r = torch.randn(n, c, 15, 15, generator=g)
    ga = torch.rand(c, generator=g) + 0.5
    be = torch.randn(c, generator=g) * 0.2
    dy = torch.randn(n, c, 15, 15, generator=g)
    rm, rv = torch.zeros(c), torch.ones(c)
    x64, r64 = x.double().requires_grad_(True), r.double().requires_grad_(True)
    ga64, be64 = ga.double().requires_grad_(gamma), be.double().requires_grad_(True)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    y64 = F.batch_norm(x64, rm64, rv64, ga64 if gamma else torch.ones(c, dtype=torch.float64), be64, training=True,
                       momentum=0.1, eps=1e-3)
    if resid:
        y64 = y64 + r64
    if relu:
        y64 = torch.relu(y64)
    y64.backward(dy.double())
    pad = pad16 if layout == 1 else (lambda t: t)
    xc, rc = pad(x).cuda(), pad(r).cuda()
    gc, bc = ga.cuda(), be.cuda()
    rmc, rvc = rm.cuda(), rv.cuda()
    y, mean, invstd = hipconv.bn_fwd(xc, gc if gamma else None, bc, rmc, rvc, rc if resid else None, relu, layout, 0.1, 1e-3)
    # dxsum: this layer's columns of a wider matrix (the trainer gives every trunk layer its own columns)
    parts = torch.full((hipconv.bn_bwd_splits(xc, layout), c + 40), 7.0, device="cuda")
    dx, dres, dgamma, dbeta = hipconv.bn_bwd(pad(dy).cuda(), xc, y, gc if gamma else None, mean, invstd, relu, resid, layout,
                                             dxsum=parts[:, 8:8 + c])
    db = hipconv.colsum(parts)
    dx_b, _, dgamma_b, dbeta_b = hipconv.bn_bwd(pad(dy).cuda(), xc, y, gc if gamma else None, mean, invstd, relu, resid, layout)
    torch.cuda.synchronize()
    # the column sums of dx = what apz_bias_grad makes of the same tensor (rounding noise around zero: dx sums to 0 per
    # channel analytically); columns outside the layer's are untouched; no atomics anywhere: a second call, the same bits
    ref_db = dx.double().sum(dim=(0, 2, 3)).cpu()
    scale = float(dx.abs().sum(dim=(0, 2, 3)).max())
    assert float((db[8:8 + c].cpu().double() - ref_db).abs().max()) < 1e-6 * scale
    assert float((hipconv.bias_grad(dx, layout).cpu().double() - ref_db).abs().max()) < 1e-6 * scale
    assert float((db[:8] - 7.0 * parts.shape[0]).abs().max()) == 0.0 and float((db[8 + c:] - 7.0 * parts.shape[0]).abs().max()) == 0.0
    assert torch.equal(dx, dx_b) and torch.equal(dbeta, dbeta_b) and torch.equal(dgamma, dgamma_b)
    cut = (lambda t: t[..., :15]) if layout == 1 else (lambda t: t)
    if layout == 1:
        assert float(y[..., 15].abs().max()) == 0.0
        assert float(dx[..., 15].abs().max()) == 0.0
    close = lambda a, b, t: float((a.cpu().double() - b).abs().max()) < t * (float(b.abs().max()) + 1e-3)
    assert close(cut(y), y64.detach(), 1e-5)
    assert close(cut(dx), x64.grad, 1e-4)
    assert close(dbeta, be64.grad, 1e-5)
    if gamma:
        assert close(dgamma, ga64.grad, 1e-5)
    if resid:
        assert close(cut(dres), r64.grad, 1e-6)
    else:
        assert dres is None
    assert close(rmc, rm64, 1e-5) and close(rvc, rv64, 1e-5)


@pytest.mark.parametrize("n", [1, 7, 64, 128, 131, 512])
def test_trunk_conv_leaves_the_batchnorm_statistics(n):
    """conv3x3_fwd_stats: the same y as conv3x3_fwd bit for bit (item = board pair x 32 or 64 channels, odd batches), and per
    (channel, board) the sum / sum of squares of the board's 225 outputs; BatchNorm from them == BatchNorm with its own
    statistics pass up to the rounding of the sums (policy_value_net_mxnet.py:41-56)."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(100 + n)
    x = pad16(torch.randn(n, 128, 15, 15, generator=g)).cuda()
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34).cuda()
    b = torch.randn(128, generator=g).cuda()
    be = (torch.randn(128, generator=g) * 0.2).cuda()
    y0 = hipconv.conv3x3_fwd(x, w, b, hipconv.ROWS16)
    y, st = hipconv.conv3x3_fwd_stats(x, w, b)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    y64 = y[..., :15].double()
    s1, s2 = y64.sum(dim=(2, 3)).t(), (y64 * y64).sum(dim=(2, 3)).t()          # [C][n]
    assert float((st[..., 0] - s1).abs().max()) < 1e-5 * float(y64.abs().sum(dim=(2, 3)).max())
    assert float((st[..., 1] - s2).abs().max()) < 1e-5 * float(s2.max())
    a0, m0, i0 = hipconv.bn_fwd(y, None, be, None, None, None, True, hipconv.ROWS16, 0.1, 1e-3)
    a1, m1, i1 = hipconv.bn_fwd(y, None, be, None, None, None, True, hipconv.ROWS16, 0.1, 1e-3, stats=st)
    torch.cuda.synchronize()
    assert float((m0 - m1).abs().max()) < 1e-6 * float(m0.abs().max()) + 1e-7
    assert float(((i0 - i1) / i0).abs().max()) < 1e-5
    assert float((a0 - a1).abs().max()) < 1e-4 * float(a0.abs().max())
    y2, st2 = hipconv.conv3x3_fwd_stats(x, w, b)                                   # no atomics: the same bits again
    torch.cuda.synchronize()
    assert torch.equal(st, st2)


@pytest.mark.parametrize("n,layout,c", [(130, 1, 128), (515, 1, 128), (300, 0, 4), (300, 0, 128)])
def test_bn_batch_splits(n, layout, c):
    """... at batch sizes that give every channel several batch splits (padded rows: four boards per trip and split): the
    consumers' sum of the per-split partials against float64, and the same bits on a second run."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, 15, 15, generator=g) * 0.8 - 0.2
    dy = torch.randn(n, c, 15, 15, generator=g)
    be = torch.randn(c, generator=g) * 0.2
    x64, be64 = x.double().requires_grad_(True), be.double().requires_grad_(True)
    y64 = torch.relu(F.batch_norm(x64, None, None, torch.ones(c, dtype=torch.float64), be64, training=True, eps=1e-3))
    y64.backward(dy.double())
    pad = pad16 if layout == 1 else (lambda t: t)
    cut = (lambda t: t[..., :15]) if layout == 1 else (lambda t: t)
    xc, dyc, bc = pad(x).cuda(), pad(dy).cuda(), be.cuda()
    assert hipconv.bn_bwd_splits(xc, layout) > 1
    outs = []
    for _ in range(2):
        y, mean, invstd = hipconv.bn_fwd(xc, None, bc, None, None, None, True, layout, 0.1, 1e-3)
        dx, _, _, dbeta = hipconv.bn_bwd(dyc, xc, y, None, mean, invstd, True, False, layout)
        outs.append((y, mean, invstd, dx, dbeta))
    torch.cuda.synchronize()
    close = lambda a, b, t: float((a.cpu().double() - b).abs().max()) < t * (float(b.abs().max()) + 1e-3)
    y, mean, invstd, dx, dbeta = outs[0]
    assert close(cut(y), y64.detach(), 1e-5) and close(cut(dx), x64.grad, 1e-4) and close(dbeta, be64.grad, 1e-5)
    assert close(mean, x.double().mean(dim=(0, 2, 3)), 1e-6)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    if layout == 1:      # the ReLU decisions as a byte per four elements: the same backward pass, bit for bit, without reading y
        y2, mean2, invstd2, mask = hipconv.bn_fwd(xc, None, bc, None, None, None, True, layout, 0.1, 1e-3, want_mask=True)
        parts_a = torch.zeros((hipconv.bn_bwd_splits(xc, layout), c), device="cuda")
        parts_b = torch.zeros_like(parts_a)
        ra = hipconv.bn_bwd(dyc, xc, y2, None, mean2, invstd2, True, True, layout, dxsum=parts_a)
        rb = hipconv.bn_bwd(dyc, xc, None, None, mean2, invstd2, True, True, layout, dxsum=parts_b, mask=mask)
        torch.cuda.synchronize()
        assert torch.equal(y2, y) and mask.dtype == torch.uint8 and tuple(mask.shape) == (n, c, 60)
        bits = torch.stack([(mask >> e) & 1 for e in range(4)], dim=-1).reshape(n, c, 15, 16).bool()
        assert torch.equal(bits, y2 > 0)
        for a, b in zip(ra, rb):
            assert torch.equal(a, b)
        assert torch.equal(parts_a, parts_b)


def test_head_conv1x1_pair_backward():
    """The two heads' 1x1 backward in one pass over their shared input == the two separate calls (the second accumulating)."""
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(5)
    n = 37
    x = pad16(torch.randn(n, 128, 15, 15, generator=g)).cuda()
    w1, w2 = (torch.randn(4, 128, 1, 1, generator=g) / 11).cuda(), (torch.randn(2, 128, 1, 1, generator=g) / 11).cuda()
    dy1, dy2 = torch.randn(n, 4, 15, 15, generator=g).cuda(), torch.randn(n, 2, 15, 15, generator=g).cuda()
    dx, dwa, dba = hipconv.conv1x1_bwd(x, w1, dy1, hipconv.ROWS16)
    dx, dwb, dbb = hipconv.conv1x1_bwd(x, w2, dy2, hipconv.ROWS16, dx=dx)
    px, pwa, pba, pwb, pbb = hipconv.conv1x1_bwd_pair(x, w1, dy1, w2, dy2, hipconv.ROWS16)
    torch.cuda.synchronize()
    assert torch.equal(pwa, dwa) and torch.equal(pwb, dwb) and torch.equal(pba, dba) and torch.equal(pbb, dbb)
    assert float((px - dx).abs().max()) < 1e-5 * float(dx.abs().max()) and float(px[..., 15].abs().max()) == 0.0


@pytest.mark.parametrize("n,c,co,hw,layout", [(9, 128, 4, 15, 1), (9, 128, 2, 15, 1), (5, 128, 4, 15, 0), (6, 256, 4, 8, 0),
                                              (6, 256, 2, 8, 0), (3, 64, 4, 15, 0), (130, 128, 4, 15, 1)])
def test_head_conv1x1_forward_backward(n, c, co, hw, layout):
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(n + c + co)
    x = torch.randn(n, c, hw, hw, generator=g)
    w = torch.randn(co, c, 1, 1, generator=g) / float(np.sqrt(c))
    b = torch.randn(co, generator=g)
    dy = torch.randn(n, co, hw, hw, generator=g)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, b64)
    y64.backward(dy.double())
    xc = (pad16(x) if layout == 1 else x).cuda()
    wc, bc, dyc = w.cuda(), b.cuda(), dy.cuda()
    y = hipconv.conv1x1_fwd(xc, wc, bc, layout)
    dx, dw, db = hipconv.conv1x1_bwd(xc, wc, dyc, layout)
    first = dx.clone()
    dx2, _, _ = hipconv.conv1x1_bwd(xc, wc, dyc, layout, dx=dx)           # accumulating call: 2 x
    torch.cuda.synchronize()
    assert tuple(y.shape) == (n, co, hw, hw)
    assert err(y, y64) < tol(y64)
    cut = (lambda t: t[..., :15]) if layout == 1 else (lambda t: t)
    if layout == 1:
        assert float(first[..., 15].abs().max()) == 0.0 and float(dx2[..., 15].abs().max()) == 0.0
    assert err(cut(first), x64.grad) < tol(x64.grad)
    assert dx2.data_ptr() == dx.data_ptr()
    assert err(cut(dx2), 2 * x64.grad) < 2 * tol(x64.grad)
    assert err(dw, w64.grad) < 5 * tol(w64.grad)
    assert err(db, b64.grad) < 5 * tol(b64.grad)


@pytest.mark.parametrize("n,k,nn", [(37, 900, 225), (5, 450, 1), (130, 256, 64), (512, 900, 225), (1, 128, 1), (64, 67, 19)])
def test_fully_connected_forward_backward(n, k, nn):
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(n + k + nn)
    x = torch.randn(n, k, generator=g)
    w = torch.randn(nn, k, generator=g) / float(np.sqrt(k))
    b = torch.randn(nn, generator=g)
    dy = torch.randn(n, nn, generator=g)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y64 = x64 @ w64.t() + b64
    y64.backward(dy.double())
    xc, wc, bc, dyc = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    y = hipconv.fc_fwd(xc, wc, bc)
    dx, dw, db = hipconv.fc_bwd(xc, wc, dyc)
    torch.cuda.synchronize()
    assert err(y, y64) < tol(y64)
    assert err(dx, x64.grad) < tol(x64.grad)
    assert err(dw, w64.grad) < 5 * tol(w64.grad)
    assert err(db, b64.grad) < 5 * tol(b64.grad)


def test_dropout_mask_is_a_stateless_hash():
    from alphapig_amd import hipconv
    x = torch.randn(300, 900).cuda()
    y = hipconv.dropout(x, 0.5, 7, 3)
    y2 = hipconv.dropout(x, 0.5, 7, 3)
    other_step = hipconv.dropout(x, 0.5, 7, 4)
    other_seed = hipconv.dropout(x, 0.5, 8, 3)
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    kept = y != 0
    assert torch.equal(y[kept], (x * 2.0)[kept])                         # kept elements scaled by 1 / keep
    frac = float(kept.float().mean())
    assert abs(frac - 0.5) < 0.01
    for o in (other_step, other_seed):
        agree = float(((o != 0) == kept).float().mean())
        assert 0.45 < agree < 0.55                                       # independent masks
    # the backward pass is the same map on the gradient
    dy = torch.randn(300, 900).cuda()
    dx = hipconv.dropout(dy, 0.5, 7, 3)
    assert torch.equal(dx != 0, kept)
    y8 = hipconv.dropout(x, 0.8, 1, 1)
    assert abs(float((y8 != 0).float().mean()) - 0.8) < 0.01
    assert torch.equal(hipconv.dropout(x, 1.0, 1, 1), x)


@pytest.mark.parametrize("n,hw", [(1, 225), (37, 225), (512, 225), (9, 64)])
def test_policy_value_loss_head(n, hw):
    from alphapig_amd import hipconv
    g = torch.Generator().manual_seed(n + hw)
    torch.manual_seed(n + hw)
    logits = torch.randn(n, hw, generator=g) * 3.0
    u = torch.randn(n, generator=g)
    pi = torch.distributions.Dirichlet(torch.ones(hw) * 0.3).sample((n,))
    pi[0] = 0.0
    pi[0, 5] = 1.0                                                        # one-hot target (temperature -> 0)
    z = torch.randint(0, 2, (n,), generator=g).float() * 2 - 1
    l64, u64 = logits.double().requires_grad_(True), u.double().requires_grad_(True)
    logp = F.log_softmax(l64, dim=1)
    v = torch.tanh(u64)
    vl = ((z.double() - v) ** 2).mean()
    pl = (-(logp * pi.double()).sum(dim=1)).mean()
    ent = (-(logp.exp() * logp).sum(dim=1)).mean()
    (vl + pl).backward()
    out = hipconv.pv_loss(logits.cuda(), u.cuda(), pi.cuda(), z.cuda(), grads=True, outputs=True)
    torch.cuda.synchronize()
    l3 = out["loss3"].cpu().double()
    assert abs(float(l3[0] - vl)) < 1e-5 * (1 + float(vl))
    assert abs(float(l3[1] - pl)) < 1e-5 * (1 + float(pl))
    assert abs(float(l3[2] - ent)) < 1e-5 * (1 + float(ent))
    assert err(out["dlogits"], l64.grad) < tol(l64.grad, 1e-4)
    assert err(out["dvlogit"], u64.grad) < tol(u64.grad, 1e-4)
    assert err(out["probs"], logp.exp()) < 1e-6
    assert err(out["values"], v) < 1e-6
    # inference form: outputs only
    out2 = hipconv.pv_loss(logits.cuda(), u.cuda(), outputs=True)
    assert set(out2) == {"probs", "values"} and torch.equal(out2["probs"], out["probs"])


def test_layout_copies_round_trip():
    from alphapig_amd import hipconv
    x = torch.randn(5, 128, 15, 15).cuda()
    r = hipconv.to_rows16(x)
    assert tuple(r.shape) == (5, 128, 15, 16)
    assert torch.equal(r[..., :15], x) and float(r[..., 15].abs().max()) == 0.0
    assert torch.equal(hipconv.from_rows16(r), x)
    y = torch.randn(3, 7, 15, 15).cuda()
    a = y.clone()
    hipconv.add_(a, x[:3, :7].contiguous())
    assert torch.allclose(a, y + x[:3, :7])


def _problem(kind, side, n, blocks, seed):
    from alphapig_amd import weights
    rs = np.random.RandomState(seed)
    prm = weights.init_params(kind, side, side, 9, blocks, 128, seed=seed + 1, style="bench")
    states = (rs.rand(n, 9, side, side) > 0.6).astype(np.float32)
    pis = rs.dirichlet(np.ones(side * side), size=n).astype(np.float32)
    zs = rs.choice([-1.0, 1.0], size=n).astype(np.float32)
    return prm, states, pis, zs


def _hip_masks(seed, keep=0.5):
    """mask_fn for the comparator: the keep masks of the HIP dropout kernel (stream 2 * step + which)."""
    from alphapig_amd import hipconv

    def fn(which, step, shape):
        ones = torch.ones(tuple(shape), dtype=torch.float32, device="cuda")
        return (hipconv.dropout(ones, keep, seed, 2 * step + which) != 0).cpu()
    return fn


# conv biases sit in front of a BatchNorm: their true gradient is exactly zero and what either implementation
# computes is rounding noise -- compared against the scale of the weights' gradient instead of their own
def _compare_grads(got, ref, rel):
    worst = {}
    for k, g64 in ref.items():
        g = got[k].astype(np.float64)
        scale = np.abs(g64).max()
        if k.endswith("_bias") and not k.startswith("fc_"):
            scale = np.abs(ref[k[:-5] + "_weight"]).max()
        worst[k] = np.abs(g - g64).max() / (scale + 1e-12)
    bad = {k: v for k, v in worst.items() if v > rel}
    assert not bad, bad


@pytest.mark.parametrize("kind,side,n,blocks", [("resnet", 15, 24, 2), ("resnet", 15, 70, 1), ("simple", 8, 12, 0),
                                                ("resnet", 8, 10, 2)])
def test_trainer_gradients_match_float64_autograd(kind, side, n, blocks):
    """Every gradient of one forward + backward of the HIP trainer (dropout on) against float64 autograd of the same
    graph with the same dropout masks and the same ReLU decisions (see torch_trainer.py).  (resnet, 15): the padded-row
    trunk on the Winograd kernels (70 boards: weight gradient through the Winograd domain); (simple, 8) and
    (resnet, 8): dense tensors on the direct kernels."""
    from alphapig_amd.train import HipTrainer
    from torch_trainer import TorchTrainer
    prm, states, pis, zs = _problem(kind, side, n, blocks, seed=side + n)
    tr = HipTrainer(prm, kind, n_blocks=blocks, batch_size=n, dropout=0.5, seed=5)
    assert tr.rows16 == (kind == "resnet" and side == 15)
    loss3 = tr.loss_and_grads(states, pis, zs, keep_tape=True).cpu().numpy().astype(np.float64)
    got = tr.get_grads()
    masks = {k: v.cpu() for k, v in tr.relu_masks().items()}
    ref = TorchTrainer(prm, kind, n_blocks=blocks, batch_size=n, device="cpu", dtype=torch.float64, dropout=0.5,
                       mask_fn=_hip_masks(5), relu_masks=masks)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    loss, ent = ref.loss(t64(states), t64(pis), t64(zs), train=True)
    loss.backward()
    assert abs(loss3[0] + loss3[1] - float(loss)) < 2e-5 * (1 + abs(float(loss)))
    assert abs(loss3[2] - float(ent)) < 2e-5 * (1 + abs(float(ent)))
    assert set(got) == set(ref.train_names)
    _compare_grads(got, ref.grads(), 1e-4)
    # the masks are the forward pass's own decisions: the plain-ReLU float64 graph agrees on all but a handful
    plain = TorchTrainer(prm, kind, n_blocks=blocks, batch_size=n, device="cpu", dtype=torch.float64, dropout=0.5,
                         mask_fn=_hip_masks(5))
    seen = {}
    plain._relu = lambda x, name: seen.setdefault(name, torch.relu(x))
    plain.loss(t64(states), t64(pis), t64(zs), train=True)
    for name, m in masks.items():
        flips = int(((seen[name] > 0) != m).sum())
        assert flips <= 2e-5 * m.numel() + 3, (name, flips)
    # moving statistics took the same step
    new, new_ref = tr.get_params(), ref.get_params()
    for k in tr.stat_names:
        np.testing.assert_allclose(new[k], new_ref[k], rtol=2e-4, atol=2e-5, err_msg=k)
    for k in tr.fixed_gamma_names:
        assert np.all(new[k] == 1.0), k
    tr.close()


def test_trainer_steps_match_the_comparator():
    """Three optimiser steps (Adam on the HIP kernel, dropout on) against the PyTorch comparator in float32 on the
    GPU with the same dropout masks and, per step, the ReLU decisions of the HIP forward pass: losses and parameters."""
    from alphapig_amd.train import HipTrainer
    from torch_trainer import TorchTrainer
    prm, states, pis, zs = _problem("resnet", 15, 24, 2, seed=1)
    tr = HipTrainer(prm, "resnet", n_blocks=2, batch_size=24, dropout=0.5, seed=3)
    ref = TorchTrainer(prm, "resnet", n_blocks=2, batch_size=24, device="cuda", dropout=0.5, mask_fn=_hip_masks(3))
    got_l, ref_l = [], []
    for _ in range(3):
        got_l.append(tr.train_step(states, pis, zs, 1e-3, keep_tape=True))
        ref.relu_masks = tr.relu_masks()
        ref_l.append(ref.train_step(states, pis, zs, 1e-3))
    np.testing.assert_allclose(got_l, ref_l, rtol=3e-4)
    a, b = tr.get_params(), ref.get_params()
    # Adam's first updates are ~lr * sign(g): where |g| is at the rounding-noise level the two implementations may
    # step in opposite directions, so a few elements per mille differ by up to 2 * lr * steps; the rest agrees
    for k in ("convA1_weight", "convB2_weight", "bnA1_gamma", "bnB2_beta", "res_conv1_weight", "bnA1_moving_var",
              "bnB2_moving_mean", "fc_3_1_1_weight", "fc_3_2_1_weight", "conv3_1_1_weight", "conv3_2_1_beta"):
        d = np.abs(a[k] - b[k])
        assert int((d > 2e-4).sum()) <= max(8, 0.01 * d.size), k
        assert float(d.max()) < 2 * 1e-3 * 3 + 2e-4, k
    tr.close()


@pytest.mark.parametrize("kind,side,n,blocks", [("resnet", 15, 70, 3), ("resnet", 15, 13, 2), ("resnet", 8, 20, 2), ("simple", 8, 20, 0)])
def test_training_steps_are_the_same_bits_on_every_run(kind, side, n, blocks):
    """No atomics anywhere in the step (BatchNorm sums, bias gradients, both weight-gradient kernels add per-slice partial
    sums in index order; dropout masks are a hash of (seed, step, element)): two trainers started from the same parameters
    hold identical parameters, moments and losses after three optimiser steps (policy_value_net_mxnet.py:282-299)."""
    from alphapig_amd.train import HipTrainer
    prm, states, pis, zs = _problem(kind, side, n, blocks, seed=9)
    runs = []
    for _ in range(2):
        tr = HipTrainer(prm, kind, n_blocks=blocks, batch_size=n, dropout=0.5, seed=5)
        losses = [tr.train_step(states, pis, zs, 2e-3) for _ in range(3)]
        runs.append((losses, tr.get_params(), {k: v.cpu().numpy() for k, v in tr.m.items()}))
        tr.close()
    assert runs[0][0] == runs[1][0]
    for k in runs[0][1]:
        np.testing.assert_array_equal(runs[0][1][k], runs[1][1][k], err_msg=k)
    for k in runs[0][2]:
        np.testing.assert_array_equal(runs[0][2][k], runs[1][2][k], err_msg=k)


def test_uploaded_mini_batch_is_the_same_step():
    """HipTrainer.upload: a policy_update's epochs train on one device copy of their mini-batch (train_mxnet.py:201-207 feeds
    the same arrays to every epoch) -- the same losses and parameters as steps fed from the host arrays."""
    from alphapig_amd.train import HipTrainer
    prm, states, pis, zs = _problem("resnet", 15, 40, 2, seed=12)
    a = HipTrainer(prm, "resnet", n_blocks=2, batch_size=40, dropout=0.5, seed=7)
    b = HipTrainer(prm, "resnet", n_blocks=2, batch_size=40, dropout=0.5, seed=7)
    batch = b.upload(states, pis, zs)
    for _ in range(3):
        assert a.train_step(states, pis, zs, 1e-3) == b.train_step(batch, None, None, 1e-3)
    pa, pb = a.get_params(), b.get_params()
    for k in pa:
        np.testing.assert_array_equal(pa[k], pb[k], err_msg=k)
    a.close()
    b.close()


def test_net_train_step_updates_the_selfplay_evaluator():
    """PolicyValueNet.train_step (policy_value_net_mxnet.py:282-299): one HIP optimiser step, then the evaluator
    answers with the new weights; the loss falls over a few steps on a fixed batch."""
    from alphapig_amd.policy_value_net import PolicyValueNet
    prm, states, pis, zs = _problem("resnet", 15, 32, 2, seed=9)
    net = PolicyValueNet(15, 15, batch_size=32, n_blocks=2, n_filter=128, model_params=prm)
    p0, _ = net.policy_value(states)
    losses = []
    for _ in range(8):
        loss, ent = net.train_step(states, pis, zs, 2e-3)
        assert loss.shape == (1,) and ent.shape == (1,) and np.isfinite(loss[0]) and np.isfinite(ent[0])
        losses.append(float(loss[0]))
    p1, _ = net.policy_value(states)
    assert np.abs(p1 - p0).max() > 1e-4
    assert min(losses[-3:]) < losses[0]
    # the evaluator holds exactly the trainer's weights
    for k, v in net._trainer.get_params().items():
        np.testing.assert_array_equal(net.params()[k], v, err_msg=k)
    net.close()


def test_policy_update_on_the_hip_trainer():
    """train_mxnet.py:194-240 on the product trainer, KL monitor through the self-play evaluator."""
    from alphapig_amd.train import HipTrainer, policy_update
    prm, states, pis, zs = _problem("resnet", 15, 32, 1, seed=4)
    tr = HipTrainer(prm, "resnet", n_blocks=1, batch_size=32, dropout=0.5, seed=1)
    batch = [(states[i], pis[i], zs[i]) for i in range(32)]
    mult, first = 1.0, None
    for _ in range(4):
        loss, ent, kl, mult = policy_update(tr, batch, learn_rate=5e-3, lr_multiplier=mult, epochs=3, kl_targ=0.02)
        first = loss if first is None else first
        assert np.isfinite(loss) and np.isfinite(ent) and kl >= -1e-6
    assert loss < first
    assert 0.05 / 1.5 <= mult <= 20 * 1.5
    tr.close()


@pytest.mark.parametrize("kind,side,blocks,nf", [("resnet", 15, 3, 128), ("simple", 8, 0, 128), ("resnet", 8, 2, 64)])
def test_device_side_weight_refresh_equals_host_load(kind, side, blocks, nf):
    """apz_load_weights_dev (BatchNorm folding, direct / Winograd / head / FC packing as kernels on the trainer's
    device tensors) against apz_load_weights from the host: the two evaluators answer alike, and the device-loaded one
    hands back the same parameter table."""
    from alphapig_amd import weights
    from alphapig_amd.policy_value_net import PolicyValueNet
    rs = np.random.RandomState(3)
    prm = weights.init_params(kind, side, side, 9, blocks, nf, seed=5, style="bench")
    other = weights.init_params(kind, side, side, 9, blocks, nf, seed=6, style="bench")
    states = (rs.rand(20, 9, side, side) > 0.6).astype(np.float32)
    host = PolicyValueNet(side, side, batch_size=32, n_blocks=blocks, n_filter=nf, model_params=prm, net_kind=kind)
    dev = PolicyValueNet(side, side, batch_size=32, n_blocks=blocks, n_filter=nf, model_params=other, net_kind=kind)
    p_other, _ = dev.policy_value(states)
    tensors = {k: torch.tensor(np.ascontiguousarray(v, dtype=np.float32), device="cuda") for k, v in prm.items()}
    dev.load_device_params(tensors, torch.cuda.current_stream().cuda_stream)
    p_host, v_host = host.policy_value(states)
    p_dev, v_dev = dev.policy_value(states)
    assert np.abs(p_other - p_host).max() > 1e-4                     # the refresh changed something
    # same double-precision maps; the device code contracts a*b+c into fma, so single results may differ by an ulp
    np.testing.assert_allclose(p_dev, p_host, rtol=0, atol=2e-7)
    np.testing.assert_allclose(v_dev, v_host, rtol=0, atol=2e-6)
    got = dev.params()
    assert set(got) == set(prm)
    for k in prm:
        np.testing.assert_array_equal(got[k], prm[k].astype(np.float32), err_msg=k)
    # a missing or mis-sized tensor is refused like on the host path
    bad = dict(tensors)
    bad.pop("fc_3_1_1_bias")
    with pytest.raises(Exception):
        dev.load_device_params(bad)
    host.close()
    dev.close()
