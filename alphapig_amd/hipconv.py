"""The training graph's operators on the hand-written HIP kernels, one thin function per
C-ABI entry point of include/alphapig_hip.h: 3x3 convolution forward / data gradient / weight gradient (direct MFMA
kernel, or the self-play path's fused Winograd kernel for the 128 -> 128 trunk shape), bias gradient, training-mode
BatchNorm (+ residual) (+ ReLU) forward and backward, the 1x1 head convolutions, FullyConnected, Dropout, the
policy-value loss, layout copies and Adam.

There is no autograd here and no PyTorch arithmetic: alphapig_amd/train.py calls these functions in the order of the
reference's graph (policy_value_net_mxnet.py:41-102, :173-212) and then in reverse.  Tensors are torch CUDA float32
tensors used as device buffers (allocation + data pointers) on torch's current stream; without the HIP library every
function raises.
"""
import ctypes as C

from . import _native
from ._native import ApzConfig

_ENGINES = {}

DENSE, ROWS16 = 0, 1       # APZ_LAYOUT_*: dense NCHW / the trunk's padded rows [n][C][15][16] (pad column zero)


def _engine(h, w, device_index):
    key = (h, w, device_index)
    if key not in _ENGINES:
        L = _native.hip()
        cfg = ApzConfig(h, w, 9, 128, 0, 0, 1, device_index)
        hnd = L.apz_create(C.byref(cfg))
        if not hnd:
            raise RuntimeError("apz_create failed: %s" % L.apz_last_error().decode())
        _ENGINES[key] = hnd
    return _ENGINES[key]


def _torch():
    import torch
    return torch


def _ck(L, rc):
    if rc < 0:
        raise RuntimeError("%s (code %d)" % (L.apz_last_error().decode(), rc))


def _ctx(x, layout=DENSE):
    """-> (library, engine handle for x's board size and device, stream pointer)"""
    torch = _torch()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("HIP operators take contiguous float32 device tensors")
    h, w = int(x.shape[2]), int(x.shape[3])
    if layout == ROWS16:
        if (h, w) != (15, 16):
            raise ValueError("padded-row tensors are [n][C][15][16]")
        w = 15
    return _native.hip(), _engine(h, w, x.device.index or 0), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _empty(shape, like):
    return _torch().empty(shape, dtype=_torch().float32, device=like.device)


def is_trunk_shape(weight, x, layout):
    """128 -> 128 channels on 15x15 boards: the shape of the self-play path's fused Winograd kernel."""
    board = (15, 16) if layout == ROWS16 else (15, 15)
    return tuple(weight.shape) == (128, 128, 3, 3) and tuple(x.shape[2:]) == board


# ---- 3x3 convolution ------------------------------------------------------------------------------------------------
def wino_pack_many(weights, out=None):
    """weights [count][128][128][3][3] (one contiguous tensor) -> [count][2][apz_wino_packed_size()]: every layer's
    transformed weights for the forward convolution ([:, 0]) and the data-gradient convolution ([:, 1]), one launch."""
    torch = _torch()
    if not (weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous() and
            tuple(weights.shape[1:]) == (128, 128, 3, 3)):
        raise ValueError("wino_pack_many takes a contiguous float32 device tensor [count][128][128][3][3]")
    L = _native.hip()
    hnd = _engine(15, 15, weights.device.index or 0)
    count = int(weights.shape[0])
    if out is None:
        out = _empty((count, 2, L.apz_wino_packed_size()), weights)
    stream = C.c_void_p(torch.cuda.current_stream(weights.device).cuda_stream)
    _ck(L, L.apz_wino_pack_many(hnd, weights.data_ptr(), count, out.data_ptr(), stream))
    return out


def _conv3x3_run(x, weight, bias, layout, flip, resid, relu, upk=None):
    """conv(x, W) (flip: the data-gradient convolution with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]) + bias + resid.
    upk: the layer's transformed weights in that orientation when the caller packed them already (wino_pack_many)."""
    L, hnd, stream = _ctx(x, layout)
    co, ci = int(weight.shape[0]), int(weight.shape[1])
    cin_p, cout_p = (co, ci) if flip else (ci, co)
    n = int(x.shape[0])
    if int(x.shape[1]) != cin_p:
        raise ValueError("channel mismatch")
    y = _empty((n, cout_p) + tuple(x.shape[2:]), x)
    # the Winograd kernel works on board pairs x channel halves: from 192 boards on it fills the chip and beats the
    # direct kernel (measured round 1: 24.2 vs 31.0 ms per training step at batch 512, 12.3 vs 11.8 ms at 128);
    # padded-row tensors are its native layout and always take it
    if is_trunk_shape(weight, x, layout) and (layout == ROWS16 or n >= 192):
        if upk is None:
            upk = _empty((L.apz_wino_packed_size(),), x)
            _ck(L, L.apz_wino_pack(hnd, weight.data_ptr(), int(flip), upk.data_ptr(), stream))
        if resid is not None and layout != ROWS16:
            _ck(L, L.apz_wino_conv_add(hnd, x.data_ptr(), upk.data_ptr(), _ptr(bias), None, y.data_ptr(), n, 0, layout, stream))
            add_(y, resid)
            if relu:
                raise ValueError("relu after a dense residual is not provided")
        else:
            _ck(L, L.apz_wino_conv_add(hnd, x.data_ptr(), upk.data_ptr(), _ptr(bias), _ptr(resid), y.data_ptr(), n, int(relu),
                                       layout, stream))
        return y
    if layout != DENSE:
        raise ValueError("padded-row layout: the 128 -> 128 trunk shape only")
    wpk = _empty((L.apz_conv3x3_packed_size(cin_p, cout_p),), x)
    _ck(L, L.apz_conv3x3_pack(hnd, weight.data_ptr(), ci, co, int(flip), wpk.data_ptr(), stream))
    _ck(L, L.apz_conv3x3_fwd(hnd, x.data_ptr(), wpk.data_ptr(), _ptr(bias), y.data_ptr(), n, cin_p, cout_p,
                             int(relu and resid is None), stream))
    if resid is not None:
        if relu:
            raise ValueError("relu after a dense residual is not provided")
        add_(y, resid)
    return y


def conv3x3_fwd(x, weight, bias=None, layout=DENSE, relu=False, upk=None):
    """y = conv2d(x, weight, bias, padding=1).  Dense: boards 15x15 or 8x8, C_out in {64, 128, 256}."""
    return _conv3x3_run(x, weight, bias, layout, False, None, relu, upk)


def conv3x3_fwd_stats(x, weight, bias, upk=None):
    """The trunk shape in the padded-row layout, for a BatchNorm that follows: -> (y, stats); stats [128][n][2] float64 =
    per (channel, board) the sum and the sum of squares of y, taken in the convolution's epilogue (bn_fwd(stats=...)
    then makes no pass of its own over y for them)."""
    torch = _torch()
    L, hnd, stream = _ctx(x, ROWS16)
    if not is_trunk_shape(weight, x, ROWS16):
        raise ValueError("conv3x3_fwd_stats: the 128 -> 128 trunk shape in the padded-row layout")
    n = int(x.shape[0])
    if upk is None:
        upk = _empty((L.apz_wino_packed_size(),), x)
        _ck(L, L.apz_wino_pack(hnd, weight.data_ptr(), 0, upk.data_ptr(), stream))
    y = _empty(tuple(x.shape), x)
    stats = torch.empty((128, n, 2), dtype=torch.float64, device=x.device)
    _ck(L, L.apz_wino_conv_stats(hnd, x.data_ptr(), upk.data_ptr(), _ptr(bias), y.data_ptr(), stats.data_ptr(), n, stream))
    return y, stats


def conv3x3_dgrad(dy, weight, layout=DENSE, add=None, upk=None):
    """dx = conv2d_input(dy, weight) (+ add: another gradient of the same tensor, e.g. the skip connection's).
    Needs C_in in {64, 128, 256} (the first layer's input gradient is never wanted)."""
    return _conv3x3_run(dy, weight, None, layout, True, add, False, upk)


def conv3x3_wgrad(x, dy, layout=DENSE):
    """dw [C_out][C_in][3][3] = sum over boards of x (*) dy."""
    L, hnd, stream = _ctx(x, layout)
    n, ci, co = int(x.shape[0]), int(x.shape[1]), int(dy.shape[1])
    dw = _empty((co, ci, 3, 3), x)
    # through the Winograd domain (3.6x fewer MFMAs; wgrad_wino3_kernel beats the direct kernel from 8 boards on: 14 against
    # 35 us at 8 boards, 22 against 48 at 32 -- round 3's kernel needed 64 boards to cover its per-slice partial sums)
    if layout == ROWS16 and (ci, co) == (128, 128):
        _ck(L, L.apz_wgrad_wino(hnd, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, stream))
    else:
        _ck(L, L.apz_conv3x3_wgrad(hnd, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, ci, co, layout, stream))
    return dw


def bias_grad(dy, layout=DENSE):
    """db[c] = sum over boards and cells of dy (pad cells of a padded-row gradient are zero)."""
    L, hnd, stream = _ctx(dy, layout)
    db = _empty((int(dy.shape[1]),), dy)
    _ck(L, L.apz_bias_grad(hnd, dy.data_ptr(), db.data_ptr(), int(dy.shape[0]), int(dy.shape[1]), layout, stream))
    return db


def add_(y, x):
    """y += x"""
    L, hnd, stream = _ctx(y, DENSE if y.shape[3] != 16 else ROWS16)
    if y.shape != x.shape:
        raise ValueError("shape mismatch")
    _ck(L, L.apz_add(hnd, y.data_ptr(), x.data_ptr(), y.numel(), stream))
    return y


# ---- BatchNorm ------------------------------------------------------------------------------------------------------
def bn_fwd(x, gamma, beta, run_mean, run_var, resid=None, relu=True, layout=DENSE, momentum=0.1, eps=1e-3, stats=None,
           want_mask=False):
    """y = act(batch_norm(x) (+ resid)) with batch statistics; gamma None = fixed at 1 (the reference's fix_gamma layers).
    run_mean / run_var are updated in place (momentum = weight of the new batch value).  -> (y, mean, invstd)
    stats: conv3x3_fwd_stats' second result for this x.  want_mask (padded rows): -> (y, mean, invstd, mask), mask = the
    ReLU decisions as uint8 [n][C][60] (4 bits per byte), for bn_bwd(mask=...)."""
    L, hnd, stream = _ctx(x, layout)
    n, c = int(x.shape[0]), int(x.shape[1])
    y = _empty(tuple(x.shape), x)
    mean, invstd = _empty((c,), x), _empty((c,), x)
    if stats is not None and (tuple(stats.shape) != (c, n, 2) or stats.dtype != _torch().float64 or not stats.is_contiguous()):
        raise ValueError("stats: contiguous float64 [C][n][2]")
    mask = _torch().empty((n, c, 60), dtype=_torch().uint8, device=x.device) if want_mask else None
    _ck(L, L.apz_bn_fwd_stats(hnd, x.data_ptr(), _ptr(resid), _ptr(gamma), beta.data_ptr(), _ptr(run_mean), _ptr(run_var),
                              y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(stats), _ptr(mask), n, c, layout, int(relu),
                              momentum, eps, stream))
    return (y, mean, invstd, mask) if want_mask else (y, mean, invstd)


def bn_bwd(dy, x, y, gamma, mean, invstd, relu=True, want_dres=False, layout=DENSE, dxsum=None, mask=None):
    """-> (dx, dres or None, dgamma, dbeta); y is the forward output (the ReLU mask) -- or None when mask (bn_fwd's
    want_mask result) carries the ReLU decisions.
    dxsum: a [bn_bwd_splits(x, layout)][C] view (row stride >= C) of a float32 matrix that receives the per-split column
    sums of dx -- colsum() of it is the bias gradient of the convolution in front (bias_parts() hands such views out)."""
    L, hnd, stream = _ctx(x, layout)
    n, c = int(x.shape[0]), int(x.shape[1])
    dx = _empty(tuple(x.shape), x)
    dres = _empty(tuple(x.shape), x) if want_dres else None
    dgamma, dbeta = _empty((c,), x), _empty((c,), x)
    ld = 0
    if dxsum is not None:
        if tuple(dxsum.shape) != (bn_bwd_splits(x, layout), c) or dxsum.stride(1) != 1:
            raise ValueError("dxsum: a [splits][C] view with unit column stride")
        ld = int(dxsum.stride(0))
    if mask is not None and (tuple(mask.shape) != (n, c, 60) or mask.dtype != _torch().uint8 or not mask.is_contiguous()):
        raise ValueError("mask: contiguous uint8 [n][C][60]")
    _ck(L, L.apz_bn_bwd(hnd, dy.data_ptr(), x.data_ptr(), _ptr(y), _ptr(mask), _ptr(gamma), mean.data_ptr(), invstd.data_ptr(),
                        dx.data_ptr(), _ptr(dres), dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dxsum), ld, n, c, layout, int(relu),
                        stream))
    return dx, dres, dgamma, dbeta


def bn_bwd_splits(x, layout=DENSE):
    """rows of bn_bwd's dxsum matrix for tensors of x's shape"""
    L, hnd, _ = _ctx(x, layout)
    s = L.apz_bn_bwd_splits(hnd, int(x.shape[0]), int(x.shape[1]), layout)
    _ck(L, s)
    return int(s)


def colsum(m, scale=1.0):
    """out[j] = scale * sum_i m[i][j] (fixed order, double accumulation); m: contiguous [rows][cols] float32"""
    torch = _torch()
    if not (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.dim() == 2):
        raise ValueError("colsum takes a contiguous float32 device matrix")
    L = _native.hip()
    hnd = _any_engine(m.device.index or 0)
    out = _empty((int(m.shape[1]),), m)
    stream = C.c_void_p(torch.cuda.current_stream(m.device).cuda_stream)
    _ck(L, L.apz_colsum(hnd, m.data_ptr(), out.data_ptr(), int(m.shape[0]), int(m.shape[1]), scale, stream))
    return out


# ---- heads ----------------------------------------------------------------------------------------------------------
def conv1x1_fwd(x, weight, bias=None, layout=DENSE):
    """y [n][CO][H][W] (dense) = W x + b, CO <= 8."""
    L, hnd, stream = _ctx(x, layout)
    n, c, co = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
    h, w = int(x.shape[2]), (15 if layout == ROWS16 else int(x.shape[3]))
    y = _empty((n, co, h, w), x)
    _ck(L, L.apz_conv1x1_fwd(hnd, x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(), n, c, co, layout, stream))
    return y


def conv1x1_bwd(x, weight, dy, layout=DENSE, dx=None):
    """-> (dx, dw, db).  dx given: the gradient is ADDED to it (the second head); else a new tensor in x's layout."""
    L, hnd, stream = _ctx(x, layout)
    n, c, co = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
    acc = dx is not None
    if dx is None:
        dx = _empty(tuple(x.shape), x)
    dw, db = _empty(tuple(weight.shape), x), _empty((co,), x)
    _ck(L, L.apz_conv1x1_bwd(hnd, x.data_ptr(), weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                             n, c, co, layout, int(acc), stream))
    return dx, dw, db


def conv1x1_bwd_pair(x, w1, dy1, w2, dy2, layout=DENSE):
    """Two 1x1 convolutions of the same input (the two heads): -> (dx, dw1, db1, dw2, db2), dx = both heads' input
    gradients added, in one pass over x."""
    L, hnd, stream = _ctx(x, layout)
    n, c, co1, co2 = int(x.shape[0]), int(x.shape[1]), int(w1.shape[0]), int(w2.shape[0])
    dx = _empty(tuple(x.shape), x)
    dw = _empty((co1 + co2, c), x)
    _ck(L, L.apz_conv1x1_bwd2(hnd, x.data_ptr(), w1.data_ptr(), dy1.data_ptr(), co1, w2.data_ptr(), dy2.data_ptr(), co2,
                               dx.data_ptr(), dw.data_ptr(), n, c, layout, 0, stream))
    return dx, dw[:co1].view(w1.shape), bias_grad(dy1, DENSE), dw[co1:].view(w2.shape), bias_grad(dy2, DENSE)


def _any_engine(device_index):
    """Operators that do not depend on the board (FullyConnected, Dropout, Adam) run on whichever engine this device
    already has (the net's own board size), else on a 15x15 one."""
    for (h, w, d), hnd in _ENGINES.items():
        if d == device_index:
            return hnd
    return _engine(15, 15, device_index)


def _ctx2(x):
    torch = _torch()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("HIP operators take contiguous float32 device tensors")
    return _native.hip(), _any_engine(x.device.index or 0), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)


def fc_fwd(x, weight, bias=None):
    """y [n][N] = x [n][K] W[N][K]^T + b"""
    L, hnd, stream = _ctx2(x)
    n, k, nn = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
    y = _empty((n, nn), x)
    _ck(L, L.apz_fc_fwd(hnd, x.data_ptr(), weight.data_ptr(), _ptr(bias), y.data_ptr(), n, k, nn, stream))
    return y


def fc_bwd(x, weight, dy):
    """-> (dx, dw, db)"""
    L, hnd, stream = _ctx2(x)
    n, k, nn = int(x.shape[0]), int(x.shape[1]), int(weight.shape[0])
    dx, dw, db = _empty((n, k), x), _empty((nn, k), x), _empty((nn,), x)
    _ck(L, L.apz_fc_bwd(hnd, x.data_ptr(), weight.data_ptr(), dy.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                        n, k, nn, stream))
    return dx, dw, db


def dropout(x, keep, seed, step):
    """y = x * mask / keep, mask = [hash(seed, step, element) < keep]; the same call on dy is the backward pass."""
    L, hnd, stream = _ctx2(x)
    y = _empty(tuple(x.shape), x)
    _ck(L, L.apz_dropout(hnd, x.data_ptr(), y.data_ptr(), x.numel(), keep, seed, step, stream))
    return y


def pv_loss(logits, vlogit, pi=None, z=None, grads=True, outputs=False):
    """The reference's loss head (policy_value_net_mxnet.py:180-193) on [n][H*W] logits and [n] value logits.
    -> dict with loss3 = (value loss, policy loss, entropy) [device, 3 floats], dlogits, dvlogit (grads),
    probs, values (outputs)."""
    L, _, stream = _ctx2(logits)
    n, hw = int(logits.shape[0]), int(logits.shape[1])
    side = int(round(hw ** 0.5))
    hnd = _engine(side, side, logits.device.index or 0)
    out = {}
    if pi is not None:
        out["loss3"] = _empty((3,), logits)
        if grads:
            out["dlogits"], out["dvlogit"] = _empty((n, hw), logits), _empty((n,), logits)
    if outputs:
        out["probs"], out["values"] = _empty((n, hw), logits), _empty((n,), logits)
    _ck(L, L.apz_pv_loss(hnd, logits.data_ptr(), vlogit.data_ptr(), _ptr(pi), _ptr(z), n, _ptr(out.get("loss3")),
                         _ptr(out.get("dlogits")), _ptr(out.get("dvlogit")), _ptr(out.get("probs")), _ptr(out.get("values")),
                         stream))
    return out


# ---- layout ---------------------------------------------------------------------------------------------------------
def to_rows16(x):
    """dense [n][C][15][15] -> padded rows [n][C][15][16] (pad column zero)"""
    L, hnd, stream = _ctx(x, DENSE)
    y = _empty((int(x.shape[0]), int(x.shape[1]), 15, 16), x)
    _ck(L, L.apz_layout_convert(hnd, x.data_ptr(), y.data_ptr(), int(x.shape[0]) * int(x.shape[1]), 1, stream))
    return y


def from_rows16(x):
    L, hnd, stream = _ctx(x, ROWS16)
    y = _empty((int(x.shape[0]), int(x.shape[1]), 15, 15), x)
    _ck(L, L.apz_layout_convert(hnd, x.data_ptr(), y.data_ptr(), int(x.shape[0]) * int(x.shape[1]), 0, stream))
    return y


# ---- Adam -----------------------------------------------------------------------------------------------------------
def adam_step(entries, lr_t, b1, b2, eps, rescale, device):
    """One launch over all tensors.  entries: [(w, grad, m, v, wd)] of float32 device tensors (apz_adam_step)."""
    import numpy as np
    torch = _torch()
    tab = np.zeros(len(entries), dtype=[("w", "u8"), ("g", "u8"), ("m", "u8"), ("v", "u8"), ("n", "i8"), ("wd", "f4"),
                                        ("pad", "i4")])
    for i, (w, g, m, v, wd) in enumerate(entries):
        if g.shape != w.shape or not g.is_contiguous():
            raise ValueError("gradient %d: shape / layout mismatch" % i)
        tab[i] = (w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), w.numel(), wd, 0)
    L = _native.hip()
    hnd = _any_engine(device.index or 0)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _ck(L, L.apz_adam_step(hnd, tab.ctypes.data_as(C.c_void_p), len(tab), lr_t, b1, b2, eps, rescale, stream))
