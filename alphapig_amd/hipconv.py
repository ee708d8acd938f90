"""3x3 convolution for the training graph on the hand-written HIP kernels:
forward, data gradient and weight gradient through include/alphapig_hip.h
(apz_conv3x3_pack / _fwd / _wgrad), wrapped as a torch.autograd.Function so that the rest of the
interim training graph (BatchNorm, ReLU, heads, loss: small element-wise work) can stay in
PyTorch while ~97 % of the training FLOPs run on this repository's kernels.

Tensors are torch CUDA float32, dense NCHW, used in place through their data pointers on torch's
current stream (PyTorch is the container, as in the self-play path).
"""
import ctypes as C

from . import _native
from ._native import ApzConfig

_ENGINES = {}


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


DENSE, ROWS16 = 0, 1       # APZ_LAYOUT_*: dense NCHW / the trunk's padded rows [n][C][15][16] (pad column zero)


def supported(x, weight, layout=DENSE):
    co, ci, kh, kw = weight.shape
    import torch
    ok = x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and kh == 3 and kw == 3
    if layout == ROWS16:
        return ok and (co, ci) == (128, 128) and tuple(x.shape[2:]) == (15, 16)
    return ok and co in (64, 128, 256) and tuple(x.shape[2:]) in ((15, 15), (8, 8))


def _wino(x, weight, layout=DENSE):
    """The trunk shape (128 -> 128 at 15x15) runs on the fused Winograd pair kernel of the self-play path once
    the batch fills the chip (one workgroup per pair of boards: >= 192 boards; measured 24.2 vs 31.0 ms per
    training step at batch 512, 12.3 vs 11.8 ms at batch 128).  APZ_TRAIN_CONV=direct / wino forces a path."""
    import os
    if layout == ROWS16:
        return True             # the padded-row layout IS the Winograd kernel's (no copies): always
    if tuple(weight.shape) != (128, 128, 3, 3) or tuple(x.shape[2:]) != (15, 15):
        return False
    mode = os.environ.get("APZ_TRAIN_CONV", "auto")
    return mode == "wino" or (mode != "direct" and x.shape[0] >= 192)


def _ck(L, rc):
    if rc < 0:
        raise RuntimeError("%s (code %d)" % (L.apz_last_error().decode(), rc))


def _function():
    import torch

    class HipConv3x3(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, weight, bias, layout):
            L = _native.hip()
            x = x.contiguous()
            weight = weight.contiguous()
            n, ci, h, w = x.shape
            co = weight.shape[0]
            ctx.layout = layout
            if layout == ROWS16:
                w = 15
            hnd = _engine(h, w, x.device.index or 0)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            y = torch.empty_like(x) if layout == ROWS16 else torch.empty((n, co, h, w), dtype=torch.float32, device=x.device)
            bptr = bias.contiguous().data_ptr() if bias is not None else None
            if _wino(x, weight, layout):
                upk = torch.empty(L.apz_wino_packed_size(), dtype=torch.float32, device=x.device)
                _ck(L, L.apz_wino_pack(hnd, weight.data_ptr(), 0, upk.data_ptr(), stream))
                _ck(L, L.apz_wino_conv(hnd, x.data_ptr(), upk.data_ptr(), bptr, y.data_ptr(), n, 0, layout, stream))
            else:
                wpk = torch.empty(L.apz_conv3x3_packed_size(ci, co), dtype=torch.float32, device=x.device)
                _ck(L, L.apz_conv3x3_pack(hnd, weight.data_ptr(), ci, co, 0, wpk.data_ptr(), stream))
                _ck(L, L.apz_conv3x3_fwd(hnd, x.data_ptr(), wpk.data_ptr(), bptr, y.data_ptr(), n, ci, co, 0, stream))
            ctx.save_for_backward(x, weight)
            ctx.has_bias = bias is not None
            return y

        @staticmethod
        def backward(ctx, dy):
            L = _native.hip()
            x, weight = ctx.saved_tensors
            dy = dy.contiguous()
            n, ci, h, w = x.shape
            co = weight.shape[0]
            layout = ctx.layout
            if layout == ROWS16:
                w = 15
            hnd = _engine(h, w, x.device.index or 0)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                if _wino(x, weight, layout):
                    upk = torch.empty(L.apz_wino_packed_size(), dtype=torch.float32, device=x.device)
                    _ck(L, L.apz_wino_pack(hnd, weight.data_ptr(), 1, upk.data_ptr(), stream))
                    dx = torch.empty_like(x)
                    _ck(L, L.apz_wino_conv(hnd, dy.data_ptr(), upk.data_ptr(), None, dx.data_ptr(), n, 0, layout, stream))
                elif ci in (64, 128, 256):
                    wpk = torch.empty(L.apz_conv3x3_packed_size(co, ci), dtype=torch.float32, device=x.device)
                    _ck(L, L.apz_conv3x3_pack(hnd, weight.data_ptr(), ci, co, 1, wpk.data_ptr(), stream))
                    dx = torch.empty_like(x)
                    _ck(L, L.apz_conv3x3_fwd(hnd, dy.data_ptr(), wpk.data_ptr(), None, dx.data_ptr(), n, co, ci, 0, stream))
                else:       # e.g. a 9-plane input that needs a gradient: rare, let torch do it
                    dx = torch.nn.grad.conv2d_input(x.shape, weight, dy, padding=1)
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                import os
                if layout == ROWS16 and n >= 64 and os.environ.get("APZ_TRAIN_WGRAD", "wino") != "direct":
                    _ck(L, L.apz_wgrad_wino(hnd, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, stream))
                else:
                    _ck(L, L.apz_conv3x3_wgrad(hnd, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, ci, co, layout, stream))
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = dy.sum(dim=(0, 2, 3))       # (pad columns of a padded-row gradient are zero)
            return dx, dw, db, None

    return HipConv3x3


_FN = None


def conv3x3(x, weight, bias=None, layout=DENSE):
    """y = conv2d(x, weight, bias, padding=1) on the HIP kernels, differentiable.  layout ROWS16: x and y are
    [n][128][15][16] with a zero pad column (the self-play kernels' activation layout; no copies)."""
    global _FN
    if _FN is None:
        _FN = _function()
    return _FN.apply(x, weight, bias, layout)


def _bn_function():
    import torch

    class HipBnAct(torch.autograd.Function):
        """Training-mode BatchNorm (+ residual) (+ ReLU) on apz_bn_fwd / apz_bn_bwd."""

        @staticmethod
        def forward(ctx, x, gamma, beta, resid, run_mean, run_var, relu, layout, momentum, eps):
            L = _native.hip()
            x = x.contiguous()
            n, c, h, w = x.shape
            hnd = _engine(h, 15 if layout == ROWS16 else w, x.device.index or 0)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            y = torch.empty_like(x)
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            invstd = torch.empty(c, dtype=torch.float32, device=x.device)
            rptr = resid.contiguous().data_ptr() if resid is not None else None
            _ck(L, L.apz_bn_fwd(hnd, x.data_ptr(), rptr, gamma.data_ptr() if gamma is not None else None, beta.data_ptr(),
                                run_mean.data_ptr(), run_var.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                n, c, layout, int(relu), momentum, eps, stream))
            ctx.save_for_backward(x, y, gamma if gamma is not None else beta, mean, invstd)
            ctx.cfg = (gamma is not None, resid is not None, bool(relu), layout)
            ctx.mark_non_differentiable(run_mean, run_var)
            return y

        @staticmethod
        def backward(ctx, dy):
            L = _native.hip()
            x, y, gamma, mean, invstd = ctx.saved_tensors
            has_gamma, has_res, relu, layout = ctx.cfg
            dy = dy.contiguous()
            n, c, h, w = x.shape
            hnd = _engine(h, 15 if layout == ROWS16 else w, x.device.index or 0)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            dx = torch.empty_like(x)
            dres = torch.empty_like(x) if has_res else None
            dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
            _ck(L, L.apz_bn_bwd(hnd, dy.data_ptr(), x.data_ptr(), y.data_ptr(), gamma.data_ptr() if has_gamma else None,
                                mean.data_ptr(), invstd.data_ptr(), dx.data_ptr(), dres.data_ptr() if has_res else None,
                                dgamma.data_ptr(), dbeta.data_ptr(), n, c, layout, int(relu), stream))
            return dx, (dgamma if has_gamma else None), dbeta, dres, None, None, None, None, None, None

    return HipBnAct


_BN = None


def bn_act(x, gamma, beta, run_mean, run_var, resid=None, relu=True, layout=DENSE, momentum=0.1, eps=1e-3):
    """act(batch_norm(x) (+ resid)) in training mode on the HIP kernels, differentiable; gamma None = fixed at 1.
    run_mean / run_var are updated in place (momentum = weight of the new batch value, as torch)."""
    global _BN
    if _BN is None:
        _BN = _bn_function()
    return _BN.apply(x, gamma, beta, resid, run_mean, run_var, relu, layout, momentum, eps)
