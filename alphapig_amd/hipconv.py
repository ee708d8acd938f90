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


def supported(x, weight):
    co, ci, kh, kw = weight.shape
    import torch
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and kh == 3 and kw == 3 and
            co in (64, 128, 256) and tuple(x.shape[2:]) in ((15, 15), (8, 8)))


def _wino(x, weight):
    """The trunk shape (128 -> 128 at 15x15) runs on the fused Winograd pair kernel of the self-play path once
    the batch fills the chip (one workgroup per pair of boards: >= 192 boards; measured 24.2 vs 31.0 ms per
    training step at batch 512, 12.3 vs 11.8 ms at batch 128).  APZ_TRAIN_CONV=direct / wino forces a path."""
    import os
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
        def forward(ctx, x, weight, bias):
            L = _native.hip()
            x = x.contiguous()
            weight = weight.contiguous()
            n, ci, h, w = x.shape
            co = weight.shape[0]
            hnd = _engine(h, w, x.device.index or 0)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            y = torch.empty((n, co, h, w), dtype=torch.float32, device=x.device)
            bptr = bias.contiguous().data_ptr() if bias is not None else None
            if _wino(x, weight):
                upk = torch.empty(L.apz_wino_packed_size(), dtype=torch.float32, device=x.device)
                _ck(L, L.apz_wino_pack(hnd, weight.data_ptr(), 0, upk.data_ptr(), stream))
                _ck(L, L.apz_wino_conv(hnd, x.data_ptr(), upk.data_ptr(), bptr, y.data_ptr(), n, 0, stream))
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
            hnd = _engine(h, w, x.device.index or 0)
            stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            dx = dw = db = None
            if ctx.needs_input_grad[0]:
                if _wino(x, weight):
                    upk = torch.empty(L.apz_wino_packed_size(), dtype=torch.float32, device=x.device)
                    _ck(L, L.apz_wino_pack(hnd, weight.data_ptr(), 1, upk.data_ptr(), stream))
                    dx = torch.empty_like(x)
                    _ck(L, L.apz_wino_conv(hnd, dy.data_ptr(), upk.data_ptr(), None, dx.data_ptr(), n, 0, stream))
                elif ci in (64, 128, 256):
                    wpk = torch.empty(L.apz_conv3x3_packed_size(co, ci), dtype=torch.float32, device=x.device)
                    _ck(L, L.apz_conv3x3_pack(hnd, weight.data_ptr(), ci, co, 1, wpk.data_ptr(), stream))
                    dx = torch.empty_like(x)
                    _ck(L, L.apz_conv3x3_fwd(hnd, dy.data_ptr(), wpk.data_ptr(), None, dx.data_ptr(), n, co, ci, 0, stream))
                else:       # e.g. a 9-plane input that needs a gradient: rare, let torch do it
                    dx = torch.nn.grad.conv2d_input(x.shape, weight, dy, padding=1)
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(weight)
                _ck(L, L.apz_conv3x3_wgrad(hnd, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, ci, co, stream))
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = dy.sum(dim=(0, 2, 3))
            return dx, dw, db

    return HipConv3x3


_FN = None


def conv3x3(x, weight, bias=None):
    """y = conv2d(x, weight, bias, padding=1) on the HIP kernels, differentiable."""
    global _FN
    if _FN is None:
        _FN = _function()
    return _FN.apply(x, weight, bias)
