"""8-fold dihedral augmentation of self-play tuples: drop-in for the reference's
`TrainPipeline.get_equi_data` (reference train_mxnet.py:115-135) as a table-driven gather, on
the host (NumPy) or on the GPU (`augment8_kernel`, include/alphapig_hip.h: apz_augment8).

Output order per tuple, as the reference emits it: rot90 x1, its left-right flip, rot90 x2, flip,
rot90 x3, flip, identity (rot90 x4), flip.  pi is stored bottom-row-first (move = h*W + w with
row 0 at the bottom) while planes are top-row-first, hence the two different index tables.
"""
import ctypes as C

import numpy as np

_TABLES = {}


def dihedral_tables(n):
    """-> (perm_s [8, n*n], perm_p [8, n*n]) with out[k].flat[i] = in.flat[perm[k][i]]."""
    if n not in _TABLES:
        idx = np.arange(n * n).reshape(n, n)
        s, p = idx, idx[::-1]                       # pi grid is flipped up-down before rotating
        ps, pp = [], []
        for _ in range(4):
            s, p = np.rot90(s), np.rot90(p)
            ps.append(s.ravel())
            pp.append(p[::-1].ravel())
            ps.append(s[:, ::-1].ravel())
            pp.append(p[:, ::-1][::-1].ravel())
        _TABLES[n] = (np.ascontiguousarray(np.stack(ps), dtype=np.int64),
                      np.ascontiguousarray(np.stack(pp), dtype=np.int64))
    return _TABLES[n]


def get_equi_data(play_data, board_height, board_width):
    """[(state [C,H,W], pi [H*W], z)] -> the 8x extended list, same order and dtypes as the
    reference (state arrays of shape [C,H,W], flattened pi, z passed through)."""
    assert board_height == board_width, "dihedral augmentation needs a square board"
    ps, pp = dihedral_tables(board_height)
    out = []
    for state, pi, z in play_data:
        st = np.asarray(state)
        flat = st.reshape(st.shape[0], -1)
        pv = np.asarray(pi).reshape(-1)
        for k in range(8):
            out.append((flat[:, ps[k]].reshape(st.shape), pv[pp[k]], z))
    return out


def augment8_gpu(net, planes, pis):
    """planes float32 [n, C, H, W], pis float32 [n, H*W] -> ([8n, C, H, W], [8n, H*W]) on the
    evaluator's GPU; rows 8i..8i+7 are the eight images of tuple i in the reference order."""
    x = np.ascontiguousarray(planes, dtype=np.float32)
    p = np.ascontiguousarray(pis, dtype=np.float32)
    n, c, h, w = x.shape
    L, hnd = net.L, net._h
    xo = np.empty((n * 8, c, h, w), dtype=np.float32)
    po = np.empty((n * 8, h * w), dtype=np.float32)
    dev = [L.apz_device_alloc(hnd, b) for b in (x.nbytes, p.nbytes, xo.nbytes, po.nbytes)]
    if not all(dev):
        raise MemoryError(L.apz_last_error().decode())
    try:
        net._ck(L.apz_memcpy_h2d(hnd, dev[0], x.ctypes.data, x.nbytes))
        net._ck(L.apz_memcpy_h2d(hnd, dev[1], p.ctypes.data, p.nbytes))
        net._ck(L.apz_augment8(hnd, dev[0], dev[1], n, c, dev[2], dev[3]))
        net._ck(L.apz_memcpy_d2h(hnd, xo.ctypes.data, dev[2], xo.nbytes))
        net._ck(L.apz_memcpy_d2h(hnd, po.ctypes.data, dev[3], po.nbytes))
    finally:
        for d in dev:
            L.apz_device_free(hnd, d)
    return xo, po
